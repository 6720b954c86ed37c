"""ctypes bindings of the C ABI in include/rln.h + include/rln_amd.h (zerokit_amd/lib/librln.so).

The library is the product; this file only declares signatures.  It is loaded with RTLD_GLOBAL so that it and torch
(when a program uses both) share one HIP runtime -- whichever of the two is loaded first brings in its
libamdhip64.so.7 and the other binds to it (same SONAME).  A program that uses both should import torch FIRST: with this
library (ROCm 7.2's runtime and RCCL) loaded before a torch wheel built for ROCm 7.0 the process aborted at exit with a
double free (measured in round 3).  Throughput does not depend on the order since the streamed inputs are staged by a
kernel instead of the copy path that was slow under the wheel's runtime (profiles/r3_rocprof_summary.md section 1);
bench.py imports torch only under torchrun."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RLNAMD_LIB") or os.path.join(_HERE, "lib", "librln.so")


class NativeMissing(RuntimeError):
    pass


class VecU8(C.Structure):
    _fields_ = [("ptr", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t)]


class CFr(C.Structure):
    _fields_ = [("le", C.c_uint8 * 32)]


class VecCFr(C.Structure):
    _fields_ = [("ptr", C.POINTER(CFr)), ("len", C.c_size_t), ("cap", C.c_size_t)]


class VecSize(C.Structure):
    _fields_ = [("ptr", C.POINTER(C.c_size_t)), ("len", C.c_size_t), ("cap", C.c_size_t)]


class VecBool(C.Structure):
    _fields_ = [("ptr", C.POINTER(C.c_bool)), ("len", C.c_size_t), ("cap", C.c_size_t)]


class CResultVecBool(C.Structure):
    _fields_ = [("ok", VecBool), ("err", VecU8)]


class VecString(C.Structure):  # repr_c::Vec<repr_c::String>
    _fields_ = [("ptr", C.POINTER(VecU8)), ("len", C.c_size_t), ("cap", C.c_size_t)]


class CBoolResult(C.Structure):
    _fields_ = [("ok", C.c_bool), ("err", VecU8)]


class CResultPtr(C.Structure):  # CResult<Box<T>, String>
    _fields_ = [("ok", C.c_void_p), ("err", VecU8)]


class CResultVecU8(C.Structure):
    _fields_ = [("ok", VecU8), ("err", VecU8)]


class CResultVecCFr(C.Structure):
    _fields_ = [("ok", VecCFr), ("err", VecU8)]


class MerkleProof(C.Structure):
    _fields_ = [("path_elements", VecCFr), ("path_index", VecU8)]


class ProverInfo(C.Structure):
    _fields_ = [("inputs_size", C.c_uint64), ("num_signals", C.c_uint64), ("domain_size", C.c_uint64),
                ("tree_depth", C.c_uint64), ("max_out", C.c_uint64), ("capacity", C.c_uint64),
                ("table_bytes", C.c_uint64), ("window_bits", C.c_int32), ("windows", C.c_int32),
                ("window_bits_g2", C.c_int32), ("windows_g2", C.c_int32), ("glv", C.c_int32), ("reserved", C.c_int32),
                ("g1_rows", C.c_uint64), ("g2_rows", C.c_uint64)]


P = C.c_void_p
PP = C.POINTER(C.c_void_p)
U8P = C.POINTER(C.c_uint8)
CFRP = C.POINTER(CFr)

# name -> (restype, argtypes); every symbol include/*.h declares
SIGNATURES = {
    # ---- rln_amd.h
    "rlnamd_last_error": (C.c_char_p, []),
    "rlnamd_device_count": (C.c_int, []),
    "rlnamd_set_device": (C.c_int, [C.c_int]),
    "rlnamd_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "rlnamd_device_name": (C.c_int, [C.c_char_p, C.c_size_t]),
    "rlnamd_poseidon_hash": (C.c_int, [C.c_char_p, C.c_size_t, C.c_size_t, C.c_char_p]),
    "rlnamd_hash_to_field_le": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p]),
    "rlnamd_hash_to_field_be": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p]),
    "rlnamd_tree_new": (C.c_int, [C.c_size_t, PP]),
    "rlnamd_tree_free": (None, [P]),
    "rlnamd_tree_set_range": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_size_t]),
    "rlnamd_tree_set_leaves": (C.c_int, [P, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "rlnamd_tree_root": (C.c_int, [P, C.c_char_p]),
    "rlnamd_tree_get_leaf": (C.c_int, [P, C.c_size_t, C.c_char_p]),
    "rlnamd_tree_proof": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p]),
    "rlnamd_tree_proofs": (C.c_int, [P, C.c_size_t, C.c_size_t, C.c_char_p, C.c_char_p]),
    "rlnamd_tree_fill_sequential": (C.c_int, [P, C.c_size_t, C.c_size_t, C.c_uint64]),
    "rlnamd_tree_bench": (C.c_int, [P, C.c_size_t, C.c_uint64, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_size_t)]),
    "rlnamd_prover_new": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, PP]),
    "rlnamd_prover_free": (None, [P]),
    "rlnamd_prover_get_info": (C.c_int, [P, C.POINTER(ProverInfo)]),
    "rlnamd_prover_input_slot": (C.c_int, [P, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rlnamd_prover_upload": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p]),
    "rlnamd_prover_run": (C.c_int, [P, C.c_size_t]),
    "rlnamd_prover_run_async": (C.c_int, [P, C.c_size_t]),
    "rlnamd_prover_sync": (C.c_int, [P]),
    "rlnamd_prover_run_mode": (C.c_int, [P, C.c_size_t, C.c_int]),
    "rlnamd_prover_run_async_mode": (C.c_int, [P, C.c_size_t, C.c_int]),
    "rlnamd_prover_upload_partial": (C.c_int, [P, C.c_size_t, C.c_char_p]),
    "rlnamd_prover_download_partial": (C.c_int, [P, C.c_size_t, C.c_char_p]),
    "rlnamd_prover_upload_witness": (C.c_int, [P, C.c_size_t, C.c_char_p]),
    "rlnamd_prover_known_mask": (C.c_int, [P, C.c_char_p]),
    "rlnamd_prover_download": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32)]),
    "rlnamd_prover_stage_ms": (C.c_int, [P, C.POINTER(C.c_float)]),
    "rlnamd_prover_walk_clock_mhz": (C.c_int, [P, C.POINTER(C.c_double)]),
    "rlnamd_prover_stage_name": (C.c_char_p, [C.c_int]),
    "rlnamd_prover_fetch_witness": (C.c_int, [P, C.c_size_t, C.c_char_p]),
    "rlnamd_prover_fetch_h": (C.c_int, [P, C.c_size_t, C.c_char_p]),
    "rlnamd_prover_residue": (C.c_int, [P, C.POINTER(C.c_uint64)]),
    "rlnamd_prover_collect_partial_cached": (C.c_int, [P, C.c_uint64, C.c_size_t, C.c_char_p, C.POINTER(C.c_uint64),
                                                       C.POINTER(C.c_uint32)]),
    "rlnamd_prover_submit_finish": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint64)]),
    "rlnamd_prover_release_partial": (C.c_int, [P, C.POINTER(C.c_uint64), C.c_size_t]),
    "rlnamd_prover_device_shared": (C.c_int, [P, C.POINTER(C.c_int)]),
    "rlnamd_prover_hint_stats": (C.c_int, [P, C.POINTER(C.c_uint64)]),
    "rlnamd_prover_hint_words": (C.c_uint32, [P]),
    "rlnamd_prover_hints_for": (C.c_int, [P, C.c_char_p, C.POINTER(C.c_uint32)]),
    "rlnamd_prover_submit_hinted": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint64)]),
    "rlnamd_prover_partial_cache_info": (C.c_int, [P, C.POINTER(C.c_uint64)]),
    "rlnamd_prover_init_ms": (C.c_int, [P, C.POINTER(C.c_float)]),
    "rlnamd_verify": (C.c_int, [P, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]),
    "rlnamd_prover_num_public": (C.c_size_t, [P]),
    "rlnamd_prover_download_public": (C.c_int, [P, C.c_size_t, C.c_char_p]),
    "rlnamd_verify_public": (C.c_int, [P, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "rlnamd_verify_with_zkey": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]),
    "rlnamd_verify_many": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p]),
    "rlnamd_verify_many_with_zkey": (C.c_int, [C.c_char_p, C.c_size_t, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t,
                                               C.c_int, C.c_char_p]),
    "rlnamd_parse_resources": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "rlnamd_proof_compress": (C.c_int, [C.c_char_p, C.c_char_p]),
    "rlnamd_proof_decompress": (C.c_int, [C.c_char_p, C.c_char_p]),
    "rlnamd_msm_new": (C.c_int, [C.c_size_t, PP]),
    "rlnamd_msm_new_g2": (C.c_int, [C.c_size_t, PP]),
    "rlnamd_msm_point_bytes": (C.c_size_t, [P]),
    "rlnamd_msm_window_sums_bytes_of": (C.c_size_t, [P]),
    "rlnamd_poseidon_params_check": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p]),
    "rlnamd_selftest_fq29": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32)]),
    "rlnamd_msm_free": (None, [P]),
    "rlnamd_msm_set": (C.c_int, [P, C.c_char_p, C.c_char_p, C.c_size_t]),
    "rlnamd_msm_generate": (C.c_int, [P, C.c_uint64, C.c_uint64, C.c_size_t]),
    "rlnamd_msm_generate_mode": (C.c_int, [P, C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint32]),
    "rlnamd_msm_fetch": (C.c_int, [P, C.c_size_t, C.c_size_t, C.c_char_p, C.c_char_p]),
    "rlnamd_msm_window_sums_bytes": (C.c_size_t, []),
    "rlnamd_msm_run": (C.c_int, [P, C.c_char_p, C.POINTER(C.c_float)]),
    "rlnamd_msm_combine": (C.c_int, [P, C.c_char_p, C.c_size_t, C.c_char_p]),
    "rlnamd_ffi_prover_info": (C.c_int, [P, C.POINTER(ProverInfo)]),
    "rlnamd_ffi_memo_stats": (C.c_int, [P, C.POINTER(C.c_uint64)]),
    "rlnamd_ffi_gather_stats": (C.c_int, [P, C.POINTER(C.c_uint64)]),
    "rlnamd_prover_slots": (C.c_int, [P]),
    "rlnamd_prover_submit": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p,
                                      C.POINTER(C.c_uint64)]),
    "rlnamd_prover_collect": (C.c_int, [P, C.c_uint64, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p,
                                       C.POINTER(C.c_uint32), C.c_char_p]),
    "rlnamd_prover_collect_public": (C.c_int, [P, C.c_uint64, C.c_size_t, C.c_char_p]),
    "rlnamd_prover_wipe": (C.c_int, [P]),
    "rlnamd_prover_describe": (C.c_int, [P, C.c_char_p, C.c_size_t]),
    "rlnamd_prover_prove_stream": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                            C.POINTER(C.c_uint32)]),
    "rlnamd_pool_new": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, C.c_int,
                                 C.POINTER(C.c_int), C.c_size_t, PP]),
    "rlnamd_pool_free": (None, [P]),
    "rlnamd_pool_size": (C.c_size_t, [P]),
    "rlnamd_pool_device": (C.c_int, [P, C.c_size_t]),
    "rlnamd_pool_get_info": (C.c_int, [P, C.POINTER(ProverInfo)]),
    "rlnamd_pool_prove": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                   C.POINTER(C.c_uint32)]),
    "rlnamd_pool_last_ms": (C.c_int, [P, C.POINTER(C.c_float)]),
    "rlnamd_pool_set_dynamic": (C.c_int, [P, C.c_int]),
    "rlnamd_pool_last_proofs": (C.c_int, [P, C.POINTER(C.c_size_t)]),
    "rlnamd_pool_inject_fault": (C.c_int, [P, C.c_size_t, C.c_size_t]),
    "rlnamd_pool_set_failover": (C.c_int, [P, C.c_int]),
    "rlnamd_pool_health": (C.c_int, [P, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "rlnamd_pool_revive": (C.c_int, [P, C.c_size_t]),
    "rlnamd_pool_set_probation": (C.c_int, [P, C.c_size_t]),
    "rlnamd_pool_verify_many": (C.c_int, [P, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p]),
    "rlnamd_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rlnamd_comm_init_rank": (C.c_int, [C.c_char_p, C.c_int, C.c_int, PP]),
    "rlnamd_comm_init_all": (C.c_int, [C.POINTER(C.c_int), C.c_size_t, PP]),
    "rlnamd_comm_free": (None, [P]),
    "rlnamd_comm_rank": (C.c_int, [P]),
    "rlnamd_comm_ranks": (C.c_int, [P]),
    "rlnamd_msm_run_sharded": (C.c_int, [P, P, C.c_char_p, C.POINTER(C.c_float)]),
    "rlnamd_msm_generated_multi": (C.c_int, [C.POINTER(C.c_int), C.c_size_t, C.c_uint64, C.c_size_t, C.c_uint32, C.c_int,
                                            C.c_char_p, C.POINTER(C.c_float)]),
    # ---- rln.h
    "ffi_rln_new": (CResultPtr, [C.c_size_t, C.c_char_p]),
    "ffi_rln_new_with_params": (CResultPtr, [C.c_size_t, C.POINTER(VecU8), C.POINTER(VecU8), C.c_char_p]),
    "ffi_rln_free": (None, [P]),
    "ffi_rln_get_tree_depth": (C.c_size_t, [PP]),
    "ffi_rln_get_max_out": (C.c_size_t, [PP]),
    "ffi_generate_rln_proof": (CResultPtr, [PP, PP]),
    "ffi_generate_rln_proof_with_rs": (CResultPtr, [PP, PP, CFRP, CFRP]),
    "ffi_generate_rln_proofs_batch": (CBoolResult, [PP, PP, C.c_size_t, CFRP, PP]),
    "ffi_finish_rln_proofs_batch": (CBoolResult, [PP, PP, PP, C.c_size_t, CFRP, PP]),
    "ffi_verify_rln_proof": (CBoolResult, [PP, PP, CFRP]),
    "ffi_verify_with_roots": (CBoolResult, [PP, PP, C.POINTER(VecCFr), CFRP]),
    "ffi_rln_proof_get_values": (P, [PP]),
    "ffi_rln_proof_get_version_byte": (C.c_uint8, [PP]),
    "ffi_rln_proof_to_bytes_le": (CResultVecU8, [PP]),
    "ffi_rln_proof_to_bytes_be": (CResultVecU8, [PP]),
    "ffi_bytes_le_to_rln_proof": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_rln_proof": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_proof_free": (None, [P]),
    "ffi_rln_witness_input_new_single": (CResultPtr, [CFRP, CFRP, CFRP, C.POINTER(VecCFr), C.POINTER(VecU8), CFRP, CFRP]),
    "ffi_rln_witness_input_new_multi": (CResultPtr, [CFRP, CFRP, C.POINTER(VecCFr), C.POINTER(VecCFr),
                                                     C.POINTER(VecU8), CFRP, CFRP, C.POINTER(VecBool)]),
    "ffi_rln_witness_input_get_message_ids": (VecCFr, [PP]),
    "ffi_rln_witness_input_get_selector_used": (VecBool, [PP]),
    "ffi_vec_bool_free": (None, [VecBool]),
    "ffi_rln_witness_input_get_version_byte": (C.c_uint8, [PP]),
    "ffi_rln_witness_input_get_identity_secret": (CFRP, [PP]),
    "ffi_rln_witness_input_get_user_message_limit": (CFRP, [PP]),
    "ffi_rln_witness_input_get_message_id": (CFRP, [PP]),
    "ffi_rln_witness_input_get_path_elements": (VecCFr, [PP]),
    "ffi_rln_witness_input_get_identity_path_index": (VecU8, [PP]),
    "ffi_rln_witness_input_get_x": (CFRP, [PP]),
    "ffi_rln_witness_input_get_external_nullifier": (CFRP, [PP]),
    "ffi_rln_witness_to_bytes_le": (CResultVecU8, [PP]),
    "ffi_rln_witness_to_bytes_be": (CResultVecU8, [PP]),
    "ffi_bytes_le_to_rln_witness": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_rln_witness": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_witness_to_bigint_json": (CResultVecU8, [PP]),
    "ffi_rln_witness_input_free": (None, [P]),
    "ffi_rln_partial_witness_input_new": (CResultPtr, [CFRP, CFRP, C.POINTER(VecCFr), C.POINTER(VecU8)]),
    "ffi_rln_witness_to_partial_witness": (P, [PP]),
    "ffi_rln_partial_witness_input_get_version_byte": (C.c_uint8, [PP]),
    "ffi_rln_partial_witness_input_get_identity_secret": (CFRP, [PP]),
    "ffi_rln_partial_witness_input_get_user_message_limit": (CFRP, [PP]),
    "ffi_rln_partial_witness_input_get_path_elements": (VecCFr, [PP]),
    "ffi_rln_partial_witness_input_get_identity_path_index": (VecU8, [PP]),
    "ffi_rln_partial_witness_to_bytes_le": (CResultVecU8, [PP]),
    "ffi_rln_partial_witness_to_bytes_be": (CResultVecU8, [PP]),
    "ffi_bytes_le_to_rln_partial_witness": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_rln_partial_witness": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_partial_witness_input_free": (None, [P]),
    "ffi_generate_rln_proof_with_witness": (CResultPtr, [PP, C.POINTER(VecString), PP]),
    "ffi_compute_id_secret": (CResultPtr, [CFRP, CFRP, CFRP, CFRP]),
    "ffi_recover_id_secret": (CResultPtr, [PP, PP]),
    "ffi_seeded_key_gen": (VecCFr, [C.POINTER(VecU8)]),
    "ffi_extended_key_gen": (VecCFr, []),
    "ffi_seeded_extended_key_gen": (VecCFr, [C.POINTER(VecU8)]),
    "ffi_generate_partial_zk_proof": (CResultPtr, [PP, PP]),
    "ffi_finish_rln_proof": (CResultPtr, [PP, PP, PP]),
    "ffi_finish_rln_proof_with_rs": (CResultPtr, [PP, PP, PP, CFRP, CFRP]),
    "ffi_rln_partial_proof_get_version_byte": (C.c_uint8, [PP]),
    "ffi_rln_partial_proof_to_bytes_le": (CResultVecU8, [PP]),
    "ffi_rln_partial_proof_to_bytes_be": (CResultVecU8, [PP]),
    "ffi_bytes_le_to_rln_partial_proof": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_rln_partial_proof": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_partial_proof_free": (None, [P]),
    "ffi_rln_proof_values_get_root": (CFRP, [PP]),
    "ffi_rln_proof_values_get_x": (CFRP, [PP]),
    "ffi_rln_proof_values_get_external_nullifier": (CFRP, [PP]),
    "ffi_rln_proof_values_get_y": (CResultPtr, [PP]),
    "ffi_rln_proof_values_get_nullifier": (CResultPtr, [PP]),
    "ffi_rln_proof_values_get_ys": (CResultVecCFr, [PP]),
    "ffi_rln_proof_values_get_nullifiers": (CResultVecCFr, [PP]),
    "ffi_rln_proof_values_get_selector_used": (CResultVecBool, [PP]),
    "ffi_rln_proof_values_get_version_byte": (C.c_uint8, [PP]),
    "ffi_rln_proof_values_to_bytes_le": (VecU8, [PP]),
    "ffi_rln_proof_values_to_bytes_be": (VecU8, [PP]),
    "ffi_bytes_le_to_rln_proof_values": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_rln_proof_values": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_proof_values_free": (None, [P]),
    "ffi_set_tree": (CBoolResult, [PP, C.c_size_t]),
    "ffi_delete_leaf": (CBoolResult, [PP, C.c_size_t]),
    "ffi_set_leaf": (CBoolResult, [PP, C.c_size_t, CFRP]),
    "ffi_get_leaf": (CResultPtr, [PP, C.c_size_t]),
    "ffi_leaves_set": (C.c_size_t, [PP]),
    "ffi_set_next_leaf": (CBoolResult, [PP, CFRP]),
    "ffi_set_leaves_from": (CBoolResult, [PP, C.c_size_t, C.POINTER(VecCFr)]),
    "ffi_init_tree_with_leaves": (CBoolResult, [PP, C.POINTER(VecCFr)]),
    "ffi_atomic_operation": (CBoolResult, [PP, C.c_size_t, C.POINTER(VecCFr), C.POINTER(VecSize)]),
    "ffi_seq_atomic_operation": (CBoolResult, [PP, C.POINTER(VecCFr), C.POINTER(VecU8)]),
    "ffi_get_root": (CFRP, [PP]),
    "ffi_get_merkle_proof": (CResultPtr, [PP, C.c_size_t]),
    "ffi_merkle_proof_free": (None, [P]),
    "ffi_set_metadata": (CBoolResult, [PP, C.POINTER(VecU8)]),
    "ffi_get_metadata": (CResultVecU8, [PP]),
    "ffi_flush": (CBoolResult, [PP]),
    "ffi_cfr_zero": (CFRP, []),
    "ffi_cfr_one": (CFRP, []),
    "ffi_cfr_to_bytes_le": (CResultVecU8, [CFRP]),
    "ffi_cfr_to_bytes_be": (CResultVecU8, [CFRP]),
    "ffi_bytes_le_to_cfr": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_cfr": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_uint_to_cfr": (CFRP, [C.c_uint32]),
    "ffi_cfr_debug": (VecU8, [CFRP]),
    "ffi_cfr_free": (None, [P]),
    "ffi_vec_cfr_new": (VecCFr, [C.c_size_t]),
    "ffi_vec_cfr_from_cfr": (VecCFr, [CFRP]),
    "ffi_vec_cfr_push": (None, [C.POINTER(VecCFr), CFRP]),
    "ffi_vec_cfr_len": (C.c_size_t, [C.POINTER(VecCFr)]),
    "ffi_vec_cfr_get": (CFRP, [C.POINTER(VecCFr), C.c_size_t]),
    "ffi_vec_cfr_to_bytes_le": (CResultVecU8, [C.POINTER(VecCFr)]),
    "ffi_vec_cfr_to_bytes_be": (CResultVecU8, [C.POINTER(VecCFr)]),
    "ffi_bytes_le_to_vec_cfr": (CResultVecCFr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_vec_cfr": (CResultVecCFr, [C.POINTER(VecU8)]),
    "ffi_vec_cfr_debug": (VecU8, [C.POINTER(VecCFr)]),
    "ffi_vec_cfr_free": (None, [VecCFr]),
    "ffi_vec_u8_to_bytes_le": (CResultVecU8, [C.POINTER(VecU8)]),
    "ffi_vec_u8_to_bytes_be": (CResultVecU8, [C.POINTER(VecU8)]),
    "ffi_bytes_le_to_vec_u8": (CResultVecU8, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_vec_u8": (CResultVecU8, [C.POINTER(VecU8)]),
    "ffi_vec_u8_debug": (VecU8, [C.POINTER(VecU8)]),
    "ffi_vec_u8_free": (None, [VecU8]),
    "ffi_hash_to_field_le": (CFRP, [C.POINTER(VecU8)]),
    "ffi_hash_to_field_be": (CFRP, [C.POINTER(VecU8)]),
    "ffi_poseidon_hash_pair": (CFRP, [CFRP, CFRP]),
    "ffi_key_gen": (VecCFr, []),
    # ---- V3 mirror (include/rln.h, ffi_rln_v3.rs:323-1609)
    "ffi_rln_v3_new_stateless_default": (P, []),
    "ffi_rln_v3_new_stateless": (CResultPtr, [C.POINTER(VecU8), C.POINTER(VecU8)]),
    "ffi_rln_v3_new_with_full_merkle_tree_default": (P, []),
    "ffi_rln_v3_new_with_full_merkle_tree": (CResultPtr, [C.c_size_t, C.POINTER(VecU8), C.POINTER(VecU8)]),
    "ffi_rln_v3_new_with_optimal_merkle_tree_default": (P, []),
    "ffi_rln_v3_new_with_optimal_merkle_tree": (CResultPtr, [C.c_size_t, C.POINTER(VecU8), C.POINTER(VecU8)]),
    "ffi_rln_v3_new_with_pm_tree_default": (P, []),
    "ffi_rln_v3_new_with_pm_tree": (CResultPtr, [C.c_size_t, C.POINTER(VecU8), C.POINTER(VecU8), C.c_char_p]),
    "ffi_rln_v3_free": (None, [P]),
    "ffi_rln_v3_generate_proof": (CResultPtr, [PP, PP]),
    "ffi_rln_v3_generate_proof_with_rs": (CResultPtr, [PP, PP, CFRP, CFRP]),
    "ffi_rln_v3_verify": (CBoolResult, [PP, PP, CFRP]),
    "ffi_rln_v3_verify_with_roots": (CBoolResult, [PP, PP, C.POINTER(VecCFr), CFRP]),
    "ffi_rln_v3_generate_partial_proof": (CResultPtr, [PP, PP]),
    "ffi_rln_v3_finish_proof": (CResultPtr, [PP, PP, PP]),
    "ffi_rln_v3_witness_input_new_single": (CResultPtr, [CFRP, CFRP, CFRP, C.POINTER(VecCFr), C.POINTER(VecU8), CFRP, CFRP]),
    "ffi_rln_v3_witness_input_new_multi": (CResultPtr, [CFRP, CFRP, C.POINTER(VecCFr), C.POINTER(VecCFr), C.POINTER(VecU8), CFRP, CFRP, C.POINTER(VecBool)]),
    "ffi_rln_v3_witness_input_get_identity_secret": (CFRP, [PP]),
    "ffi_rln_v3_witness_input_get_user_message_limit": (CFRP, [PP]),
    "ffi_rln_v3_witness_input_get_message_id": (CResultPtr, [PP]),
    "ffi_rln_v3_witness_input_get_message_ids": (CResultVecCFr, [PP]),
    "ffi_rln_v3_witness_input_get_path_elements": (VecCFr, [PP]),
    "ffi_rln_v3_witness_input_get_identity_path_index": (VecU8, [PP]),
    "ffi_rln_v3_witness_input_get_x": (CFRP, [PP]),
    "ffi_rln_v3_witness_input_get_external_nullifier": (CFRP, [PP]),
    "ffi_rln_v3_witness_input_get_selector_used": (CResultVecBool, [PP]),
    "ffi_rln_v3_witness_to_bytes_le": (CResultVecU8, [PP]),
    "ffi_rln_v3_witness_to_bytes_be": (CResultVecU8, [PP]),
    "ffi_bytes_le_to_rln_v3_witness": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_rln_v3_witness": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_v3_witness_input_free": (None, [P]),
    "ffi_rln_v3_partial_witness_input_new": (CResultPtr, [CFRP, CFRP, C.POINTER(VecCFr), C.POINTER(VecU8)]),
    "ffi_rln_v3_partial_witness_input_get_identity_secret": (CFRP, [PP]),
    "ffi_rln_v3_partial_witness_input_get_user_message_limit": (CFRP, [PP]),
    "ffi_rln_v3_partial_witness_input_get_path_elements": (VecCFr, [PP]),
    "ffi_rln_v3_partial_witness_input_get_identity_path_index": (VecU8, [PP]),
    "ffi_rln_v3_witness_to_partial_witness": (P, [PP]),
    "ffi_rln_v3_partial_witness_to_bytes_le": (CResultVecU8, [PP]),
    "ffi_rln_v3_partial_witness_to_bytes_be": (CResultVecU8, [PP]),
    "ffi_bytes_le_to_rln_v3_partial_witness": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_rln_v3_partial_witness": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_v3_partial_witness_input_free": (None, [P]),
    "ffi_rln_v3_proof_get_values": (P, [PP]),
    "ffi_rln_v3_proof_to_bytes_le": (CResultVecU8, [PP]),
    "ffi_rln_v3_proof_to_bytes_mixed": (CResultVecU8, [PP]),
    "ffi_bytes_le_to_rln_v3_proof": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_mixed_to_rln_v3_proof": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_v3_proof_free": (None, [P]),
    "ffi_rln_v3_partial_proof_to_bytes_le": (CResultVecU8, [PP]),
    "ffi_bytes_le_to_rln_v3_partial_proof": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_v3_partial_proof_free": (None, [P]),
    "ffi_rln_v3_proof_values_get_root": (CFRP, [PP]),
    "ffi_rln_v3_proof_values_get_x": (CFRP, [PP]),
    "ffi_rln_v3_proof_values_get_external_nullifier": (CFRP, [PP]),
    "ffi_rln_v3_proof_values_get_y": (CResultPtr, [PP]),
    "ffi_rln_v3_proof_values_get_nullifier": (CResultPtr, [PP]),
    "ffi_rln_v3_proof_values_get_selector_used": (CResultVecBool, [PP]),
    "ffi_rln_v3_proof_values_get_ys": (CResultVecCFr, [PP]),
    "ffi_rln_v3_proof_values_get_nullifiers": (CResultVecCFr, [PP]),
    "ffi_rln_v3_proof_values_to_bytes_le": (CResultVecU8, [PP]),
    "ffi_rln_v3_proof_values_to_bytes_be": (CResultVecU8, [PP]),
    "ffi_bytes_le_to_rln_v3_proof_values": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_bytes_be_to_rln_v3_proof_values": (CResultPtr, [C.POINTER(VecU8)]),
    "ffi_rln_v3_proof_values_free": (None, [P]),
    "ffi_rln_v3_compute_id_secret": (CResultPtr, [CFRP, CFRP, CFRP, CFRP]),
    "ffi_rln_v3_recover_id_secret": (CResultPtr, [PP, PP]),
    "ffi_rln_v3_merkle_proof_free": (None, [P]),
    "ffi_rln_v3_delete_leaf": (CBoolResult, [PP, C.c_size_t]),
    "ffi_rln_v3_set_leaf": (CBoolResult, [PP, C.c_size_t, CFRP]),
    "ffi_rln_v3_get_leaf": (CResultPtr, [PP, C.c_size_t]),
    "ffi_rln_v3_leaves_set": (C.c_size_t, [PP]),
    "ffi_rln_v3_set_next_leaf": (CBoolResult, [PP, CFRP]),
    "ffi_rln_v3_set_leaves_from": (CBoolResult, [PP, C.c_size_t, C.POINTER(VecCFr)]),
    "ffi_rln_v3_init_tree_with_leaves": (CBoolResult, [PP, C.POINTER(VecCFr)]),
    "ffi_rln_v3_atomic_operation": (CBoolResult, [PP, C.c_size_t, C.POINTER(VecCFr), C.POINTER(VecSize)]),
    "ffi_rln_v3_seq_atomic_operation": (CBoolResult, [PP, C.POINTER(VecCFr), C.POINTER(VecU8)]),
    "ffi_rln_v3_get_root": (CFRP, [PP]),
    "ffi_rln_v3_get_merkle_proof": (CResultPtr, [PP, C.c_size_t]),
    "ffi_rln_v3_set_metadata": (CBoolResult, [PP, C.POINTER(VecU8)]),
    "ffi_rln_v3_get_metadata": (CResultVecU8, [PP]),
    "ffi_rln_v3_flush": (CBoolResult, [PP]),
    "ffi_c_string_free": (None, [VecU8]),
}

_lib = None


def lib():
    """The loaded C-ABI library; raises NativeMissing when it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeMissing("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(make -C zerokit_amd/csrc).  There is no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def last_error():
    return (lib().rlnamd_last_error() or b"").decode("utf-8", "replace")


class RLNError(RuntimeError):
    """Mirrors RLNError (rln/src/error.rs:99-114): the message is the string the C ABI returned."""


def check(rc):
    if rc != 0:
        raise RLNError(last_error())
