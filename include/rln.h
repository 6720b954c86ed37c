/* rln.h -- zerokit-compatible C ABI of the MI355X RLN backend (drop-in for `librln`).
 *
 * Same symbols, argument meaning, struct layouts, ownership and error convention as the header safer-ffi
 * generates from /root/reference/rln/src/ffi/{ffi_rln,ffi_tree,ffi_utils}.rs (cargo run --features headers
 * --bin generate_headers, rln/src/ffi/mod.rs:11-14).  Each declaration cites the #[ffi_export] it replaces.
 *
 *  - `T* const*` / `T**` arguments are safer-ffi's `&repr_c::Box<T>` / `&mut repr_c::Box<T>`: the address of
 *    the caller's pointer variable (see rln/ffi_c_examples/basic_proof.c:48-49).
 *  - Vec_T = { ptr, len, cap } (rln/ffi_nim_examples/rln.nim:28-46); strings are Vec_uint8_t, NUL-terminated
 *    when produced by the library.
 *  - CResult: `ok` is NULL (or an empty Vec) on failure and `err.ptr` is NULL on success
 *    (ffi_utils.rs:15-29).  Everything returned is caller-owned and released with its `ffi_*_free`.
 *  - CFr is opaque; elements of a Vec_CFr_t are contiguous CFr values reachable through ffi_vec_cfr_get
 *    (ffi_utils.rs:183-185).  In this implementation a CFr is the 32-byte little-endian canonical value.
 *
 * Scope: single and multi message-id circuits, full and partial proofs, the HBM-resident tree with the
 * PmTreeConfig semantics of config_path (path / temporary / tree_depth; a persistent tree is one snapshot file
 * <path>/rlnamd_tree.bin written by ffi_flush and when the object is freed -- not sled's on-disk format, which belongs
 * to a third-party crate).  Depth <= 30: the dense tree in HBM; depth 31 .. 63: a sparse tree (only written nodes are
 * kept, hashes in device batches -- OptimalMerkleTree's semantics, utils/src/merkle_tree/optimal_merkle_tree.rs);
 * depth >= 64 is the reference's InvalidDepth.  Replacing a dense tree (ffi_set_tree, ffi_init_tree_with_leaves) builds
 * the new one beside the old when the device can hold both and releases the old one first otherwise.  config_path may also carry "window_bits" and "max_batch" (prover sizing, see
 * INTEGRATION.md).  The ffi_rln_v3_* mirror is at the end of this file.
 * Extensions that the reference lacks are marked EXT (deterministic blinding, batch: any n, streamed through the
 * workspace slots).
 */
#ifndef RLN_H
#define RLN_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct CFr CFr_t;                                 /* ffi_utils.rs:34-36  #[repr(opaque)] */
typedef struct FFI_RLN FFI_RLN_t;                         /* ffi_rln.rs:16-18 */
typedef struct FFI_RLNProof FFI_RLNProof_t;               /* ffi_rln.rs:153-155 */
typedef struct FFI_RLNWitnessInput FFI_RLNWitnessInput_t; /* ffi_rln.rs:322-324 */
typedef struct FFI_RLNProofValues FFI_RLNProofValues_t;   /* ffi_rln.rs:714-716 */
typedef struct FFI_RLNPartialWitnessInput FFI_RLNPartialWitnessInput_t; /* ffi_rln.rs:562-564 */
typedef struct FFI_RLNPartialProof FFI_RLNPartialProof_t; /* ffi_rln.rs:239-241 */

typedef struct Vec_uint8 { uint8_t* ptr; size_t len; size_t cap; } Vec_uint8_t;
typedef struct Vec_CFr { CFr_t* ptr; size_t len; size_t cap; } Vec_CFr_t;
typedef struct Vec_size { size_t* ptr; size_t len; size_t cap; } Vec_size_t;
typedef struct Vec_bool { bool* ptr; size_t len; size_t cap; } Vec_bool_t;
typedef struct Vec_String { Vec_uint8_t* ptr; size_t len; size_t cap; } Vec_String_t; /* repr_c::Vec<repr_c::String> */

typedef struct CBoolResult { bool ok; Vec_uint8_t err; } CBoolResult_t;                  /* ffi_utils.rs:24-29 */
typedef struct FFI_MerkleProof { Vec_CFr_t path_elements; Vec_uint8_t path_index; } FFI_MerkleProof_t; /* ffi_tree.rs:13-18 */

typedef struct { FFI_RLN_t* ok; Vec_uint8_t err; } CResult_FFI_RLN_ptr_Vec_uint8_t;
typedef struct { FFI_RLNProof_t* ok; Vec_uint8_t err; } CResult_FFI_RLNProof_ptr_Vec_uint8_t;
typedef struct { FFI_RLNWitnessInput_t* ok; Vec_uint8_t err; } CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t;
typedef struct { FFI_RLNProofValues_t* ok; Vec_uint8_t err; } CResult_FFI_RLNProofValues_ptr_Vec_uint8_t;
typedef struct { FFI_MerkleProof_t* ok; Vec_uint8_t err; } CResult_FFI_MerkleProof_ptr_Vec_uint8_t;
typedef struct { FFI_RLNPartialWitnessInput_t* ok; Vec_uint8_t err; } CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t;
typedef struct { FFI_RLNPartialProof_t* ok; Vec_uint8_t err; } CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t;
typedef struct { CFr_t* ok; Vec_uint8_t err; } CResult_CFr_ptr_Vec_uint8_t;
typedef struct { Vec_CFr_t ok; Vec_uint8_t err; } CResult_Vec_CFr_Vec_uint8_t;
typedef struct { Vec_uint8_t ok; Vec_uint8_t err; } CResult_Vec_uint8_Vec_uint8_t;
typedef struct { Vec_bool_t ok; Vec_uint8_t err; } CResult_Vec_bool_Vec_uint8_t;

/* ---- RLN object ------------------------------------------------------------------------------------ */
CResult_FFI_RLN_ptr_Vec_uint8_t ffi_rln_new(size_t tree_depth, const char* config_path);      /* ffi_rln.rs:24-57 */
CResult_FFI_RLN_ptr_Vec_uint8_t ffi_rln_new_with_params(size_t tree_depth, const Vec_uint8_t* zkey_data,
                                                        const Vec_uint8_t* graph_data,
                                                        const char* config_path);             /* ffi_rln.rs:76-116 */
void ffi_rln_free(FFI_RLN_t* rln);                                                            /* ffi_rln.rs:137 */
size_t ffi_rln_get_tree_depth(FFI_RLN_t* const* rln);                                         /* ffi_rln.rs:142 */
size_t ffi_rln_get_max_out(FFI_RLN_t* const* rln);                                            /* ffi_rln.rs:147 */

/* ---- proofs ---------------------------------------------------------------------------------------- */
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_generate_rln_proof(FFI_RLN_t* const* rln,
                                                            FFI_RLNWitnessInput_t* const* witness); /* ffi_rln.rs:852-872 */
CBoolResult_t ffi_verify_rln_proof(FFI_RLN_t* const* rln, FFI_RLNProof_t* const* proof, const CFr_t* x); /* :966-984 */
CBoolResult_t ffi_verify_with_roots(FFI_RLN_t* const* rln, FFI_RLNProof_t* const* proof, const Vec_CFr_t* roots,
                                    const CFr_t* x);                                          /* ffi_rln.rs:987-1010 */
/* proof from an externally calculated witness (decimal strings, one per witness signal) */
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_generate_rln_proof_with_witness(FFI_RLN_t* const* rln,
                                                                         const Vec_String_t* calculated_witness,
                                                                         FFI_RLNWitnessInput_t* const* witness); /* ffi_rln.rs:875-920 */
/* slashing (protocol/slashing.rs:12-100) */
CResult_CFr_ptr_Vec_uint8_t ffi_compute_id_secret(const CFr_t* share1_x, const CFr_t* share1_y, const CFr_t* share2_x,
                                                  const CFr_t* share2_y);                       /* ffi_rln.rs:1015 */
CResult_CFr_ptr_Vec_uint8_t ffi_recover_id_secret(FFI_RLNProofValues_t* const* proof_values_1,
                                                  FFI_RLNProofValues_t* const* proof_values_2); /* ffi_rln.rs:1036 */
/* EXT: generate_zk_proof_with_rs (protocol/proof.rs:753-777) -- explicit blinding scalars r, s */
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_generate_rln_proof_with_rs(FFI_RLN_t* const* rln,
                                                                    FFI_RLNWitnessInput_t* const* witness,
                                                                    const CFr_t* r, const CFr_t* s);
/* EXT: n independent proofs in one device batch.  witnesses: array of n witness pointers; rs: NULL (random
 * blinding) or 2n CFr (r_0, s_0, r_1, ...); out: array of n proof pointers filled on success. */
CBoolResult_t ffi_generate_rln_proofs_batch(FFI_RLN_t* const* rln, FFI_RLNWitnessInput_t* const* witnesses, size_t n,
                                            const CFr_t* rs, FFI_RLNProof_t** out);

/* EXT: n finishes in one call (single message-id): partials[i] is the partial proof witnesses[i] is finished from -- the
 * same pointer may repeat, one member's partial proof finished for n messages being what partial proofs are for
 * (rln/README.md:360-375); rs: NULL or 2n CFr; out: n proof pointers filled on success.  Any n: chunks of the
 * workspace's capacity, streamed.  The first request that cannot be finished fails the call. */
CBoolResult_t ffi_finish_rln_proofs_batch(FFI_RLN_t* const* rln, FFI_RLNPartialProof_t* const* partials,
                                          FFI_RLNWitnessInput_t* const* witnesses, size_t n, const CFr_t* rs,
                                          FFI_RLNProof_t** out);

FFI_RLNProofValues_t* ffi_rln_proof_get_values(FFI_RLNProof_t* const* proof);                 /* ffi_rln.rs:158 */
uint8_t ffi_rln_proof_get_version_byte(FFI_RLNProof_t* const* proof);                         /* ffi_rln.rs:165 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_proof_to_bytes_le(FFI_RLNProof_t* const* proof);        /* ffi_rln.rs:170 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_proof_to_bytes_be(FFI_RLNProof_t* const* proof);        /* ffi_rln.rs:186 */
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_bytes_le_to_rln_proof(const Vec_uint8_t* bytes);     /* ffi_rln.rs:202 */
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_bytes_be_to_rln_proof(const Vec_uint8_t* bytes);     /* ffi_rln.rs:218 */
void ffi_rln_proof_free(FFI_RLNProof_t* proof);                                               /* ffi_rln.rs:234 */

/* ---- witness input --------------------------------------------------------------------------------- */
CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t ffi_rln_witness_input_new_single(
    const CFr_t* identity_secret, const CFr_t* user_message_limit, const CFr_t* message_id,
    const Vec_CFr_t* path_elements, const Vec_uint8_t* identity_path_index, const CFr_t* x,
    const CFr_t* external_nullifier);                                                          /* ffi_rln.rs:327-358 */
CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t ffi_rln_witness_input_new_multi(
    const CFr_t* identity_secret, const CFr_t* user_message_limit, const Vec_CFr_t* message_ids,
    const Vec_CFr_t* path_elements, const Vec_uint8_t* identity_path_index, const CFr_t* x,
    const CFr_t* external_nullifier, const Vec_bool_t* selector_used);                          /* ffi_rln.rs:361-396 */
Vec_CFr_t ffi_rln_witness_input_get_message_ids(FFI_RLNWitnessInput_t* const* w);              /* ffi_rln.rs:425 */
Vec_bool_t ffi_rln_witness_input_get_selector_used(FFI_RLNWitnessInput_t* const* w);           /* ffi_rln.rs:470 */
void ffi_vec_bool_free(Vec_bool_t v);
uint8_t ffi_rln_witness_input_get_version_byte(FFI_RLNWitnessInput_t* const* w);               /* ffi_rln.rs:399 */
CFr_t* ffi_rln_witness_input_get_identity_secret(FFI_RLNWitnessInput_t* const* w);             /* ffi_rln.rs:404 */
CFr_t* ffi_rln_witness_input_get_user_message_limit(FFI_RLNWitnessInput_t* const* w);          /* ffi_rln.rs:411 */
CFr_t* ffi_rln_witness_input_get_message_id(FFI_RLNWitnessInput_t* const* w);                  /* ffi_rln.rs:418 */
Vec_CFr_t ffi_rln_witness_input_get_path_elements(FFI_RLNWitnessInput_t* const* w);            /* ffi_rln.rs:438 */
Vec_uint8_t ffi_rln_witness_input_get_identity_path_index(FFI_RLNWitnessInput_t* const* w);    /* ffi_rln.rs:451 */
CFr_t* ffi_rln_witness_input_get_x(FFI_RLNWitnessInput_t* const* w);                           /* ffi_rln.rs:458 */
CFr_t* ffi_rln_witness_input_get_external_nullifier(FFI_RLNWitnessInput_t* const* w);          /* ffi_rln.rs:463 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_witness_to_bytes_le(FFI_RLNWitnessInput_t* const* w);    /* ffi_rln.rs:477 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_witness_to_bytes_be(FFI_RLNWitnessInput_t* const* w);    /* ffi_rln.rs:493 */
CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t ffi_bytes_le_to_rln_witness(const Vec_uint8_t* b); /* ffi_rln.rs:509 */
CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t ffi_bytes_be_to_rln_witness(const Vec_uint8_t* b); /* ffi_rln.rs:525 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_witness_to_bigint_json(FFI_RLNWitnessInput_t* const* w); /* ffi_rln.rs:541 */
void ffi_rln_witness_input_free(FFI_RLNWitnessInput_t* w);                                     /* ffi_rln.rs:557 */

/* ---- partial proofs (proof split: precompute the member-dependent part, finish per message) ----------- */
CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t ffi_rln_partial_witness_input_new(
    const CFr_t* identity_secret, const CFr_t* user_message_limit, const Vec_CFr_t* path_elements,
    const Vec_uint8_t* identity_path_index);                                                    /* ffi_rln.rs:567-592 */
FFI_RLNPartialWitnessInput_t* ffi_rln_witness_to_partial_witness(FFI_RLNWitnessInput_t* const* w); /* ffi_rln.rs:636 */
uint8_t ffi_rln_partial_witness_input_get_version_byte(FFI_RLNPartialWitnessInput_t* const* w);  /* ffi_rln.rs:595 */
CFr_t* ffi_rln_partial_witness_input_get_identity_secret(FFI_RLNPartialWitnessInput_t* const* w); /* ffi_rln.rs:602 */
CFr_t* ffi_rln_partial_witness_input_get_user_message_limit(FFI_RLNPartialWitnessInput_t* const* w); /* :609 */
Vec_CFr_t ffi_rln_partial_witness_input_get_path_elements(FFI_RLNPartialWitnessInput_t* const* w); /* :616 */
Vec_uint8_t ffi_rln_partial_witness_input_get_identity_path_index(FFI_RLNPartialWitnessInput_t* const* w); /* :629 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_partial_witness_to_bytes_le(FFI_RLNPartialWitnessInput_t* const* w); /* :644 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_partial_witness_to_bytes_be(FFI_RLNPartialWitnessInput_t* const* w); /* :660 */
CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t ffi_bytes_le_to_rln_partial_witness(const Vec_uint8_t* b); /* :676 */
CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t ffi_bytes_be_to_rln_partial_witness(const Vec_uint8_t* b); /* :692 */
void ffi_rln_partial_witness_input_free(FFI_RLNPartialWitnessInput_t* w);                       /* ffi_rln.rs:708 */
CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t ffi_generate_partial_zk_proof(
    FFI_RLN_t* const* rln, FFI_RLNPartialWitnessInput_t* const* partial_witness);               /* ffi_rln.rs:922-936 */
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_finish_rln_proof(FFI_RLN_t* const* rln, FFI_RLNPartialProof_t* const* partial,
                                                          FFI_RLNWitnessInput_t* const* witness); /* ffi_rln.rs:939-960 */
/* EXT: finish_zk_proof_with_rs (protocol/proof.rs:821-849) */
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_finish_rln_proof_with_rs(FFI_RLN_t* const* rln,
                                                                  FFI_RLNPartialProof_t* const* partial,
                                                                  FFI_RLNWitnessInput_t* const* witness,
                                                                  const CFr_t* r, const CFr_t* s);
uint8_t ffi_rln_partial_proof_get_version_byte(FFI_RLNPartialProof_t* const* partial);          /* ffi_rln.rs:245 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_partial_proof_to_bytes_le(FFI_RLNPartialProof_t* const* partial); /* ffi_rln.rs:252 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_partial_proof_to_bytes_be(FFI_RLNPartialProof_t* const* partial); /* ffi_rln.rs:289 */
CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t ffi_bytes_le_to_rln_partial_proof(const Vec_uint8_t* bytes); /* ffi_rln.rs:268 */
CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t ffi_bytes_be_to_rln_partial_proof(const Vec_uint8_t* bytes); /* ffi_rln.rs:305 */
void ffi_rln_partial_proof_free(FFI_RLNPartialProof_t* partial);                                /* ffi_rln.rs:284 */

/* ---- proof values ---------------------------------------------------------------------------------- */
CFr_t* ffi_rln_proof_values_get_root(FFI_RLNProofValues_t* const* pv);                         /* ffi_rln.rs:719 */
CFr_t* ffi_rln_proof_values_get_x(FFI_RLNProofValues_t* const* pv);                            /* ffi_rln.rs:724 */
CFr_t* ffi_rln_proof_values_get_external_nullifier(FFI_RLNProofValues_t* const* pv);           /* ffi_rln.rs:729 */
CResult_CFr_ptr_Vec_uint8_t ffi_rln_proof_values_get_y(FFI_RLNProofValues_t* const* pv);       /* ffi_rln.rs:736 */
CResult_CFr_ptr_Vec_uint8_t ffi_rln_proof_values_get_nullifier(FFI_RLNProofValues_t* const* pv); /* ffi_rln.rs:746 */
CResult_Vec_bool_Vec_uint8_t ffi_rln_proof_values_get_selector_used(FFI_RLNProofValues_t* const* pv); /* ffi_rln.rs:756 */
CResult_Vec_CFr_Vec_uint8_t ffi_rln_proof_values_get_ys(FFI_RLNProofValues_t* const* pv);      /* ffi_rln.rs:766 */
CResult_Vec_CFr_Vec_uint8_t ffi_rln_proof_values_get_nullifiers(FFI_RLNProofValues_t* const* pv); /* ffi_rln.rs:782 */
uint8_t ffi_rln_proof_values_get_version_byte(FFI_RLNProofValues_t* const* pv);                /* ffi_rln.rs:798 */
Vec_uint8_t ffi_rln_proof_values_to_bytes_le(FFI_RLNProofValues_t* const* pv);                 /* ffi_rln.rs:803 */
Vec_uint8_t ffi_rln_proof_values_to_bytes_be(FFI_RLNProofValues_t* const* pv);                 /* ffi_rln.rs:808 */
CResult_FFI_RLNProofValues_ptr_Vec_uint8_t ffi_bytes_le_to_rln_proof_values(const Vec_uint8_t* b); /* ffi_rln.rs:813 */
CResult_FFI_RLNProofValues_ptr_Vec_uint8_t ffi_bytes_be_to_rln_proof_values(const Vec_uint8_t* b); /* ffi_rln.rs:829 */
void ffi_rln_proof_values_free(FFI_RLNProofValues_t* pv);                                      /* ffi_rln.rs:845 */

/* ---- Merkle tree (ffi_tree.rs) --------------------------------------------------------------------- */
CBoolResult_t ffi_set_tree(FFI_RLN_t** rln, size_t tree_depth);                                /* ffi_tree.rs:28-41 */
CBoolResult_t ffi_delete_leaf(FFI_RLN_t** rln, size_t index);                                  /* ffi_tree.rs:44 */
CBoolResult_t ffi_set_leaf(FFI_RLN_t** rln, size_t index, const CFr_t* leaf);                  /* ffi_tree.rs:58 */
CResult_CFr_ptr_Vec_uint8_t ffi_get_leaf(FFI_RLN_t* const* rln, size_t index);                 /* ffi_tree.rs:72 */
size_t ffi_leaves_set(FFI_RLN_t* const* rln);                                                  /* ffi_tree.rs:89 */
CBoolResult_t ffi_set_next_leaf(FFI_RLN_t** rln, const CFr_t* leaf);                           /* ffi_tree.rs:94 */
CBoolResult_t ffi_set_leaves_from(FFI_RLN_t** rln, size_t index, const Vec_CFr_t* leaves);     /* ffi_tree.rs:108 */
CBoolResult_t ffi_init_tree_with_leaves(FFI_RLN_t** rln, const Vec_CFr_t* leaves);             /* ffi_tree.rs:127 */
CBoolResult_t ffi_atomic_operation(FFI_RLN_t** rln, size_t index, const Vec_CFr_t* leaves,
                                   const Vec_size_t* indices);                                 /* ffi_tree.rs:147 */
CBoolResult_t ffi_seq_atomic_operation(FFI_RLN_t** rln, const Vec_CFr_t* leaves,
                                       const Vec_uint8_t* indices);                            /* ffi_tree.rs:170 */
CFr_t* ffi_get_root(FFI_RLN_t* const* rln);                                                    /* ffi_tree.rs:191 */
CResult_FFI_MerkleProof_ptr_Vec_uint8_t ffi_get_merkle_proof(FFI_RLN_t* const* rln, size_t index); /* ffi_tree.rs:196 */
void ffi_merkle_proof_free(FFI_MerkleProof_t* proof);                                          /* ffi_tree.rs:21 */
CBoolResult_t ffi_set_metadata(FFI_RLN_t** rln, const Vec_uint8_t* metadata);                  /* ffi_tree.rs:231 */
CResult_Vec_uint8_Vec_uint8_t ffi_get_metadata(FFI_RLN_t* const* rln);                         /* ffi_tree.rs:244 */
CBoolResult_t ffi_flush(FFI_RLN_t** rln);                                                      /* ffi_tree.rs:257 */

/* ---- field / vector helpers (ffi_utils.rs) --------------------------------------------------------- */
CFr_t* ffi_cfr_zero(void);                                                                     /* ffi_utils.rs:70 */
CFr_t* ffi_cfr_one(void);                                                                      /* ffi_utils.rs:75 */
CResult_Vec_uint8_Vec_uint8_t ffi_cfr_to_bytes_le(const CFr_t* cfr);                           /* ffi_utils.rs:80 */
CResult_Vec_uint8_Vec_uint8_t ffi_cfr_to_bytes_be(const CFr_t* cfr);                           /* ffi_utils.rs:95 */
CResult_CFr_ptr_Vec_uint8_t ffi_bytes_le_to_cfr(const Vec_uint8_t* bytes);                     /* ffi_utils.rs:110 */
CResult_CFr_ptr_Vec_uint8_t ffi_bytes_be_to_cfr(const Vec_uint8_t* bytes);                     /* ffi_utils.rs:124 */
CFr_t* ffi_uint_to_cfr(uint32_t value);                                                        /* ffi_utils.rs:138 */
Vec_uint8_t ffi_cfr_debug(const CFr_t* cfr);                                                   /* ffi_utils.rs:143 */
void ffi_cfr_free(CFr_t* cfr);                                                                 /* ffi_utils.rs:151 */
Vec_CFr_t ffi_vec_cfr_new(size_t capacity);                                                    /* ffi_utils.rs:158 */
Vec_CFr_t ffi_vec_cfr_from_cfr(const CFr_t* cfr);                                              /* ffi_utils.rs:163 */
void ffi_vec_cfr_push(Vec_CFr_t* v, const CFr_t* cfr);                                         /* ffi_utils.rs:168 */
size_t ffi_vec_cfr_len(const Vec_CFr_t* v);                                                    /* ffi_utils.rs:178 */
const CFr_t* ffi_vec_cfr_get(const Vec_CFr_t* v, size_t i);                                    /* ffi_utils.rs:183 */
CResult_Vec_uint8_Vec_uint8_t ffi_vec_cfr_to_bytes_le(const Vec_CFr_t* v);                     /* ffi_utils.rs:188 */
CResult_Vec_uint8_Vec_uint8_t ffi_vec_cfr_to_bytes_be(const Vec_CFr_t* v);                     /* ffi_utils.rs:204 */
CResult_Vec_CFr_Vec_uint8_t ffi_bytes_le_to_vec_cfr(const Vec_uint8_t* bytes);                 /* ffi_utils.rs:220 */
CResult_Vec_CFr_Vec_uint8_t ffi_bytes_be_to_vec_cfr(const Vec_uint8_t* bytes);                 /* ffi_utils.rs:240 */
Vec_uint8_t ffi_vec_cfr_debug(const Vec_CFr_t* v);                                             /* ffi_utils.rs:259 */
void ffi_vec_cfr_free(Vec_CFr_t v);                                                            /* ffi_utils.rs:270 */
CResult_Vec_uint8_Vec_uint8_t ffi_vec_u8_to_bytes_le(const Vec_uint8_t* v);                    /* ffi_utils.rs:277 */
CResult_Vec_uint8_Vec_uint8_t ffi_vec_u8_to_bytes_be(const Vec_uint8_t* v);                    /* ffi_utils.rs:292 */
CResult_Vec_uint8_Vec_uint8_t ffi_bytes_le_to_vec_u8(const Vec_uint8_t* bytes);                /* ffi_utils.rs:307 */
CResult_Vec_uint8_Vec_uint8_t ffi_bytes_be_to_vec_u8(const Vec_uint8_t* bytes);                /* ffi_utils.rs:321 */
Vec_uint8_t ffi_vec_u8_debug(const Vec_uint8_t* v);                                            /* ffi_utils.rs:335 */
void ffi_vec_u8_free(Vec_uint8_t v);                                                           /* ffi_utils.rs:343 */
CFr_t* ffi_hash_to_field_le(const Vec_uint8_t* input);                                         /* ffi_utils.rs:349 */
CFr_t* ffi_hash_to_field_be(const Vec_uint8_t* input);                                         /* ffi_utils.rs:354 */
/* The five functions below have no error channel in the reference ABI.  A device failure (no GPU, out of memory)
 * is reported on stderr and by a NULL pointer / an empty vector {NULL, 0, 0} -- never by a zero "result". */
CFr_t* ffi_poseidon_hash_pair(const CFr_t* a, const CFr_t* b);                                 /* ffi_utils.rs:359 */
Vec_CFr_t ffi_key_gen(void);                                                                   /* ffi_utils.rs:366 */
Vec_CFr_t ffi_seeded_key_gen(const Vec_uint8_t* seed);                                         /* ffi_utils.rs:372 */
Vec_CFr_t ffi_extended_key_gen(void);                                                          /* ffi_utils.rs:380 */
Vec_CFr_t ffi_seeded_extended_key_gen(const Vec_uint8_t* seed);                                /* ffi_utils.rs:392 */
void ffi_c_string_free(Vec_uint8_t s);                                                         /* ffi_utils.rs:407 */

/* ==== V3 mirror (rln/src/ffi/ffi_rln_v3.rs:323-1609) -- what rln/ffi_c_examples and rln/ffi_nim_examples call.
 * Same objects as above behind V3 names; differences: stateless / tree-flavour constructors, V3 error texts
 * (rln/src/error.rs:104-208), ffi_rln_v3_verify returning plain `false` on a signal mismatch, and the V3 wire
 * formats (protocol/serialize.rs): enum tag 0/1, values ordered y|root|nullifier|x|ext, proofs without a leading
 * version byte, "mixed" = LE proof + BE values; deserialisers accept trailing bytes.  All three tree flavours map
 * to the one device-resident tree; the pm-tree variant keeps no on-disk state. */
typedef struct FFI_RLNV3 FFI_RLNV3_t;                                   /* ffi_rln_v3.rs:310-312 */
typedef struct FFI_RLNV3WitnessInput FFI_RLNV3WitnessInput_t;           /* ffi_rln_v3.rs:612-614 */
typedef struct FFI_RLNV3PartialWitnessInput FFI_RLNV3PartialWitnessInput_t; /* ffi_rln_v3.rs:864-866 */
typedef struct FFI_RLNV3Proof FFI_RLNV3Proof_t;                         /* ffi_rln_v3.rs:1011-1013 */
typedef struct FFI_RLNV3PartialProof FFI_RLNV3PartialProof_t;           /* ffi_rln_v3.rs:1095-1097 */
typedef struct FFI_RLNV3ProofValues FFI_RLNV3ProofValues_t;             /* ffi_rln_v3.rs:1139-1141 */
typedef struct FFI_RLNV3MerkleProof { Vec_CFr_t path_elements; Vec_uint8_t path_index; } FFI_RLNV3MerkleProof_t; /* :1363-1368 */
typedef struct { FFI_RLNV3_t* ok; Vec_uint8_t err; } CResult_FFI_RLNV3_ptr_Vec_uint8_t;
typedef struct { FFI_RLNV3WitnessInput_t* ok; Vec_uint8_t err; } CResult_FFI_RLNV3WitnessInput_ptr_Vec_uint8_t;
typedef struct { FFI_RLNV3PartialWitnessInput_t* ok; Vec_uint8_t err; } CResult_FFI_RLNV3PartialWitnessInput_ptr_Vec_uint8_t;
typedef struct { FFI_RLNV3Proof_t* ok; Vec_uint8_t err; } CResult_FFI_RLNV3Proof_ptr_Vec_uint8_t;
typedef struct { FFI_RLNV3PartialProof_t* ok; Vec_uint8_t err; } CResult_FFI_RLNV3PartialProof_ptr_Vec_uint8_t;
typedef struct { FFI_RLNV3ProofValues_t* ok; Vec_uint8_t err; } CResult_FFI_RLNV3ProofValues_ptr_Vec_uint8_t;
typedef struct { FFI_RLNV3MerkleProof_t* ok; Vec_uint8_t err; } CResult_FFI_RLNV3MerkleProof_ptr_Vec_uint8_t;

FFI_RLNV3_t* ffi_rln_v3_new_stateless_default(void); /* ffi_rln_v3.rs:324 */
CResult_FFI_RLNV3_ptr_Vec_uint8_t ffi_rln_v3_new_stateless(const Vec_uint8_t* zkey_data, const Vec_uint8_t* graph_data); /* ffi_rln_v3.rs:330 */
FFI_RLNV3_t* ffi_rln_v3_new_with_full_merkle_tree_default(void); /* ffi_rln_v3.rs:350 */
CResult_FFI_RLNV3_ptr_Vec_uint8_t ffi_rln_v3_new_with_full_merkle_tree(size_t tree_depth, const Vec_uint8_t* zkey_data, const Vec_uint8_t* graph_data); /* ffi_rln_v3.rs:357 */
FFI_RLNV3_t* ffi_rln_v3_new_with_optimal_merkle_tree_default(void); /* ffi_rln_v3.rs:391 */
CResult_FFI_RLNV3_ptr_Vec_uint8_t ffi_rln_v3_new_with_optimal_merkle_tree(size_t tree_depth, const Vec_uint8_t* zkey_data, const Vec_uint8_t* graph_data); /* ffi_rln_v3.rs:399 */
FFI_RLNV3_t* ffi_rln_v3_new_with_pm_tree_default(void); /* ffi_rln_v3.rs:433 */
CResult_FFI_RLNV3_ptr_Vec_uint8_t ffi_rln_v3_new_with_pm_tree(size_t tree_depth, const Vec_uint8_t* zkey_data, const Vec_uint8_t* graph_data, const char* config_path); /* ffi_rln_v3.rs:440 */
void ffi_rln_v3_free(FFI_RLNV3_t* rln); /* ffi_rln_v3.rs:606 */
CResult_FFI_RLNV3Proof_ptr_Vec_uint8_t ffi_rln_v3_generate_proof(FFI_RLNV3_t* const* rln, FFI_RLNV3WitnessInput_t* const* witness); /* ffi_rln_v3.rs:507 */
CResult_FFI_RLNV3Proof_ptr_Vec_uint8_t ffi_rln_v3_generate_proof_with_rs(FFI_RLNV3_t* const* rln, FFI_RLNV3WitnessInput_t* const* witness, const CFr_t* r, const CFr_t* s); /* EXT */
CBoolResult_t ffi_rln_v3_verify(FFI_RLNV3_t* const* rln, FFI_RLNV3Proof_t* const* proof, const CFr_t* x); /* ffi_rln_v3.rs:524 */
CBoolResult_t ffi_rln_v3_verify_with_roots(FFI_RLNV3_t* const* rln, FFI_RLNV3Proof_t* const* proof, const Vec_CFr_t* roots, const CFr_t* x); /* ffi_rln_v3.rs:548 */
CResult_FFI_RLNV3PartialProof_ptr_Vec_uint8_t ffi_rln_v3_generate_partial_proof( FFI_RLNV3_t* const* rln, FFI_RLNV3PartialWitnessInput_t* const* partial_witness); /* ffi_rln_v3.rs:571 */
CResult_FFI_RLNV3Proof_ptr_Vec_uint8_t ffi_rln_v3_finish_proof(FFI_RLNV3_t* const* rln, FFI_RLNV3PartialProof_t* const* partial, FFI_RLNV3WitnessInput_t* const* witness); /* ffi_rln_v3.rs:588 */
CResult_FFI_RLNV3WitnessInput_ptr_Vec_uint8_t ffi_rln_v3_witness_input_new_single( const CFr_t* identity_secret, const CFr_t* user_message_limit, const CFr_t* message_id, const Vec_CFr_t* path_elements, const Vec_uint8_t* identity_path_index, const CFr_t* x, const CFr_t* external_nullifier); /* ffi_rln_v3.rs:617 */
CResult_FFI_RLNV3WitnessInput_ptr_Vec_uint8_t ffi_rln_v3_witness_input_new_multi( const CFr_t* identity_secret, const CFr_t* user_message_limit, const Vec_CFr_t* message_ids, const Vec_CFr_t* path_elements, const Vec_uint8_t* identity_path_index, const CFr_t* x, const CFr_t* external_nullifier, const Vec_bool_t* selector_used); /* ffi_rln_v3.rs:652 */
CFr_t* ffi_rln_v3_witness_input_get_identity_secret(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:691 */
CFr_t* ffi_rln_v3_witness_input_get_user_message_limit(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:698 */
CResult_CFr_ptr_Vec_uint8_t ffi_rln_v3_witness_input_get_message_id(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:705 */
CResult_Vec_CFr_Vec_uint8_t ffi_rln_v3_witness_input_get_message_ids(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:721 */
Vec_CFr_t ffi_rln_v3_witness_input_get_path_elements(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:742 */
Vec_uint8_t ffi_rln_v3_witness_input_get_identity_path_index(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:755 */
CFr_t* ffi_rln_v3_witness_input_get_x(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:762 */
CFr_t* ffi_rln_v3_witness_input_get_external_nullifier(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:769 */
CResult_Vec_bool_Vec_uint8_t ffi_rln_v3_witness_input_get_selector_used(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:776 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_witness_to_bytes_le(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:792 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_witness_to_bytes_be(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:809 */
CResult_FFI_RLNV3WitnessInput_ptr_Vec_uint8_t ffi_bytes_le_to_rln_v3_witness(const Vec_uint8_t* bytes); /* ffi_rln_v3.rs:826 */
CResult_FFI_RLNV3WitnessInput_ptr_Vec_uint8_t ffi_bytes_be_to_rln_v3_witness(const Vec_uint8_t* bytes); /* ffi_rln_v3.rs:842 */
void ffi_rln_v3_witness_input_free(FFI_RLNV3WitnessInput_t* w); /* ffi_rln_v3.rs:858 */
CResult_FFI_RLNV3PartialWitnessInput_ptr_Vec_uint8_t ffi_rln_v3_partial_witness_input_new( const CFr_t* identity_secret, const CFr_t* user_message_limit, const Vec_CFr_t* path_elements, const Vec_uint8_t* identity_path_index); /* ffi_rln_v3.rs:869 */
CFr_t* ffi_rln_v3_partial_witness_input_get_identity_secret(FFI_RLNV3PartialWitnessInput_t* const* w); /* ffi_rln_v3.rs:897 */
CFr_t* ffi_rln_v3_partial_witness_input_get_user_message_limit(FFI_RLNV3PartialWitnessInput_t* const* w); /* ffi_rln_v3.rs:904 */
Vec_CFr_t ffi_rln_v3_partial_witness_input_get_path_elements(FFI_RLNV3PartialWitnessInput_t* const* w); /* ffi_rln_v3.rs:911 */
Vec_uint8_t ffi_rln_v3_partial_witness_input_get_identity_path_index(FFI_RLNV3PartialWitnessInput_t* const* w); /* ffi_rln_v3.rs:924 */
FFI_RLNV3PartialWitnessInput_t* ffi_rln_v3_witness_to_partial_witness(FFI_RLNV3WitnessInput_t* const* w); /* ffi_rln_v3.rs:931 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_partial_witness_to_bytes_le(FFI_RLNV3PartialWitnessInput_t* const* w); /* ffi_rln_v3.rs:939 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_partial_witness_to_bytes_be(FFI_RLNV3PartialWitnessInput_t* const* w); /* ffi_rln_v3.rs:956 */
CResult_FFI_RLNV3PartialWitnessInput_ptr_Vec_uint8_t ffi_bytes_le_to_rln_v3_partial_witness(const Vec_uint8_t* bytes); /* ffi_rln_v3.rs:973 */
CResult_FFI_RLNV3PartialWitnessInput_ptr_Vec_uint8_t ffi_bytes_be_to_rln_v3_partial_witness(const Vec_uint8_t* bytes); /* ffi_rln_v3.rs:989 */
void ffi_rln_v3_partial_witness_input_free(FFI_RLNV3PartialWitnessInput_t* w); /* ffi_rln_v3.rs:1005 */
FFI_RLNV3ProofValues_t* ffi_rln_v3_proof_get_values(FFI_RLNV3Proof_t* const* proof); /* ffi_rln_v3.rs:1016 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_proof_to_bytes_le(FFI_RLNV3Proof_t* const* proof); /* ffi_rln_v3.rs:1023 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_proof_to_bytes_mixed(FFI_RLNV3Proof_t* const* proof); /* ffi_rln_v3.rs:1040 */
CResult_FFI_RLNV3Proof_ptr_Vec_uint8_t ffi_bytes_le_to_rln_v3_proof(const Vec_uint8_t* bytes); /* ffi_rln_v3.rs:1057 */
CResult_FFI_RLNV3Proof_ptr_Vec_uint8_t ffi_bytes_mixed_to_rln_v3_proof(const Vec_uint8_t* bytes); /* ffi_rln_v3.rs:1073 */
void ffi_rln_v3_proof_free(FFI_RLNV3Proof_t* proof); /* ffi_rln_v3.rs:1089 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_partial_proof_to_bytes_le(FFI_RLNV3PartialProof_t* const* partial); /* ffi_rln_v3.rs:1100 */
CResult_FFI_RLNV3PartialProof_ptr_Vec_uint8_t ffi_bytes_le_to_rln_v3_partial_proof(const Vec_uint8_t* bytes); /* ffi_rln_v3.rs:1117 */
void ffi_rln_v3_partial_proof_free(FFI_RLNV3PartialProof_t* partial); /* ffi_rln_v3.rs:1133 */
CFr_t* ffi_rln_v3_proof_values_get_root(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1144 */
CFr_t* ffi_rln_v3_proof_values_get_x(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1151 */
CFr_t* ffi_rln_v3_proof_values_get_external_nullifier(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1156 */
CResult_CFr_ptr_Vec_uint8_t ffi_rln_v3_proof_values_get_y(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1163 */
CResult_CFr_ptr_Vec_uint8_t ffi_rln_v3_proof_values_get_nullifier(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1179 */
CResult_Vec_bool_Vec_uint8_t ffi_rln_v3_proof_values_get_selector_used(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1195 */
CResult_Vec_CFr_Vec_uint8_t ffi_rln_v3_proof_values_get_ys(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1211 */
CResult_Vec_CFr_Vec_uint8_t ffi_rln_v3_proof_values_get_nullifiers(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1232 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_proof_values_to_bytes_le(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1253 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_proof_values_to_bytes_be(FFI_RLNV3ProofValues_t* const* pv); /* ffi_rln_v3.rs:1270 */
CResult_FFI_RLNV3ProofValues_ptr_Vec_uint8_t ffi_bytes_le_to_rln_v3_proof_values(const Vec_uint8_t* bytes); /* ffi_rln_v3.rs:1287 */
CResult_FFI_RLNV3ProofValues_ptr_Vec_uint8_t ffi_bytes_be_to_rln_v3_proof_values(const Vec_uint8_t* bytes); /* ffi_rln_v3.rs:1303 */
void ffi_rln_v3_proof_values_free(FFI_RLNV3ProofValues_t* pv); /* ffi_rln_v3.rs:1319 */
CResult_CFr_ptr_Vec_uint8_t ffi_rln_v3_compute_id_secret(const CFr_t* share1_x, const CFr_t* share1_y, const CFr_t* share2_x, const CFr_t* share2_y); /* ffi_rln_v3.rs:1324 */
CResult_CFr_ptr_Vec_uint8_t ffi_rln_v3_recover_id_secret(FFI_RLNV3ProofValues_t* const* pv1, FFI_RLNV3ProofValues_t* const* pv2); /* ffi_rln_v3.rs:1345 */
void ffi_rln_v3_merkle_proof_free(FFI_RLNV3MerkleProof_t* proof); /* ffi_rln_v3.rs:1371 */
CBoolResult_t ffi_rln_v3_delete_leaf(FFI_RLNV3_t** rln, size_t index); /* ffi_rln_v3.rs:1376 */
CBoolResult_t ffi_rln_v3_set_leaf(FFI_RLNV3_t** rln, size_t index, const CFr_t* leaf); /* ffi_rln_v3.rs:1390 */
CResult_CFr_ptr_Vec_uint8_t ffi_rln_v3_get_leaf(FFI_RLNV3_t* const* rln, size_t index); /* ffi_rln_v3.rs:1408 */
size_t ffi_rln_v3_leaves_set(FFI_RLNV3_t* const* rln); /* ffi_rln_v3.rs:1425 */
CBoolResult_t ffi_rln_v3_set_next_leaf(FFI_RLNV3_t** rln, const CFr_t* leaf); /* ffi_rln_v3.rs:1430 */
CBoolResult_t ffi_rln_v3_set_leaves_from(FFI_RLNV3_t** rln, size_t index, const Vec_CFr_t* leaves); /* ffi_rln_v3.rs:1444 */
CBoolResult_t ffi_rln_v3_init_tree_with_leaves(FFI_RLNV3_t** rln, const Vec_CFr_t* leaves); /* ffi_rln_v3.rs:1463 */
CBoolResult_t ffi_rln_v3_atomic_operation(FFI_RLNV3_t** rln, size_t index, const Vec_CFr_t* leaves, const Vec_size_t* indices); /* ffi_rln_v3.rs:1481 */
CBoolResult_t ffi_rln_v3_seq_atomic_operation(FFI_RLNV3_t** rln, const Vec_CFr_t* leaves, const Vec_uint8_t* indices); /* ffi_rln_v3.rs:1502 */
CFr_t* ffi_rln_v3_get_root(FFI_RLNV3_t* const* rln); /* ffi_rln_v3.rs:1531 */
CResult_FFI_RLNV3MerkleProof_ptr_Vec_uint8_t ffi_rln_v3_get_merkle_proof(FFI_RLNV3_t* const* rln, size_t index); /* ffi_rln_v3.rs:1537 */
CBoolResult_t ffi_rln_v3_set_metadata(FFI_RLNV3_t** rln, const Vec_uint8_t* metadata); /* ffi_rln_v3.rs:1565 */
CResult_Vec_uint8_Vec_uint8_t ffi_rln_v3_get_metadata(FFI_RLNV3_t* const* rln); /* ffi_rln_v3.rs:1582 */
CBoolResult_t ffi_rln_v3_flush(FFI_RLNV3_t** rln); /* ffi_rln_v3.rs:1598 */

#ifdef __cplusplus
}
#endif
#endif /* RLN_H */
