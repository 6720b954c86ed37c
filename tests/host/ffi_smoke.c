/* C caller of the zerokit-compatible ABI (include/rln.h), modelled on the flow of the reference's
 * rln/ffi_c_examples/basic_proof.c (V1 names): build against the header with a plain C compiler, link the
 * backend as `-lrln`, create an RLN object, register a member, prove, serialise, verify.  Exit code 0 on success.
 * Used by tests/test_gpu_ffi.py::test_c_program_links_and_proves. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rln.h"

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL: %s\n", msg); return 1; } } while (0)

int main(void) {
  CResult_FFI_RLN_ptr_Vec_uint8_t r = ffi_rln_new(20, "");
  if (!r.ok) { fprintf(stderr, "ffi_rln_new: %s\n", r.err.ptr); ffi_c_string_free(r.err); return 2; }
  FFI_RLN_t* rln = r.ok;
  CHECK(ffi_rln_get_tree_depth(&rln) == 20, "tree depth");

  /* identity: secret = hash_to_field("c-harness"), commitment = H(secret), leaf = H(commitment, limit) */
  uint8_t seed[] = "c-harness";
  Vec_uint8_t seedv = {seed, sizeof(seed) - 1, sizeof(seed) - 1};
  CFr_t* secret = ffi_hash_to_field_le(&seedv);
  CFr_t* limit = ffi_uint_to_cfr(100);
  CFr_t* zero = ffi_cfr_zero();
  Vec_CFr_t one_in = ffi_vec_cfr_from_cfr(secret);
  (void)one_in;
  /* commitment via the pair hash is not the 1-input hash; use key material from ffi_key_gen instead */
  Vec_CFr_t keys = ffi_key_gen();
  CHECK(ffi_vec_cfr_len(&keys) == 2, "key_gen returns (secret, commitment)");
  const CFr_t* id_secret = ffi_vec_cfr_get(&keys, 0);
  const CFr_t* id_commitment = ffi_vec_cfr_get(&keys, 1);
  CFr_t* rate_commitment = ffi_poseidon_hash_pair(id_commitment, limit);

  CBoolResult_t ok = ffi_set_leaf(&rln, 3, rate_commitment);
  CHECK(ok.ok && !ok.err.ptr, "set_leaf");
  CHECK(ffi_leaves_set(&rln) == 4, "leaves_set");
  CResult_FFI_MerkleProof_ptr_Vec_uint8_t mp = ffi_get_merkle_proof(&rln, 3);
  CHECK(mp.ok && mp.ok->path_elements.len == 20 && mp.ok->path_index.len == 20, "merkle proof shape");
  CHECK(mp.ok->path_index.ptr[0] == 1 && mp.ok->path_index.ptr[1] == 1 && mp.ok->path_index.ptr[2] == 0, "index bits");

  uint8_t sig[] = "hey hey";
  Vec_uint8_t sigv = {sig, sizeof(sig) - 1, sizeof(sig) - 1};
  CFr_t* x = ffi_hash_to_field_le(&sigv);
  CFr_t* ext = ffi_uint_to_cfr(424242);
  CFr_t* msg_id = ffi_uint_to_cfr(1);
  CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t w =
      ffi_rln_witness_input_new_single(id_secret, limit, msg_id, &mp.ok->path_elements, &mp.ok->path_index, x, ext);
  CHECK(w.ok, "witness");
  /* invalid witness: message id == limit */
  CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t bad =
      ffi_rln_witness_input_new_single(id_secret, limit, limit, &mp.ok->path_elements, &mp.ok->path_index, x, ext);
  CHECK(!bad.ok && bad.err.ptr && strstr((char*)bad.err.ptr, "not within user_message_limit"), "invalid witness error");
  ffi_c_string_free(bad.err);

  CResult_FFI_RLNProof_ptr_Vec_uint8_t p = ffi_generate_rln_proof(&rln, &w.ok);
  if (!p.ok) { fprintf(stderr, "prove: %s\n", p.err.ptr); return 3; }
  CResult_Vec_uint8_Vec_uint8_t bytes = ffi_rln_proof_to_bytes_le(&p.ok);
  CHECK(bytes.ok.ptr && bytes.ok.len == 290, "proof is 290 bytes");
  CResult_FFI_RLNProof_ptr_Vec_uint8_t p2 = ffi_bytes_le_to_rln_proof(&bytes.ok);
  CHECK(p2.ok, "proof round trip");

  CBoolResult_t v = ffi_verify_rln_proof(&rln, &p2.ok, x);
  if (!v.ok) { fprintf(stderr, "verify: %s\n", v.err.ptr ? (char*)v.err.ptr : "false"); return 4; }
  CBoolResult_t v2 = ffi_verify_rln_proof(&rln, &p2.ok, ext); /* wrong signal */
  CHECK(!v2.ok && v2.err.ptr && strstr((char*)v2.err.ptr, "Signal value does not match"), "wrong signal rejected");
  ffi_c_string_free(v2.err);

  FFI_RLNProofValues_t* pv = ffi_rln_proof_get_values(&p.ok);
  CFr_t* root = ffi_rln_proof_values_get_root(&pv);
  CFr_t* tree_root = ffi_get_root(&rln);
  CResult_Vec_uint8_Vec_uint8_t rb = ffi_cfr_to_bytes_le(root), tb = ffi_cfr_to_bytes_le(tree_root);
  CHECK(rb.ok.len == 32 && memcmp(rb.ok.ptr, tb.ok.ptr, 32) == 0, "proof root equals tree root");
  Vec_uint8_t dbg = ffi_cfr_debug(root);
  printf("root = %s\n", dbg.ptr);
  ffi_c_string_free(dbg);

  ffi_vec_u8_free(rb.ok); ffi_vec_u8_free(tb.ok); ffi_vec_u8_free(bytes.ok);
  ffi_cfr_free(root); ffi_cfr_free(tree_root); ffi_rln_proof_values_free(pv);
  ffi_rln_proof_free(p.ok); ffi_rln_proof_free(p2.ok); ffi_rln_witness_input_free(w.ok);
  ffi_merkle_proof_free(mp.ok); ffi_vec_cfr_free(keys); ffi_vec_cfr_free(one_in);
  ffi_cfr_free(secret); ffi_cfr_free(limit); ffi_cfr_free(zero); ffi_cfr_free(rate_commitment);
  ffi_cfr_free(x); ffi_cfr_free(ext); ffi_cfr_free(msg_id);
  ffi_rln_free(rln);
  printf("C harness: OK\n");
  return 0;
}
