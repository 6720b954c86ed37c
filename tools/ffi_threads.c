/* T threads of a plain C program share ONE RLN object and call ffi_generate_rln_proof in a loop (generate_rln_proof takes
 * &self, rln/src/public.rs:624): calls per second and milliseconds per call for T = 1 .. 64, every proof verified at the
 * end of a run's first and last call.  The library gathers the calls that arrive while a proof is on the device into one
 * batch (include/rln_amd.h: rlnamd_ffi_gather_stats); RLNAMD_GATHER_CALLS=0 in the environment gives the behaviour
 * before (one call at a time).  With the argument `finish` the threads finish the member's partial proof instead
 * (ffi_finish_rln_proof; its ms_proving_per_batch is not reported); a second argument is the object's config_path ("" for
 * none), a third the number of members to prove for in turn (1: one member, its chain of hints remembered).
 * One JSON line.
 *   gcc -O2 -std=c11 -I include tools/ffi_threads.c -L zerokit_amd/lib -lrln -lpthread -Wl,-rpath,$PWD/zerokit_amd/lib -o tools/ffi_threads */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rln.h"

extern int rlnamd_ffi_gather_stats(const void* ffi_rln, uint64_t out[8]);

static FFI_RLN_t* rln;
static const CFr_t *id_secret, *limit_c;
static CFr_t* ext;
static FFI_MerkleProof_t* mp;
/* argv[3] = M > 1: M members are registered and every call proves for another one ((thread, call) -> member), as a service
 * that proves for its users would: the chains of hints are not remembered from call to call */
#define MAX_MEMBERS 4096
static int n_members = 1;
static Vec_CFr_t member_keys[MAX_MEMBERS];
static FFI_MerkleProof_t* member_mp[MAX_MEMBERS];
static int calls_per_thread;
static int failures;
static FFI_RLNPartialProof_t* partial;   /* argv[1] = "finish": the threads finish this member's partial proof instead */

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static void* work(void* arg) {
  const long tid = (long)arg;
  for (int j = 0; j < calls_per_thread; j++) {
    CFr_t* x = ffi_uint_to_cfr((uint32_t)(1 + tid * 100000 + j));
    CFr_t* msg = ffi_uint_to_cfr((uint32_t)((tid + j) % 100));
    const int mem = n_members > 1 ? (int)((tid * 7919 + j * 104729L) % n_members) : 0;
    const CFr_t* sec = n_members > 1 ? ffi_vec_cfr_get(&member_keys[mem], 0) : id_secret;
    FFI_MerkleProof_t* path = n_members > 1 ? member_mp[mem] : mp;
    CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t w =
        ffi_rln_witness_input_new_single(sec, limit_c, msg, &path->path_elements, &path->path_index, x, ext);
    if (!w.ok) {
      __sync_fetch_and_add(&failures, 1);
      return NULL;
    }
    CResult_FFI_RLNProof_ptr_Vec_uint8_t p =
        partial ? ffi_finish_rln_proof(&rln, &partial, &w.ok) : ffi_generate_rln_proof(&rln, &w.ok);
    if (!p.ok) {
      __sync_fetch_and_add(&failures, 1);
      ffi_c_string_free(p.err);
    } else {
      if (j == 0 || j == calls_per_thread - 1) {
        CBoolResult_t v = ffi_verify_rln_proof(&rln, &p.ok, x);
        if (!v.ok) __sync_fetch_and_add(&failures, 1);
        if (v.err.ptr) ffi_c_string_free(v.err);
      }
      ffi_rln_proof_free(p.ok);
    }
    ffi_rln_witness_input_free(w.ok);
    ffi_cfr_free(x);
    ffi_cfr_free(msg);
  }
  return NULL;
}

int main(int argc, char** argv) {
  /* argv[2]: a config_path JSON for the object, e.g. one holding {"profile": "throughput"} */
  CResult_FFI_RLN_ptr_Vec_uint8_t r = ffi_rln_new(20, argc > 2 ? argv[2] : "");
  if (!r.ok) {
    fprintf(stderr, "ffi_rln_new: %s\n", r.err.ptr);
    return 2;
  }
  rln = r.ok;
  Vec_CFr_t keys = ffi_key_gen();
  id_secret = ffi_vec_cfr_get(&keys, 0);
  CFr_t* limit = ffi_uint_to_cfr(100);
  limit_c = limit;
  CFr_t* rate = ffi_poseidon_hash_pair(ffi_vec_cfr_get(&keys, 1), limit);
  CBoolResult_t ok = ffi_set_leaf(&rln, 7, rate);
  if (!ok.ok) return 3;
  CResult_FFI_MerkleProof_ptr_Vec_uint8_t m = ffi_get_merkle_proof(&rln, 7);
  if (!m.ok) return 4;
  mp = m.ok;
  ext = ffi_uint_to_cfr(424242);
  if (argc > 3 && atoi(argv[3]) > 1) {
    n_members = atoi(argv[3]) > MAX_MEMBERS ? MAX_MEMBERS : atoi(argv[3]);
    for (int i = 0; i < n_members; i++) {
      member_keys[i] = ffi_key_gen();
      CFr_t* rc = ffi_poseidon_hash_pair(ffi_vec_cfr_get(&member_keys[i], 1), limit);
      CBoolResult_t o = ffi_set_leaf(&rln, 100 + (size_t)i, rc);
      if (!o.ok) return 7;
      ffi_cfr_free(rc);
    }
    for (int i = 0; i < n_members; i++) {
      CResult_FFI_MerkleProof_ptr_Vec_uint8_t q = ffi_get_merkle_proof(&rln, 100 + (size_t)i);
      if (!q.ok) return 8;
      member_mp[i] = q.ok;
    }
  }
  if (argc > 1 && strcmp(argv[1], "finish") == 0) {
    CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t pw =
        ffi_rln_partial_witness_input_new(id_secret, limit_c, &mp->path_elements, &mp->path_index);
    if (!pw.ok) return 5;
    CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t pp = ffi_generate_partial_zk_proof(&rln, &pw.ok);
    if (!pp.ok) return 6;
    partial = pp.ok;
    ffi_rln_partial_witness_input_free(pw.ok);
  }
  calls_per_thread = 4;
  work((void*)99);   /* warm */
  const int ts[] = {1, 2, 4, 8, 16, 32, 64};
  printf("{\"calls\": \"%s\", \"members\": %d, \"gather_calls_env\": \"%s\", \"threads\": {", partial ? "ffi_finish_rln_proof" : "ffi_generate_rln_proof", n_members,
         getenv("RLNAMD_GATHER_CALLS") ? getenv("RLNAMD_GATHER_CALLS") : "");
  for (unsigned k = 0; k < sizeof ts / sizeof ts[0]; k++) {
    const int T = ts[k];
    calls_per_thread = T <= 8 ? 300 : 100;
    pthread_t th[64];
    uint64_t g0[8] = {0};
    rlnamd_ffi_gather_stats(rln, g0);
    const double t0 = now_s();
    for (long t = 0; t < T; t++) pthread_create(&th[t], NULL, work, (void*)t);
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    const double dt = now_s() - t0;
    uint64_t g1[8] = {0};
    rlnamd_ffi_gather_stats(rln, g1);
    const int bi = partial ? 6 : 0, ci = partial ? 7 : 1;
    const double nb = (double)(g1[bi] - g0[bi]);
    printf("%s\"%d\": {\"calls_per_s\": %.1f, \"ms_per_call\": %.3f, \"calls_per_batch\": %.2f, \"ms_proving_per_batch\": %.3f, "
           "\"ms_between_batches\": %.3f}",
           k ? ", " : "", T, T * calls_per_thread / dt, dt / calls_per_thread * 1e3, nb ? (g1[ci] - g0[ci]) / nb : 0.0,
           nb ? (g1[5] - g0[5]) * 1e-6 / nb : 0.0, nb ? (dt * 1e3 - (g1[5] - g0[5]) * 1e-6) / nb : 0.0);
  }
  uint64_t st[8] = {0};
  rlnamd_ffi_gather_stats(rln, st);
  printf("}, \"gather_stats\": {\"batches\": %llu, \"calls\": %llu, \"largest\": %llu, \"cap\": %llu, \"waited\": %llu}, "
         "\"failures\": %d}\n",
         (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3],
         (unsigned long long)st[4], failures);
  if (partial) ffi_rln_partial_proof_free(partial);
  ffi_merkle_proof_free(mp);
  ffi_vec_cfr_free(keys);
  ffi_cfr_free(limit);
  ffi_cfr_free(rate);
  ffi_cfr_free(ext);
  ffi_rln_free(rln);
  return failures ? 1 : 0;
}
