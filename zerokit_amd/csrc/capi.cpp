// C ABI of the extension surface declared in include/rln_amd.h.
#include "../../include/rln_amd.h"

#include <string.h>

#include <algorithm>
#include <atomic>
#include <system_error>
#include <memory>
#include <mutex>
#include <thread>
#include <string>

#include "common.h"
#include "keccak.h"
#include "merkle.h"
#include "msm.h"
#include "pairing.h"
#include "poseidon.h"
#include "prover.h"
#include "capi_util.h"

using namespace rlnamd;

namespace rlnamd {
thread_local std::string g_last_error;
int fail(const std::exception& e) {
  g_last_error = e.what();
  return g_last_error.find("no HIP device") != std::string::npos ? RLNAMD_ERR_NO_DEVICE : RLNAMD_ERR;
}
}  // namespace rlnamd

struct rlnamd_tree {
  // every rlnamd_tree_* call on one handle takes this: the tree has ONE stream and shared scratch (proof_host's
  // staging buffer, the host chain's pinned buffers), so readers mutate object state too
  std::mutex mu;
  MerkleTreeDev t;
  size_t host_max = 0;   // rlnamd_tree_set_leaves: up to this many distinct leaves take MerkleTreeDev::set_few
  DevBuf<uint8_t> bench_elems, bench_bits;
};

static void verify_common(const Zkey& zk, const uint8_t proof[128], const uint8_t* values_le, int* ok, size_t nv = 5);

extern "C" {

const char* rlnamd_last_error(void) { return rlnamd::g_last_error.c_str(); }

int rlnamd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rlnamd_set_device(int ordinal) {
  RLN_TRY
  RLN_HIP(hipSetDevice(ordinal));
  RLN_CATCH
}

int rlnamd_get_device(int* ordinal) {
  RLN_TRY
  RLN_HIP(hipGetDevice(ordinal));
  RLN_CATCH
}

int rlnamd_device_name(char* buf, size_t cap) {
  RLN_TRY
  require_gpu();
  int dev = 0;
  RLN_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  RLN_HIP(hipGetDeviceProperties(&prop, dev));
  std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
  if (cap) {
    strncpy(buf, s.c_str(), cap - 1);
    buf[cap - 1] = 0;
  }
  RLN_CATCH
}

int rlnamd_poseidon_hash(const uint8_t* inputs_le, size_t n, size_t arity, uint8_t* out_le) {
  RLN_TRY
  require_gpu();
  // PoseidonError texts of utils/src/poseidon/error.rs:4-8; widths t = 2..9 (rln/src/hashers.rs:14-23)
  if (arity == 0) throw Error("Empty input provided");
  if (arity > 8) throw Error("No parameters found for input length " + std::to_string(arity));
  if (n == 0) return RLNAMD_OK;
  for (size_t i = 0; i < n * arity; i++) {
    uint32_t tmp[8];
    memcpy(tmp, inputs_le + 32 * i, 32);
    if (limbs_geq(tmp, FrParams::MOD)) throw Error("Non-canonical field element: value is not in [0, r-1]");
  }
  DevBuf<uint8_t> din(n * arity * 32), dout(n * 32);
  RLN_HIP(hipMemcpy(din.p, inputs_le, n * arity * 32, hipMemcpyHostToDevice));
  poseidon_hash_batch_device(din.p, n, (int)arity, dout.p, 0);
  RLN_HIP(hipMemcpy(out_le, dout.p, n * 32, hipMemcpyDeviceToHost));
  RLN_CATCH
}

int rlnamd_hash_to_field_le(const uint8_t* data, size_t len, uint8_t out_le[32]) {
  RLN_TRY
  hash_to_field_le(data, len, out_le);
  RLN_CATCH
}
int rlnamd_hash_to_field_be(const uint8_t* data, size_t len, uint8_t out_le[32]) {
  RLN_TRY
  hash_to_field_be(data, len, out_le);
  RLN_CATCH
}

// ---------------------------------------------------------------------------------------------- tree
int rlnamd_tree_new(size_t depth, rlnamd_tree** out) {
  RLN_TRY
  std::unique_ptr<rlnamd_tree> h(new rlnamd_tree);
  uint8_t zero[32] = {0};
  if (depth > 30) throw Error("InvalidDepth");
  h->t.init((int)depth, zero);
  h->host_max = MerkleTreeDev::host_max_from_env();
  *out = h.release();
  RLN_CATCH
}
void rlnamd_tree_free(rlnamd_tree* t) { delete t; }

int rlnamd_tree_set_range(rlnamd_tree* t, size_t start, const uint8_t* leaves_le, size_t n) {
  RLN_TRY
  std::lock_guard<std::mutex> tree_lk(t->mu);
  for (size_t i = 0; i < n; i++) {
    uint32_t tmp[8];
    memcpy(tmp, leaves_le + 32 * i, 32);
    if (limbs_geq(tmp, FrParams::MOD)) throw Error("field element is not canonical (>= modulus)");
  }
  t->t.set_range_host(start, leaves_le, n);
  RLN_CATCH
}
int rlnamd_tree_set_leaves(rlnamd_tree* t, const uint64_t* indices, const uint8_t* leaves_le, size_t k) {
  RLN_TRY
  std::lock_guard<std::mutex> tree_lk(t->mu);
  // any order, later entries win: brought to the strictly increasing form the pass wants
  std::vector<std::pair<uint64_t, size_t>> ord(k);
  for (size_t i = 0; i < k; i++) {
    uint32_t tmp[8];
    memcpy(tmp, leaves_le + 32 * i, 32);
    if (limbs_geq(tmp, FrParams::MOD)) throw Error("field element is not canonical (>= modulus)");
    if (indices[i] >= t->t.capacity()) throw Error("TooManySet");
    ord[i] = {indices[i], i};
  }
  std::stable_sort(ord.begin(), ord.end(), [](const std::pair<uint64_t, size_t>& a, const std::pair<uint64_t, size_t>& b) { return a.first < b.first; });
  std::vector<uint64_t> idx;
  std::vector<uint8_t> leaves;
  for (size_t i = 0; i < k; i++) {
    if (i + 1 < k && ord[i + 1].first == ord[i].first) continue;   // the last write to an index wins
    idx.push_back(ord[i].first);
    leaves.insert(leaves.end(), leaves_le + 32 * ord[i].second, leaves_le + 32 * ord[i].second + 32);
  }
  if (idx.size() <= t->host_max) t->t.set_few(idx.data(), leaves.data(), idx.size());
  else t->t.set_scattered(idx.data(), leaves.data(), idx.size());
  RLN_HIP(hipStreamSynchronize(t->t.stream));
  RLN_CATCH
}
int rlnamd_tree_root(rlnamd_tree* t, uint8_t out_le[32]) {
  RLN_TRY
  std::lock_guard<std::mutex> tree_lk(t->mu);
  t->t.get_node_host(0, out_le);
  RLN_CATCH
}
int rlnamd_tree_get_leaf(rlnamd_tree* t, size_t index, uint8_t out_le[32]) {
  RLN_TRY
  std::lock_guard<std::mutex> tree_lk(t->mu);
  if (index >= t->t.capacity()) throw Error("InvalidLeaf");
  t->t.get_node_host(t->t.capacity() - 1 + index, out_le);
  RLN_CATCH
}
int rlnamd_tree_proof(rlnamd_tree* t, size_t index, uint8_t* elems_le, uint8_t* bits) {
  RLN_TRY
  std::lock_guard<std::mutex> tree_lk(t->mu);
  t->t.proof_host(index, elems_le, bits);
  RLN_CATCH
}
int rlnamd_tree_proofs(rlnamd_tree* t, size_t first, size_t count, uint8_t* elems_le, uint8_t* bits) {
  RLN_TRY
  std::lock_guard<std::mutex> tree_lk(t->mu);
  if (count == 0 || t->t.depth == 0) return RLNAMD_OK;
  size_t d = t->t.depth;
  DevBuf<uint8_t> e(count * d * 32), b(count * d);
  t->t.proofs_device(first, count, e.p, b.p);
  RLN_HIP(hipMemcpyAsync(elems_le, e.p, count * d * 32, hipMemcpyDeviceToHost, t->t.stream));
  RLN_HIP(hipMemcpyAsync(bits, b.p, count * d, hipMemcpyDeviceToHost, t->t.stream));
  RLN_HIP(hipStreamSynchronize(t->t.stream));
  RLN_CATCH
}
int rlnamd_tree_fill_sequential(rlnamd_tree* t, size_t start, size_t n, uint64_t first_value) {
  RLN_TRY
  std::lock_guard<std::mutex> tree_lk(t->mu);
  t->t.fill_sequential_device(start, n, first_value);
  RLN_HIP(hipStreamSynchronize(t->t.stream));
  RLN_CATCH
}
int rlnamd_tree_bench(rlnamd_tree* t, size_t n_leaves, uint64_t first_value, int verify, float ms[2], size_t* bad) {
  RLN_TRY
  std::lock_guard<std::mutex> tree_lk(t->mu);
  MerkleTreeDev& T = t->t;
  if (n_leaves > T.capacity()) throw Error("TooManySet");
  size_t d = T.depth;
  if (t->bench_elems.n < n_leaves * d * 32) t->bench_elems.alloc(n_leaves * d * 32);
  if (t->bench_bits.n < n_leaves * d) t->bench_bits.alloc(n_leaves * d);
  hipEvent_t e0, e1, e2;
  RLN_HIP(hipEventCreate(&e0));
  RLN_HIP(hipEventCreate(&e1));
  RLN_HIP(hipEventCreate(&e2));
  RLN_HIP(hipEventRecord(e0, T.stream));
  T.fill_sequential_device(0, n_leaves, first_value);
  RLN_HIP(hipEventRecord(e1, T.stream));
  T.proofs_device(0, n_leaves, t->bench_elems.p, t->bench_bits.p);
  RLN_HIP(hipEventRecord(e2, T.stream));
  RLN_HIP(hipStreamSynchronize(T.stream));
  RLN_HIP(hipEventElapsedTime(&ms[0], e0, e1));
  RLN_HIP(hipEventElapsedTime(&ms[1], e1, e2));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipEventDestroy(e2);
  if (bad) *bad = verify ? T.verify_proofs_device(0, n_leaves, t->bench_elems.p, t->bench_bits.p) : 0;
  RLN_CATCH
}

// -------------------------------------------------------------------------------------------- prover
int rlnamd_prover_new(const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len, size_t max_batch,
                      int window_bits, rlnamd_prover** out) {
  RLN_TRY
  ProverConfig cfg;
  cfg.max_batch = max_batch ? max_batch : 64;
  cfg.window_bits = window_bits;
  std::unique_ptr<rlnamd_prover> h(new rlnamd_prover);
  h->p.reset(new Prover(zkey, zkey_len, graph, graph_len, cfg));
  *out = h.release();
  RLN_CATCH
}
void rlnamd_prover_free(rlnamd_prover* p) { delete p; }

int rlnamd_prover_get_info(rlnamd_prover* p, rlnamd_prover_info* info) {
  RLN_TRY
  rlnamd::fill_prover_info(*p->p, info);
  RLN_CATCH
}
}  // extern "C"
namespace rlnamd {
void fill_prover_info(const Prover& P, rlnamd_prover_info* info) {
  info->inputs_size = P.inputs_per_proof();
  info->num_signals = P.graph().signals.size();
  uint64_t dom = 1;
  while (dom < P.zkey().num_constraints + P.zkey().num_instance_variables) dom <<= 1;
  info->domain_size = dom;
  info->tree_depth = P.graph().tree_depth;
  info->max_out = P.graph().max_out;
  info->capacity = P.capacity();
  info->table_bytes = P.table_bytes();
  info->window_bits = P.window_bits();
  info->windows = P.windows();
  info->window_bits_g2 = P.window_bits_g2();
  info->windows_g2 = P.windows_g2();
  info->glv = P.glv() ? 1 : 0;
  info->reserved = 0;
  info->g1_rows = P.g1_rows();
  info->g2_rows = P.g2_rows();
}
}  // namespace rlnamd
extern "C" {
int rlnamd_prover_input_slot(rlnamd_prover* p, const char* name, uint32_t* offset, uint32_t* len) {
  RLN_TRY
  auto& m = p->p->graph().input_mapping;
  auto it = m.find(name);
  if (it == m.end()) throw Error(std::string("MissingInput: ") + name);
  *offset = it->second.first;
  *len = it->second.second;
  RLN_CATCH
}
int rlnamd_prover_upload(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le) {
  RLN_TRY
  p->p->upload(n, inputs_le, rs_le);
  RLN_CATCH
}
int rlnamd_prover_run(rlnamd_prover* p, size_t n) {
  RLN_TRY
  p->p->run(n);
  RLN_CATCH
}
int rlnamd_prover_run_async(rlnamd_prover* p, size_t n) {
  RLN_TRY
  p->p->run_async(n);
  RLN_CATCH
}
int rlnamd_prover_sync(rlnamd_prover* p) {
  RLN_TRY
  p->p->sync();
  RLN_CATCH
}
int rlnamd_prover_slots(rlnamd_prover* p) { return p->p->slots(); }
int rlnamd_prover_submit(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le, int mode,
                         const uint8_t* partial320, uint64_t* ticket) {
  RLN_TRY
  *ticket = p->p->submit(n, inputs_le, rs_le, mode, partial320);
  RLN_CATCH
}
int rlnamd_prover_collect(rlnamd_prover* p, uint64_t ticket, size_t n, uint8_t* proofs, uint8_t* coords, uint8_t* values,
                          uint32_t* errors, uint8_t* partial320) {
  RLN_TRY
  p->p->collect(ticket, n, proofs, values, errors, coords, partial320);
  RLN_CATCH
}
uint32_t rlnamd_prover_hint_words(rlnamd_prover* p) { return p->p->hint_words(); }
int rlnamd_prover_hints_for(rlnamd_prover* p, const uint8_t* inputs_le, uint32_t* hints) {
  RLN_TRY
  p->p->hints_for(inputs_le, hints);
  RLN_CATCH
}
int rlnamd_prover_submit_hinted(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le,
                                const uint32_t* hints, uint64_t* ticket) {
  RLN_TRY
  *ticket = p->p->submit_hinted(n, inputs_le, rs_le, hints);
  RLN_CATCH
}
int rlnamd_prover_hint_stats(rlnamd_prover* p, uint64_t out[7]) {
  RLN_TRY
  static_assert(Prover::HINT_STATS_FIELDS == 7, "rln_amd.h states seven fields");
  p->p->hint_stats(out);
  RLN_CATCH
}
int rlnamd_prover_device_shared(rlnamd_prover* p, int* who) {
  RLN_TRY
  *who = p->p->device_shared();
  RLN_CATCH
}
int rlnamd_prover_collect_partial_cached(rlnamd_prover* p, uint64_t ticket, size_t n, uint8_t* partial320, uint64_t* handles,
                                         uint32_t* errors) {
  RLN_TRY
  p->p->collect_partial_cached(ticket, n, partial320, handles, errors);
  RLN_CATCH
}
int rlnamd_prover_submit_finish(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le,
                                const uint8_t* partial320, const uint64_t* handles, uint64_t* ticket) {
  RLN_TRY
  *ticket = p->p->submit_finish(n, inputs_le, rs_le, partial320, handles);
  RLN_CATCH
}
int rlnamd_prover_release_partial(rlnamd_prover* p, const uint64_t* handles, size_t n) {
  RLN_TRY
  p->p->release_partial(handles, n);
  RLN_CATCH
}
int rlnamd_prover_partial_cache_info(rlnamd_prover* p, uint64_t out[8]) {
  RLN_TRY
  static_assert(Prover::PARTIAL_CACHE_FIELDS == 8, "rln_amd.h states eight fields");
  p->p->partial_cache_info(out);
  RLN_CATCH
}
int rlnamd_prover_collect_public(rlnamd_prover* p, uint64_t ticket, size_t n, uint8_t* out_le) {
  RLN_TRY
  std::vector<uint8_t> v;
  p->p->collect_public(ticket, n, &v);
  memcpy(out_le, v.data(), v.size());
  RLN_CATCH
}
int rlnamd_prover_prove_stream(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le,
                               uint8_t* proofs, uint8_t* values, uint32_t* errors) {
  RLN_TRY
  p->p->prove_stream(n, inputs_le, rs_le, proofs, values, errors);
  RLN_CATCH
}
int rlnamd_prover_run_mode(rlnamd_prover* p, size_t n, int mode) {
  RLN_TRY
  p->p->run(n, mode);
  RLN_CATCH
}
int rlnamd_prover_run_async_mode(rlnamd_prover* p, size_t n, int mode) {
  RLN_TRY
  p->p->run_async(n, mode);
  RLN_CATCH
}
int rlnamd_prover_upload_partial(rlnamd_prover* p, size_t n, const uint8_t* coords320) {
  RLN_TRY
  p->p->upload_partial(n, coords320);
  RLN_CATCH
}
int rlnamd_prover_upload_witness(rlnamd_prover* p, size_t n, const uint8_t* witness_le) {
  RLN_TRY
  p->p->upload_witness(n, witness_le);
  RLN_CATCH
}
int rlnamd_prover_download_partial(rlnamd_prover* p, size_t n, uint8_t* coords320) {
  RLN_TRY
  p->p->download_partial(n, coords320);
  RLN_CATCH
}
int rlnamd_prover_known_mask(rlnamd_prover* p, uint8_t* out) {
  RLN_TRY
  const std::vector<uint8_t>& m = p->p->known_mask();
  memcpy(out, m.data(), m.size());
  RLN_CATCH
}
int rlnamd_prover_download(rlnamd_prover* p, size_t n, uint8_t* proofs, uint8_t* coords, uint8_t* values,
                           uint32_t* errors) {
  RLN_TRY
  std::vector<ProofOut> out(n);
  p->p->download(n, out.data());
  for (size_t i = 0; i < n; i++) {
    if (proofs) memcpy(proofs + i * 128, out[i].compressed, 128);
    if (coords) memcpy(coords + i * 256, out[i].coords, 256);
    if (values) memcpy(values + i * 160, out[i].values, 160);
    if (errors) errors[i] = out[i].error;
  }
  RLN_CATCH
}
int rlnamd_prover_stage_ms(rlnamd_prover* p, float ms[RLNAMD_PROVER_STAGES]) {
  RLN_TRY
  p->p->stage_ms(ms);
  RLN_CATCH
}
int rlnamd_prover_walk_clock_mhz(rlnamd_prover* p, double mhz[2]) {
  RLN_TRY
  p->p->walk_clock_mhz(mhz);
  RLN_CATCH
}
const char* rlnamd_prover_stage_name(int i) { return (i >= 0 && i < PROVER_STAGES) ? kProverStageNames[i] : ""; }

int rlnamd_prover_describe(rlnamd_prover* p, char* buf, size_t cap) {
  RLN_TRY
  std::string d = p->p->tuning().describe();
  if (cap) {
    strncpy(buf, d.c_str(), cap - 1);
    buf[cap - 1] = 0;
  }
  RLN_CATCH
}
int rlnamd_prover_init_ms(rlnamd_prover* p, float ms[4]) {
  RLN_TRY
  for (int k = 0; k < 4; k++) ms[k] = p->p->init_ms()[k];
  RLN_CATCH
}
int rlnamd_prover_wipe(rlnamd_prover* p) {
  RLN_TRY
  p->p->wipe(0);
  RLN_CATCH
}
int rlnamd_prover_fetch_witness(rlnamd_prover* p, size_t index, uint8_t* out_le) {
  RLN_TRY
  std::vector<uint8_t> w;
  p->p->fetch_witness(index, &w);
  memcpy(out_le, w.data(), w.size());
  RLN_CATCH
}
int rlnamd_prover_fetch_h(rlnamd_prover* p, size_t index, uint8_t* out_le) {
  RLN_TRY
  std::vector<uint8_t> h;
  p->p->fetch_h(index, &h);
  memcpy(out_le, h.data(), h.size());
  RLN_CATCH
}
int rlnamd_prover_residue(rlnamd_prover* p, uint64_t out[6]) {
  RLN_TRY
  static_assert(Prover::RESIDUE_FIELDS == 6, "rln_amd.h states six counters");
  p->p->residue(out);
  RLN_CATCH
}

int rlnamd_verify(rlnamd_prover* p, const uint8_t proof[128], const uint8_t values_le[160], int* ok) {
  RLN_TRY
  verify_common(p->p->zkey(), proof, values_le, ok);
  RLN_CATCH
}

int rlnamd_msm_new(size_t capacity, rlnamd_msm** out) {
  RLN_TRY
  std::unique_ptr<rlnamd_msm> h(new rlnamd_msm);
  h->m.reset(new MsmG1(capacity ? capacity : 1));
  *out = h.release();
  RLN_CATCH
}
int rlnamd_msm_new_g2(size_t capacity, rlnamd_msm** out) {
  RLN_TRY
  std::unique_ptr<rlnamd_msm> h(new rlnamd_msm);
  h->m2.reset(new MsmG2(capacity ? capacity : 1));
  *out = h.release();
  RLN_CATCH
}
void rlnamd_msm_free(rlnamd_msm* m) { delete m; }
size_t rlnamd_msm_point_bytes(rlnamd_msm* m) { return m->m2 ? MsmG2::POINT_BYTES : MsmG1::POINT_BYTES; }
size_t rlnamd_msm_window_sums_bytes_of(rlnamd_msm* m) { return m->m2 ? MsmG2::window_sums_bytes() : MsmG1::window_sums_bytes(); }
int rlnamd_poseidon_params_check(const uint8_t* inputs_le, size_t arity, uint8_t out_dense_le[32], uint8_t out_sparse_le[32]) {
  RLN_TRY
  if (arity == 0) throw Error("Empty input provided");
  if (arity > 8) throw Error("No parameters found for input length " + std::to_string(arity));
  PoseidonParams P = poseidon_derive_params((int)arity + 1);
  std::vector<Fr> in(arity);
  for (size_t i = 0; i < arity; i++) {
    uint32_t tmp[8];
    memcpy(tmp, inputs_le + 32 * i, 32);
    if (limbs_geq(tmp, FrParams::MOD)) throw Error("Non-canonical field element: value is not in [0, r-1]");
    in[i] = Fr::from_canonical(tmp);
  }
  Fr d, sp;
  poseidon_params_eval_host(P, in.data(), &d, &sp);
  // the host hash of the tree's single-path chain (MerkleTreeDev::set_few) is that sparse form with fixed-size state
  if (poseidon_hash_host(poseidon_host_params((int)arity + 1), in.data()) != sp)
    throw Error("internal: poseidon_hash_host differs from the sparse evaluation");
  uint32_t c[8];
  d.to_canonical(c);
  memcpy(out_dense_le, c, 32);
  sp.to_canonical(c);
  memcpy(out_sparse_le, c, 32);
  RLN_CATCH
}
int rlnamd_selftest_fq29(int group, uint32_t threads, uint32_t iters, const uint8_t* g2_gen_xy_le, uint32_t* mismatches) {
  RLN_TRY
  *mismatches = selftest_fq29(group, threads, iters, g2_gen_xy_le);
  RLN_CATCH
}
int rlnamd_msm_set(rlnamd_msm* m, const uint8_t* points_xy_le, const uint8_t* scalars_le, size_t n) {
  RLN_TRY
  if (m->m2) m->m2->set_host(points_xy_le, scalars_le, n);
  else m->m->set_host(points_xy_le, scalars_le, n);
  RLN_CATCH
}
int rlnamd_msm_generate(rlnamd_msm* m, uint64_t seed, uint64_t first_index, size_t n) {
  RLN_TRY
  if (m->m2) m->m2->generate(seed, first_index, n);
  else m->m->generate(seed, first_index, n);
  RLN_CATCH
}
int rlnamd_msm_generate_mode(rlnamd_msm* m, uint64_t seed, uint64_t first_index, size_t n, uint32_t mode) {
  RLN_TRY
  if (mode > 3) throw Error("rlnamd_msm_generate_mode: unknown mode bits");
  if (m->m2) m->m2->generate(seed, first_index, n, mode);
  else m->m->generate(seed, first_index, n, mode);
  RLN_CATCH
}
int rlnamd_msm_fetch(rlnamd_msm* m, size_t first, size_t count, uint8_t* points_xy_le, uint8_t* scalars_le) {
  RLN_TRY
  if (m->m2) m->m2->fetch(first, count, points_xy_le, scalars_le);
  else m->m->fetch(first, count, points_xy_le, scalars_le);
  RLN_CATCH
}
size_t rlnamd_msm_window_sums_bytes(void) { return MsmG1::window_sums_bytes(); }
int rlnamd_msm_run(rlnamd_msm* m, uint8_t* window_sums, float ms[3]) {
  RLN_TRY
  if (m->m2) m->m2->run_windows(window_sums, ms);
  else m->m->run_windows(window_sums, ms);
  RLN_CATCH
}
int rlnamd_msm_run_sharded(rlnamd_msm* m, rlnamd_comm* c, uint8_t out_xy_le[64], float ms[4]) {
  RLN_TRY
  if (!c) throw Error("rlnamd_msm_run_sharded: no communicator");
  if (m->m2) m->m2->run_sharded(rlnamd_comm_handle(c), rlnamd_comm_size(c), out_xy_le, ms);
  else m->m->run_sharded(rlnamd_comm_handle(c), rlnamd_comm_size(c), out_xy_le, ms);
  RLN_CATCH
}
int rlnamd_msm_combine(rlnamd_msm* m, const uint8_t* window_sums, size_t contributors, uint8_t out_xy_le[64]) {
  RLN_TRY
  if (m->m2) m->m2->combine(window_sums, contributors, out_xy_le);
  else m->m->combine(window_sums, contributors, out_xy_le);
  RLN_CATCH
}

static void verify_common(const Zkey& zk, const uint8_t proof[128], const uint8_t* values_le, int* ok, size_t nv) {
  G1Affine A, C;
  G2Affine B;
  *ok = 0;
  // Validate::Yes of the reference's proof deserialiser: on the curve AND, for G2, in the order-r subgroup
  if (!g1_decompress(proof, &A) || !g2_decompress(proof + 32, &B) || !g1_decompress(proof + 96, &C) ||
      !g2_in_subgroup(B))
    throw Error("Proof serialization error: the input buffer contained invalid data");
  std::vector<Fr> in;
  for (size_t i = 0; i < nv; i++) {
    uint32_t c[8];
    memcpy(c, values_le + 32 * i, 32);
    if (limbs_geq(c, FrParams::MOD)) throw Error("Non-canonical field element: value is not in [0, r-1]");
    in.push_back(Fr::from_canonical(c));
  }
  *ok = groth16_verify(zk, A, B, C, in) ? 1 : 0;
}

// n independent verifications spread over host threads (verification stays on the CPU; a relay node
// verifies every message it forwards).  ok[i] = 1 valid, 0 invalid or malformed.
}  // extern "C"
namespace rlnamd {
void verify_many_common(const Zkey& zk, size_t n, const uint8_t* proofs, const uint8_t* values_le, size_t nv,
                        int threads, uint8_t* ok) {
  if (nv + 1 != zk.gamma_abc_g1.size()) throw Error("MalformedVerifyingKey");
  (void)prepared(zk);  // once, before the workers only read it
  unsigned hw = std::thread::hardware_concurrency();
  if (!hw) hw = 1;
  size_t nt = threads > 0 ? (size_t)threads : hw;
  nt = std::max<size_t>(1, std::min({nt, n, (size_t)4 * hw}));   // more threads than that only cost memory
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
      int v = 0;
      try {
        verify_common(zk, proofs + 128 * i, values_le + 32 * nv * i, &v, nv);
      } catch (const std::exception&) {
        v = 0;
      }
      ok[i] = (uint8_t)v;
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(nt);
  for (size_t t = 1; t < nt; t++) {
    try {
      pool.emplace_back(work);
    } catch (const std::system_error&) {
      break;  // no more threads to be had: the ones running (and this one) drain the queue
    }
  }
  work();
  for (auto& t : pool) t.join();
}
}  // namespace rlnamd
extern "C" {
int rlnamd_verify_many(rlnamd_prover* p, size_t n, const uint8_t* proofs, const uint8_t* values_le, size_t n_values,
                       int threads, uint8_t* ok) {
  RLN_TRY
  if (n_values + 1 != p->p->zkey().gamma_abc_g1.size()) throw Error("MalformedVerifyingKey");
  verify_many_common(p->p->zkey(), n, proofs, values_le, n_values, threads, ok);
  RLN_CATCH
}
int rlnamd_verify_many_with_zkey(const uint8_t* zkey, size_t zkey_len, size_t n, const uint8_t* proofs,
                                 const uint8_t* values_le, size_t n_values, int threads, uint8_t* ok) {
  RLN_TRY
  Zkey zk = parse_arkzkey(zkey, zkey_len);
  if (n_values + 1 != zk.gamma_abc_g1.size()) throw Error("MalformedVerifyingKey");
  verify_many_common(zk, n, proofs, values_le, n_values, threads, ok);
  RLN_CATCH
}

size_t rlnamd_prover_num_public(rlnamd_prover* p) { return p->p->num_public(); }
int rlnamd_prover_download_public(rlnamd_prover* p, size_t n, uint8_t* out_le) {
  RLN_TRY
  std::vector<uint8_t> v;
  p->p->fetch_public(n, &v);
  memcpy(out_le, v.data(), v.size());
  RLN_CATCH
}
int rlnamd_verify_public(rlnamd_prover* p, const uint8_t proof[128], const uint8_t* values_le, size_t n_values, int* ok) {
  RLN_TRY
  verify_common(p->p->zkey(), proof, values_le, ok, n_values);
  RLN_CATCH
}

int rlnamd_verify_with_zkey(const uint8_t* zkey, size_t zkey_len, const uint8_t proof[128],
                            const uint8_t values_le[160], int* ok) {
  RLN_TRY
  Zkey zk = parse_arkzkey(zkey, zkey_len);
  verify_common(zk, proof, values_le, ok);
  RLN_CATCH
}

int rlnamd_parse_resources(const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len,
                           uint64_t counts[13]) {
  RLN_TRY
  Zkey zk = parse_arkzkey(zkey, zkey_len);
  Graph g = parse_graph(graph, graph_len);
  counts[0] = zk.num_instance_variables;
  counts[1] = zk.num_witness_variables;
  counts[2] = zk.num_constraints;
  counts[3] = zk.a_nnz;
  counts[4] = zk.b_nnz;
  counts[5] = zk.a_query.size();
  counts[6] = zk.h_query.size();
  counts[7] = zk.l_query.size();
  counts[8] = g.nodes.size();
  counts[9] = g.signals.size();
  counts[10] = g.tree_depth;
  counts[11] = g.max_out;
  counts[12] = g.inputs_size;
  RLN_CATCH
}

int rlnamd_proof_compress(const uint8_t coords_le[256], uint8_t proof[128]) {
  RLN_TRY
  auto ld = [&](int k) {
    uint32_t c[8];
    memcpy(c, coords_le + 32 * k, 32);
    if (limbs_geq(c, FqParams::MOD)) throw Error("Non-canonical field element");
    return Fq::from_canonical(c);
  };
  G1Affine A{ld(0), ld(1)}, C{ld(6), ld(7)};
  G2Affine B{{ld(2), ld(3)}, {ld(4), ld(5)}};
  g1_compress(A, proof);
  g2_compress(B, proof + 32);
  g1_compress(C, proof + 96);
  RLN_CATCH
}
int rlnamd_proof_decompress(const uint8_t proof[128], uint8_t coords_le[256]) {
  RLN_TRY
  G1Affine A, C;
  G2Affine B;
  if (!g1_decompress(proof, &A) || !g2_decompress(proof + 32, &B) || !g1_decompress(proof + 96, &C))
    throw Error("Proof serialization error: the input buffer contained invalid data");
  auto st = [&](int k, const Fq& v) {
    uint32_t c[8];
    v.to_canonical(c);
    memcpy(coords_le + 32 * k, c, 32);
  };
  st(0, A.x); st(1, A.y); st(2, B.x.c0); st(3, B.x.c1); st(4, B.y.c0); st(5, B.y.c1); st(6, C.x); st(7, C.y);
  RLN_CATCH
}

}  // extern "C"
