#include "tree_any.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#include "poseidon.h"

namespace rlnamd {

// n hashes of two inputs on the device (pairs: n x 64 bytes canonical LE -> out: n x 32)
static void hash_pairs(const std::vector<uint8_t>& pairs, std::vector<uint8_t>& out) {
  const size_t n = pairs.size() / 64;
  out.resize(n * 32);
  if (!n) return;
  // a handful of hashes (a single update is one per level, 31 .. 63 levels one after the other): the library's host
  // Poseidon instead of an H2D + kernel + D2H round trip per level (see MerkleTreeDev::set_few; same threshold)
  if (n <= MerkleTreeDev::host_max_from_env()) {
    const PoseidonParams& P = poseidon_host_params(3);
    for (size_t i = 0; i < n; i++) {
      uint32_t c[8];
      Fr in[2];
      memcpy(c, pairs.data() + 64 * i, 32);
      in[0] = Fr::from_canonical(c);
      memcpy(c, pairs.data() + 64 * i + 32, 32);
      in[1] = Fr::from_canonical(c);
      poseidon_hash_host(P, in).to_canonical(c);
      memcpy(out.data() + 32 * i, c, 32);
    }
    return;
  }
  DevBuf<uint8_t> din(pairs.size()), dout(n * 32);
  RLN_HIP(hipMemcpy(din.p, pairs.data(), pairs.size(), hipMemcpyHostToDevice));
  poseidon_hash_batch_device(din.p, n, 2, dout.p, 0);
  RLN_HIP(hipMemcpy(out.data(), dout.p, n * 32, hipMemcpyDeviceToHost));
}

void SparseTree::init(int depth, const uint8_t default_leaf_le[32]) {
  require_gpu();
  if (depth < 0 || depth >= 64) throw Error("InvalidDepth");
  depth_ = depth;
  lv_.assign(depth + 1, {});
  zero_.assign(depth + 1, Node{});
  memcpy(zero_[depth].data(), default_leaf_le, 32);
  for (int l = depth - 1; l >= 0; l--) {   // cached_nodes (optimal_merkle_tree.rs:86-96): one hash per level
    std::vector<uint8_t> in(64), out;
    memcpy(in.data(), zero_[l + 1].data(), 32);
    memcpy(in.data() + 32, zero_[l + 1].data(), 32);
    hash_pairs(in, out);
    memcpy(zero_[l].data(), out.data(), 32);
  }
}

void SparseTree::node(int level, uint64_t index, uint8_t out_le[32]) const {
  auto it = lv_[level].find(index);
  memcpy(out_le, it == lv_[level].end() ? zero_[level].data() : it->second.data(), 32);
}

size_t SparseTree::stored_nodes() const {
  size_t s = 0;
  for (const auto& m : lv_) s += m.size();
  return s;
}

void SparseTree::set_range(size_t start, const uint8_t* leaves_le, size_t n) {
  std::vector<uint64_t> idx(n);
  for (size_t i = 0; i < n; i++) idx[i] = start + i;
  set_many(idx.data(), leaves_le, n);
}

void SparseTree::set_many(const uint64_t* idx, const uint8_t* leaves_le, size_t n) {
  if (!n) return;
  std::vector<uint64_t> touched(n);
  for (size_t i = 0; i < n; i++) {
    Node v;
    memcpy(v.data(), leaves_le + 32 * i, 32);
    lv_[depth_][idx[i]] = v;
    touched[i] = idx[i];
  }
  // update_hashes (optimal_merkle_tree.rs:296-330): the parents of everything touched, level by level, each level one
  // device batch
  for (int l = depth_; l > 0; l--) {
    std::vector<uint64_t> parents;
    for (uint64_t t : touched)
      if (parents.empty() || parents.back() != (t >> 1)) parents.push_back(t >> 1);   // touched is sorted
    std::vector<uint8_t> in(parents.size() * 64), out;
    for (size_t k = 0; k < parents.size(); k++) {
      node(l, 2 * parents[k], in.data() + 64 * k);
      node(l, 2 * parents[k] + 1, in.data() + 64 * k + 32);
    }
    hash_pairs(in, out);
    for (size_t k = 0; k < parents.size(); k++) {
      Node v;
      memcpy(v.data(), out.data() + 32 * k, 32);
      lv_[l - 1][parents[k]] = v;
    }
    touched.swap(parents);
  }
}

void SparseTree::proof(size_t leaf, uint8_t* elems_le, uint8_t* bits) const {
  uint64_t idx = leaf;
  for (int l = depth_; l > 0; l--) {
    const int k = depth_ - l;
    node(l, idx ^ 1, elems_le + 32 * k);
    bits[k] = (uint8_t)(idx & 1);   // odd index at its level == right child (full_merkle_tree.rs:296-300)
    idx >>= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
void TreeAny::init(int depth_, const uint8_t default_leaf_le[32]) {
  if (depth_ < 0 || depth_ >= 64) throw Error("InvalidDepth");
  // RLNAMD_TREE_SPARSE_ABOVE (tests): take the sparse tree above a smaller depth, so that the reference's tree tests and
  // known answers at depth 20 run against it as well
  static const int dense_max = [] {
    const char* v = getenv("RLNAMD_TREE_SPARSE_ABOVE");
    return (v && *v) ? std::min(atoi(v), MAX_DENSE_DEPTH) : MAX_DENSE_DEPTH;
  }();
  sparse = depth_ > dense_max;
  if (sparse) {
    sp.init(depth_, default_leaf_le);
    dense = MerkleTreeDev();
  } else {
    dense.init(depth_, default_leaf_le);
    sp = SparseTree();
  }
  depth = depth_;
  pend.reset(new Pending);
  // up to host_max dirty leaves per pass take the host-core chain (MerkleTreeDev::set_few); 0 = always the device pass
  host_max = MerkleTreeDev::host_max_from_env();
}

size_t TreeAny::pending_writes() const {
  if (!pend) return 0;
  std::lock_guard<std::mutex> lk(pend->mu);
  return pend->writes.size();
}

// caller holds pend->mu
static void flush_locked(TreeAny& t) {
  auto& w = t.pend->writes;
  if (w.empty()) return;
  std::vector<uint64_t> idx;
  std::vector<uint8_t> leaves;
  idx.reserve(w.size());
  leaves.reserve(w.size() * 32);
  for (const auto& kv : w) {
    idx.push_back(kv.first);
    leaves.insert(leaves.end(), kv.second.begin(), kv.second.end());
  }
  // The writes leave the queue only when the pass has gone through: their writers were told "done" long ago (the FFI
  // object has advanced its leaf bookkeeping), so a pass that throws (a HIP error, no pinned memory) must not lose them --
  // the error reaches the reader that triggered it, and the next reader tries again.
  if (t.sparse) t.sp.set_many(idx.data(), leaves.data(), idx.size());
  else if (idx.size() <= t.host_max) t.dense.set_few(idx.data(), leaves.data(), idx.size());
  else t.dense.set_scattered(idx.data(), leaves.data(), idx.size());
  w.clear();
}

void TreeAny::flush_pending() {
  if (!pend) return;
  std::lock_guard<std::mutex> lk(pend->mu);
  flush_locked(*this);
}

void TreeAny::set_leaf(size_t index, const uint8_t leaf_le[32]) {
  if (index >= capacity()) throw Error("TooManySet");
  std::lock_guard<std::mutex> lk(pend->mu);
  std::array<uint8_t, 32> v;
  memcpy(v.data(), leaf_le, 32);
  pend->writes[index] = v;
  if (pend->writes.size() >= MAX_PENDING) flush_locked(*this);
}

void TreeAny::set_range_host(size_t start, const uint8_t* leaves_le, size_t n) {
  if (start + n > capacity() || start + n < start) throw Error("TooManySet");
  if (n == 0) return;
  std::lock_guard<std::mutex> lk(pend->mu);
  if (n <= DEFER_RANGE_MAX) {   // a short range is a handful of single writes
    for (size_t i = 0; i < n; i++) {
      std::array<uint8_t, 32> v;
      memcpy(v.data(), leaves_le + 32 * i, 32);
      pend->writes[start + i] = v;
    }
    if (pend->writes.size() >= MAX_PENDING) flush_locked(*this);
    return;
  }
  // a bulk write: earlier single writes inside the range are overwritten by it, the others keep waiting (their paths
  // and the range's meet only in nodes both passes recompute from the leaves below -- but the range pass would hash
  // over leaves that are not written yet, so the pending ones go first)
  flush_locked(*this);
  if (sparse) sp.set_range(start, leaves_le, n); else dense.set_range_host(start, leaves_le, n);
}

void TreeAny::get_node_host(size_t node, uint8_t out_le[32]) {
  std::lock_guard<std::mutex> lk(pend->mu);
  if (depth > 0 && node >= capacity() - 1) {   // a leaf: pending writes answer for themselves, the others are current
    auto it = pend->writes.find(node - (capacity() - 1));
    if (it != pend->writes.end()) {
      memcpy(out_le, it->second.data(), 32);
      return;
    }
  } else {
    flush_locked(*this);
  }
  if (!sparse) return dense.get_node_host(node, out_le);
  if (node == 0) return sp.root(out_le);
  if (node < capacity() - 1) throw Error("sparse tree: only the root and the leaves are addressed by heap index");
  sp.leaf(node - (capacity() - 1), out_le);
}

void TreeAny::get_leaves_host(size_t first, size_t n, uint8_t* out_le) {
  std::lock_guard<std::mutex> lk(pend->mu);
  flush_locked(*this);
  if (!sparse) return dense.get_leaves_host(first, n, out_le);
  for (size_t i = 0; i < n; i++) sp.leaf(first + i, out_le + 32 * i);
}

void TreeAny::proof_host(size_t leaf, uint8_t* elems_le, uint8_t* bits) {
  std::lock_guard<std::mutex> lk(pend->mu);
  flush_locked(*this);
  if (sparse) sp.proof(leaf, elems_le, bits); else dense.proof_host(leaf, elems_le, bits);
}

}  // namespace rlnamd
