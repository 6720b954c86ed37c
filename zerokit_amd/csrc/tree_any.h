// The membership tree behind the FFI object: the dense HBM-resident tree (merkle.h, depth <= 30: 2^(depth+1) - 1 nodes of
// 32 bytes) or, for the depths only a sparse structure can hold (31 .. 63), a host-indexed tree that keeps the nodes
// that were ever written and hashes on the device in per-level batches -- the semantics of OptimalMerkleTree
// (/root/reference/utils/src/merkle_tree/optimal_merkle_tree.rs:15-41, 120-200: nodes in a HashMap keyed (depth,
// index), cached default hashes per level, update_hashes over the touched range).  Roots, proofs and leaves are the same
// bytes where both structures exist (rln/tests/poseidon_tree.rs).
#pragma once
#include <stdint.h>

#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "merkle.h"

namespace rlnamd {

class SparseTree {
 public:
  void init(int depth, const uint8_t default_leaf_le[32]);
  int depth() const { return depth_; }
  void set_range(size_t start, const uint8_t* leaves_le, size_t n);   // then update_hashes over the touched indices
  // k leaves at strictly increasing indices, then ONE update_hashes pass over the union of the touched paths
  void set_many(const uint64_t* idx, const uint8_t* leaves_le, size_t k);
  void root(uint8_t out_le[32]) const { node(0, 0, out_le); }
  void leaf(size_t index, uint8_t out_le[32]) const { node(depth_, index, out_le); }
  void proof(size_t leaf, uint8_t* elems_le, uint8_t* bits) const;     // bottom-up, bit = 1: the node is a right child
  size_t stored_nodes() const;

 private:
  typedef std::array<uint8_t, 32> Node;
  void node(int level, uint64_t index, uint8_t out_le[32]) const;      // level 0 = root ... depth = leaves
  int depth_ = 0;
  std::vector<std::unordered_map<uint64_t, Node>> lv_;                  // [level] index -> canonical LE value
  std::vector<Node> zero_;                                              // [level] hash of an empty subtree
};

// what ffi.cpp drives: the subset of MerkleTreeDev's interface it uses, dispatched on the depth
struct TreeAny {
  int depth = 0;
  bool sparse = false;
  MerkleTreeDev dense;
  SparseTree sp;
  static constexpr int MAX_DENSE_DEPTH = 30;

  // Deferred, coalesced single-leaf updates.  The reference pays `depth` hashes inside every set()
  // (full_merkle_tree.rs:336-399) -- on a CPU core that is ~0.5 ms; a chain of 20 dependent hashes on a GPU is ~2 ms
  // whatever is done to it.  So set_leaf() only records the write; the first READER of anything a write can change (root,
  // proof, a bulk range write, a snapshot) runs ONE bottom-up pass over the union of the dirty paths: k updates between two
  // reads cost one pass, not k.  A leaf read is answered from the pending writes.  Observable behaviour is the
  // reference's: every read sees all earlier writes, later writes to an index win.
  struct Pending {
    std::mutex mu;                                        // readers (&self in the reference) may come from several threads
    std::map<uint64_t, std::array<uint8_t, 32>> writes;   // ordered: the flush wants increasing indices
  };
  std::unique_ptr<Pending> pend{new Pending};
  static constexpr size_t MAX_PENDING = (size_t)1 << 18;  // bounds host memory; a burst above it flushes early
  static constexpr size_t DEFER_RANGE_MAX = 64;           // ranges up to this length ride in the pending set
  // A pass over at most host_max dirty leaves runs its dependent chain on a host core (MerkleTreeDev::set_few: one
  // update + root 3.1 -> 0.24 ms).  RLNAMD_TREE_HOST_MAX overrides (read by init(); 0: never).
  size_t host_max = MerkleTreeDev::HOST_MAX_DEFAULT;

  void init(int depth_, const uint8_t default_leaf_le[32]);
  size_t capacity() const { return (size_t)1 << depth; }
  void set_leaf(size_t index, const uint8_t leaf_le[32]);   // deferred
  void flush_pending();                                     // explicit (ffi_flush, snapshots); readers call it themselves
  size_t pending_writes() const;
  void set_range_host(size_t start, const uint8_t* leaves_le, size_t n);
  // heap index as in MerkleTreeDev: 0 = root, capacity() - 1 + i = leaf i (the only two forms the FFI uses)
  void get_node_host(size_t node, uint8_t out_le[32]);
  void get_leaves_host(size_t first, size_t n, uint8_t* out_le);
  void proof_host(size_t leaf, uint8_t* elems_le, uint8_t* bits);
};

}  // namespace rlnamd
