// Dense Poseidon Merkle tree resident in HBM.
//
// Device-side counterpart of FullMerkleTree (/root/reference/utils/src/merkle_tree/full_merkle_tree.rs):
// same heap layout (node i has children 2i+1, 2i+2; leaves at 2^d - 1 + idx; :20-40,336-358), same
// level-by-level rehash of the touched parent range (update_hashes :360-399), same proof shape
// (bottom-up siblings + index bits, bit = 1 when the node is a right child, :288-304,420-439).
// Nodes are kept as Montgomery residues in HBM (2^(d+1) - 1 x 32 B: 64 MiB at depth 20).
#pragma once
#include <vector>

#include "common.h"
#include "field.h"

namespace rlnamd {

struct MerkleTreeDev {
  int depth = 0;
  DevBuf<Fr> nodes;
  std::vector<Fr> zero_hashes;  // [level] Montgomery, level 0 = root ... depth = leaf
  hipStream_t stream = 0;
  // staging of set_scattered: leaf indices, dirty-node lists and leaves of one pass (pinned host + device copy)
  DevBuf<uint32_t> scat_dev;
  uint32_t* scat_host = nullptr;
  size_t scat_cap = 0;
  bool root_known = false;      // root_host is the root of what the stream will have written (set_few)
  Fr root_host;
  DevBuf<uint8_t> proof_dev;    // proof_host's device scratch (depth * 33 bytes), allocated once
  void reserve_staging(size_t words);

  MerkleTreeDev() = default;
  ~MerkleTreeDev();
  MerkleTreeDev(MerkleTreeDev&&) noexcept;
  MerkleTreeDev& operator=(MerkleTreeDev&&) noexcept;
  MerkleTreeDev(const MerkleTreeDev&) = delete;
  MerkleTreeDev& operator=(const MerkleTreeDev&) = delete;

  void init(int depth_, const uint8_t default_leaf_le[32]);  // FullMerkleTree::new :82-115
  size_t capacity() const { return (size_t)1 << depth; }
  size_t num_nodes() const { return ((size_t)2 << depth) - 1; }

  // write `n` canonical LE leaves starting at leaf index `start`, then rehash (set_range :197-223).
  void set_range_host(size_t start, const uint8_t* leaves_le, size_t n);
  void set_range_device(size_t start, const uint8_t* d_leaves_le, size_t n);
  // k leaves at strictly increasing indices `idx`, then ONE bottom-up pass over the union of their paths (the parents of
  // what changed, level by level; the top levels in a single launch).  Stream-ordered: returns without a host wait.
  void set_scattered(const uint64_t* idx, const uint8_t* leaves_le, size_t k);
  // The same update for a HANDFUL of leaves, the dependent chain on a host core: one gather of the <= depth * k clean
  // siblings of the dirty paths (a kernel writing pinned host memory), the hashes of the paths with the library's own
  // host Poseidon (poseidon_hash_host: ~20 us each, against 146 us per link of the same chain on a lone wave), one
  // scatter of the rewritten nodes behind it (stream-ordered, nobody waits for it).  The tree stays in HBM and every
  // other reader finds it updated; the root is remembered on the host (root_known) so that the usual
  // set_leaf -> get_root of a membership contract costs no second round trip.  (FullMerkleTree::set + update_hashes,
  // full_merkle_tree.rs:197-223,336-399.)
  void set_few(const uint64_t* idx, const uint8_t* leaves_le, size_t k);
  // how many dirty leaves a pass may have and still take set_few: the device pass costs ~2.2 ms whatever k is (up to
  // a few thousand: 20 levels x 0.108 ms with four lanes per hash), set_few 0.24 ms for one leaf and ~0.18 ms for every
  // further one (measured, EPYC 9575F host: k = 8 1.5 ms, k = 12 2.1 ms).  RLNAMD_TREE_HOST_MAX overrides (0: never; tests force 0 / 8 / 4096).
  static constexpr size_t HOST_MAX_DEFAULT = 11, HOST_MAX_LIMIT = 4096;
  static size_t host_max_from_env();
  // leaves i -> Fr(first + i): synthetic fill generated on the device (bench / config 3), then rehash
  void fill_sequential_device(size_t start, size_t n, uint64_t first);
  void rehash(size_t lo_node, size_t hi_node);  // update_hashes :360-399

  void get_node_host(size_t node, uint8_t out_le[32]);
  // leaves [first, first + n) as canonical 32-byte LE values (tree snapshots)
  void get_leaves_host(size_t first, size_t n, uint8_t* out_le);
  // one proof -> host (elems: depth*32 B canonical LE bottom-up, bits: depth bytes)
  void proof_host(size_t leaf, uint8_t* elems_le, uint8_t* bits);
  // `count` proofs for leaves [first, first+count) written to device buffers
  // d_elems: [count][depth][32] canonical LE, d_bits: [count][depth]
  void proofs_device(size_t first, size_t count, uint8_t* d_elems, uint8_t* d_bits);
  // recompute the root from each emitted proof + its leaf on the device (compute_root_from :441-446);
  // returns the number of proofs whose root differs from the tree root.
  size_t verify_proofs_device(size_t first, size_t count, const uint8_t* d_elems, const uint8_t* d_bits);
};

}  // namespace rlnamd
