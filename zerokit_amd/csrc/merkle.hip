#include "merkle.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "poseidon.h"

namespace rlnamd {

// parents [first, first+count): nodes[p] = H(nodes[2p+1], nodes[2p+2])   (hash_parent, :373-376)
__global__ void __launch_bounds__(256) k_hash_parents(Fr* __restrict__ nodes, size_t first, size_t count,
                                                      PoseidonView pv) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  size_t p = first + i;
  Fr in[2] = {nodes[2 * p + 1], nodes[2 * p + 2]};
  nodes[p] = poseidon_hash_dev<3>(in, pv);
}

// the same for a level with fewer parents than the chip has lanes: four lanes per hash (poseidon_hash4_lanes), 16
// parents per single-wave workgroup -- the hash latency, which is all such a level costs, drops by 2.3 x
constexpr uint32_t LPH = POSEIDON_LANES_PER_HASH, HPW = POSEIDON_HASHES_PER_WAVE;
__global__ void __launch_bounds__(64) k_hash_parents_l3(Fr* __restrict__ nodes, size_t first, size_t count, PoseidonView pv) {
  __shared__ __attribute__((aligned(16))) uint32_t sh[64 * 12];
  const uint32_t lane = threadIdx.x, g = lane / LPH, j = lane % LPH;
  const size_t i = (size_t)blockIdx.x * HPW + g;
  const bool active = i < count;
  const size_t p = first + (active ? i : 0);
  Fr in = Fr::zero();
  if (active && (j == 1 || j == 2)) in = nodes[2 * p + j];   // j = 1: left child 2 p + 1, j = 2: right child 2 p + 2
  const Fr h = poseidon_hash4_lanes(in, pv, sh);
  if (active && j == 0) nodes[p] = h;
}

// ---- union of dirty paths (deferred single-leaf updates, MerkleTreeDev::set_scattered): a level is a LIST of parents
__global__ void __launch_bounds__(256) k_scatter_leaves(Fr* __restrict__ nodes, size_t first_leaf_node,
                                                        const uint32_t* __restrict__ idx, const uint32_t* __restrict__ leaves_le,
                                                        uint32_t k) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  nodes[first_leaf_node + idx[i]] = Fr::from_canonical(leaves_le + (size_t)i * 8);
}
__global__ void __launch_bounds__(256) k_hash_parents_list(Fr* __restrict__ nodes, const uint32_t* __restrict__ list,
                                                           uint32_t count, PoseidonView pv) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  size_t p = list[i];
  Fr in[2] = {nodes[2 * p + 1], nodes[2 * p + 2]};
  nodes[p] = poseidon_hash_dev<3>(in, pv);
}
__global__ void __launch_bounds__(64) k_hash_parents_l3_list(Fr* __restrict__ nodes, const uint32_t* __restrict__ list,
                                                             uint32_t count, PoseidonView pv) {
  __shared__ __attribute__((aligned(16))) uint32_t sh[64 * 12];
  const uint32_t lane = threadIdx.x, g = lane / LPH, j = lane % LPH;
  const uint32_t i = blockIdx.x * HPW + g;
  const bool active = i < count;
  const size_t p = active ? list[i] : 0;
  Fr in = Fr::zero();
  if (active && (j == 1 || j == 2)) in = nodes[2 * p + j];
  const Fr h = poseidon_hash4_lanes(in, pv, sh);
  if (active && j == 0) nodes[p] = h;
}
// The top of a dirty-path pass -- every level from `first` up to the root -- in ONE launch: a single update is 20
// dependent hashes, and as 20 launches each paid its launch latency on top of the hash.  One workgroup of TAIL_WAVES
// waves, 16 four-lane hashes per wave and step; a level's results reach the next level through HBM (the workgroup's
// own CU: workgroup-scope visibility after the barrier).  levels[l] = [off[l], off[l + 1]) of `list`, bottom-up.
// (WAVES = 1 when no level holds more than 16 parents -- a single update, a handful of them: the barriers inside the hash
// then cost nothing; WAVES = 4 otherwise: four waves meeting at ~260 barriers per hash run each level ~40 % slower)
constexpr uint32_t TAIL_WAVES_MAX = 4, TAIL_HASHES = HPW * TAIL_WAVES_MAX;
struct LevelOffsets {
  uint32_t off[34];
};
template <uint32_t WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_hash_tail_list(Fr* __restrict__ nodes, const uint32_t* __restrict__ list,
                                                               LevelOffsets lo, int first, int nlevels, PoseidonView pv) {
  __shared__ __attribute__((aligned(16))) uint32_t sh_all[WAVES][64 * 12];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane / LPH, j = lane % LPH;
  uint32_t* sh = sh_all[wave];
  for (int l = first; l < nlevels; l++) {
    const uint32_t b = lo.off[l], cnt = lo.off[l + 1] - b;
    for (uint32_t i0 = 0; i0 < cnt; i0 += HPW * WAVES) {
      const uint32_t i = i0 + wave * HPW + g;
      const bool active = i < cnt;
      const size_t p = active ? list[b + i] : 0;
      Fr in = Fr::zero();
      if (active && (j == 1 || j == 2)) in = nodes[2 * p + j];
      const Fr h = poseidon_hash4_lanes(in, pv, sh);
      if (active && j == 0) nodes[p] = h;
    }
    __threadfence();
    __syncthreads();
  }
}

// set_few: the clean siblings of a handful of dirty paths out (into pinned host memory), the rewritten nodes back in
__global__ void __launch_bounds__(64) k_gather_nodes(const Fr* __restrict__ nodes, const uint32_t* __restrict__ list, uint32_t count,
                                                     Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i < count) out[i] = nodes[list[i]];
}
__global__ void __launch_bounds__(64) k_scatter_nodes(Fr* __restrict__ nodes, const uint32_t* __restrict__ list, uint32_t count,
                                                      const Fr* __restrict__ vals) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i < count) nodes[list[i]] = vals[i];
}

// zero_hashes[depth] = default leaf; zero_hashes[l] = H(z[l+1], z[l+1])  -- single lane, init only
__global__ void k_zero_hashes(Fr* zh, int depth, PoseidonView pv) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int l = depth - 1; l >= 0; l--) {
    Fr in[2] = {zh[l + 1], zh[l + 1]};
    zh[l] = poseidon_hash_dev<3>(in, pv);
  }
}

// nodes of level l (2^l of them starting at 2^l - 1) <- zero_hashes[l]
__global__ void __launch_bounds__(256) k_fill_levels(Fr* __restrict__ nodes, const Fr* __restrict__ zh, int depth,
                                                     size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int level = 63 - __clzll((unsigned long long)(i + 1));
  nodes[i] = zh[level];
}

__global__ void __launch_bounds__(256) k_set_leaves(Fr* __restrict__ nodes, size_t first_node,
                                                    const uint32_t* __restrict__ leaves_le, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  nodes[first_node + i] = Fr::from_canonical(leaves_le + i * 8);
}

__global__ void __launch_bounds__(256) k_fill_seq(Fr* __restrict__ nodes, size_t first_node, size_t n,
                                                  uint64_t first) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t v = first + i;
  uint32_t c[8] = {(uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0};
  nodes[first_node + i] = Fr::from_canonical(c);
}

// one lane per (proof, level): coalesced 32-byte stores of the sibling in canonical form
__global__ void __launch_bounds__(256) k_proofs(const Fr* __restrict__ nodes, int depth, size_t first, size_t count,
                                                uint32_t* __restrict__ elems, uint8_t* __restrict__ bits) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * (size_t)depth) return;
  size_t pi = t / depth;
  int lvl = (int)(t % depth);  // 0 = leaf level
  size_t node = (((size_t)1 << depth) - 1 + first + pi);
  // ancestor of the leaf at height lvl: ((node + 1) >> lvl) - 1
  size_t cur = ((node + 1) >> lvl) - 1;
  bool right = (cur & 1) == 0;  // even heap index == right child (:296-300)
  size_t sib = right ? cur - 1 : cur + 1;
  uint32_t c[8];
  nodes[sib].to_canonical(c);
  uint4* o = reinterpret_cast<uint4*>(elems + t * 8);  // two 16-byte stores per lane: a wave writes 2 KiB contiguous
  o[0] = make_uint4(c[0], c[1], c[2], c[3]);
  o[1] = make_uint4(c[4], c[5], c[6], c[7]);
  bits[t] = right ? 1 : 0;
}

constexpr uint32_t PROOFS_PER_BLOCK = 64;
// Bulk emission.  A gather -> store loop (round 2: canonical copies of all nodes, then one gather per 16 bytes) makes every
// 16-byte store wait for an L2 round trip, and with 32 waves per CU that caps the chip near 4 TB/s (0.19 ms for 692 MB; a
// fill reaches 6.8 TB/s, a copy 2 x 2.7 TB/s: tools/fill_bw.py).  A block
// of PROOFS_PER_BLOCK consecutive proofs only needs ~170 distinct nodes (level l: the siblings of <= 64 >> l + 1
// ancestors, a contiguous heap range), so they are converted to canonical form ONCE into LDS (no separate conversion
// pass, no 67 MB round trip through HBM) and the emission loop is LDS read -> 16-byte store: stores issue back to back, every
// store instruction of a wave covers 1 KiB without gaps, and the path bits of the block (1 280 contiguous bytes) leave
// as 16-byte stores instead of one byte per element.
constexpr uint32_t PATHS_LDS_NODES = PROOFS_PER_BLOCK * 2 + 4 * 32;   // sum over levels of <= (63 >> l) + 4 nodes, depth <= 30
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_proofs_lds(const Fr* __restrict__ nodes, uint32_t depth, uint32_t magic,
                                                    size_t first, size_t count, u32x4_t* __restrict__ elems,
                                                    uint8_t* __restrict__ bits, uint32_t bits_vec) {
  __shared__ u32x4_t sh[PATHS_LDS_NODES * 2];
  __shared__ uint32_t lvl_off[32];       // LDS slot of the first node kept for a level
  __shared__ unsigned long long lvl_base[32];  // 1-based heap index of that node
  const size_t p0 = (size_t)blockIdx.x * PROOFS_PER_BLOCK;
  const uint32_t np = (uint32_t)(count - p0 < PROOFS_PER_BLOCK ? count - p0 : PROOFS_PER_BLOCK);
  const size_t leaf1 = ((size_t)1 << depth) + first + p0;   // 1-based heap index of the block's first leaf
  if (threadIdx.x < depth) {
    // ancestors at height l: [leaf1 >> l, (leaf1 + np - 1) >> l]; their siblings: that range widened to even..odd
    uint32_t off = 0;
    for (uint32_t l = 0; l < threadIdx.x; l++) {
      size_t lo = (leaf1 >> l) & ~(size_t)1, hi = ((leaf1 + np - 1) >> l) | 1;
      off += (uint32_t)(hi - lo + 1);
    }
    lvl_off[threadIdx.x] = off;
    lvl_base[threadIdx.x] = (leaf1 >> threadIdx.x) & ~(size_t)1;
  }
  __syncthreads();
  {
    const uint32_t last = depth - 1;
    const uint32_t total = lvl_off[last] + (uint32_t)((((leaf1 + np - 1) >> last) | 1) - lvl_base[last] + 1);
    for (uint32_t i = threadIdx.x; i < total; i += 256) {
      uint32_t l = 0;
      while (l + 1 < depth && lvl_off[l + 1] <= i) l++;
      const size_t h1 = lvl_base[l] + (i - lvl_off[l]);   // 1-based heap index
      uint32_t c[8];
      nodes[h1 - 1].to_canonical(c);
      sh[2 * i] = u32x4_t{c[0], c[1], c[2], c[3]};
      sh[2 * i + 1] = u32x4_t{c[4], c[5], c[6], c[7]};
    }
  }
  __syncthreads();
  const uint32_t n2 = np * depth * 2;
  const size_t out0 = p0 * depth;
#pragma unroll 2
  for (uint32_t e2 = threadIdx.x; e2 < n2; e2 += 256) {
    const uint32_t e = e2 >> 1, half = e2 & 1;
    const uint32_t pl = __umulhi(e, magic);
    const uint32_t lvl = e - pl * depth;
    const size_t cur1 = (leaf1 + pl) >> lvl;               // 1-based ancestor; its sibling is cur1 ^ 1
    const uint32_t slot = lvl_off[lvl] + (uint32_t)((cur1 ^ 1) - lvl_base[lvl]);
    __builtin_nontemporal_store(sh[2 * slot + half], &elems[(out0 << 1) + e2]);   // written once, never read back here
  }
  // bits: bit = 1 when the node on the path is a right child, i.e. its 1-based heap index is odd (:296-300)
  const uint32_t nb = np * depth;
  if (bits_vec) {
    for (uint32_t q = threadIdx.x; q * 16 < nb; q += 256) {
      uint32_t w[4] = {0, 0, 0, 0};
      for (uint32_t k = 0; k < 16; k++) {
        const uint32_t e = q * 16 + k;
        const uint32_t pl = __umulhi(e, magic);
        const uint32_t lvl = e - pl * depth;
        const uint32_t b = e < nb ? (uint32_t)(((leaf1 + pl) >> lvl) & 1) : 0u;
        w[k >> 2] |= b << (8 * (k & 3));
      }
      if (q * 16 + 16 <= nb) {
        *reinterpret_cast<uint4*>(bits + out0 + q * 16) = make_uint4(w[0], w[1], w[2], w[3]);
      } else {
        for (uint32_t k = 0; q * 16 + k < nb; k++) bits[out0 + q * 16 + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
      }
    }
  } else {
    for (uint32_t e = threadIdx.x; e < nb; e += 256) {
      const uint32_t pl = __umulhi(e, magic);
      const uint32_t lvl = e - pl * depth;
      bits[out0 + e] = (uint8_t)(((leaf1 + pl) >> lvl) & 1);
    }
  }
}

__global__ void __launch_bounds__(256) k_verify_proofs(const Fr* __restrict__ nodes, int depth, size_t first,
                                                       size_t count, const uint32_t* __restrict__ elems,
                                                       const uint8_t* __restrict__ bits, PoseidonView pv,
                                                       unsigned long long* __restrict__ bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Fr h = nodes[((size_t)1 << depth) - 1 + first + i];
  for (int l = 0; l < depth; l++) {
    Fr s = Fr::from_canonical(elems + (i * depth + l) * 8);
    Fr in[2];
    if (bits[i * depth + l]) {
      in[0] = s;
      in[1] = h;
    } else {
      in[0] = h;
      in[1] = s;
    }
    h = poseidon_hash_dev<3>(in, pv);
  }
  if (h != nodes[0]) atomicAdd(bad, 1ULL);
}

void MerkleTreeDev::init(int depth_, const uint8_t default_leaf_le[32]) {
  require_gpu();
  if (depth_ < 0 || depth_ > 30) throw Error("InvalidDepth: tree depth must be in [0, 30] for the HBM-resident tree");
  depth = depth_;
  root_known = false;
  nodes.alloc(num_nodes());
  PoseidonView pv = poseidon_view(3);
  DevBuf<Fr> zh(depth + 1);
  uint32_t c[8];
  memcpy(c, default_leaf_le, 32);
  Fr leaf = Fr::from_canonical(c);
  RLN_HIP(hipMemcpyAsync(zh.p + depth, &leaf, sizeof(Fr), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(k_zero_hashes, dim3(1), dim3(64), 0, stream, zh.p, depth, pv);
  size_t total = num_nodes();
  hipLaunchKernelGGL(k_fill_levels, dim3(div_up(total, 256)), dim3(256), 0, stream, nodes.p, zh.p, depth, total);
  RLN_HIP(hipGetLastError());
  zero_hashes.resize(depth + 1);
  RLN_HIP(hipMemcpyAsync(zero_hashes.data(), zh.p, (depth + 1) * sizeof(Fr), hipMemcpyDeviceToHost, stream));
  RLN_HIP(hipStreamSynchronize(stream));
}

void MerkleTreeDev::rehash(size_t lo, size_t hi) {
  root_known = false;
  PoseidonView pv = poseidon_view(3);
  const size_t l3_max = HPW * 1024;
  while (lo > 0) {
    lo = ((lo + 1) >> 1) - 1;
    hi = ((hi + 1) >> 1) - 1;
    size_t cnt = hi - lo + 1;
    // a level that cannot fill the chip anyway (at most one 16-hash wave per SIMD) costs one hash latency: four
    // lanes per hash there
    if (cnt <= l3_max)
      hipLaunchKernelGGL(k_hash_parents_l3, dim3(div_up(cnt, HPW)), dim3(64), 0, stream, nodes.p, lo, cnt, pv);
    else
      hipLaunchKernelGGL(k_hash_parents, dim3(div_up(cnt, 256)), dim3(256), 0, stream, nodes.p, lo, cnt, pv);
  }
  RLN_HIP(hipGetLastError());
}

void MerkleTreeDev::set_range_device(size_t start, const uint8_t* d_leaves_le, size_t n) {
  if (start + n > capacity() || start + n < start) throw Error("TooManySet");
  if (n == 0) return;
  size_t first = capacity() - 1 + start;
  hipLaunchKernelGGL(k_set_leaves, dim3(div_up(n, 256)), dim3(256), 0, stream, nodes.p, first,
                     (const uint32_t*)d_leaves_le, n);
  rehash(first, first + n - 1);
}

void MerkleTreeDev::set_range_host(size_t start, const uint8_t* leaves_le, size_t n) {
  if (start + n > capacity() || start + n < start) throw Error("TooManySet");
  if (n == 0) return;
  DevBuf<uint8_t> tmp(n * 32);
  RLN_HIP(hipMemcpyAsync(tmp.p, leaves_le, n * 32, hipMemcpyHostToDevice, stream));
  set_range_device(start, tmp.p, n);
  RLN_HIP(hipStreamSynchronize(stream));
}

// One bottom-up pass over the UNION of the dirty paths of k scattered leaves: level by level only the parents of what
// changed below are rehashed (update_hashes :360-399 applied to a set instead of a range).  k single-leaf updates cost
// one ~depth-level pass instead of k of them.  idx: strictly increasing leaf indices; stream-ordered, no host wait.
void MerkleTreeDev::set_scattered(const uint64_t* idx, const uint8_t* leaves_le, size_t k) {
  if (k == 0) return;
  if (k > 0xFFFFFFFFull) throw Error("TooManySet");
  for (size_t i = 0; i < k; i++)
    if (idx[i] >= capacity() || (i && idx[i] <= idx[i - 1])) throw Error("set_scattered: indices must be increasing and inside the tree");
  root_known = false;
  // host side: the dirty node lists, bottom-up.  Heap indices fit 32 bits (depth <= 30).
  LevelOffsets lo{};
  std::vector<uint32_t> lists;
  {
    std::vector<uint64_t> cur(k);
    for (size_t i = 0; i < k; i++) cur[i] = capacity() - 1 + idx[i];
    for (int l = 0; l < depth; l++) {
      lo.off[l] = (uint32_t)lists.size();
      size_t m = 0;
      for (size_t i = 0; i < cur.size(); i++) {
        uint64_t p = (cur[i] - 1) >> 1;
        if (m == 0 || cur[m - 1] != p) cur[m++] = p;
      }
      cur.resize(m);
      for (uint64_t p : cur) lists.push_back((uint32_t)p);
    }
    lo.off[depth] = (uint32_t)lists.size();
  }
  // staging: [k leaf indices][lists][k x 32 B leaves], pinned; the previous pass may still be reading it
  const size_t words = k + lists.size() + 8 * k;
  reserve_staging(words);
  for (size_t i = 0; i < k; i++) scat_host[i] = (uint32_t)idx[i];
  if (!lists.empty()) memcpy(scat_host + k, lists.data(), lists.size() * 4);
  memcpy(scat_host + k + lists.size(), leaves_le, 32 * k);
  RLN_HIP(hipMemcpyAsync(scat_dev.p, scat_host, words * 4, hipMemcpyHostToDevice, stream));
  const uint32_t* d_idx = scat_dev.p;
  const uint32_t* d_list = scat_dev.p + k;
  const uint32_t* d_leaves = scat_dev.p + k + lists.size();
  hipLaunchKernelGGL(k_scatter_leaves, dim3(div_up(k, 256)), dim3(256), 0, stream, nodes.p, capacity() - 1, d_idx, d_leaves,
                     (uint32_t)k);
  PoseidonView pv = poseidon_view(3);
  int l = 0;
  for (; l < depth; l++) {
    const uint32_t cnt = lo.off[l + 1] - lo.off[l];
    if (cnt <= TAIL_HASHES) break;   // counts only shrink towards the root: the rest is one launch
    if (cnt <= HPW * 1024)
      hipLaunchKernelGGL(k_hash_parents_l3_list, dim3(div_up(cnt, HPW)), dim3(64), 0, stream, nodes.p, d_list + lo.off[l], cnt, pv);
    else
      hipLaunchKernelGGL(k_hash_parents_list, dim3(div_up(cnt, 256)), dim3(256), 0, stream, nodes.p, d_list + lo.off[l], cnt, pv);
  }
  if (l < depth) {
    if (lo.off[l + 1] - lo.off[l] <= HPW)   // counts only shrink: one wave covers every remaining level
      hipLaunchKernelGGL(k_hash_tail_list<1>, dim3(1), dim3(64), 0, stream, nodes.p, d_list, lo, l, depth, pv);
    else
      hipLaunchKernelGGL(k_hash_tail_list<TAIL_WAVES_MAX>, dim3(1), dim3(64 * TAIL_WAVES_MAX), 0, stream, nodes.p, d_list, lo,
                         l, depth, pv);
  }
  RLN_HIP(hipGetLastError());
}

// pinned staging of the dirty-path passes; the previous pass may still be reading it
void MerkleTreeDev::reserve_staging(size_t words) {
  RLN_HIP(hipStreamSynchronize(stream));
  if (scat_cap < words) {
    if (scat_host) (void)hipHostFree(scat_host);
    scat_host = nullptr;
    scat_cap = std::max<size_t>(words, 4096);
    RLN_HIP(hipHostMalloc((void**)&scat_host, scat_cap * 4, hipHostMallocDefault));
    scat_dev.alloc(scat_cap);
  }
}

size_t MerkleTreeDev::host_max_from_env() {
  const char* v = getenv("RLNAMD_TREE_HOST_MAX");
  const size_t m = (v && *v) ? (size_t)strtoull(v, nullptr, 10) : HOST_MAX_DEFAULT;
  return std::min(m, HOST_MAX_LIMIT);
}

void MerkleTreeDev::set_few(const uint64_t* idx, const uint8_t* leaves_le, size_t k) {
  if (k == 0) return;
  if (k > HOST_MAX_LIMIT) throw Error("set_few: a handful of leaves only (set_scattered takes the rest)");
  for (size_t i = 0; i < k; i++)
    if (idx[i] >= capacity() || (i && idx[i] <= idx[i - 1])) throw Error("set_few: indices must be increasing and inside the tree");
  // Level by level (bottom-up), the dirty nodes of the level in increasing heap order: cur[l]; a dirty node's sibling is
  // either dirty too (then it is its neighbour in cur[l]) or clean: fetched.  All of it follows from the indices alone.
  struct Dirty { uint32_t node; int sib; };   // sib >= 0: index into the fetch list; -1: the neighbour in cur is the sibling
  std::vector<std::vector<Dirty>> lv(depth + 1);
  std::vector<uint32_t> fetch;
  lv[0].reserve(k);
  for (size_t i = 0; i < k; i++) lv[0].push_back({(uint32_t)(capacity() - 1 + idx[i]), -1});
  for (int l = 0; l < depth; l++) {
    std::vector<Dirty>& c = lv[l];
    for (size_t i = 0; i < c.size(); i++) {
      const uint32_t nd = c[i].node, sb = (nd & 1) ? nd + 1 : nd - 1;   // children of p: 2p + 1 (odd), 2p + 2
      const bool paired = (nd & 1) ? (i + 1 < c.size() && c[i + 1].node == sb) : (i > 0 && c[i - 1].node == sb);
      if (!paired) {
        c[i].sib = (int)fetch.size();
        fetch.push_back(sb);
      }
      const uint32_t p = (nd - 1) >> 1;
      if (lv[l + 1].empty() || lv[l + 1].back().node != p) lv[l + 1].push_back({p, -1});
    }
  }
  size_t nwrite = 0;
  for (int l = 0; l <= depth; l++) nwrite += lv[l].size();
  // staging (32-bit words): [fetch list][fetched Fr x nf][write list][written Fr x nwrite]
  const size_t nf = fetch.size();
  const size_t o_fetched = (nf + 7) / 8 * 8, o_wlist = o_fetched + 8 * nf, o_wvals = (o_wlist + nwrite + 7) / 8 * 8;
  reserve_staging(o_wvals + 8 * nwrite);
  Fr* fetched = reinterpret_cast<Fr*>(scat_host + o_fetched);
  uint32_t* wlist = scat_host + o_wlist;
  Fr* wvals = reinterpret_cast<Fr*>(scat_host + o_wvals);
  if (nf) {
    memcpy(scat_host, fetch.data(), nf * 4);
    hipLaunchKernelGGL(k_gather_nodes, dim3(div_up(nf, 64)), dim3(64), 0, stream, nodes.p, scat_host, (uint32_t)nf, fetched);
    RLN_HIP(hipGetLastError());
  }
  // (the host work that does not need the siblings overlaps the gather)
  const PoseidonParams& P = poseidon_host_params(3);
  std::vector<std::vector<Fr>> val(depth + 1);
  val[0].resize(k);
  for (size_t i = 0; i < k; i++) {
    uint32_t c[8];
    memcpy(c, leaves_le + 32 * i, 32);
    val[0][i] = Fr::from_canonical(c);
  }
  if (nf) RLN_HIP(hipStreamSynchronize(stream));
  for (int l = 0; l < depth; l++) {
    const std::vector<Dirty>& c = lv[l];
    val[l + 1].reserve(lv[l + 1].size());
    for (size_t i = 0; i < c.size(); i++) {
      const bool left = (c[i].node & 1) != 0;
      Fr in[2];
      if (c[i].sib >= 0) {
        in[left ? 0 : 1] = val[l][i];
        in[left ? 1 : 0] = fetched[c[i].sib];
      } else if (left) {   // both children dirty: hashed once, at the left one
        in[0] = val[l][i];
        in[1] = val[l][i + 1];
      } else {
        continue;
      }
      val[l + 1].push_back(poseidon_hash_host(P, in));
    }
    if (val[l + 1].size() != lv[l + 1].size()) throw Error("internal: dirty-path bookkeeping");
  }
  size_t w = 0;
  for (int l = 0; l <= depth; l++)
    for (size_t i = 0; i < lv[l].size(); i++) {
      wlist[w] = lv[l][i].node;
      wvals[w] = val[l][i];
      w++;
    }
  hipLaunchKernelGGL(k_scatter_nodes, dim3(div_up(nwrite, 64)), dim3(64), 0, stream, nodes.p, wlist, (uint32_t)nwrite, wvals);
  RLN_HIP(hipGetLastError());
  root_host = val[depth][0];
  root_known = true;
}

MerkleTreeDev::~MerkleTreeDev() {
  // (a scatter kernel of set_few / set_scattered may still be reading the pinned staging or writing the nodes)
  if (scat_host || nodes.p) (void)hipStreamSynchronize(stream);
  if (scat_host) (void)hipHostFree(scat_host);
}
MerkleTreeDev::MerkleTreeDev(MerkleTreeDev&& o) noexcept { *this = std::move(o); }
MerkleTreeDev& MerkleTreeDev::operator=(MerkleTreeDev&& o) noexcept {
  if (this != &o) {
    if (scat_host || nodes.p) (void)hipStreamSynchronize(stream);
    if (scat_host) (void)hipHostFree(scat_host);
    depth = o.depth;
    nodes = std::move(o.nodes);
    zero_hashes = std::move(o.zero_hashes);
    stream = o.stream;
    scat_dev = std::move(o.scat_dev);
    scat_host = o.scat_host;
    scat_cap = o.scat_cap;
    root_known = o.root_known;
    root_host = o.root_host;
    proof_dev = std::move(o.proof_dev);
    o.root_known = false;
    o.scat_host = nullptr;
    o.scat_cap = 0;
    o.depth = 0;
  }
  return *this;
}

void MerkleTreeDev::fill_sequential_device(size_t start, size_t n, uint64_t first_value) {
  if (start + n > capacity()) throw Error("TooManySet");
  if (n == 0) return;
  size_t first = capacity() - 1 + start;
  hipLaunchKernelGGL(k_fill_seq, dim3(div_up(n, 256)), dim3(256), 0, stream, nodes.p, first, n, first_value);
  rehash(first, first + n - 1);
}

void MerkleTreeDev::get_node_host(size_t node, uint8_t out_le[32]) {
  if (node >= num_nodes()) throw Error("InvalidLeaf");
  Fr v;
  if (node == 0 && root_known) {
    v = root_host;
  } else {
    RLN_HIP(hipMemcpyAsync(&v, nodes.p + node, sizeof(Fr), hipMemcpyDeviceToHost, stream));
    RLN_HIP(hipStreamSynchronize(stream));
  }
  uint32_t c[8];
  v.to_canonical(c);
  memcpy(out_le, c, 32);
}

void MerkleTreeDev::get_leaves_host(size_t first, size_t n, uint8_t* out_le) {
  if (first + n > capacity() || first + n < first) throw Error("InvalidLeaf");
  if (n == 0) return;
  std::vector<Fr> v(n);
  RLN_HIP(hipMemcpyAsync(v.data(), nodes.p + (capacity() - 1 + first), n * sizeof(Fr), hipMemcpyDeviceToHost, stream));
  RLN_HIP(hipStreamSynchronize(stream));
  for (size_t i = 0; i < n; i++) {
    uint32_t c[8];
    v[i].to_canonical(c);
    memcpy(out_le + 32 * i, c, 32);
  }
}

void MerkleTreeDev::proofs_device(size_t first, size_t count, uint8_t* d_elems, uint8_t* d_bits) {
  if (first + count > capacity()) throw Error("InvalidLeaf");
  if (count == 0 || depth == 0) return;
  if ((uintptr_t)d_elems & 15) throw Error("proofs_device: the path-element buffer must be 16-byte aligned");
  size_t total = count * (size_t)depth;
  if (total >= num_nodes()) {   // bulk: fewer distinct nodes than path elements
    // (p0 * depth) bytes into d_bits is a multiple of 16 for every block when the base is 16-byte aligned
    const uint32_t bits_vec = (((uintptr_t)d_bits & 15) == 0 && (PROOFS_PER_BLOCK * (size_t)depth) % 16 == 0) ? 1u : 0u;
    const uint32_t magic = (uint32_t)(0x100000000ull / (uint32_t)depth) + 1;
    hipLaunchKernelGGL(k_proofs_lds, dim3(div_up(count, PROOFS_PER_BLOCK)), dim3(256), 0, stream, nodes.p, (uint32_t)depth,
                       magic, first, count, (u32x4_t*)d_elems, d_bits, bits_vec);
  } else {
    hipLaunchKernelGGL(k_proofs, dim3(div_up(total, 256)), dim3(256), 0, stream, nodes.p, depth, first, count,
                       (uint32_t*)d_elems, d_bits);
  }
  RLN_HIP(hipGetLastError());
}

void MerkleTreeDev::proof_host(size_t leaf, uint8_t* elems_le, uint8_t* bits) {
  if (leaf >= capacity()) throw Error("InvalidLeaf");
  if (depth == 0) return;
  // (the index bits follow from the leaf index -- full_merkle_tree.rs:296-300: 1 when the node on the path is a right
  // child; a hipMalloc per call cost more than the proof)
  if (proof_dev.n < (size_t)depth * 33 + 16) proof_dev.alloc((size_t)depth * 33 + 16);
  proofs_device(leaf, 1, proof_dev.p, proof_dev.p + (size_t)depth * 32);
  RLN_HIP(hipMemcpyAsync(elems_le, proof_dev.p, (size_t)depth * 32, hipMemcpyDeviceToHost, stream));
  RLN_HIP(hipStreamSynchronize(stream));
  for (int l = 0; l < depth; l++) bits[l] = (uint8_t)(((leaf + capacity()) >> l) & 1);
}

size_t MerkleTreeDev::verify_proofs_device(size_t first, size_t count, const uint8_t* d_elems, const uint8_t* d_bits) {
  if (count == 0) return 0;
  DevBuf<unsigned long long> bad(1);
  RLN_HIP(hipMemsetAsync(bad.p, 0, sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(k_verify_proofs, dim3(div_up(count, 256)), dim3(256), 0, stream, nodes.p, depth, first, count,
                     (const uint32_t*)d_elems, d_bits, poseidon_view(3), bad.p);
  unsigned long long h = 0;
  RLN_HIP(hipMemcpyAsync(&h, bad.p, sizeof(h), hipMemcpyDeviceToHost, stream));
  RLN_HIP(hipStreamSynchronize(stream));
  return (size_t)h;
}

}  // namespace rlnamd
