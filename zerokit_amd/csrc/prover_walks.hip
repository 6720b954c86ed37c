// prover_walks.hip -- the fixed-base tables and their walks: table build (8 x 32 slabs converted into the packed 9 x 29
// form), the walk kernel k_msm29 (body: walk29_impl.h) in its four launched forms, and the reductions of the partial sums.
#include "prover_kernels.h"

#include "glv.h"
#include "walk29_impl.h"

namespace rlnamd {

// =====================================================================================================
// 5. table-driven MSM: acc += +-T[point][window][|digit|-1]
// =====================================================================================================
// slab of 8 x 32 rows -> the packed 9 x 29 entries in place.  Points below `npaired` (global index k0 + local index) are
// pair members (walk29.h ROW_PAIRED): entry x of point 2 q + m goes to (2 q) stride + 2 x + m.
template <class A, class E>
__global__ void __launch_bounds__(256) k_table_to29(const A* __restrict__ src, E* __restrict__ dst, size_t n, uint32_t stride,
                                                    uint32_t k0, uint32_t npaired) {
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const uint32_t kl = (uint32_t)(t / stride), x = (uint32_t)(t % stride), k = k0 + kl;
  size_t o = t;   // dst points at the slab's first point (k0, even)
  if (k < npaired) o = (size_t)(kl & ~1u) * stride + 2 * (size_t)x + (kl & 1u);
  dst[o] = to_table29(src[t]);
}

// dst[r][p] = sum of src[i][p] over ranges[r] -- used twice (chunks -> groups -> segments) so the
// per-proof reduction is a two-level tree instead of one long serial chain
template <class F>
__global__ void __launch_bounds__(64) k_sum_ranges(const XYZZ<F>* __restrict__ src, const ChunkDesc* __restrict__ ranges,
                                                   uint32_t nranges, XYZZ<F>* __restrict__ dst, uint32_t B, uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  uint32_t r = blockIdx.y;
  if (p >= nb || r >= nranges) return;
  ChunkDesc cd = ranges[r];
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t i = cd.pt_begin; i < cd.pt_end; i++) acc.add(src[(size_t)i * B + p]);
  dst[(size_t)r * B + p] = acc;
}

// The same reduction for small batches, lanes = partial sums instead of lanes = proofs: with one proof in the batch
// k_sum_ranges leaves 63 lanes idle and a G2 segment is a serial chain of 16 + 30 general additions (~ 2 ms); here the
// 64 lanes of the wave of (proof, segment) each add their share of the segment's chunks and meet in a six-level tree
// through LDS: 8 + 6 additions.
// which segments / tasks a launch covers (grid.y = n): a small batch finishes s A and r B1 from the h-independent rows
// while the h rows are still being walked, so the back-end kernels run twice on disjoint task lists
// (the tree adds in the 9 x 29 form of fq29.h -- G1Acc29::add / G2Acc29::add: a level is one general addition of a lone
// lane, 14 field products, and that form needs 0.55 x the instructions of the 8 x 32 one; partial sums are converted once on
// the way in and once on the way out, and cross the levels through LDS as they are: 144 / 288 bytes per point, dynamic LDS)
template <class F> struct Acc29Of;
template <> struct Acc29Of<Fq> { typedef G1Acc29 type; };
template <> struct Acc29Of<Fq2> { typedef G2Acc29 type; };
extern __shared__ uint4 sum_tree_lds[];

// Acc = Acc29Of<F>::type (a lane per point) or, for G2, G2AccPair29: a lane PAIR per point (fq29.h: Fq2PairOps) -- half
// the points per workgroup, each addition 1.65 x shorter; what a lone proof's G2 chain wants.
template <class F, class Acc>
__global__ void __launch_bounds__(SUM_TREE_LANES) k_sum_tree(const XYZZ<F>* __restrict__ part, const ChunkDesc* __restrict__ segchunks,
                                                  XYZZ<F>* __restrict__ dst, uint32_t B, uint32_t PB, TaskSel sel) {
  // 512 lanes per (proof, segment): the short chunks of the small-batch plans leave ~2 000 partial sums per segment;
  // four per lane and a nine-level tree (part stride PB, result stride B).  The additions are a dependent chain for the
  // lone waves of a single proof, so the lane count is what sets the kernel's length: 256 lanes were 8 + 8 additions.
  // Only the upper half of a level passes through LDS.
  constexpr uint32_t NP = SUM_TREE_LANES / Acc::LPP;   // points in flight
  Acc* sh = reinterpret_cast<Acc*>(sum_tree_lds);
  __builtin_amdgcn_s_setprio(3);
  const uint32_t p = blockIdx.x, sgi = sel.id[blockIdx.y], l = threadIdx.x / Acc::LPP;
  const ChunkDesc cd = segchunks[sgi];
  Acc acc = Acc::inf();
  for (uint32_t i = cd.pt_begin + l; i < cd.pt_end; i += NP) acc.add(Acc::load_xyzz(&part[(size_t)i * PB + p]));
#pragma unroll 1
  for (uint32_t stride = NP / 2; stride >= 1; stride >>= 1) {
    if (l >= stride && l < 2 * stride) acc.store_lds(&sh[l - stride]);
    __syncthreads();
    if (l < stride) acc.add(sh[l]);   // (G2: read from LDS coordinate by coordinate, see G2AccT::add)
    __syncthreads();
  }
  if (l == 0) acc.store_xyzz(&dst[(size_t)sgi * B + p]);
}

// First stage of the two-stage sum (tiny batches): block z of segment sel.id[y] -- NP consecutive partial sums -- to
// one point, dst[(segblocks[seg].begin + z) * PB + p]; k_sum_tree over dst (segment ranges = segblocks) is the second.
template <class F, class Acc>
__global__ void __launch_bounds__(SUM_TREE_LANES) k_sum_blocks(const XYZZ<F>* __restrict__ part, const ChunkDesc* __restrict__ segchunks,
                                                    const ChunkDesc* __restrict__ segblocks, XYZZ<F>* __restrict__ dst, uint32_t PB,
                                                    TaskSel sel) {
  constexpr uint32_t NP = SUM_TREE_LANES / Acc::LPP;
  Acc* sh = reinterpret_cast<Acc*>(sum_tree_lds);
  __builtin_amdgcn_s_setprio(3);
  const uint32_t p = blockIdx.x, sgi = sel.id[blockIdx.y], l = threadIdx.x / Acc::LPP;
  const ChunkDesc cd = segchunks[sgi], bd = segblocks[sgi];
  if (blockIdx.z >= bd.pt_end - bd.pt_begin) return;   // (uniform for the workgroup)
  const uint32_t i = cd.pt_begin + blockIdx.z * NP + l;
  Acc acc = Acc::inf();
  if (i < cd.pt_end) acc = Acc::load_xyzz(&part[(size_t)i * PB + p]);
#pragma unroll 1
  for (uint32_t stride = NP / 2; stride >= 1; stride >>= 1) {
    if (l >= stride && l < 2 * stride) acc.store_lds(&sh[l - stride]);
    __syncthreads();
    if (l < stride) acc.add(sh[l]);   // (G2: read from LDS coordinate by coordinate, see G2AccT::add)
    __syncthreads();
  }
  if (l == 0) acc.store_xyzz(&dst[(size_t)(bd.pt_begin + blockIdx.z) * PB + p]);
}

// GLV: segment t holds sum k1_i P_i, segment nseg + t holds sum k2_i P_i; the result is the first plus phi of the
// second, phi(X, Y, ZZ, ZZZ) = (beta X, Y, ZZ, ZZZ) (x = X / ZZ).  One Fq product per output point and proof.
__global__ void __launch_bounds__(64) k_glv_fold(G1XYZZ* __restrict__ sums1, G2XYZZ* __restrict__ sums2, uint32_t nseg1,
                                                 uint32_t B, uint32_t nb, TaskSel sel) {
  __builtin_amdgcn_s_setprio(3);
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t t = sel.id[blockIdx.y];
  if (t < nseg1) {
    G1XYZZ a = sums1[(size_t)t * B + p], b = sums1[(size_t)(nseg1 + t) * B + p];
    b.X = b.X * Fq::from_canonical(GlvParams::BETA_G1);
    a.add(b);
    sums1[(size_t)t * B + p] = a;
  } else {
    G2XYZZ a = sums2[p], b = sums2[(size_t)B + p];
    b.X = b.X.mul_fq(Fq::from_canonical(GlvParams::BETA_G2));
    a.add(b);
    sums2[p] = a;
  }
}

// one-time comb table: row (k, j) = { d * 2^(c j) * P_k : d = 1..2^(c-1) } in affine form.
// Built by doubling the known prefix (multiples 1..m -> m+1..2m are "T[i] + T[m]" and one doubling) with ONE inversion
// per level (Montgomery's trick; prefix products parked in `scratch`).  A WAVE per row (round 4; a lane per row left half
// the SIMDs idle and made every level a serial chain of up to 16 384 products: 4.6 s for the 228 GiB schedule): the m
// additions of a level are dealt to the lanes round-robin (item i -> lane (i - 1) mod 64: coalesced reads and writes),
// every lane multiplies up the denominators of its own items, lane 0 inverts the 64 lane totals with the same trick
// (189 products + 1 inversion), and every lane walks its items backwards with the inverse of its own total.  Which
// denominators share an inversion does not change a quotient: the table entries are the same field elements.
template <class F>
__global__ void __launch_bounds__(64) k_table_build(const Affine<F>* __restrict__ pts, uint32_t npts, WinSched ws,
                                                    Affine<F>* __restrict__ table, F* __restrict__ scratch) {
  __shared__ F tot[64], itot[64];
  const uint32_t W = (uint32_t)ws.W, l = threadIdx.x;
  const size_t r = blockIdx.x;
  if (r >= (size_t)npts * W) return;   // (uniform)
  const uint32_t k = (uint32_t)(r / W), j = (uint32_t)(r % W);
  const uint32_t E = 1u << (ws.cw[j] - 1);
  const size_t off = (size_t)k * ws.stride + ws.ro[j];  // even: every row has >= 2 entries (cw >= 2)
  Affine<F>* row = table + off;
  F* pre = scratch + off / 2;
  if (l == 0) {
    XYZZ<F> b = XYZZ<F>::from_affine(pts[k]);
    for (uint32_t i = 0; i < (uint32_t)ws.bo[j]; i++) b = b.dbl();
    row[0] = b.to_affine();
  }
  __threadfence_block();
  __syncthreads();
  for (uint32_t m = 1; m < E; m <<= 1) {
    const Affine<F> Pm = row[m - 1];
    // items i = 1 .. m: T[m + i - 1] = T[i - 1] + T[m - 1] (i < m), T[2 m - 1] = 2 T[m - 1] (i = m)
    F run = F::one();
    for (uint32_t i = 1 + l; i <= m; i += 64) {
      const F den = (i < m) ? (row[i - 1].x - Pm.x) : Pm.y.dbl();
      pre[i - 1] = run;
      run = run * den;
    }
    tot[l] = run;
    __syncthreads();
    if (l == 0) {   // 1 / tot[q] for all 64 lanes: one inversion
      F acc = F::one();
      for (uint32_t q = 0; q < 64; q++) {
        itot[q] = acc;          // product of the totals before q
        acc = acc * tot[q];
      }
      F inv = acc.inv();
      for (uint32_t q = 64; q-- > 0;) {
        const F t = tot[q];
        itot[q] = inv * itot[q];
        inv = inv * t;
      }
    }
    __syncthreads();
    F inv = itot[l];
    const uint32_t cnt = m > l ? (m - 1 - l) / 64 + 1 : 0;   // items of this lane: l + 1, l + 65, ...
    for (uint32_t t = cnt; t-- > 0;) {
      const uint32_t i = 1 + l + 64 * t;
      F den, lam, x3, y3;
      if (i < m) {
        const Affine<F> Pi = row[i - 1];
        den = Pi.x - Pm.x;
        const F di = inv * pre[i - 1];
        lam = (Pi.y - Pm.y) * di;
        x3 = lam.sqr() - Pi.x - Pm.x;
        y3 = lam * (Pi.x - x3) - Pi.y;
      } else {
        den = Pm.y.dbl();
        const F di = inv * pre[i - 1];
        const F x2 = Pm.x.sqr();
        lam = (x2.dbl() + x2) * di;
        x3 = lam.sqr() - Pm.x.dbl();
        y3 = lam * (Pm.x - x3) - Pm.y;
      }
      inv = inv * den;
      row[m + i - 1] = {x3, y3};
    }
    __threadfence_block();   // the next level reads what this one wrote (other lanes' entries)
    __syncthreads();
  }
}


// ---- explicit instantiations: every form the host launches
template __global__ void k_table_to29<G1Affine, G1Affine29>(const G1Affine* __restrict__ src, G1Affine29* __restrict__ dst, size_t n, uint32_t stride, uint32_t k0, uint32_t npaired);
template __global__ void k_table_to29<G2Affine, G2Affine29>(const G2Affine* __restrict__ src, G2Affine29* __restrict__ dst, size_t n, uint32_t stride, uint32_t k0, uint32_t npaired);
template __global__ void k_sum_ranges<Fq>(const XYZZ<Fq>* __restrict__ src, const ChunkDesc* __restrict__ ranges, uint32_t nranges, XYZZ<Fq>* __restrict__ dst, uint32_t B, uint32_t nb);
template __global__ void k_sum_ranges<Fq2>(const XYZZ<Fq2>* __restrict__ src, const ChunkDesc* __restrict__ ranges, uint32_t nranges, XYZZ<Fq2>* __restrict__ dst, uint32_t B, uint32_t nb);
template __global__ void k_sum_tree<Fq, G1Acc29>(const XYZZ<Fq>* __restrict__ part, const ChunkDesc* __restrict__ segchunks, XYZZ<Fq>* __restrict__ dst, uint32_t B, uint32_t PB, TaskSel sel);
template __global__ void k_sum_tree<Fq, G1AccPair29>(const XYZZ<Fq>* __restrict__ part, const ChunkDesc* __restrict__ segchunks, XYZZ<Fq>* __restrict__ dst, uint32_t B, uint32_t PB, TaskSel sel);
template __global__ void k_sum_tree<Fq2, G2Acc29>(const XYZZ<Fq2>* __restrict__ part, const ChunkDesc* __restrict__ segchunks, XYZZ<Fq2>* __restrict__ dst, uint32_t B, uint32_t PB, TaskSel sel);
template __global__ void k_sum_blocks<Fq, G1Acc29>(const XYZZ<Fq>* __restrict__ part, const ChunkDesc* __restrict__ segchunks, const ChunkDesc* __restrict__ segblocks, XYZZ<Fq>* __restrict__ dst, uint32_t PB, TaskSel sel);
template __global__ void k_sum_blocks<Fq, G1AccPair29>(const XYZZ<Fq>* __restrict__ part, const ChunkDesc* __restrict__ segchunks, const ChunkDesc* __restrict__ segblocks, XYZZ<Fq>* __restrict__ dst, uint32_t PB, TaskSel sel);
template __global__ void k_sum_blocks<Fq2, G2Acc29>(const XYZZ<Fq2>* __restrict__ part, const ChunkDesc* __restrict__ segchunks, const ChunkDesc* __restrict__ segblocks, XYZZ<Fq2>* __restrict__ dst, uint32_t PB, TaskSel sel);
template __global__ void k_sum_tree<Fq2, G2AccPair29>(const XYZZ<Fq2>* __restrict__ part, const ChunkDesc* __restrict__ segchunks, XYZZ<Fq2>* __restrict__ dst, uint32_t B, uint32_t PB, TaskSel sel);
template __global__ void k_sum_blocks<Fq2, G2AccPair29>(const XYZZ<Fq2>* __restrict__ part, const ChunkDesc* __restrict__ segchunks, const ChunkDesc* __restrict__ segblocks, XYZZ<Fq2>* __restrict__ dst, uint32_t PB, TaskSel sel);
template __global__ void k_table_build<Fq>(const Affine<Fq>* __restrict__ pts, uint32_t npts, WinSched ws, Affine<Fq>* __restrict__ table, Fq* __restrict__ scratch);
template __global__ void k_table_build<Fq2>(const Affine<Fq2>* __restrict__ pts, uint32_t npts, WinSched ws, Affine<Fq2>* __restrict__ table, Fq2* __restrict__ scratch);
template __global__ void k_msm29<G1Acc29, G1Affine29, G1XYZZ, 4, false>(const G1Affine29* __restrict__ table, const uint32_t* __restrict__ sid, const uint32_t* __restrict__ rows, const ChunkDesc* __restrict__ chunks, uint32_t nchunks, const int16_t* __restrict__ digits, G1XYZZ* __restrict__ part, WinSched ws, uint32_t B, uint32_t pgroups, uint32_t nh, unsigned long long* __restrict__ clk, const uint32_t* __restrict__ chunk_ids, uint32_t pstride, PairPlan pairs);
template __global__ void k_msm29<G1Acc29, G1Affine29, G1XYZZ, 2, true>(const G1Affine29* __restrict__ table, const uint32_t* __restrict__ sid, const uint32_t* __restrict__ rows, const ChunkDesc* __restrict__ chunks, uint32_t nchunks, const int16_t* __restrict__ digits, G1XYZZ* __restrict__ part, WinSched ws, uint32_t B, uint32_t pgroups, uint32_t nh, unsigned long long* __restrict__ clk, const uint32_t* __restrict__ chunk_ids, uint32_t pstride, PairPlan pairs);
template __global__ void k_msm29<G2Acc29, G2Affine29, G2XYZZ, 2, false>(const G2Affine29* __restrict__ table, const uint32_t* __restrict__ sid, const uint32_t* __restrict__ rows, const ChunkDesc* __restrict__ chunks, uint32_t nchunks, const int16_t* __restrict__ digits, G2XYZZ* __restrict__ part, WinSched ws, uint32_t B, uint32_t pgroups, uint32_t nh, unsigned long long* __restrict__ clk, const uint32_t* __restrict__ chunk_ids, uint32_t pstride, PairPlan pairs);
template __global__ void k_msm29<G2Acc29, G2Affine29, G2XYZZ, 1, true>(const G2Affine29* __restrict__ table, const uint32_t* __restrict__ sid, const uint32_t* __restrict__ rows, const ChunkDesc* __restrict__ chunks, uint32_t nchunks, const int16_t* __restrict__ digits, G2XYZZ* __restrict__ part, WinSched ws, uint32_t B, uint32_t pgroups, uint32_t nh, unsigned long long* __restrict__ clk, const uint32_t* __restrict__ chunk_ids, uint32_t pstride, PairPlan pairs);
template __global__ void k_msm29<G2AccPair29, G2Affine29, G2XYZZ, 1, true>(const G2Affine29* __restrict__ table, const uint32_t* __restrict__ sid, const uint32_t* __restrict__ rows, const ChunkDesc* __restrict__ chunks, uint32_t nchunks, const int16_t* __restrict__ digits, G2XYZZ* __restrict__ part, WinSched ws, uint32_t B, uint32_t pgroups, uint32_t nh, unsigned long long* __restrict__ clk, const uint32_t* __restrict__ chunk_ids, uint32_t pstride, PairPlan pairs);

}  // namespace rlnamd
