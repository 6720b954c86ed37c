// Variable-base MSM for large n (BASELINE config 5: one 2^24-point MSM, split across GPUs by point index) -- G1 in this unit,
// G2 in msm_g2.hip, the kernels and the host class in msm_impl.h (templated on the group).
//
// Replaces ark-ec 0.5.0 `VariableBaseMSM::msm_bigint` (third party; call sites
// /root/reference/rln/src/partial_proof.rs:98-104,255-256) for the case the fixed-base tables of prover.hip
// do not cover: bases that are not known in advance.  Classic Pippenger, laid out for a GPU:
//   1. k_digits      signed 16-bit digits of every scalar (16 windows), packed uint16
//   2. k_hist / k_tile_prefix / k_scan_*   per-(window, tile) bucket counts in LDS, prefix sums -> bucket offsets
//   3. k_part1 / k_part2   two-level counting sort: point indices grouped by (window, bucket)
//   4. k_slice_acc / k_slice_fix   equal slices of the sorted list, one per lane: mixed additions in the 9 x 29-bit
//                    limb form (fq29.h), pieces of a bucket joined afterwards     <- the multiplier-bound kernel
//   5. k_bucket_red  per (window, 32-bucket chunk): running-sum trick -> sum and weighted sum
//      k_chunk_fix   weighted sum + chunk_base * sum
//      k_range_sum   two levels of 32-to-1 wave trees over the chunks -> one point per window
//   6. the fold      adds the per-window sums of all contributing devices and runs Horner over the windows: 240 dependent
//                    doublings of ONE point -- on a host core (0.1 ms; k_combine, the same on a lone GPU lane, 1.75 ms)
// Multi-GPU: every rank runs 1-5 on its slice of the points; the 16 window sums (2 KB) are
// exchanged with one all-gather and step 6 runs on every rank.  RCCL has no elliptic-curve reduce op, so the
// "all-reduce of bucket partials" is realised as gather + local add.
#include "msm_impl.h"

namespace rlnamd {

// ---------------------------------------------------------------------------------------------------------------
// Self-test of the 9 x 29-bit group law (fq29.h) against the 8 x 32-bit one (curve.h) on the device: every thread
// walks the same pseudo-random sequence of signed table points with both accumulators, including a repeated point
// (doubling), a point followed by its negation (cancellation) and additions onto infinity.  Used by
// rlnamd_selftest_fq29 (tests/test_gpu_parity.py); `bad` counts threads whose affine results differ.
template <class Aff, class XY, class Aff29, class Acc29>
__global__ void __launch_bounds__(64) k_selftest29(const Aff* __restrict__ pts, const Aff29* __restrict__ pts29,
                                                   uint32_t npts, uint32_t iters, uint32_t* __restrict__ bad) {
  uint32_t t = (blockIdx.x * 64 + threadIdx.x) / Acc29::LPP;   // (G2AccPair29: two adjacent lanes run the same sequence)
  XY a = XY::inf(), sa = XY::inf();
  Acc29 b = Acc29::inf(), sb = Acc29::inf();
  uint32_t st = t * 2654435761u + 12345u;
  for (uint32_t i = 0; i < iters; i++) {
    st = st * 1664525u + 1013904223u;
    uint32_t k = (st >> 8) % npts;
    bool neg = (st >> 7) & 1;
    if (i % 7 == 3) k = (t + i) % npts;            // then again at i % 7 == 4: same point twice
    if (i % 7 == 4) { k = (t + i - 1) % npts; neg = (t >> 1) & 1; }
    if (i % 11 == 5 && (t & 3) == 0) { a = XY::inf(); b = Acc29::inf(); }   // restart from infinity
    Aff p = pts[k];
    if (neg) p.y = p.y.neg();
    a.madd(p);
    b.madd(pts29[k], neg);
    // general additions (Acc29::add, the small batches' sum trees): a snapshot of the walk is added back in later -- and once
    // straight away, which is the doubling case of the general law
    if (i % 9 == 2) { sa = a; sb = b; }
    if (i % 9 == 6 || (i % 27 == 2 && (t & 1))) { a.add(sa); b.add(sb); }
    if (i % 13 == 7) {                              // a point and its negation back to back: cancels to the previous sum
      Aff q = pts[(k + 1) % npts];
      a.madd(q);
      b.madd(pts29[(k + 1) % npts], false);
      q.y = q.y.neg();
      a.madd(q);
      b.madd(pts29[(k + 1) % npts], true);
    }
  }
  XY c = b.to_xyzz();
  Aff x = a.to_affine(), y = c.to_affine();
  if (!(x.x == y.x) || !(x.y == y.y)) atomicAdd(bad, 1u);
  // the pair form's own ways in and out (one component per lane): through memory and back
  if (Acc29::LPP == 2) {
    __shared__ XY box[32];
    XY* slot = &box[threadIdx.x / 2];
    b.store_xyzz(slot);
    __builtin_amdgcn_wave_barrier();
    Acc29 b2 = Acc29::load_xyzz(slot);
    Aff z = b2.to_xyzz().to_affine();
    if (!(x.x == z.x) || !(x.y == z.y)) atomicAdd(bad, 1u);
  }
}
// G1AccPair29 (the lane-pair general addition of the small batches' G1 sum trees) against G1Acc29::add: sums of a few
// table points added to each other, to themselves (doubling), to their negatives (cancellation) and to infinity, and the
// pair form's one-coordinate-pair-per-lane loads and stores
__global__ void __launch_bounds__(64) k_selftest_pair_g1(const G1Affine29* __restrict__ pts29, uint32_t npts, uint32_t* __restrict__ bad) {
  const uint32_t t = (blockIdx.x * 64 + threadIdx.x) / 2;
  uint32_t st = t * 2654435761u + 777u;
  auto next = [&]() { st = st * 1664525u + 1013904223u; return (st >> 8) % npts; };
  G1Acc29 u = G1Acc29::inf(), v = G1Acc29::inf();
  const uint32_t k1 = next(), k2 = next(), k3 = next(), k4 = next(), k5 = next();
  u.madd(pts29[k1], false);
  u.madd(pts29[k2], true);
  u.madd(pts29[k3], false);
  const uint32_t kind = t % 5;
  if (kind == 0) { v.madd(pts29[k4], false); v.madd(pts29[k5], true); }                                      // generic
  else if (kind == 1) { v.madd(pts29[k1], false); v.madd(pts29[k2], true); v.madd(pts29[k3], false); }     // v = u: doubling
  else if (kind == 2) { v.madd(pts29[k1], true); v.madd(pts29[k2], false); v.madd(pts29[k3], true); }      // v = -u: cancellation
  else if (kind == 3) { u = G1Acc29::inf(); v.madd(pts29[k4], false); }                                      // infinity + v
  // kind 4: u + infinity
  const G1XYZZ ux = u.to_xyzz(), vx = v.to_xyzz();
  G1AccPair29 pu = G1AccPair29::from_xyzz(ux), pv = G1AccPair29::from_xyzz(vx);
  u.add(v);
  pu.add(pv);
  const G1Affine want = u.to_xyzz().to_affine(), got = pu.to_xyzz().to_affine();
  if (!(want.x == got.x) || !(want.y == got.y)) atomicAdd(bad, 1u);
  __shared__ G1XYZZ box[32];
  __shared__ G1AccPair29 lbox[32];
  pu.store_xyzz(&box[threadIdx.x / 2]);
  pu.store_lds(&lbox[threadIdx.x / 2]);
  __builtin_amdgcn_wave_barrier();
  const G1Affine a1 = G1AccPair29::load_xyzz(&box[threadIdx.x / 2]).to_xyzz().to_affine();
  const G1Affine a2 = lbox[threadIdx.x / 2].to_xyzz().to_affine();
  if (!(want.x == a1.x) || !(want.y == a1.y) || !(want.x == a2.x) || !(want.y == a2.y)) atomicAdd(bad, 1u);
}
template <class Aff, class Aff29>
__global__ void k_selftest_conv(const Aff* __restrict__ src, Aff29* __restrict__ dst, uint32_t n) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = to_table29(src[t]);
}
template <class F>
__global__ void k_selftest_points(Affine<F> g, Affine<F>* __restrict__ out, uint32_t n) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint32_t k[8] = {t * 2654435761u + 1u, t ^ 0x9e3779b9u, t + 17u, 0, 0, 0, 0, 0};
  out[t] = scalar_mul(g, k).to_affine();
}

// returns the number of mismatching threads (0 = the two representations agree); group 1 = G1, 2 = G2, 3 = G2 by lane pairs, 4 = G1 general additions by lane pairs
uint32_t selftest_fq29(int group, uint32_t threads, uint32_t iters, const uint8_t* g2_gen_xy_le) {
  require_gpu();
  const uint32_t NP = 1024;
  DevBuf<uint32_t> bad(1);
  RLN_HIP(hipMemset(bad.p, 0, 4));
  threads = (threads + 63) / 64 * 64;
  if (group == 1 || group == 4) {
    DevBuf<G1Affine> p(NP);
    DevBuf<G1Affine29> p29(NP);
    G1Affine g{Fq::from_u32(1), Fq::from_u32(2)};
    hipLaunchKernelGGL(k_selftest_points<Fq>, dim3(NP / 64), dim3(64), 0, 0, g, p.p, NP);
    hipLaunchKernelGGL((k_selftest_conv<G1Affine, G1Affine29>), dim3(NP / 64), dim3(64), 0, 0, p.p, p29.p, NP);
    if (group == 4)   // the lane-pair general addition (fq29.h: G1AccPair29)
      hipLaunchKernelGGL(k_selftest_pair_g1, dim3(2 * threads / 64), dim3(64), 0, 0, p29.p, NP, bad.p);
    else
    hipLaunchKernelGGL((k_selftest29<G1Affine, G1XYZZ, G1Affine29, G1Acc29>), dim3(threads / 64), dim3(64), 0, 0, p.p,
                       p29.p, NP, iters, bad.p);
  } else {
    G2Affine g;
    uint32_t c[4][8];
    memcpy(c, g2_gen_xy_le, 128);
    g.x = {Fq::from_canonical(c[0]), Fq::from_canonical(c[1])};
    g.y = {Fq::from_canonical(c[2]), Fq::from_canonical(c[3])};
    DevBuf<G2Affine> p(NP);
    DevBuf<G2Affine29> p29(NP);
    hipLaunchKernelGGL(k_selftest_points<Fq2>, dim3(NP / 64), dim3(64), 0, 0, g, p.p, NP);
    hipLaunchKernelGGL((k_selftest_conv<G2Affine, G2Affine29>), dim3(NP / 64), dim3(64), 0, 0, p.p, p29.p, NP);
    if (group == 3)   // the lane-pair form of the small batches' G2 chain (fq29.h: Fq2PairOps)
      hipLaunchKernelGGL((k_selftest29<G2Affine, G2XYZZ, G2Affine29, G2AccPair29>), dim3(2 * threads / 64), dim3(64), 0, 0, p.p,
                         p29.p, NP, iters, bad.p);
    else
    hipLaunchKernelGGL((k_selftest29<G2Affine, G2XYZZ, G2Affine29, G2Acc29>), dim3(threads / 64), dim3(64), 0, 0, p.p,
                       p29.p, NP, iters, bad.p);
  }
  RLN_HIP(hipGetLastError());
  uint32_t h = 0;
  RLN_HIP(hipMemcpy(&h, bad.p, 4, hipMemcpyDeviceToHost));
  return h;
}

struct MsmG1::Impl : MsmImpl<MsmOpsG1> {
  using MsmImpl<MsmOpsG1>::MsmImpl;
};
RLN_MSM_WRAPPERS(MsmG1, MsmOpsG1)

}  // namespace rlnamd
