// Experiment: batch-affine additions against the XYZZ chain of the fixed-base walk (k_msm29<G1>).
//
// The walk adds one table entry per step into a lane's XYZZ accumulator: 8 M + 2 S per mixed addition, a dependent chain.
// The alternative priced (never measured) in round 2: sum a lane's m picks as a binary tree of AFFINE additions,
//     lambda = (yQ - yP) / (xQ - xP),  xR = lambda^2 - xP - xQ,  yR = lambda (xP - xR) - yP,
// with the level's n / 2 denominators inverted together (Montgomery's trick: 3 M per element and ONE inversion per lane
// and level): 5 M + 1 S per addition.  The prefix products and the level's results do not fit registers or LDS for the
// n that amortises an inversion (hundreds), so they go through HBM, lanes interleaved (a wave's access is contiguous).
//
//   forward   t = 0 .. n/2-1:  pre[t] = run;  run *= delta_t                     (1 M, 48 B written)
//   invert    inv = 1 / run                                                        (modinv30.h: ~13 k instructions)
//   backward  t = n/2-1 .. 0:  li = inv pre[t];  inv *= delta_t;  lambda = nu_t li;  xR, yR   (4 M + 1 S, 80 B written)
//
// Every case of the group law is handled as the product would have to: an operand at infinity, P = Q (the tangent:
// delta = 2 y, nu = 3 x^2 go through the same chain), P = -Q (the result is infinity; delta = 1).  Results are brought
// back below 2 q after each addition (x and y are inputs of the next level's subtractions, unlike the XYZZ chain whose
// subtrahends are fresh products).  The last TAIL points of a lane are added as the walk would (XYZZ).
//
// The table holds valid curve points (the tree and the chain associate differently: only on the curve do they agree), and
// the two kernels' sums are compared lane by lane on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=400000 -I zerokit_amd/csrc \
//        tools/microbench_batch_affine.hip -o tools/microbench_batch_affine
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "fq29.h"
using namespace rlnamd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

namespace {
constexpr uint32_t K6D[9] = {0x52edefaau, 0x461a4446u, 0x4aafd3d8u, 0x50fed0e3u, 0x412318ceu, 0x51238482u, 0x43e94784u, 0x5628e536u, 0x012259d4u};  // 6 q, limbs >= 2^30
constexpr uint32_t Z7[9] = {0xe0000000u, 0xdffffff9u, 0xdffffff9u, 0xdffffff9u, 0xdffffff9u, 0xdffffff9u, 0xdffffff9u, 0xdffffff9u, 0xfffffff9u};   // 0, limbs >= 7 2^29 - 7
constexpr uint32_t QTOP_MAGIC = 0x54au;   // floor(2^32 / (q >> 232 + 1))

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void k_make_table(const G1Affine* __restrict__ valid, uint32_t nvalid, G1Affine29* __restrict__ table, size_t nt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nt) return;
  table[i] = to_table29(valid[mix64(i * 0x9E3779B97F4A7C15ull + 77) % nvalid]);
}

// the pick of lane gl at leaf t: a table index and a sign (as the signed window digits give)
__device__ __forceinline__ void pick(uint64_t gl, uint32_t m, uint32_t t, size_t nt_mask, size_t* idx, bool* neg) {
  const uint64_t h = mix64((gl * m + t) * 0x9E3779B97F4A7C15ull + 0x1234567);
  *idx = (size_t)(h >> 8) & nt_mask;
  *neg = (h & 1) != 0;
}

// ---- the chain: what k_msm29<G1> does per step --------------------------------------------------------------------------
template <int WAVES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_chain(const G1Affine29* __restrict__ table, size_t nt_mask, uint32_t m, G1XYZZ* __restrict__ out) {
  const uint64_t gl = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  G1Acc29 acc = G1Acc29::inf();
#pragma unroll 1
  for (uint32_t t = 0; t < m; t++) {
    size_t idx;
    bool neg;
    pick(gl, m, t, nt_mask, &idx, &neg);
    acc.madd(table[idx], neg);
  }
  out[gl] = acc.to_xyzz();
}

// ---- the tree -------------------------------------------------------------------------------------------------------------
struct Pt {   // affine, Fq29: x, y normalised and < 2.1 q; inf != 0: the point at infinity
  Fq29 x, y;
  uint32_t inf;
};
// a (normalised, < 8 q) -> the same residue, normalised, < 2 q: k = floor(a / q) or one less from the top limb
__device__ __forceinline__ Fq29 below_2q(const Fq29& a) {
  const uint32_t k = __umulhi(a.v[8], QTOP_MAGIC);
  Fq29 r;
#pragma unroll
  for (int j = 0; j < 9; j++) r.v[j] = a.v[j] + Z7[j] - k * Fq29C::P[j];
  r.normalize();
  return r;
}
struct Scratch {   // per launch: uint4 planes, lanes of a wave adjacent
  uint4* bufA;     // [wave][slot < m/2][5][64]
  uint4* bufB;     // [wave][slot < m/4][5][64]
  uint4* pre;      // [wave][slot < m/2][2][64] + [wave][slot < m/2][64] words behind them (36 B per value)
};
__device__ __forceinline__ void st_pt(uint4* buf, uint32_t slot, const Pt& p) {
  uint4* b = buf + (size_t)slot * 5 * 64 + threadIdx.x;
  b[0] = make_uint4(p.x.v[0], p.x.v[1], p.x.v[2], p.x.v[3]);
  b[64] = make_uint4(p.x.v[4], p.x.v[5], p.x.v[6], p.x.v[7]);
  b[128] = make_uint4(p.x.v[8], p.y.v[0], p.y.v[1], p.y.v[2]);
  b[192] = make_uint4(p.y.v[3], p.y.v[4], p.y.v[5], p.y.v[6]);
  b[256] = make_uint4(p.y.v[7], p.y.v[8], p.inf, 0u);
}
__device__ __forceinline__ Pt ld_pt(const uint4* buf, uint32_t slot) {
  const uint4* b = buf + (size_t)slot * 5 * 64 + threadIdx.x;
  const uint4 a = b[0], c = b[64], d = b[128], e = b[192], f = b[256];
  Pt p;
  p.x.v[0] = a.x, p.x.v[1] = a.y, p.x.v[2] = a.z, p.x.v[3] = a.w;
  p.x.v[4] = c.x, p.x.v[5] = c.y, p.x.v[6] = c.z, p.x.v[7] = c.w;
  p.x.v[8] = d.x, p.y.v[0] = d.y, p.y.v[1] = d.z, p.y.v[2] = d.w;
  p.y.v[3] = e.x, p.y.v[4] = e.y, p.y.v[5] = e.z, p.y.v[6] = e.w;
  p.y.v[7] = f.x, p.y.v[8] = f.y, p.inf = f.z;
  return p;
}
// x and the flag only (48 of the 80 bytes): what the forward pass needs unless the pair has one x
__device__ __forceinline__ Pt ld_pt_x(const uint4* buf, uint32_t slot) {
  const uint4* b = buf + (size_t)slot * 5 * 64 + threadIdx.x;
  const uint4 a = b[0], c = b[64], d = b[128];
  Pt p;
  p.x.v[0] = a.x, p.x.v[1] = a.y, p.x.v[2] = a.z, p.x.v[3] = a.w;
  p.x.v[4] = c.x, p.x.v[5] = c.y, p.x.v[6] = c.z, p.x.v[7] = c.w;
  p.x.v[8] = d.x;
  p.inf = ((const uint32_t*)(b + 256))[2];
  p.y = Fq29::zero();
  return p;
}
__device__ __forceinline__ void st_fq(uint4* buf, uint32_t half, uint32_t slot, const Fq29& a) {
  uint4* b = buf + (size_t)slot * 2 * 64 + threadIdx.x;
  b[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  b[64] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
  ((uint32_t*)(buf + (size_t)half * 2 * 64))[(size_t)slot * 64 + threadIdx.x] = a.v[8];
}
__device__ __forceinline__ Fq29 ld_fq(const uint4* buf, uint32_t half, uint32_t slot) {
  const uint4* b = buf + (size_t)slot * 2 * 64 + threadIdx.x;
  const uint4 a = b[0], c = b[64];
  const uint32_t d = ((const uint32_t*)(buf + (size_t)half * 2 * 64))[(size_t)slot * 64 + threadIdx.x];
  Fq29 r;
  r.v[0] = a.x, r.v[1] = a.y, r.v[2] = a.z, r.v[3] = a.w;
  r.v[4] = c.x, r.v[5] = c.y, r.v[6] = c.z, r.v[7] = c.w;
  r.v[8] = d;
  return r;
}
__device__ __forceinline__ Pt leaf(const G1Affine29* __restrict__ table, uint64_t gl, uint32_t m, uint32_t t, size_t nt_mask) {
  size_t idx;
  bool neg;
  pick(gl, m, t, nt_mask, &idx, &neg);
  const G1Affine29 e = table[idx];
  Pt p;
  p.x = unpack29(e.x);
  p.y = unpack29(e.y);
  if (neg) {
    p.y = Fq29::neg_lazy(Fq29C::K2, p.y);
    p.y.normalize();
  }
  p.inf = 0;
  return p;
}
// the denominator and numerator of the pair's slope; kind: 0 the chord, 1 the tangent, 2 nothing to divide (an operand at
// infinity, or P = -Q)
__device__ __forceinline__ int slope(const Pt& P, const Pt& Q, Fq29* delta, Fq29* nu) {
  *delta = Fq29::sub(Q.x, Fq29C::K4, P.x);   // in (1.9 q, 6.1 q), normalised
  *nu = Fq29::sub(Q.y, Fq29C::K4, P.y);
  if (P.inf | Q.inf) {
    *delta = Fq29::from_const(Fq29C::ONE);
    return 2;
  }
  if (delta->is_zero_mod_q()) {
    if (nu->is_zero_mod_q()) {   // the tangent: 3 x^2 / 2 y
#pragma unroll
      for (int j = 0; j < 9; j++) delta->v[j] = 2 * P.y.v[j];
      delta->normalize();        // < 4.2 q
      const Fq29 x2 = Fq29::sqr(P.x);
#pragma unroll
      for (int j = 0; j < 9; j++) nu->v[j] = 3 * x2.v[j];
      nu->normalize();           // < 3.3 q
      return 1;
    }
    *delta = Fq29::from_const(Fq29C::ONE);
    return 2;
  }
  return 0;
}
__device__ __forceinline__ Pt combine(const Pt& P, const Pt& Q, int kind, const Fq29& nu, const Fq29& li) {
  if (kind == 2) {
    if (P.inf) return Q;
    if (Q.inf) return P;
    Pt r = P;
    r.inf = 1;
    return r;
  }
  const Fq29 lam = Fq29::mul(nu, li);          // < 1.05 q
  Fq29 kT;
#pragma unroll
  for (int j = 0; j < 9; j++) kT.v[j] = K6D[j] - (P.x.v[j] + Q.x.v[j]);
  Pt r;
  r.x = below_2q(Fq29::sqr_add(lam, &kT));     // lambda^2 - xP - xQ  (< 7.1 q before)
  Fq29 D;
#pragma unroll
  for (int j = 0; j < 9; j++) D.v[j] = P.x.v[j] + Fq29C::K4[j] - r.x.v[j];   // limbs < 3 2^29
  const Fq29 nY = Fq29::neg_lazy(Fq29C::K4, P.y);
  r.y = below_2q(Fq29::mul_add(lam, D, nY));   // lambda (xP - xR) - yP  (< 5.1 q before)
  r.inf = 0;
  return r;
}
__device__ __noinline__ Fq29 inverse29(Fq29 a) {   // a = z 2^261  ->  z^-1 2^261
  return Fq29::from_fq(a.to_fq().inv());
}

template <bool LEAF>
__device__ __forceinline__ void level(const G1Affine29* __restrict__ table, uint64_t gl, uint32_t m, size_t nt_mask,
                                      const uint4* __restrict__ src, uint4* __restrict__ dst, uint4* __restrict__ pre,
                                      uint32_t half, uint32_t smask) {
  Fq29 run = Fq29::from_const(Fq29C::ONE);
#pragma unroll 1
  for (uint32_t t = 0; t < half; t++) {
    Pt P = LEAF ? leaf(table, gl, m, 2 * t, nt_mask) : ld_pt_x(src, (2 * t) & smask);
    Pt Q = LEAF ? leaf(table, gl, m, 2 * t + 1, nt_mask) : ld_pt_x(src, (2 * t + 1) & smask);
    Fq29 delta, nu;
    if (!LEAF) {
      delta = Fq29::sub(Q.x, Fq29C::K4, P.x);
      if ((P.inf | Q.inf) || delta.is_zero_mod_q()) {   // rare: the whole points
        P = ld_pt(src, (2 * t) & smask);
        Q = ld_pt(src, (2 * t + 1) & smask);
        slope(P, Q, &delta, &nu);
      }
    } else {
      slope(P, Q, &delta, &nu);
    }
    st_fq(pre, half, t & smask, run);
    run = Fq29::mul(run, delta);
  }
  Fq29 inv = inverse29(run);
#pragma unroll 1
  for (uint32_t t = half; t-- > 0;) {
    const Pt P = LEAF ? leaf(table, gl, m, 2 * t, nt_mask) : ld_pt(src, (2 * t) & smask);
    const Pt Q = LEAF ? leaf(table, gl, m, 2 * t + 1, nt_mask) : ld_pt(src, (2 * t + 1) & smask);
    const Fq29 pr = ld_fq(pre, half, t & smask);
    Fq29 delta, nu;
    const int kind = slope(P, Q, &delta, &nu);
    const Fq29 li = Fq29::mul(inv, pr);
    inv = Fq29::mul(inv, delta);
    st_pt(dst, t & smask, combine(P, Q, kind, nu, li));
  }
}

template <int WAVES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_tree(const G1Affine29* __restrict__ table, size_t nt_mask, uint32_t m, uint32_t tail, Scratch S, G1XYZZ* __restrict__ out, uint32_t smask) {
  const uint64_t gl = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  uint4* A = S.bufA + (size_t)blockIdx.x * (m / 2) * 5 * 64;
  uint4* B = S.bufB + (size_t)blockIdx.x * (m / 4) * 5 * 64;
  uint4* pre = S.pre + (size_t)blockIdx.x * ((m / 2) * 2 * 64 + (m / 2) * 16);
  uint32_t n = m;
  level<true>(table, gl, m, nt_mask, nullptr, A, pre, n / 2, smask);
  n /= 2;
  uint4 *src = A, *dst = B;
  while (n > tail) {
    level<false>(table, gl, m, nt_mask, src, dst, pre, n / 2, smask);
    n /= 2;
    uint4* t = src;
    src = dst;
    dst = t;
  }
  G1Acc29 acc = G1Acc29::inf();
#pragma unroll 1
  for (uint32_t t = 0; t < n; t++) {
    const Pt p = ld_pt(src, t & smask);
    if (p.inf) continue;
    G1Affine29 e;
    pack29_reduced(p.x, e.x);
    pack29_reduced(p.y, e.y);
    acc.madd(e, false);
  }
  out[gl] = acc.to_xyzz();
}

G1Affine host_affine(const G1XYZZ& p) { return p.to_affine(); }
}  // namespace

template <int WAVES>
static int run(const G1Affine29* table, size_t nt_mask, uint32_t m, uint32_t tail, uint32_t waves, int reps, bool check,
               uint32_t smask = 0xFFFFFFFFu) {
  G1XYZZ *o_chain, *o_tree;
  const size_t lanes = (size_t)waves * 64;
  CK(hipMalloc(&o_chain, lanes * sizeof(G1XYZZ)));
  CK(hipMalloc(&o_tree, lanes * sizeof(G1XYZZ)));
  Scratch S;
  CK(hipMalloc(&S.bufA, (size_t)waves * (m / 2) * 5 * 64 * sizeof(uint4)));
  CK(hipMalloc(&S.bufB, (size_t)waves * (m / 4) * 5 * 64 * sizeof(uint4)));
  CK(hipMalloc(&S.pre, (size_t)waves * ((m / 2) * 2 * 64 + (m / 2) * 16) * sizeof(uint4)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms_chain = 1e30f, ms_tree = 1e30f;
  for (int r = 0; r < reps; r++) {
    float ms;
    CK(hipEventRecord(e0));
    k_chain<4><<<waves, 64>>>(table, nt_mask, m, o_chain);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < ms_chain) ms_chain = ms;
    CK(hipEventRecord(e0));
    k_tree<WAVES><<<waves, 64>>>(table, nt_mask, m, tail, S, o_tree, smask);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < ms_tree) ms_tree = ms;
  }
  CK(hipDeviceSynchronize());
  const double adds = (double)lanes * m;
  // bytes per lane: level 0 two 64-byte gathers per leaf (forward and backward), 36 + 36 B of prefix and 80 B of result per
  // addition; the levels above 2 x 48 B forward, 2 x 80 B backward, the same prefix and result
  const double l0 = (double)m / 2, up = (double)m / 2 - tail;
  const double tree_bytes = (double)lanes * (l0 * (4 * 64 + 72 + 80) + up * (96 + 160 + 72 + 80));
  int bad = 0, inf = 0;
  if (check) {
    const size_t nchk = lanes < 512 ? lanes : 512;
    std::vector<G1XYZZ> hc(nchk), ht(nchk);
    CK(hipMemcpy(hc.data(), o_chain, nchk * sizeof(G1XYZZ), hipMemcpyDeviceToHost));
    CK(hipMemcpy(ht.data(), o_tree, nchk * sizeof(G1XYZZ), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nchk; i++) {
      const G1Affine a = host_affine(hc[i]), b = host_affine(ht[i]);
      if (a.is_inf()) inf++;
      if (!(a.x == b.x) || !(a.y == b.y)) bad++;
    }
  }
  printf("{\"scratch_slots_aliased\": %s, \"waves_per_simd_tree\": %d, \"waves\": %u, \"m\": %u, \"tail\": %u, \"chain_ms\": %.3f, \"tree_ms\": %.3f, "
         "\"chain_ns_per_add\": %.4f, \"tree_ns_per_add\": %.4f, \"tree_over_chain\": %.3f, \"tree_bytes_per_add\": %.0f, \"tree_TB_per_s\": %.2f, \"checked_lanes_differing\": %d, "
         "\"sums_at_infinity\": %d}\n",
         smask == 0xFFFFFFFFu ? "false" : "true", WAVES, waves, m, tail, ms_chain, ms_tree, ms_chain * 1e6 / adds, ms_tree * 1e6 / adds, ms_tree / ms_chain,
         tree_bytes / adds, tree_bytes / (ms_tree * 1e-3) * 1e-12, check ? bad : -1, inf);
  CK(hipFree(o_chain));
  CK(hipFree(o_tree));
  CK(hipFree(S.bufA));
  CK(hipFree(S.bufB));
  CK(hipFree(S.pre));
  return bad;
}

int main(int argc, char** argv) {
  const uint32_t lg_nt = argc > 1 ? atoi(argv[1]) : 24;      // table entries (2^lg_nt x 64 B)
  const uint32_t nvalid = argc > 2 ? atoi(argv[2]) : 4096;   // distinct curve points behind them
  const size_t nt = (size_t)1 << lg_nt;
  // valid points: (s + i) G on the host
  std::vector<G1Affine> valid(nvalid);
  {
    G1Affine g;
    g.x = Fq::from_u32(1);
    g.y = Fq::from_u32(2);
    uint32_t k[8] = {0x9E3779B9u, 0x7F4A7C15u, 0x12345u, 0, 0, 0, 0, 0};
    G1XYZZ acc = scalar_mul(g, k);
    for (uint32_t i = 0; i < nvalid; i++) {
      valid[i] = acc.to_affine();
      acc.madd(g);
    }
  }
  G1Affine* d_valid;
  G1Affine29* table;
  CK(hipMalloc(&d_valid, nvalid * sizeof(G1Affine)));
  CK(hipMemcpy(d_valid, valid.data(), nvalid * sizeof(G1Affine), hipMemcpyHostToDevice));
  CK(hipMalloc(&table, nt * sizeof(G1Affine29)));
  k_make_table<<<(unsigned)((nt + 255) / 256), 256>>>(d_valid, nvalid, table, nt);
  CK(hipDeviceSynchronize());
  int bad = 0;
  // a small table of few distinct points first: equal and opposite picks meet at every level
  bad += run<4>(table, 15, 64, 4, 8, 1, true);
  bad += run<4>(table, nt - 1, 256, 8, 64, 1, true);
  const uint32_t ms[] = {256, 512, 1024, 2048};
  for (uint32_t m : ms) {
    bad += run<4>(table, nt - 1, m, 8, 4096, 3, true);
    bad += run<3>(table, nt - 1, m, 8, 3072, 3, true);
    bad += run<2>(table, nt - 1, m, 8, 2048, 3, true);
  }
  // the same with every pick in a 64 KiB corner of the table (the gathers hit the L2: what is left is the scratch traffic)
  bad += run<4>(table, 1023, 1024, 8, 4096, 3, true);
  bad += run<2>(table, 1023, 1024, 8, 2048, 3, true);
  // the issue-bound floor: every scratch slot aliased onto four (wrong sums, the same instructions; the scratch stays in
  // the L2) with the picks cached as well -- what the tree would cost if memory were free
  run<4>(table, 1023, 1024, 8, 4096, 3, false, 3u);
  run<4>(table, nt - 1, 1024, 8, 4096, 3, false, 3u);
  printf("%s\n", bad ? "MISMATCH" : "all checked lanes equal");
  return bad ? 2 : 0;
}
