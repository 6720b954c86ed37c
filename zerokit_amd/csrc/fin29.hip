// k_fin_smul29: s A and r B1 for batches below a wave of proofs, where the kernel is a lone lane's dependent chain
// (k_fin_smul: 1.5 ms of the 3 ms a single proof spends behind the witness interpreter).  Same result as k_fin_smul
// (prover.hip) -- the group element k P, handed on as XYZZ<Fq> -- by a different route:
//   * arithmetic in the 9 x 29-bit form (fq29.h: a product ~205 instead of ~375 instructions, squarings 45 products);
//   * no table of multiples: the GLV halves k = +-k1 +- lambda k2 (|k1|, |k2| < 2^126) are recoded to non-adjacent form
//     and one ladder of 127 doublings adds +-P and +-phi(P) -- both AFFINE, so every addition is the mixed addition of the
//     table walks (G1Acc29::madd, 8M + 2S) instead of a general one, and a third of the digits are non-zero.  With lanes
//     = proofs a wave would execute every branch some lane takes, which is why the batch kernel keeps fixed windows.
// Bounds: the doubling below is G1Acc29::dbl_affine's formula sequence with (X, Y) in place of the affine (x, y) and the
// two products ZZ3 = V ZZ, ZZZ3 = W ZZZ added; X < 5.2 q, Y < 2.1 q (the accumulator invariants) enter exactly where
// x < q, y < 2 q did: U = 2 Y has limbs < 2^30 (lazy), x2 = X^2 / 2^261 + q < 1.2 q, M = 3 x2 < 3.6 q.
#include "fin29.h"

#include "fq29.h"
#include "glv.h"

namespace rlnamd {

// 2 (X : Y : ZZ : ZZZ), dbl-2008-s-1 for XYZZ coordinates (a = 0)
__device__ __forceinline__ void g1acc29_dbl(G1Acc29& a) {
  if (a.is_inf()) return;
  Fq29 U;
#pragma unroll
  for (int j = 0; j < 9; j++) U.v[j] = 2 * a.Y.v[j];          // lazy, < 4.2 q
  const Fq29 V = Fq29::sqr(U);
  const Fq29 W = Fq29::mul(U, V);
  const Fq29 S = Fq29::mul(a.X, V);
  const Fq29 x2 = Fq29::sqr(a.X);
  Fq29 Mm;
#pragma unroll
  for (int j = 0; j < 9; j++) Mm.v[j] = 3 * x2.v[j];
  Mm.normalize();                                              // < 3.6 q
  const Fq29 M2 = Fq29::sqr(Mm);
  Fq29 X3;
#pragma unroll
  for (int j = 0; j < 9; j++) X3.v[j] = M2.v[j] + Fq29C::K4T[j] - 2 * S.v[j];
  X3.normalize();                                              // < 5.1 q
  const Fq29 D = Fq29::sub(S, Fq29C::K6, X3);
  const Fq29 nY = Fq29::neg_lazy(Fq29C::K4, a.Y);
  const Fq29 Y3 = Fq29::dot2(Mm, D, nY, W);
  a.ZZ = Fq29::mul(V, a.ZZ);
  a.ZZZ = Fq29::mul(W, a.ZZZ);
  a.X = X3;
  a.Y = Y3;
}

// non-adjacent form of a 126-bit magnitude: bit i of `nz` = digit i non-zero, bit i of `neg` = digit i is -1
// (3 k = k + 2 k; the digits are ((3 k) xor k) >> 1, positive where 3 k has the bit)
__device__ __forceinline__ void naf128(const uint32_t k[4], uint32_t nz[5], uint32_t ng[5]) {
  uint32_t k3[5], kk[5] = {k[0], k[1], k[2], k[3], 0};
  uint64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const uint64_t v = (uint64_t)kk[i] * 3 + carry;
    k3[i] = (uint32_t)v;
    carry = v >> 32;
  }
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const uint32_t x = k3[i] ^ kk[i], n = ~k3[i] & kk[i];
    const uint32_t xh = i + 1 < 5 ? (k3[i + 1] ^ kk[i + 1]) : 0, nh = i + 1 < 5 ? (~k3[i + 1] & kk[i + 1]) : 0;
    nz[i] = (x >> 1) | (xh << 31);
    ng[i] = (n >> 1) | (nh << 31);
  }
}

__global__ void __launch_bounds__(64) k_fin_smul29(const G1Affine* __restrict__ affA, const G1Affine* __restrict__ affB1,
                                                   const uint32_t* __restrict__ rs, G1XYZZ* __restrict__ prod, uint32_t B,
                                                   uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);
  const uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t task = blockIdx.y;  // 0: s A, 1: r B1
  const G1Affine P = task == 0 ? affA[p] : affB1[p];
  if (P.is_inf()) {   // A or B1 at infinity: the product is infinity
    prod[(size_t)task * B + p] = G1XYZZ::inf();
    return;
  }
  const uint32_t* k = rs + (size_t)p * 16 + (task == 0 ? 8 : 0);
  uint32_t kk[8], k1[4], k2[4], n1, n2;
#pragma unroll
  for (int i = 0; i < 8; i++) kk[i] = k[i];
  glv_split(kk, k1, &n1, k2, &n2);
  uint32_t nz1[5], ng1[5], nz2[5], ng2[5];
  naf128(k1, nz1, ng1);
  naf128(k2, nz2, ng2);
  // the two affine addends as table entries: P and phi(P) = (beta x, y)
  const G1Affine29 e1 = to_table29(P);
  G1Affine Q = P;
  Q.x = Q.x * Fq::from_canonical(GlvParams::BETA_G1);
  const G1Affine29 e2 = to_table29(Q);
  G1Acc29 acc = G1Acc29::inf();
#pragma unroll 1
  for (int i = 128; i >= 0; i--) {
    g1acc29_dbl(acc);
    const uint32_t w = (uint32_t)i >> 5, b = (uint32_t)i & 31;
    if ((nz1[w] >> b) & 1) acc.madd(e1, (((ng1[w] >> b) & 1) != 0) != (n1 != 0));
    if ((nz2[w] >> b) & 1) acc.madd(e2, (((ng2[w] >> b) & 1) != 0) != (n2 != 0));
  }
  prod[(size_t)task * B + p] = acc.to_xyzz();   // k == 0 gives infinity, matching g1_b = 0 (partial_proof.rs:242-248)
}

// ---------------------------------------------------------------------------------------------------------------------
// Finish from a cached partial proof (round 6): s A + r B1 = [s pi_a + r rho] + sum_u (s w_i) A_i + sum_u (r w_i) B1_i + 2 r s delta
// with A = pi_a + A_u + r delta, B1 = rho + B_u + s delta (partial_proof.rs:236-260).  Everything behind the bracket is
// rows of the fixed-base walk under product scalars (the fused plan); the bracket is the only variable-base work left,
// and its bases are known when the PARTIAL proof is made.  k_pp_powers (partial time, off the caller's path) stores
// 2^(8 k) P and phi(2^(8 k) P) for P = pi_a, rho, k = 0 .. 15, affine, in the table form, beside the partial run's cached
// values; k_pp_smul (finish time) gives every (point, GLV half, 8-bit chunk) of the two scalars a lane -- 8 doublings and
// at most 8 mixed additions instead of the 129-step ladder of k_fin_smul29 (0.93 ms of a lone finish) -- and sums the 64
// lane results in a tree through LDS.  Same group element: bytes cannot change.
constexpr uint32_t PP_CHUNKS = 16;                        // 8-bit chunks of a 128-bit GLV half
// cache entry, in uint4 units from `off16`: [point t][half h][chunk k] G1Affine29 (4 uint4 each)
__global__ void __launch_bounds__(64) k_pp_powers(const uint32_t* __restrict__ pp, const uint32_t* __restrict__ entry_of,
                                                  uint4* __restrict__ cache, uint32_t stride16, uint32_t off16) {
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  if (lane >= 2 * PP_CHUNKS) return;
  const uint32_t t = lane / PP_CHUNKS, k = lane % PP_CHUNKS;
  const uint32_t* d = pp + (size_t)p * 80 + (t == 0 ? 0 : 16);     // pi_a | rho (k_partial_out)
  G1Affine P{Fq::from_canonical(d), Fq::from_canonical(d + 8)};
  G1Affine29* out = (G1Affine29*)(cache + (size_t)entry_of[p] * stride16 + off16);
  G1Affine29 e1{}, e2{};   // infinity stays (0, 0)
  if (!P.is_inf()) {
    G1Affine Q = P;
    if (k) {
      G1Acc29 acc = G1Acc29::inf();
      acc.madd(to_table29(P), false);
      for (uint32_t i = 0; i < 8 * k; i++) g1acc29_dbl(acc);
      Q = acc.to_xyzz().to_affine();
    }
    if (!Q.is_inf()) {
      e1 = to_table29(Q);
      Q.x = Q.x * Fq::from_canonical(GlvParams::BETA_G1);
      e2 = to_table29(Q);
    }
  }
  out[(t * 2 + 0) * PP_CHUNKS + k] = e1;
  out[(t * 2 + 1) * PP_CHUNKS + k] = e2;
}

// out[p] = s pi_a + r rho (XYZZ); rs: n x (r | s) canonical LE words
__global__ void __launch_bounds__(64) k_pp_smul(const uint4* __restrict__ cache, const uint32_t* __restrict__ entry_of,
                                                uint32_t stride16, uint32_t off16, const uint32_t* __restrict__ rs,
                                                G1XYZZ* __restrict__ out) {
  __shared__ G1Acc29 sh[64];
  __builtin_amdgcn_s_setprio(3);
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  const uint32_t t = lane >> 5, h = (lane >> 4) & 1, k = lane & 15;
  const uint32_t* sc = rs + (size_t)p * 16 + (t == 0 ? 8 : 0);   // pi_a under s, rho under r
  uint32_t kk[8], k1[4], k2[4], n1, n2;
#pragma unroll
  for (int i = 0; i < 8; i++) kk[i] = sc[i];
  glv_split(kk, k1, &n1, k2, &n2);
  uint32_t word = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) word = (k >> 2) == (uint32_t)i ? (h ? k2[i] : k1[i]) : word;
  const uint32_t chunk = (word >> ((k & 3) * 8)) & 0xFF;
  const bool neg = (h ? n2 : n1) != 0;
  const G1Affine29 e = ((const G1Affine29*)(cache + (size_t)entry_of[p] * stride16 + off16))[(t * 2 + h) * PP_CHUNKS + k];
  G1Acc29 acc = G1Acc29::inf();
  if (chunk && !e.is_inf()) {
#pragma unroll 1
    for (int b = 7; b >= 0; b--) {
      g1acc29_dbl(acc);
      if ((chunk >> b) & 1) acc.madd(e, neg);
    }
  }
#pragma unroll 1
  for (uint32_t w = 32; w >= 1; w >>= 1) {
    sh[lane] = acc;
    __syncthreads();
    if (lane < w) acc.add(sh[lane + w]);
    __syncthreads();
  }
  if (lane == 0) out[p] = acc.to_xyzz();
}

void launch_pp_powers(hipStream_t s, const uint32_t* pp, const uint32_t* entry_of, uint4* cache, uint32_t stride16,
                      uint32_t off16, uint32_t n) {
  hipLaunchKernelGGL(k_pp_powers, dim3(n), dim3(64), 0, s, pp, entry_of, cache, stride16, off16);
}
void launch_pp_smul(hipStream_t s, const uint4* cache, const uint32_t* entry_of, uint32_t stride16, uint32_t off16,
                    const uint32_t* rs, G1XYZZ* out, uint32_t n) {
  hipLaunchKernelGGL(k_pp_smul, dim3(n), dim3(64), 0, s, cache, entry_of, stride16, off16, rs, out);
}

void launch_fin_smul29(hipStream_t s, const G1Affine* affA, const G1Affine* affB1, const uint32_t* rs, G1XYZZ* prod,
                       uint32_t B, uint32_t nb) {
  hipLaunchKernelGGL(k_fin_smul29, dim3((nb + 63) / 64, 2), dim3(64), 0, s, affA, affB1, rs, prod, B, nb);
}

}  // namespace rlnamd
