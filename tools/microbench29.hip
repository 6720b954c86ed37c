// Experiment: Fq as 9 x 29-bit unsaturated limbs (R = 2^261) against the shipped 8 x 32-bit saturated form.
// Column sums of 29x29-bit products fit a 64-bit accumulator, so v_mad_u64_u32 needs no carry capture (no v_addc)
// and the 7 spare bits make the final conditional subtraction unnecessary.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zerokit_amd/csrc tools/microbench29.hip -o tools/microbench29
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "fq29.h"
using namespace rlnamd;
#ifndef VG
#define VG 128
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr uint32_t M29 = (1u << 29) - 1;
struct P29 { uint32_t p[9]; uint32_t inv; };   // modulus limbs, -p^-1 mod 2^29
__constant__ P29 c_p29;

struct L29 {
  uint32_t v[9];
  static __device__ __forceinline__ L29 mul(const L29& a, const L29& b) {
    uint32_t p[9];
#pragma unroll
    for (int j = 0; j < 9; j++) p[j] = __builtin_amdgcn_readfirstlane(c_p29.p[j]);
    const uint32_t inv = __builtin_amdgcn_readfirstlane(c_p29.inv);
    uint64_t t[10];
#pragma unroll
    for (int j = 0; j < 10; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
      for (int j = 0; j < 9; j++) t[j] += (uint64_t)a.v[j] * b.v[i];
      uint32_t m = ((uint32_t)t[0] * inv) & M29;
#pragma unroll
      for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * p[j];
      uint64_t carry = t[0] >> 29;
#pragma unroll
      for (int j = 0; j < 9; j++) t[j] = t[j + 1];
      t[0] += carry;
      t[9] = 0;
    }
    L29 r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      r.v[j] = (uint32_t)t[j] & M29;
      t[j + 1] += t[j] >> 29;
    }
    r.v[8] = (uint32_t)t[8];
    return r;
  }
  static __device__ __forceinline__ L29 add_lazy(const L29& a, const L29& b) {
    L29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.v[j] = a.v[j] + b.v[j];
    return r;
  }
  __device__ __forceinline__ void normalize() {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      v[j + 1] += v[j] >> 29;
      v[j] &= M29;
    }
  }
};

template <int ITER> __global__ void __launch_bounds__(256) k_mul29(L29* out, const L29* in) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  L29 x = in[t], y = in[t + 1];
#pragma unroll 1
  for (int i = 0; i < ITER; i++) { x = L29::mul(x, y); y = L29::mul(y, x); }
  out[t] = L29::add_lazy(x, y);
}
template <int ITER> __global__ void __launch_bounds__(256) k_fqmul(Fq* out, const Fq* in) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = in[t], y = in[t + 1];
#pragma unroll 1
  for (int i = 0; i < ITER; i++) { x = x * y; y = y * x; }
  out[t] = x + y;
}
// ---- madd in both representations: same walk (with a forced doubling and a forced cancellation), results compared
template <int ITER> __global__ void __launch_bounds__(64) k_madd_ref(G1XYZZ* out, const G1Affine* pts, int npts) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  G1XYZZ acc = G1XYZZ::inf();
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
    G1Affine p = pts[(t * 7 + i * 13) % npts];
    if ((t + i) & 1) p.y = p.y.neg();
    acc.madd(p);
  }
  out[t] = acc;
}
template <int ITER> __global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(VG))) k_madd_29(G1XYZZ* out, const G1Affine29* pts, int npts) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  G1Acc29 acc = G1Acc29::inf();
#pragma unroll 1
  for (int i = 0; i < ITER; i++) acc.madd(pts[(t * 7 + i * 13) % npts], (t + i) & 1);
  out[t] = acc.to_xyzz();
}
__global__ void k_conv(const G1Affine* in, G1Affine29* out, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = to_table29(in[t]);
}
__global__ void k_affine(const G1XYZZ* in, G1Affine* out, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = in[t].to_affine();
}
// special cases: P + P (doubling), P - P (infinity), inf + P, then keep adding
__global__ void k_special(const G1Affine* pts, const G1Affine29* pts29, G1XYZZ* out_ref, G1XYZZ* out_29) {
  int t = threadIdx.x;
  G1XYZZ a = G1XYZZ::inf();
  G1Acc29 b = G1Acc29::inf();
  auto both = [&](int k, bool neg) {
    G1Affine p = pts[k];
    if (neg) p.y = p.y.neg();
    a.madd(p);
    b.madd(pts29[k], neg);
  };
  both(t, false); both(t, false);          // doubling
  both(t + 1, true); both(t + 2, false);
  out_ref[2 * t] = a; out_29[2 * t] = b.to_xyzz();
  a = G1XYZZ::inf(); b = G1Acc29::inf();
  both(t, true); both(t, false);           // cancellation -> infinity
  both(t + 3, false); both(t + 3, false); both(t + 3, true);
  out_ref[2 * t + 1] = a; out_29[2 * t + 1] = b.to_xyzz();
}

template <class K> static float time_kernel(K launch, int reps = 5) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}
typedef unsigned __int128 u128;
int main() {
  // p as 9 x 29-bit limbs and -p^-1 mod 2^29 from the canonical 8 x 32 limbs
  uint32_t pm[8];
  memcpy(pm, FqParams::MOD, 32);
  P29 hp;
  for (int j = 0; j < 9; j++) {
    int bit = 29 * j, w = bit >> 5, s = bit & 31;
    uint64_t lo = pm[w], hi = w + 1 < 8 ? pm[w + 1] : 0;
    hp.p[j] = (uint32_t)(((lo | (hi << 32)) >> s) & M29);
  }
  uint32_t x = 1;
  for (int i = 0; i < 6; i++) x = x * (2 - hp.p[0] * x);
  hp.inv = (0u - x) & M29;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_p29), &hp, sizeof hp));
  const int BLOCKS = 256 * 16, T = 256, N = BLOCKS * T; constexpr int MI = 512;
  // correctness on a few values: mul(a, b) == a*b*2^-261 mod p, checked on the host with python-free 128-bit loops is long;
  // instead check the algebraic identity mul(mul(a,b),c) == mul(a,mul(b,c)) and mul(a, R mod p) == a on device results.
  std::vector<L29> h(N + 1);
  for (int i = 0; i <= N; i++) for (int k = 0; k < 9; k++) h[i].v[k] = (uint32_t)(i * 2654435761u + k * 40503u + 12345u) & (k == 8 ? 0x1FFFFF : M29);
  L29 *din, *dout; CK(hipMalloc(&din, (N + 1) * sizeof(L29))); CK(hipMalloc(&dout, N * sizeof(L29)));
  CK(hipMemcpy(din, h.data(), (N + 1) * sizeof(L29), hipMemcpyHostToDevice));
  float ms = time_kernel([&] { hipLaunchKernelGGL(k_mul29<MI>, dim3(BLOCKS), dim3(T), 0, 0, dout, din); });
  printf("Fq 9x29 mont mul   : %8.3f ms  %8.2f Gmul/s\n", ms, (double)N * MI * 2 / ms / 1e6);
  std::vector<Fq> g(N + 1);
  for (int i = 0; i <= N; i++) { uint32_t c[8]; for (int k = 0; k < 8; k++) c[k] = (uint32_t)(i * 2654435761u + k * 40503u + 12345u); c[7] &= 0x0FFFFFFF; g[i] = Fq::from_canonical(c); }
  Fq *ein, *eout; CK(hipMalloc(&ein, (N + 1) * sizeof(Fq))); CK(hipMalloc(&eout, N * sizeof(Fq)));
  CK(hipMemcpy(ein, g.data(), (N + 1) * sizeof(Fq), hipMemcpyHostToDevice));
  ms = time_kernel([&] { hipLaunchKernelGGL(k_fqmul<MI>, dim3(BLOCKS), dim3(T), 0, 0, eout, ein); });
  printf("Fq 8x32 mont mul   : %8.3f ms  %8.2f Gmul/s\n", ms, (double)N * MI * 2 / ms / 1e6);
  // latency of one dependent product when a single wave has the SIMD to itself (the graph interpreter's situation)
  ms = time_kernel([&] { hipLaunchKernelGGL(k_mul29<MI>, dim3(1), dim3(64), 0, 0, dout, din); });
  printf("single wave 9x29   : %8.3f us per dependent product\n", ms * 1e3 / (MI * 2));
  ms = time_kernel([&] { hipLaunchKernelGGL(k_fqmul<MI>, dim3(1), dim3(64), 0, 0, eout, ein); });
  printf("single wave 8x32   : %8.3f us per dependent product\n", ms * 1e3 / (MI * 2));
  {
    const int NP = 512;
    std::vector<G1Affine> pts(NP);
    G1Affine g{Fq::from_u32(1), Fq::from_u32(2)};
    G1XYZZ acc = G1XYZZ::from_affine(g);
    for (int i = 0; i < NP; i++) { pts[i] = acc.to_affine(); acc.madd(g); if (i % 3 == 0) acc = acc.dbl(); }
    G1Affine *dp, *da_ref, *da_29; G1Affine29* dp29; G1XYZZ *dr, *d29;
    const int TH = 256 * 64 * 16; constexpr int AI = 256;
    CK(hipMalloc(&dp, NP * sizeof(G1Affine))); CK(hipMalloc(&dp29, NP * sizeof(G1Affine29)));
    CK(hipMalloc(&dr, TH * sizeof(G1XYZZ))); CK(hipMalloc(&d29, TH * sizeof(G1XYZZ)));
    CK(hipMalloc(&da_ref, TH * sizeof(G1Affine))); CK(hipMalloc(&da_29, TH * sizeof(G1Affine)));
    CK(hipMemcpy(dp, pts.data(), NP * sizeof(G1Affine), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_conv, dim3((NP + 63) / 64), dim3(64), 0, 0, dp, dp29, NP);
    float m0 = time_kernel([&] { hipLaunchKernelGGL(k_madd_ref<AI>, dim3(TH / 64), dim3(64), 0, 0, dr, dp, NP); });
    float m1 = time_kernel([&] { hipLaunchKernelGGL(k_madd_29<AI>, dim3(TH / 64), dim3(64), 0, 0, d29, dp29, NP); });
    printf("G1 madd 8x32       : %8.3f ms  %8.2f Mmadd/s\n", m0, (double)TH * AI / m0 / 1e3);
    printf("G1 madd 9x29       : %8.3f ms  %8.2f Mmadd/s\n", m1, (double)TH * AI / m1 / 1e3);
    const int NC = 4096;
    hipLaunchKernelGGL(k_affine, dim3(NC / 64), dim3(64), 0, 0, dr, da_ref, NC);
    hipLaunchKernelGGL(k_affine, dim3(NC / 64), dim3(64), 0, 0, d29, da_29, NC);
    std::vector<G1Affine> h0(NC), h1(NC);
    CK(hipMemcpy(h0.data(), da_ref, NC * sizeof(G1Affine), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), da_29, NC * sizeof(G1Affine), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < NC; i++) bad += memcmp(&h0[i], &h1[i], sizeof(G1Affine)) != 0;
    printf("madd walk parity   : %d / %d mismatches\n", bad, NC);
    hipLaunchKernelGGL(k_special, dim3(1), dim3(64), 0, 0, dp, dp29, dr, d29);
    hipLaunchKernelGGL(k_affine, dim3(2), dim3(64), 0, 0, dr, da_ref, 128);
    hipLaunchKernelGGL(k_affine, dim3(2), dim3(64), 0, 0, d29, da_29, 128);
    CK(hipMemcpy(h0.data(), da_ref, 128 * sizeof(G1Affine), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), da_29, 128 * sizeof(G1Affine), hipMemcpyDeviceToHost));
    bad = 0;
    for (int i = 0; i < 128; i++) bad += memcmp(&h0[i], &h1[i], sizeof(G1Affine)) != 0;
    printf("special-case parity: %d / 128 mismatches\n", bad);
  }
  return 0;
}
