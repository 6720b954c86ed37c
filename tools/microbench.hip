// Instruction-level microbenchmarks that decide how the 256-bit Montgomery product is written on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zerokit_amd/csrc [-DRLN_NOINLINE_MUL] tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "curve.h"
using namespace rlnamd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ITER> __global__ void __launch_bounds__(256) k_mad64(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t x0 = t, x1 = t + 1, x2 = t + 2, x3 = t + 3;
  uint32_t m0 = a + t, m1 = b + t;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      x0 = (uint64_t)(uint32_t)x0 * m0 + x0;
      x1 = (uint64_t)(uint32_t)x1 * m1 + x1;
      x2 = (uint64_t)(uint32_t)x2 * m0 + x2;
      x3 = (uint64_t)(uint32_t)x3 * m1 + x3;
    }
  }
  out[t] = (uint32_t)(x0 ^ x1 ^ x2 ^ x3) ^ (uint32_t)((x0 ^ x1 ^ x2 ^ x3) >> 32);
}
template <int ITER> __global__ void __launch_bounds__(256) k_mullo(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x0 = t, x1 = t + 1, x2 = t + 2, x3 = t + 3, m0 = a | 1, m1 = b | 1;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { x0 = x0 * m0 + 1; x1 = x1 * m1 + 1; x2 = x2 * m0 + 3; x3 = x3 * m1 + 5; }
  }
  out[t] = x0 ^ x1 ^ x2 ^ x3;
}
template <int ITER> __global__ void __launch_bounds__(256) k_mulhi(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x0 = t | 0x80000000u, x1 = ~t, x2 = t * 77 + 0xdeadbeef, x3 = t + 0xc0000000u, m0 = a | 0xF0000000u, m1 = b | 0xE0000000u;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { x0 = __umulhi(x0, m0) | 0x80000000u; x1 = __umulhi(x1, m1) | 0x80000000u; x2 = __umulhi(x2, m0) | 0x80000000u; x3 = __umulhi(x3, m1) | 0x80000000u; }
  }
  out[t] = x0 ^ x1 ^ x2 ^ x3;
}
template <int ITER> __global__ void __launch_bounds__(256) k_add32(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x0 = t, x1 = t + 1, x2 = t + 2, x3 = t + 3;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { x0 = (x0 + a) ^ x1; x1 = (x1 + b) ^ x2; x2 = (x2 + a) ^ x3; x3 = (x3 + b) ^ x0; }
  }
  out[t] = x0 ^ x1 ^ x2 ^ x3;
}
template <int ITER> __global__ void __launch_bounds__(256) k_mul24(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x0 = t, x1 = t + 1, x2 = t + 2, x3 = t + 3, m0 = a & 0xFFFFFF, m1 = b & 0xFFFFFF;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      x0 = __umul24(x0 & 0xFFFFFF, m0) + 1; x1 = __umul24(x1 & 0xFFFFFF, m1) + 1;
      x2 = __umul24(x2 & 0xFFFFFF, m0) + 1; x3 = __umul24(x3 & 0xFFFFFF, m1) + 1;
    }
  }
  out[t] = x0 ^ x1 ^ x2 ^ x3;
}
template <int ITER> __global__ void __launch_bounds__(256) k_fma64(double* out, double a, double b) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  double x0 = t, x1 = t + 1, x2 = t + 2, x3 = t + 3;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b); }
  }
  out[t] = x0 + x1 + x2 + x3;
}
template <int ITER> __global__ void __launch_bounds__(256) k_fqmul(Fq* out, const Fq* in) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = in[t], y = in[t + 1];
#pragma unroll 1
  for (int i = 0; i < ITER; i++) { x = x * y; y = y * x; }
  out[t] = x + y;
}
template <int ITER> __global__ void __launch_bounds__(256) k_fqadd(Fq* out, const Fq* in) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = in[t], y = in[t + 1];
#pragma unroll 1
  for (int i = 0; i < ITER; i++) { x = x + y; y = y - x; }
  out[t] = x + y;
}
template <class F, int ITER> __global__ void __launch_bounds__(64) k_madd(XYZZ<F>* out, const Affine<F>* pts, int npts) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ<F> acc = XYZZ<F>::from_affine(pts[t % npts]);
#pragma unroll 1
  for (int i = 0; i < ITER; i++) acc.madd(pts[(t + 1 + i) % npts]);
  out[t] = acc;
}

template <class K> static float time_kernel(K launch, int reps = 5) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s %s CUs %d clock %d kHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  const int BLOCKS = 256 * 16, T = 256, N = BLOCKS * T; constexpr int IT = 256;
  uint32_t* d32; double* d64; CK(hipMalloc(&d32, N * 8)); CK(hipMalloc(&d64, N * 8));
  double ops = (double)N * IT * 64;
  float ms;
  ms = time_kernel([&] { hipLaunchKernelGGL(k_mad64<IT>, dim3(BLOCKS), dim3(T), 0, 0, d32, 12345u, 67891u); });
  printf("v_mad_u64_u32      : %8.3f ms  %8.2f Gop/s\n", ms, ops / ms / 1e6);
  ms = time_kernel([&] { hipLaunchKernelGGL(k_mullo<IT>, dim3(BLOCKS), dim3(T), 0, 0, d32, 12345u, 67891u); });
  printf("v_mul_lo_u32(+add) : %8.3f ms  %8.2f Gop/s\n", ms, ops / ms / 1e6);
  ms = time_kernel([&] { hipLaunchKernelGGL(k_mulhi<IT>, dim3(BLOCKS), dim3(T), 0, 0, d32, 12345u, 67891u); });
  printf("v_mul_hi_u32(+or)  : %8.3f ms  %8.2f Gop/s\n", ms, ops / ms / 1e6);
  ms = time_kernel([&] { hipLaunchKernelGGL(k_mul24<IT>, dim3(BLOCKS), dim3(T), 0, 0, d32, 12345u, 67891u); });
  printf("v_mul_u32_u24(+2)  : %8.3f ms  %8.2f Gop/s\n", ms, ops / ms / 1e6);
  ms = time_kernel([&] { hipLaunchKernelGGL(k_add32<IT>, dim3(BLOCKS), dim3(T), 0, 0, d32, 12345u, 67891u); });
  printf("v_add+v_xor pairs  : %8.3f ms  %8.2f Gop/s (pairs)\n", ms, ops / ms / 1e6);
  ms = time_kernel([&] { hipLaunchKernelGGL(k_fma64<IT>, dim3(BLOCKS), dim3(T), 0, 0, d64, 1.000001, 0.5); });
  printf("v_fma_f64          : %8.3f ms  %8.2f Gop/s\n", ms, ops / ms / 1e6);

  std::vector<Fq> h(N + 1);
  for (int i = 0; i <= N; i++) { uint32_t c[8]; for (int k = 0; k < 8; k++) c[k] = (uint32_t)(i * 2654435761u + k * 40503u + 12345u); c[7] &= 0x0FFFFFFF; h[i] = Fq::from_canonical(c); }
  Fq *din, *dout; CK(hipMalloc(&din, (N + 1) * sizeof(Fq))); CK(hipMalloc(&dout, N * sizeof(Fq)));
  CK(hipMemcpy(din, h.data(), (N + 1) * sizeof(Fq), hipMemcpyHostToDevice));
  constexpr int MI = 512;
  ms = time_kernel([&] { hipLaunchKernelGGL(k_fqmul<MI>, dim3(BLOCKS), dim3(T), 0, 0, dout, din); });
  printf("Fq mont mul        : %8.3f ms  %8.2f Gmul/s\n", ms, (double)N * MI * 2 / ms / 1e6);
  ms = time_kernel([&] { hipLaunchKernelGGL(k_fqadd<MI>, dim3(BLOCKS), dim3(T), 0, 0, dout, din); });
  printf("Fq add/sub         : %8.3f ms  %8.2f Gop/s\n", ms, (double)N * MI * 2 / ms / 1e6);

  // madd throughput: G1 generator multiples as table (host computes a few points)
  {
    const int NP = 64;
    std::vector<G1Affine> pts(NP);
    G1Affine g{Fq::from_u32(1), Fq::from_u32(2)};
    G1XYZZ acc = G1XYZZ::from_affine(g);
    for (int i = 0; i < NP; i++) { pts[i] = acc.to_affine(); acc.madd(g); acc = acc.dbl(); }
    G1Affine* dp; G1XYZZ* da; const int TH = 256 * 64 * 4; constexpr int AI = 256;
    CK(hipMalloc(&dp, NP * sizeof(G1Affine))); CK(hipMalloc(&da, TH * sizeof(G1XYZZ)));
    CK(hipMemcpy(dp, pts.data(), NP * sizeof(G1Affine), hipMemcpyHostToDevice));
    ms = time_kernel([&] { hipLaunchKernelGGL((k_madd<Fq, AI>), dim3(TH / 64), dim3(64), 0, 0, da, dp, NP); });
    printf("G1 madd (XYZZ)     : %8.3f ms  %8.2f Mmadd/s  (= %.2f Gmul/s at 10 mul/madd)\n", ms, (double)TH * AI / ms / 1e3, (double)TH * AI * 10 / ms / 1e6);
  }
  return 0;
}
