// What a LONE wave (one per SIMD, as the graph interpreter runs) pays: VALU issue cadence for independent and
// dependent 64-bit multiply-adds and adds, LDS round trips, uniform-branch hops.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) k(unsigned long long* out, uint32_t* sink, int iters, uint32_t seed) {
  extern __shared__ uint32_t lds[];
  const uint32_t lane = threadIdx.x;
  for (int i = lane; i < 16384; i += 64) lds[i] = i * 2654435761u + seed;
  __syncthreads();
  uint64_t t[9];
  uint32_t a[9], b = seed | 1;
  for (int j = 0; j < 9; j++) { t[j] = j + lane; a[j] = seed * (j + 3) + lane; }
  unsigned long long c0, c1;
  // 1. nine independent chains of mads (as a field product's columns)
  c0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int j = 0; j < 9; j++) t[j] += (uint64_t)a[j] * b;
  }
  c1 = clock64();
  if (lane == 0 && blockIdx.x == 0) out[0] = c1 - c0;
  // 2. one dependent chain of mads
  uint64_t d = t[0];
  c0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 72; r++) d = (uint64_t)(uint32_t)d * b + d;
  }
  c1 = clock64();
  if (lane == 0 && blockIdx.x == 0) out[1] = c1 - c0;
  // 3. nine independent chains of 32-bit adds
  c0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int j = 0; j < 9; j++) a[j] += a[(j + 1) % 9] ^ b;
  }
  c1 = clock64();
  if (lane == 0 && blockIdx.x == 0) out[2] = c1 - c0;
  // 4. LDS round trip: nine strided reads whose address depends on the previous result
  uint32_t idx = lane;
  c0 = clock64();
  for (int i = 0; i < iters; i++) {
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) s += lds[(idx & 63) + j * 64];
    idx = s;
  }
  c1 = clock64();
  if (lane == 0 && blockIdx.x == 0) out[3] = c1 - c0;
  // 5. the same with 18 reads (two operands)
  c0 = clock64();
  for (int i = 0; i < iters; i++) {
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 18; j++) s += lds[(idx & 63) + j * 64];
    idx = s;
  }
  c1 = clock64();
  if (lane == 0 && blockIdx.x == 0) out[4] = c1 - c0;
  // 6. LDS write then dependent read
  c0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 9; j++) lds[8192 + lane + j * 64] = idx + j;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) s += lds[8192 + ((lane + 1) & 63) + j * 64];
    idx = s;
  }
  c1 = clock64();
  if (lane == 0 && blockIdx.x == 0) out[5] = c1 - c0;
  // 7. uniform branch hops (scalar compare + branch on a wave-uniform value)
  uint32_t u = __builtin_amdgcn_readfirstlane(idx) | 1, acc = 0;
  c0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll 1
    for (int r = 0; r < 16; r++) {
      if ((u >> r) & 1) acc += 3; else acc ^= 5;
      u = u * 1664525u + 1013904223u;
    }
  }
  c1 = clock64();
  if (lane == 0 && blockIdx.x == 0) out[6] = c1 - c0;
  uint64_t s = d + acc + idx;
  for (int j = 0; j < 9; j++) s += t[j] + a[j];
  sink[blockIdx.x * 64 + lane] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
int main() {
  unsigned long long* out;
  uint32_t* sink;
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&sink, 256 * 64 * 4));
  const int iters = 200;
  for (int waves : {1, 16, 256}) {
    CK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL(k, dim3(waves), dim3(64), 65536, 0, out, sink, iters, 12345u);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
    printf("workgroups %3d: indep mads %.2f cyc/instr | dep mads %.2f | indep adds %.2f | LDS 9 reads RT %.0f cyc | 18 reads RT %.0f | write9+read9 %.0f | branch hop %.1f cyc\n",
           waves, h[0] / (double)(iters * 72), h[1] / (double)(iters * 72), h[2] / (double)(iters * 72 * 2), h[3] / (double)iters,
           h[4] / (double)iters, h[5] / (double)iters, h[6] / (double)(iters * 16));
  }
  return 0;
}
