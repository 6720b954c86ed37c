// Experiment: what bounds the fixed-base table walk (prover.hip k_msm29<G1>) -- the integer multiplier or the random
// 64-byte table gathers?  Same loop structure as the kernel (lanes = proofs, a wave-uniform 256 KiB table row per step,
// one int16 digit load and one dependent 64-byte gather per mixed addition) over a table far larger than L2 + MALL.
// Variants: cached gathers (ALU-bound reference), gather only (memory-bound reference), digit prefetch, entry
// prefetch into registers.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zerokit_amd/csrc tools/microbench_gather.hip -o tools/microbench_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fq29.h"
using namespace rlnamd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_fill(uint32_t* t, size_t nwords) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nwords; i += stride) {
    uint64_t z = i * 0x9E3779B97F4A7C15ull + 0x1234567;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    uint32_t w = (uint32_t)z;
    if ((i & 7) == 7) w &= 0x0FFFFFFF;  // < 2^252 < q
    t[i] = w | 1u;
  }
}
__global__ void k_fill_digits(int16_t* d, size_t n, int c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t z = i * 0x9E3779B97F4A7C15ull + 0xABCDEF;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z ^= z >> 31;
  int v = (int)(z & ((1u << c) - 1)) - (1 << (c - 1));  // [-2^(c-1), 2^(c-1))
  if (v == 0) v = 1;
  d[i] = (int16_t)v;
}

// MODE 0: as k_msm29   1: all lanes take entry 0 of the row (cache hits)   2: gather only, no group law
// MODE 3: digit of step i+1 loaded before the addition of step i
// MODE 4: digit two ahead + entry one ahead held in registers
template <int MODE, int WAVES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_walk(const G1Affine29* __restrict__ table, size_t rows_total, const int16_t* __restrict__ digits, G1XYZZ* __restrict__ out,
       uint32_t steps, uint32_t B, uint32_t pgroups, int cs, unsigned long long* __restrict__ clk) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  uint32_t L = blockIdx.x;
  uint32_t xcd = L & 7, q = L >> 3;
  uint32_t chunk = (q / pgroups) * 8 + xcd, pg = q % pgroups;
  uint32_t p = pg * 64 + threadIdx.x;
  G1Acc29 acc = G1Acc29::inf();
  const int16_t* dg = digits + (size_t)(chunk % 64) * steps * B + p;
  size_t row0 = ((size_t)chunk * steps) % (rows_total - steps);
  const G1Affine29* base = table + (row0 << cs);
  if (MODE == 0 || MODE == 1 || MODE == 2) {
#pragma unroll 1
    for (uint32_t j = 0; j < steps; j++) {
      int d = dg[(size_t)j * B];
      if (d != 0) {
        uint32_t e = MODE == 1 ? 0u : (uint32_t)(d < 0 ? -d : d) - 1;
        if (MODE == 2) {
          const G1Affine29 en = base[((size_t)j << cs) + e];
#pragma unroll
          for (int k = 0; k < 8; k++) {
            acc.X.v[k] ^= en.x[k];
            acc.Y.v[k] += en.y[k];
          }
        } else {
          acc.madd(base[((size_t)j << cs) + e], d < 0);
        }
      }
    }
  } else if (MODE == 5) {  // as MODE 0 with non-temporal (streaming) loads of the entry: it is never reused
#pragma unroll 1
    for (uint32_t j = 0; j < steps; j++) {
      int d = dg[(size_t)j * B];
      if (d != 0) {
        uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* src = reinterpret_cast<const u32x4*>(base + ((size_t)j << cs) + e);
        G1Affine29 en;
        u32x4 q0 = __builtin_nontemporal_load(src), q1 = __builtin_nontemporal_load(src + 1),
              q2 = __builtin_nontemporal_load(src + 2), q3 = __builtin_nontemporal_load(src + 3);
        en.x[0] = q0.x; en.x[1] = q0.y; en.x[2] = q0.z; en.x[3] = q0.w;
        en.x[4] = q1.x; en.x[5] = q1.y; en.x[6] = q1.z; en.x[7] = q1.w;
        en.y[0] = q2.x; en.y[1] = q2.y; en.y[2] = q2.z; en.y[3] = q2.w;
        en.y[4] = q3.x; en.y[5] = q3.y; en.y[6] = q3.z; en.y[7] = q3.w;
        acc.madd(en, d < 0);
      }
    }
  } else if (MODE == 6) {  // two accumulators fed from ONE 128-byte line per step (entries 2e, 2e+1): what pairing the
    // A_i / B1_i (or A_i / L_i) entries that share the scalar w_i would do -- half the HBM requests per addition
    G1Acc29 acc2 = G1Acc29::inf();
#pragma unroll 1
    for (uint32_t j = 0; j < steps / 2; j++) {
      int d = dg[(size_t)j * B];
      if (d != 0) {
        uint32_t e = ((uint32_t)(d < 0 ? -d : d) - 1) & ~1u;
        const G1Affine29* pr = base + ((size_t)j << cs) + e;
        const G1Affine29 e0 = pr[0], e1 = pr[1];
        acc.madd(e0, d < 0);
        acc2.madd(e1, d < 0);
      }
    }
    out[(size_t)(chunk * pgroups + pg) * 64 + threadIdx.x] = acc2.to_xyzz();
  } else if (MODE == 7) {  // control for MODE 6: two accumulators, but their entries come from different rows
    G1Acc29 acc2 = G1Acc29::inf();
#pragma unroll 1
    for (uint32_t j = 0; j < steps / 2; j++) {
      int d = dg[(size_t)j * B];
      if (d != 0) {
        uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        const G1Affine29 e0 = base[((size_t)j << cs) + e];
        const G1Affine29 e1 = base[((size_t)(j + steps / 2) << cs) + (e ^ 0x155)];
        acc.madd(e0, d < 0);
        acc2.madd(e1, d < 0);
      }
    }
    out[(size_t)(chunk * pgroups + pg) * 64 + threadIdx.x] = acc2.to_xyzz();
  } else if (MODE == 8) {  // LANE PAIRS share a 128-byte line: lane 2k accumulates the first entry of the line, lane 2k+1 the
    // second, both for proof k of the wave (32 proofs per wave) under the SAME digit -- what storing A_i / B1_i (same scalar
    // w_i) side by side would do with one accumulator per lane: VGPRs and waves per SIMD as in the shipped walk, the two
    // 64-byte halves of a line leave as one 128-byte request
    const uint32_t pp = (pg * 32 + (threadIdx.x >> 1)) % B, m = threadIdx.x & 1;
    const int16_t* dgp = digits + (size_t)(chunk % 64) * steps * B + pp;
#pragma unroll 1
    for (uint32_t j = 0; j < steps; j++) {
      int d = dgp[(size_t)j * B];
      if (d != 0) {
        uint32_t e = (((uint32_t)(d < 0 ? -d : d) - 1) & ~1u) | m;
        acc.madd(base[((size_t)j << cs) + e], d < 0);
      }
    }
  } else if (MODE == 3) {
    int dn = dg[0];
#pragma unroll 1
    for (uint32_t j = 0; j < steps; j++) {
      int d = dn;
      if (j + 1 < steps) dn = dg[(size_t)(j + 1) * B];
      if (d != 0) {
        uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        acc.madd(base[((size_t)j << cs) + e], d < 0);
      }
    }
  } else {
    int d0 = dg[0];
    int d1 = steps > 1 ? dg[B] : 1;
    G1Affine29 en = base[(uint32_t)(d0 < 0 ? -d0 : d0) - 1];
#pragma unroll 1
    for (uint32_t j = 0; j < steps; j++) {
      const G1Affine29 cur = en;
      const int d = d0;
      d0 = d1;
      if (j + 1 < steps) en = base[((size_t)(j + 1) << cs) + (uint32_t)(d0 < 0 ? -d0 : d0) - 1];
      if (j + 2 < steps) d1 = dg[(size_t)(j + 2) * B];
      acc.madd(cur, d < 0);
    }
  }
  if (MODE == 2) {
    acc.ZZ = acc.X;
    acc.ZZZ = acc.Y;
  }
  out[(size_t)(chunk * pgroups + pg) * 64 + threadIdx.x] = acc.to_xyzz();
  if (threadIdx.x == 0 && (L & 63) == 0) {  // shader clock cycles and 100 MHz wall ticks of this workgroup
    atomicAdd(clk, clock64() - c0);
    atomicAdd(clk + 1, wall_clock64() - w0);
  }
}

template <class K> static float time_kernel(K launch, int reps = 3) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const int c = 13, cs = c - 1;
  const uint32_t B = 1024, pgroups = B / 64;
  const uint32_t steps = argc > 1 ? atoi(argv[1]) : 160;      // 8 points x 20 windows per chunk
  const uint32_t nchunks = argc > 2 ? atoi(argv[2]) : 2960;   // 23 675 points / 8
  const size_t gib = argc > 3 ? atoi(argv[3]) : 96;
  const size_t entries = (gib << 30) / 64, rows_total = entries >> cs;
  G1Affine29* table;
  CK(hipMalloc(&table, entries * 64));
  hipLaunchKernelGGL(k_fill, dim3(256 * 64), dim3(256), 0, 0, (uint32_t*)table, entries * 16);
  int16_t* digits;
  const size_t nd = (size_t)64 * steps * B;
  CK(hipMalloc(&digits, nd * 2));
  hipLaunchKernelGGL(k_fill_digits, dim3((nd + 255) / 256), dim3(256), 0, 0, digits, nd, c);
  G1XYZZ* out;
  const uint32_t blocks = (nchunks + 7) / 8 * 8 * pgroups;
  CK(hipMalloc(&out, (size_t)blocks * 64 * sizeof(G1XYZZ)));
  unsigned long long* clk;
  CK(hipMalloc(&clk, 16));
  CK(hipDeviceSynchronize());
  const double adds = (double)blocks * 64 * steps;
  printf("table %zu GiB, %u chunks x %u steps x %u proofs = %.1f M additions per launch\n", gib, nchunks, steps, B, adds / 1e6);
#define RUN(MODE, WAVES, label)                                                                                        \
  {                                                                                                                    \
    hipMemset(clk, 0, 16);                                                                                             \
    float ms = time_kernel([&] {                                                                                       \
      hipLaunchKernelGGL((k_walk<MODE, WAVES>), dim3(blocks), dim3(64), 0, 0, table, rows_total, digits, out, steps, B, \
                         pgroups, cs, clk);                                                                            \
    });                                                                                                                \
    unsigned long long hc[2];                                                                                          \
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);                                                                     \
    printf("%-58s %8.3f ms  %7.2f G/s   shader clock %6.0f MHz\n", label, ms, adds / ms / 1e6,                         \
           hc[1] ? (double)hc[0] / hc[1] * 100.0 : 0.0);                                                               \
  }
  RUN(0, 4, "walk as k_msm29 (4 waves/SIMD)");
  RUN(1, 4, "same, every lane takes entry 0 of the row (cached)");
  RUN(2, 4, "gather only, no group law (4 waves/SIMD)");
  RUN(2, 8, "gather only, no group law (8 waves/SIMD)");
  RUN(3, 4, "next digit loaded ahead of the addition (4 waves)");
  RUN(4, 4, "next entry in registers, digit two ahead (4 waves)");
  RUN(4, 3, "next entry in registers, digit two ahead (3 waves)");
  RUN(0, 2, "walk as k_msm29 (2 waves/SIMD)");
  RUN(7, 2, "two accumulators, two separate 64-byte entries per step (2 waves)");
  RUN(6, 3, "two accumulators, one 128-byte line per two additions (3 waves)");
  RUN(6, 2, "two accumulators, one 128-byte line per two additions (2 waves)");
  RUN(8, 4, "lane pairs share a 128-byte line, one accumulator per lane (4 waves)");
  RUN(0, 4, "walk as k_msm29 (4 waves/SIMD), again");
  RUN(8, 4, "lane pairs share a 128-byte line (4 waves), again");
  RUN(5, 4, "non-temporal entry loads (4 waves)");
  RUN(0, 4, "walk as k_msm29 again (4 waves/SIMD)");
  RUN(0, 3, "walk as k_msm29 (3 waves/SIMD)");
  RUN(3, 3, "next digit ahead (3 waves)");
  CK(hipDeviceSynchronize());
  return 0;
}
