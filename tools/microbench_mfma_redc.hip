// Experiment: can the idle matrix pipe take the reduction half of a Montgomery product?
//
// The reduction multiplies by CONSTANTS: m = (T mod R) p' mod R and U = m p.  Across the 64 lanes of a wave U is an integer
// contraction  D[c][lane] = sum_k Toeplitz(p)[c][k] * m[k][lane]  -- eligible for v_mfma_i32_32x32x32_i8 while the VALU
// does a * b.  The matrix pipe takes signed 8-bit operands, so m is cut into 7-bit limbs (261 bits = 38 limbs, K padded
// to 64), p likewise; D has 76 columns of <= 38 * 127^2 < 2^20.  What this file measures is everything the VALU still
// has to do around the 12 MFMAs of one wave: cutting m (9 x 29 bits) into 7-bit limbs and packing them four to a
// register, moving them into the B-operand layout (lane l supplies column l % 32, k-block l / 32: v_permlane32_swap),
// gathering the 32 x 32 result blocks back to "lane = element", and folding 76 twenty-bit columns at 7-bit spacing back
// into 29-bit limbs.
//
//   mul29        Fq29::mul, the shipped product (VALU only): the baseline
//   hybrid       a * b on the VALU (81 multiply-adds, full 18 columns), m on the VALU (45 multiply-adds), U = m p through
//                the matrix pipe as described, t = (T + U) / 2^261; checked against mul29 (same residue mod q)
//   hybrid_valu  the same with the MFMAs removed (accumulators left as they are): the VALU cost of the hybrid alone --
//                if this is already slower than mul29, no amount of overlap with the matrix pipe can help
// Each kernel runs a dependent chain of ITERS products per lane, 4 waves per SIMD on every CU.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=400000 -I zerokit_amd/csrc \
//              tools/microbench_mfma_redc.hip -o tools/microbench_mfma_redc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "fq29.h"
using namespace rlnamd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int NL7 = 38;     // 7-bit limbs of a 261-bit value
constexpr int NC = 76;      // columns of the product of two 38-limb values (75 + carry room)
// Toeplitz(p) as the A operand of v_mfma_i32_32x32x32_i8, prepared on the host: for row block rb (3 blocks of 32 output
// columns) and k block kb (2 blocks of 32 limbs of m), lane l holds row rb * 32 + l % 32, k = kb * 32 + (l / 32) * 16 .. + 15
__constant__ int c_toep[3][2][64][4];
__constant__ uint32_t c_pinv[9];   // p' = -p^-1 mod 2^261 as 9 x 29-bit limbs

// T = a * b, all 18 columns (unnormalised 64-bit sums of 29 x 29-bit products)
__device__ __forceinline__ void full_product(const Fq29& a, const Fq29& b, uint64_t (&t)[18]) {
#pragma unroll
  for (int c = 0; c < 18; c++) t[c] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) t[i + j] += (uint64_t)a.v[j] * b.v[i];
}
// m = (T mod 2^261) * p' mod 2^261 as normalised 29-bit limbs: low half product, 45 multiply-adds
__device__ __forceinline__ void low_times_pinv(const uint64_t (&t)[18], uint32_t (&m)[9]) {
  uint32_t lo[9];
  uint64_t carry = 0;
#pragma unroll
  for (int c = 0; c < 9; c++) {   // normalise the low columns (values mod 2^261)
    uint64_t x = t[c] + carry;
    lo[c] = (uint32_t)x & Fq29::M;
    carry = x >> 29;
  }
  uint64_t u[9];
#pragma unroll
  for (int c = 0; c < 9; c++) u[c] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j + i < 9; j++) u[i + j] += (uint64_t)lo[j] * c_pinv[i];
  carry = 0;
#pragma unroll
  for (int c = 0; c < 9; c++) {
    uint64_t x = u[c] + carry;
    m[c] = (uint32_t)x & Fq29::M;
    carry = x >> 29;
  }
}
// 9 x 29-bit limbs -> 38 seven-bit limbs packed four to a register (k = 4 r .. 4 r + 3 in register r), 16 registers
// (K padded to 64 with zeros)
__device__ __forceinline__ void cut7(const uint32_t (&m)[9], int (&pk)[16]) {
  // the 261-bit integer as 32-bit words first (9 words), then 7-bit fields
  uint32_t w[9];
#pragma unroll
  for (int k = 0; k < 9; k++) w[k] = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    const int bit = 29 * j, k = bit >> 5, s = bit & 31;
    w[k] |= m[j] << s;
    if (s > 3 && k + 1 < 9) w[k + 1] |= m[j] >> (32 - s);
  }
#pragma unroll
  for (int r = 0; r < 16; r++) {
    uint32_t v = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int limb = 4 * r + q;
      if (limb < NL7) {
        const int bit = 7 * limb, k = bit >> 5, s = bit & 31;
        uint32_t f = w[k] >> s;
        if (s > 25 && k + 1 < 9) f |= w[k + 1] << (32 - s);
        v |= (f & 127u) << (8 * q);
      }
    }
    pk[r] = (int)v;
  }
}
// 76 columns of < 2^20 at 7-bit spacing -> value / 2^261 added to the high half of T, result as 29-bit limbs.
// acc[c] for this lane's element; the sum over c of acc[c] 2^(7 c) is U = m p (exactly).
__device__ __forceinline__ Fq29 fold_and_reduce(const uint64_t (&t)[18], const int (&col)[96]) {
  // U as 29-bit columns: column c contributes to limb (7 c) / 29 at shift (7 c) % 29, and spills into the next limb
  uint64_t u[19];
#pragma unroll
  for (int k = 0; k < 19; k++) u[k] = 0;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int bit = 7 * c, k = bit / 29, s = bit % 29;
    u[k] += (uint64_t)(uint32_t)col[c] << s;    // < 2^20 << 28 = 2^48; a limb collects at most 5 columns
  }
  // T + U: the low 261 bits cancel (they sum to 0 or 2^261); carry them through
  uint64_t carry = 0;
#pragma unroll
  for (int c = 0; c < 9; c++) {
    uint64_t x = t[c] + u[c] + carry;
    carry = x >> 29;
  }
  Fq29 r;
#pragma unroll
  for (int c = 9; c < 18; c++) {
    uint64_t x = (c < 17 ? t[c] : 0) + u[c] + carry;
    if (c == 17) x += u[18] << 29;
    r.v[c - 9] = c < 17 ? ((uint32_t)x & Fq29::M) : (uint32_t)x;
    carry = x >> 29;
  }
  return r;
}

// value of lane l ^ 32: one v_permlane32_swap_b32 (gfx950) and a select
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int xor32(int v, bool upper) {
  const v2u r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return (int)(upper ? r[0] : r[1]);
}

template <bool WITH_MFMA>
__device__ __forceinline__ Fq29 hybrid_mul(const Fq29& a, const Fq29& b) {
  uint64_t t[18];
  full_product(a, b, t);
  t[17] = 0;
  uint32_t m[9];
  low_times_pinv(t, m);
  int pk[16];
  cut7(m, pk);
  // B operand of block nb (elements 32 nb .. 32 nb + 31) and k block kb: lane l supplies element 32 nb + l % 32, limbs
  // kb * 32 + (l / 32) * 16 .. + 15 = registers pk[8 kb + 4 (l / 32) .. + 3] of THAT element's lane.
  // own = the registers this lane would supply if it were in the right half; the other half comes by v_permlane32_swap.
  int col[96];
#pragma unroll
  for (int c = 0; c < 96; c++) col[c] = 0;
  const bool upper = threadIdx.x & 32;
#pragma unroll
  for (int nb = 0; nb < 2; nb++) {
    v16i acc[3];
#pragma unroll
    for (int rb = 0; rb < 3; rb++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[rb][q] = 0;
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      // lanes 0..31 need registers [8 kb .. 8 kb + 3] of element 32 nb + l, lanes 32..63 registers [8 kb + 4 .. + 7] of
      // element 32 nb + l - 32.  nb = 0: the elements live in the lower half: the upper half fetches from the lower.
      v4i bop;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        int lo_regs = pk[8 * kb + q], hi_regs = pk[8 * kb + 4 + q];
        // value wanted by the lower half from "its" element, and by the upper half from the element 32 lanes below / above
        int mine = nb == 0 ? lo_regs : hi_regs;     // what the half that owns the elements uses directly
        int give = nb == 0 ? hi_regs : lo_regs;     // what the owning half hands to the other half
        // swap `give` across the halves: after it, the non-owning half holds the owner's registers
        int other = xor32(give, upper);
        const bool owner = nb == 0 ? !upper : upper;
        bop[q] = owner ? mine : other;
      }
      if (WITH_MFMA) {
#pragma unroll
        for (int rb = 0; rb < 3; rb++) {
          v4i aop;
#pragma unroll
          for (int q = 0; q < 4; q++) aop[q] = c_toep[rb][kb][threadIdx.x & 63][q];
          acc[rb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aop, bop, acc[rb], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int rb = 0; rb < 3; rb++) acc[rb][0] += bop[0] + bop[1] + bop[2] + bop[3];   // keep the operands alive
      }
    }
    // D block (rb): lane l holds element 32 nb + l % 32, output columns rb * 32 + 8 g + 4 (l / 32) + i for register
    // 4 g + i.  The lane that owns the element needs all 32: its own 16 and the 16 of the lane 32 away.
#pragma unroll
    for (int rb = 0; rb < 3; rb++)
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int mine = acc[rb][4 * g + i];
          const int theirs = xor32(mine, upper);
          const bool owner = nb == 0 ? !upper : upper;
          if (owner) {
            const int c_lo = rb * 32 + 8 * g + i, c_hi = c_lo + 4;     // lower half rows, upper half rows
            col[c_lo] = upper ? theirs : mine;
            col[c_hi] = upper ? mine : theirs;
          }
        }
  }
  return fold_and_reduce(t, col);
}

template <int MODE>
__global__ void __launch_bounds__(64) k_chain(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters) {
  const size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
  Fq29 x, y;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    x.v[j] = in[g * 18 + j];
    y.v[j] = in[g * 18 + 9 + j];
  }
  for (int it = 0; it < iters; it++) {
    Fq29 z = MODE == 0 ? Fq29::mul(x, y) : MODE == 1 ? hybrid_mul<true>(x, y) : hybrid_mul<false>(x, y);
    x = y;
    y = z;
  }
#pragma unroll
  for (int j = 0; j < 9; j++) out[g * 9 + j] = y.v[j];
}

// ------------------------------------------------------------------------------------------------ host
typedef unsigned __int128 u128;
static void to_canon_host(const uint32_t v[9], uint64_t o[5]) {   // sum v[j] 2^(29 j) as 5 x 64-bit words
  for (int k = 0; k < 5; k++) o[k] = 0;
  for (int j = 0; j < 9; j++) {
    const int bit = 29 * j, k = bit / 64, s = bit % 64;
    u128 x = (u128)v[j] << s;
    u128 lo = (u128)o[k] + (uint64_t)x;
    o[k] = (uint64_t)lo;
    u128 hi = (x >> 64) + (lo >> 64);
    for (int q = k + 1; q < 5 && hi; q++) {
      u128 y = (u128)o[q] + (uint64_t)hi;
      o[q] = (uint64_t)y;
      hi = y >> 64;
    }
  }
}
// value mod q by repeated subtraction of q 2^k (values are a few q at most after a Montgomery product)
static void mod_q(uint64_t o[5], const uint64_t q[5]) {
  for (;;) {
    bool ge = true;
    for (int k = 4; k >= 0; k--)
      if (o[k] != q[k]) { ge = o[k] > q[k]; break; }
    if (!ge) return;
    u128 borrow = 0;
    for (int k = 0; k < 5; k++) {
      u128 d = (u128)o[k] - q[k] - (uint64_t)borrow;
      o[k] = (uint64_t)d;
      borrow = (d >> 64) & 1;
    }
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  // p as 7-bit limbs, Toeplitz in the A-operand layout
  uint32_t p29[9];
  memcpy(p29, Fq29C::P, sizeof(p29));
  uint64_t pw[5];
  to_canon_host(p29, pw);
  int p7[NL7];
  for (int l = 0; l < NL7; l++) {
    const int bit = 7 * l, k = bit / 64, s = bit % 64;
    uint64_t f = pw[k] >> s;
    if (s > 57 && k + 1 < 5) f |= pw[k + 1] << (64 - s);
    p7[l] = (int)(f & 127);
  }
  static int toep[3][2][64][4];
  for (int rb = 0; rb < 3; rb++)
    for (int kb = 0; kb < 2; kb++)
      for (int l = 0; l < 64; l++)
        for (int q = 0; q < 4; q++) {
          uint32_t v = 0;
          for (int b = 0; b < 4; b++) {
            const int row = rb * 32 + l % 32, k = kb * 32 + (l / 32) * 16 + 4 * q + b;
            const int idx = row - k;   // coefficient of m_k in output column `row` is p7[row - k]
            const int coef = (idx >= 0 && idx < NL7 && k < NL7) ? p7[idx] : 0;
            v |= (uint32_t)(coef & 0xFF) << (8 * b);
          }
          toep[rb][kb][l][q] = (int)v;
        }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_toep), toep, sizeof(toep)));
  // p' = -p^-1 mod 2^261 by Newton iteration on 29-bit limbs, done with plain 320-bit host arithmetic
  auto mul_lo = [](const uint32_t a[9], const uint32_t b[9], uint32_t o[9]) {   // (a b) mod 2^261, 29-bit limbs
    u128 acc = 0;
    for (int c = 0; c < 9; c++) {
      for (int i = 0; i <= c; i++) acc += (u128)a[i] * b[c - i];
      o[c] = (uint32_t)(acc & ((1u << 29) - 1));
      acc >>= 29;
    }
  };
  uint32_t inv[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0};   // x <- x (2 - p x)
  for (int it = 0; it < 10; it++) {
    uint32_t px[9], two_m[9], nx[9];
    mul_lo(p29, inv, px);
    // 2 - px mod 2^261
    int64_t borrow = 0;
    for (int c = 0; c < 9; c++) {
      int64_t d = (int64_t)(c == 0 ? 2 : 0) - (int64_t)px[c] + borrow;
      two_m[c] = (uint32_t)(d & ((1 << 29) - 1));
      borrow = d >> 29;
    }
    mul_lo(inv, two_m, nx);
    memcpy(inv, nx, sizeof(inv));
  }
  uint32_t pinv[9];   // -inv mod 2^261
  {
    int64_t borrow = 0;
    for (int c = 0; c < 9; c++) {
      int64_t d = -(int64_t)inv[c] + borrow;
      pinv[c] = (uint32_t)(d & ((1 << 29) - 1));
      borrow = d >> 29;
    }
  }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_pinv), pinv, sizeof(pinv)));

  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int waves = prop.multiProcessorCount * 4 * 4;   // 4 waves per SIMD
  const size_t n = (size_t)waves * 64;
  std::vector<uint32_t> h_in(n * 18);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (auto& v : h_in) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    v = (uint32_t)s & ((1u << 29) - 1);
  }
  for (size_t g = 0; g < n; g++) { h_in[g * 18 + 8] &= 0xFFFFF; h_in[g * 18 + 17] &= 0xFFFFF; }   // values < 2^252
  uint32_t *d_in, *d_out[3];
  CK(hipMalloc(&d_in, h_in.size() * 4));
  CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
  for (int k = 0; k < 3; k++) CK(hipMalloc(&d_out[k], n * 9 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[3] = {"mul29 (VALU only)", "hybrid (VALU a*b, m; MFMA m*p)", "hybrid, VALU part only"};
  float ms[3];
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(waves), dim3(64), 0, 0, d_in, d_out[0], iters);
      if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(waves), dim3(64), 0, 0, d_in, d_out[1], iters);
      if (mode == 2) hipLaunchKernelGGL(k_chain<2>, dim3(waves), dim3(64), 0, 0, d_in, d_out[2], iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[mode], e0, e1));
    }
    printf("%-34s %8.3f ms  %7.1f G products/s\n", names[mode], ms[mode], (double)n * iters / (ms[mode] * 1e-3) / 1e9);
  }
  // numerics: the hybrid's results are the same residues mod q as the shipped product's
  std::vector<uint32_t> r0(n * 9), r1(n * 9);
  CK(hipMemcpy(r0.data(), d_out[0], n * 9 * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r1.data(), d_out[1], n * 9 * 4, hipMemcpyDeviceToHost));
  uint64_t qw[5];
  to_canon_host(p29, qw);
  size_t bad = 0;
  for (size_t g = 0; g < n; g++) {
    uint64_t a[5], b[5];
    to_canon_host(&r0[g * 9], a);
    to_canon_host(&r1[g * 9], b);
    mod_q(a, qw);
    mod_q(b, qw);
    if (memcmp(a, b, sizeof(a)) != 0) bad++;
  }
  printf("hybrid vs mul29 after %d chained products: %zu of %zu lanes differ mod q\n", iters, bad, n);
  printf("ratio hybrid / mul29 time: %.3f   (VALU part alone / mul29: %.3f)\n", ms[1] / ms[0], ms[2] / ms[0]);
  return bad ? 2 : 0;
}
