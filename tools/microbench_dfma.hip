// Experiment: the BN254 Fq Montgomery product on the FP64 pipe -- 5 x 52-bit limbs held as doubles,
// every 52 x 52 partial product split into its exact high and low halves by a pair of round-toward-zero v_fma_f64
// (Emmart, "Faster modular exponentiation using double precision floating point arithmetic on the GPU", ARITH 2018):
//     hi  = fma_rz(x, y, 2^104)                = 2^104 + floor(x y / 2^52) 2^52
//     sub = (2^104 + 2^52) - hi                                                      (exact)
//     lo  = fma_rz(x, y, sub)                  = 2^52 + (x y mod 2^52)               (exact)
// the mantissas of hi / lo ARE the two 52-bit halves: they are accumulated column by column as 64-bit integers (bit
// patterns minus the bias patterns).  Beside it, in the same binary and on the same box: the shipped 9 x 29-bit
// v_mad_u64_u32 product of fq29.h, and the bare issue rates of the instructions both forms are made of.
//
// Gate: >= 1.25 x products/s at <= 128 VGPRs per G1 accumulator => port the walk; below: record and stop.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zerokit_amd/csrc tools/microbench_dfma.hip -o tools/microbench_dfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "fq29.h"
using namespace rlnamd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---------------------------------------------------------------------------------------------------- the 5 x 52 form
struct P52 { double p[5]; double np; };   // modulus limbs, -p^-1 mod 2^52
__constant__ P52 c_p52;

constexpr unsigned long long K1 = 0x4670000000000000ull;   // bit pattern of 2^104
constexpr unsigned long long K3 = 0x4330000000000000ull;   // bit pattern of 2^52
constexpr unsigned long long MASK52 = (1ull << 52) - 1;

static __device__ __forceinline__ void set_round_toward_zero() {
  // MODE.FP_ROUND[3:2] (f64 / f16 rounding) = 3: hwreg(HW_REG_MODE = 1, offset 2, size 2)
  __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);
}
// the FMAs as asm: no constant folding or re-association under the default rounding mode the compiler assumes.
// gfx9 takes ONE scalar operand per VOP3 instruction: the bias 2^104 rides in VGPRs, the modulus limbs in SGPRs.
static __device__ __forceinline__ double fma_rz(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
static __device__ __forceinline__ double fma_rz_s(double a, double b_sgpr, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_sgpr), "v"(c));
  return d;
}
static __device__ __forceinline__ double sub_exact(double a, double b) {   // a - b
  double d;
  asm("v_add_f64 %0, %1, -%2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
static __device__ __forceinline__ double sub_exact_s(double a, double b_sgpr) {
  double d;
  asm("v_add_f64 %0, %1, -%2" : "=v"(d) : "v"(a), "s"(b_sgpr));
  return d;
}
static __device__ __forceinline__ double opaque_sgpr(unsigned long long bits) {
  asm volatile("" : "+s"(bits));
  return __longlong_as_double((long long)bits);
}
static __device__ __forceinline__ double opaque_vgpr(unsigned long long bits) {
  asm volatile("" : "+v"(bits));
  return __longlong_as_double((long long)bits);
}
// x y into two raw column accumulators (bit patterns; the bias patterns are taken off per row)
template <bool YS>
static __device__ __forceinline__ void mac52(double x, double y, unsigned long long& lo_acc, unsigned long long& hi_acc, double C1, double C2) {
  const double hi = YS ? fma_rz_s(x, y, C1) : fma_rz(x, y, C1);
  const double sb = sub_exact(C2, hi);
  const double lo = YS ? fma_rz_s(x, y, sb) : fma_rz(x, y, sb);
  hi_acc += (unsigned long long)__double_as_longlong(hi);
  lo_acc += (unsigned long long)__double_as_longlong(lo);
}

struct L52 {
  double v[5];
  // a b / 2^260 mod p, in [0, p + a b / 2^260): closed for operands below 2^256, no final subtraction
  static __device__ __forceinline__ L52 mul(const L52& a, const L52& b) {
    double p[5];
#pragma unroll
    for (int j = 0; j < 5; j++) p[j] = c_p52.p[j];
    const double np = c_p52.np;
    // opaque constants (scalar registers): otherwise every use re-materialises a 64-bit literal
    const double C1 = opaque_vgpr(K1), C2 = opaque_vgpr(K1 + 1);   // 2^104; 2^104 + 2^52 (one ulp above)
    const double C3 = opaque_sgpr(K3);                               // 2^52
    unsigned long long t[6];   // column sums modulo 2^64 (the true sums stay below 2^57)
    // the biases one row adds: position 0 takes two low halves, 1..4 two low + two high halves, 5 two high halves
#pragma unroll
    for (int j = 0; j < 6; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      t[0] -= K3;
#pragma unroll
      for (int j = 1; j < 5; j++) t[j] -= 2 * (K1 + K3);
      t[5] -= 2 * K1;
#pragma unroll
      for (int j = 0; j < 5; j++) mac52<false>(a.v[j], b.v[i], t[j], t[j + 1], C1, C2);
      // q = t[0] np mod 2^52
      const double t0 = sub_exact_s(__longlong_as_double((long long)((t[0] & MASK52) | K3)), C3);
      const double qh = fma_rz_s(t0, np, C1);
      const double ql = fma_rz_s(t0, np, sub_exact(C2, qh));
      const double q = sub_exact_s(ql, C3);
      t[0] -= K3;
#pragma unroll
      for (int j = 0; j < 5; j++) mac52<true>(q, p[j], t[j], t[j + 1], C1, C2);
      const unsigned long long carry = t[0] >> 52;   // t[0] = 0 mod 2^52 now
#pragma unroll
      for (int j = 0; j < 5; j++) t[j] = t[j + 1];
      t[0] += carry;
      t[5] = 0;
    }
    L52 r;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      r.v[j] = sub_exact_s(__longlong_as_double((long long)((t[j] & MASK52) | K3)), C3);
      if (j < 4) t[j + 1] += t[j] >> 52;
    }
    return r;
  }
};

template <int ITER> __global__ void __launch_bounds__(256) k_mul52(L52* out, const L52* in) {
  set_round_toward_zero();
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  L52 x = in[t], y = in[t + 1];
#pragma unroll 1
  for (int i = 0; i < ITER; i++) { x = L52::mul(x, y); y = L52::mul(y, x); }
  L52 r;
#pragma unroll
  for (int j = 0; j < 5; j++) r.v[j] = x.v[j] + y.v[j] * 0x1p60;   // both results reach memory (limbs < 2^52: exact pack is not needed here)
  out[t] = r;
}
// one product per lane, for the check against the host
__global__ void __launch_bounds__(64) k_mul52_once(L52* out, const L52* a, const L52* b, int n) {
  set_round_toward_zero();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = L52::mul(a[t], b[t]);
}
// the shipped product (fq29.h: wide reduction rounds, v_mad_u64_u32 columns), same chain shape
template <int ITER> __global__ void __launch_bounds__(256) k_mul29(Fq29* out, const Fq29* in) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 x = in[t], y = in[t + 1];
#pragma unroll 1
  for (int i = 0; i < ITER; i++) { x = Fq29::mul(x, y); y = Fq29::mul(y, x); }
  Fq29 r;
#pragma unroll
  for (int j = 0; j < 9; j++) r.v[j] = x.v[j] + y.v[j];
  out[t] = r;
}

// ------------------------------------------------------------------------------------------- bare instruction rates
// eight independent dependency chains per lane, 256 lanes x 4096 workgroups: the issue rate, not the latency
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int ITER> __global__ void __launch_bounds__(256) k_rate_fma64(double* out, double a, double b) {
  double x[8];
  for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
    REP8(X) REP8(X)
#undef X
  }
  double s = 0;
  for (int k = 0; k < 8; k++) s += x[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ITER> __global__ void __launch_bounds__(256) k_rate_add64(double* out, double a) {
  double x[8];
  for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(a));
    REP8(X) REP8(X)
#undef X
  }
  double s = 0;
  for (int k = 0; k < 8; k++) s += x[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ITER> __global__ void __launch_bounds__(256) k_rate_mad(unsigned long long* out, uint32_t a, uint32_t b) {
  unsigned long long x[8];
  for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
  uint32_t va = a + threadIdx.x, vb = b + threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[k]) : "v"(va), "v"(vb) : "vcc");
    REP8(X) REP8(X)
#undef X
  }
  unsigned long long s = 0;
  for (int k = 0; k < 8; k++) s += x[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ITER> __global__ void __launch_bounds__(256) k_rate_add_u64(unsigned long long* out, unsigned long long a) {
  unsigned long long x[8];
  for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
  unsigned long long va = a + threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#define X(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[k]) : "v"(va));
    REP8(X) REP8(X)
#undef X
  }
  unsigned long long s = 0;
  for (int k = 0; k < 8; k++) s += x[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ITER> __global__ void __launch_bounds__(256) k_rate_add_u32(uint32_t* out, uint32_t a) {
  uint32_t x[8];
  for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
  uint32_t va = a + threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(va));
    REP8(X) REP8(X)
#undef X
  }
  uint32_t s = 0;
  for (int k = 0; k < 8; k++) s += x[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// more of the instructions the products are made of, same shape (32-bit destination / two sources; 64-bit shifts)
#define RATE32(NAME, ASM)                                                                                     \
  template <int ITER> __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a) {                \
    uint32_t x[8];                                                                                            \
    for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;                                                       \
    uint32_t va = a + threadIdx.x;                                                                            \
    _Pragma("unroll 1") for (int i = 0; i < ITER; i++) {                                                      \
      for (int r = 0; r < 2; r++) {                                                                           \
        asm volatile(ASM : "+v"(x[0]) : "v"(va) : "vcc"); asm volatile(ASM : "+v"(x[1]) : "v"(va) : "vcc");   \
        asm volatile(ASM : "+v"(x[2]) : "v"(va) : "vcc"); asm volatile(ASM : "+v"(x[3]) : "v"(va) : "vcc");   \
        asm volatile(ASM : "+v"(x[4]) : "v"(va) : "vcc"); asm volatile(ASM : "+v"(x[5]) : "v"(va) : "vcc");   \
        asm volatile(ASM : "+v"(x[6]) : "v"(va) : "vcc"); asm volatile(ASM : "+v"(x[7]) : "v"(va) : "vcc");   \
      }                                                                                                       \
    }                                                                                                         \
    uint32_t s = 0;                                                                                           \
    for (int k = 0; k < 8; k++) s += x[k];                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                                  \
  }
RATE32(k_rate_mul_lo, "v_mul_lo_u32 %0, %0, %1")
RATE32(k_rate_mul_hi, "v_mul_hi_u32 %0, %0, %1")
RATE32(k_rate_and, "v_and_b32 %0, %0, %1")
RATE32(k_rate_add_co, "v_add_co_u32 %0, vcc, %0, %1")
RATE32(k_rate_addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
RATE32(k_rate_alignbit, "v_alignbit_b32 %0, %0, %1, 29")
RATE32(k_rate_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
RATE32(k_rate_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %1")
RATE32(k_rate_sub, "v_sub_u32 %0, %0, %1")
RATE32(k_rate_or, "v_or_b32 %0, %0, %1")
RATE32(k_rate_lshr32, "v_lshrrev_b32 %0, 3, %0")
RATE32(k_rate_lshl32, "v_lshlrev_b32 %0, 1, %0")
RATE32(k_rate_mov, "v_mov_b32 %0, %1")
RATE32(k_rate_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
RATE32(k_rate_or3, "v_or3_b32 %0, %0, %1, %1")
RATE32(k_rate_add3, "v_add3_u32 %0, %0, %1, %1")
template <int ITER> __global__ void __launch_bounds__(256) k_rate_lshr64(unsigned long long* out, unsigned long long a) {
  unsigned long long x[8];
  for (int k = 0; k < 8; k++) x[k] = a + threadIdx.x + k;
#pragma unroll 1
  for (int i = 0; i < ITER; i++) {
#define X(k) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(x[k]));
    REP8(X) REP8(X)
#undef X
  }
  unsigned long long s = 0;
  for (int k = 0; k < 8; k++) s += x[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K> static float time_kernel(K launch, int reps = 5) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}

// ------------------------------------------------------------------------------------------------------- host side
// 256-bit little-endian words <-> 5 x 52-bit limbs as doubles
static void to52(const uint32_t w[8], double out[5]) {
  unsigned long long q[5] = {0, 0, 0, 0, 0};
  for (int bit = 0; bit < 256; bit++)
    if ((w[bit >> 5] >> (bit & 31)) & 1) q[bit / 52] |= 1ull << (bit % 52);
  for (int j = 0; j < 5; j++) out[j] = (double)q[j];
}
static void from52(const double in[5], uint32_t w[9]) {   // 260 bits
  memset(w, 0, 36);
  for (int j = 0; j < 5; j++) {
    const unsigned long long q = (unsigned long long)in[j];
    for (int b = 0; b < 52; b++)
      if ((q >> b) & 1) { const int bit = 52 * j + b; w[bit >> 5] |= 1u << (bit & 31); }
  }
}
static bool geq(const uint32_t* a, const uint32_t* b, int n) {
  for (int i = n - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i];
  return true;
}
static void sub_n(uint32_t* a, const uint32_t* b, int n) {
  long long br = 0;
  for (int i = 0; i < n; i++) { long long d = (long long)a[i] - b[i] - br; a[i] = (uint32_t)d; br = d < 0; }
}

int main() {
  uint32_t pm[9] = {0};
  memcpy(pm, FqParams::MOD, 32);
  P52 hp;
  to52(pm, hp.p);
  {   // -p^-1 mod 2^52 (Newton on the low 64 bits)
    const unsigned long long p0 = (unsigned long long)pm[0] | ((unsigned long long)pm[1] << 32);
    unsigned long long x = 1;
    for (int i = 0; i < 7; i++) x = x * (2 - p0 * x);
    hp.np = (double)((0 - x) & ((1ull << 52) - 1));
  }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_p52), &hp, sizeof hp));

  // ---- correctness: a b 2^-260 mod p against the library's host Fq (Montgomery, R = 2^256)
  {
    const int NC = 4096;
    std::vector<L52> ha(NC), hb(NC), hr(NC);
    std::vector<Fq> fa(NC), fb(NC);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int i = 0; i < NC; i++) {
      uint32_t a[8], b[8];
      for (int k = 0; k < 8; k++) { a[k] = next(); b[k] = next(); }
      a[7] &= 0x1FFFFFFF; b[7] &= 0x1FFFFFFF;            // < 2^253 < p
      if (i == 0) { memset(a, 0, 32); }
      if (i == 1) { const uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0}; memcpy(a, pm, 32); sub_n(a, one, 8); memcpy(b, a, 32); }   // (p-1)^2
      if (i == 2) { memset(a, 0xFF, 32); a[7] = 0x1FFFFFFF; memcpy(b, a, 32); }
      to52(a, ha[i].v); to52(b, hb[i].v);
      fa[i] = Fq::from_canonical(a); fb[i] = Fq::from_canonical(b);
    }
    L52 *da, *db, *dr;
    CK(hipMalloc(&da, NC * sizeof(L52))); CK(hipMalloc(&db, NC * sizeof(L52))); CK(hipMalloc(&dr, NC * sizeof(L52)));
    CK(hipMemcpy(da, ha.data(), NC * sizeof(L52), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), NC * sizeof(L52), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mul52_once, dim3(NC / 64), dim3(64), 0, 0, dr, da, db, NC);
    CK(hipMemcpy(hr.data(), dr, NC * sizeof(L52), hipMemcpyDeviceToHost));
    Fq half = Fq::from_u32(2).inv(), inv260 = Fq::one();
    for (int i = 0; i < 260; i++) inv260 = inv260 * half;
    int bad = 0, above_p = 0;
    for (int i = 0; i < NC; i++) {
      uint32_t w[9];
      from52(hr[i].v, w);
      for (int k = 0; k < 5; k++) if (hr[i].v[k] < 0 || hr[i].v[k] >= 0x1p52 || hr[i].v[k] != (double)(unsigned long long)hr[i].v[k]) bad++;
      if (geq(w, pm, 9)) { above_p++; sub_n(w, pm, 9); }
      if (geq(w, pm, 9)) { bad++; continue; }
      uint32_t c[8];
      (fa[i] * fb[i] * inv260).to_canonical(c);
      bad += memcmp(c, w, 32) != 0;
    }
    printf("5x52 DFMA product vs host Fq   : %d / %d mismatches (%d results in [p, 2p))\n", bad, NC, above_p);
    if (bad) return 2;
  }

  // ---- products per second, same chain shape for both forms
  const int BLOCKS = 256 * 16, T = 256, N = BLOCKS * T;
  constexpr int MI = 512;
  {
    std::vector<L52> h(N + 1);
    for (int i = 0; i <= N; i++)
      for (int k = 0; k < 5; k++) h[i].v[k] = (double)(((unsigned long long)(i * 2654435761u + k * 40503u + 12345u) * 0x9E3779B1ull) & (k == 4 ? (1ull << 44) - 1 : (1ull << 52) - 1));
    L52 *din, *dout; CK(hipMalloc(&din, (N + 1) * sizeof(L52))); CK(hipMalloc(&dout, N * sizeof(L52)));
    CK(hipMemcpy(din, h.data(), (N + 1) * sizeof(L52), hipMemcpyHostToDevice));
    float ms = time_kernel([&] { hipLaunchKernelGGL(k_mul52<MI>, dim3(BLOCKS), dim3(T), 0, 0, dout, din); });
    printf("Fq 5x52 DFMA mont mul          : %8.3f ms  %8.2f Gmul/s\n", ms, (double)N * MI * 2 / ms / 1e6);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_mul52<MI>, dim3(1), dim3(64), 0, 0, dout, din); });
    printf("single wave 5x52               : %8.3f us per dependent product\n", ms * 1e3 / (MI * 2));
  }
  {
    std::vector<Fq29> h(N + 1);
    for (int i = 0; i <= N; i++) for (int k = 0; k < 9; k++) h[i].v[k] = (uint32_t)(i * 2654435761u + k * 40503u + 12345u) & (k == 8 ? 0x1FFFFF : (1u << 29) - 1);
    Fq29 *din, *dout; CK(hipMalloc(&din, (N + 1) * sizeof(Fq29))); CK(hipMalloc(&dout, N * sizeof(Fq29)));
    CK(hipMemcpy(din, h.data(), (N + 1) * sizeof(Fq29), hipMemcpyHostToDevice));
    float ms = time_kernel([&] { hipLaunchKernelGGL(k_mul29<MI>, dim3(BLOCKS), dim3(T), 0, 0, dout, din); });
    printf("Fq 9x29 v_mad_u64_u32 (shipped): %8.3f ms  %8.2f Gmul/s\n", ms, (double)N * MI * 2 / ms / 1e6);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_mul29<MI>, dim3(1), dim3(64), 0, 0, dout, din); });
    printf("single wave 9x29               : %8.3f us per dependent product\n", ms * 1e3 / (MI * 2));
  }
  // ---- bare issue rates: wave-instructions per second over the whole chip
  {
    constexpr int RI = 1024;
    const double winst = (double)BLOCKS * (T / 64) * RI * 16;
    void* buf; CK(hipMalloc(&buf, (size_t)N * 8));
    float ms = time_kernel([&] { hipLaunchKernelGGL(k_rate_fma64<RI>, dim3(BLOCKS), dim3(T), 0, 0, (double*)buf, 1.0000001, 1e-9); });
    printf("v_fma_f64      : %8.1f G wave-instr/s\n", winst / ms / 1e6);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_rate_add64<RI>, dim3(BLOCKS), dim3(T), 0, 0, (double*)buf, 1e-9); });
    printf("v_add_f64      : %8.1f G wave-instr/s\n", winst / ms / 1e6);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_rate_mad<RI>, dim3(BLOCKS), dim3(T), 0, 0, (unsigned long long*)buf, 3u, 5u); });
    printf("v_mad_u64_u32  : %8.1f G wave-instr/s\n", winst / ms / 1e6);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_rate_add_u64<RI>, dim3(BLOCKS), dim3(T), 0, 0, (unsigned long long*)buf, 3ull); });
    printf("v_lshl_add_u64 : %8.1f G wave-instr/s\n", winst / ms / 1e6);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_rate_add_u32<RI>, dim3(BLOCKS), dim3(T), 0, 0, (uint32_t*)buf, 3u); });
    printf("v_add_u32      : %8.1f G wave-instr/s\n", winst / ms / 1e6);
#define RUN32(NAME, LABEL)                                                                                          \
    ms = time_kernel([&] { hipLaunchKernelGGL(NAME<RI>, dim3(BLOCKS), dim3(T), 0, 0, (uint32_t*)buf, 3u); });       \
    printf("%-15s: %8.1f G wave-instr/s\n", LABEL, winst / ms / 1e6);
    RUN32(k_rate_mul_lo, "v_mul_lo_u32")
    RUN32(k_rate_mul_hi, "v_mul_hi_u32")
    RUN32(k_rate_mad_u32_u24, "v_mad_u32_u24")
    RUN32(k_rate_and, "v_and_b32")
    RUN32(k_rate_add_co, "v_add_co_u32")
    RUN32(k_rate_addc_co, "v_addc_co_u32")
    RUN32(k_rate_alignbit, "v_alignbit_b32")
    RUN32(k_rate_lshl_add, "v_lshl_add_u32")
    RUN32(k_rate_sub, "v_sub_u32")
    RUN32(k_rate_or, "v_or_b32")
    RUN32(k_rate_lshr32, "v_lshrrev_b32")
    RUN32(k_rate_lshl32, "v_lshlrev_b32")
    RUN32(k_rate_mov, "v_mov_b32")
    RUN32(k_rate_cndmask, "v_cndmask_b32")
    RUN32(k_rate_or3, "v_or3_b32")
    RUN32(k_rate_add3, "v_add3_u32")
    ms = time_kernel([&] { hipLaunchKernelGGL(k_rate_lshr64<RI>, dim3(BLOCKS), dim3(T), 0, 0, (unsigned long long*)buf, 3ull); });
    printf("%-15s: %8.1f G wave-instr/s\n", "v_lshrrev_b64", winst / ms / 1e6);
  }
  return 0;
}
