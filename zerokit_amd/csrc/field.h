// BN254 prime-field arithmetic for gfx950: 8 x 32-bit limbs, Montgomery form (R = 2^256).
//
// Replaces what the reference gets from ark-ff 0.5.0 `Fp256<MontBackend<..,4>>` (third party, pinned in
// /root/reference/Cargo.lock; used everywhere below rln/src/circuit/mod.rs:88-96).  arkworks uses
// 4 x 64-bit limbs with x86 mulx/adx; CDNA4's widest integer multiply is v_mad_u64_u32
// (32 x 32 + 64 -> 64), so the natural limb on this chip is 32 bits and every product below is written
// to lower onto exactly that instruction.  Both moduli are 254 bits, which gives the two spare bits the
// "no-carry" CIOS variant needs (the running top word never overflows 32 bits).
//
// Everything is `__host__ __device__` so the same code is unit-tested on the CPU (tests/test_host_math.py)
// and used for one-time table derivation at init; the proving path runs it on the GPU only.
#pragma once
#include <stdint.h>

#include "bn254_constants.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RLN_HD __host__ __device__ __forceinline__
#else
#define RLN_HD inline
#endif

#include "modinv30.h"

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__) && !defined(RLN_NO_HOST_ADX)
// Host product of the 4 x 64-bit Montgomery form with mulx / adcx / adox (the sequence ark-ff-asm emits: no-carry CIOS for a
// modulus with a spare top bit, two independent carry chains) -- 1.5 x the portable form below on the dependent chains the
// host runs: the hints of a lone proof (22 Poseidon hashes), the <= 11-leaf tree pass, the pairing of the verifier.  The
// library is built for generic x86-64, so the instructions are enabled for this one function and it is only called
// where the CPU has them (rln_host_has_adx: one cached cpuid test); both forms are compared by tests/test_host_math.py.
static inline bool rln_host_has_adx() {
  static const bool v = __builtin_cpu_supports("adx") && __builtin_cpu_supports("bmi2");
  return v;
}
__attribute__((target("adx,bmi2"))) static inline void rln_mont_mul_adx(const uint64_t* a, const uint64_t* b, const uint64_t* p,
                                                                        uint64_t inv, uint64_t* out) {
  uint64_t t0, t1, t2, t3, A, lo, hi;
#define RLN_RED_ROW \
    "movq %[inv], %%rdx\n\t imulq %[t0], %%rdx\n\t" \
    "xorq %[lo], %[lo]\n\t" \
    "mulxq 0(%[p]), %[lo], %[hi]\n\t adcxq %[t0], %[lo]\n\t movq %[hi], %[t0]\n\t" \
    "adcxq %[t1], %[t0]\n\t mulxq 8(%[p]), %[lo], %[t1]\n\t adoxq %[lo], %[t0]\n\t" \
    "adcxq %[t2], %[t1]\n\t mulxq 16(%[p]), %[lo], %[t2]\n\t adoxq %[lo], %[t1]\n\t" \
    "adcxq %[t3], %[t2]\n\t mulxq 24(%[p]), %[lo], %[t3]\n\t adoxq %[lo], %[t2]\n\t" \
    "movl $0, %k[lo]\n\t adcxq %[lo], %[t3]\n\t adoxq %[A], %[t3]\n\t"
#define RLN_MUL_ROW(OFF) \
    "xorq %[lo], %[lo]\n\t movq " #OFF "(%[b]), %%rdx\n\t" \
    "mulxq 0(%[a]), %[lo], %[A]\n\t adoxq %[lo], %[t0]\n\t" \
    "adcxq %[A], %[t1]\n\t mulxq 8(%[a]), %[lo], %[A]\n\t adoxq %[lo], %[t1]\n\t" \
    "adcxq %[A], %[t2]\n\t mulxq 16(%[a]), %[lo], %[A]\n\t adoxq %[lo], %[t2]\n\t" \
    "adcxq %[A], %[t3]\n\t mulxq 24(%[a]), %[lo], %[A]\n\t adoxq %[lo], %[t3]\n\t" \
    "movl $0, %k[lo]\n\t adcxq %[lo], %[A]\n\t adoxq %[lo], %[A]\n\t"
  __asm__(
    "movq 0(%[b]), %%rdx\n\t"
    "xorq %[lo], %[lo]\n\t"
    "mulxq 0(%[a]), %[t0], %[t1]\n\t"
    "mulxq 8(%[a]), %[lo], %[t2]\n\t adoxq %[lo], %[t1]\n\t"
    "mulxq 16(%[a]), %[lo], %[t3]\n\t adoxq %[lo], %[t2]\n\t"
    "mulxq 24(%[a]), %[lo], %[A]\n\t adoxq %[lo], %[t3]\n\t"
    "movl $0, %k[lo]\n\t adoxq %[lo], %[A]\n\t"
    RLN_RED_ROW RLN_MUL_ROW(8) RLN_RED_ROW RLN_MUL_ROW(16) RLN_RED_ROW RLN_MUL_ROW(24) RLN_RED_ROW
    : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [A] "=&r"(A), [lo] "=&r"(lo), [hi] "=&r"(hi)
    : [a] "r"(a), [b] "r"(b), [p] "r"(p), [inv] "r"(inv), "m"(*(const uint64_t(*)[4])a), "m"(*(const uint64_t(*)[4])b)
    : "rdx", "cc");
#undef RLN_RED_ROW
#undef RLN_MUL_ROW
  const uint64_t t[4] = {t0, t1, t2, t3};
  uint64_t d[4];
  unsigned __int128 br = 0;   // result < 2 p: one conditional subtraction
  for (int i = 0; i < 4; i++) {
    const unsigned __int128 x = (unsigned __int128)t[i] - p[i] - (uint64_t)br;
    d[i] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
  const uint64_t keep = (uint64_t)0 - (uint64_t)br;
  for (int i = 0; i < 4; i++) out[i] = (t[i] & keep) | (d[i] & ~keep);
}
#define RLN_HOST_ADX 1
#endif

namespace rlnamd {

#if defined(RLN_COUNT_HOST_MUL)
inline unsigned long long rln_host_mul_count = 0;
#endif

template <class P>
struct Fp {
  uint32_t v[8];

  static RLN_HD Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
  }
  static RLN_HD Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = P::R1[i];
    return r;
  }
  RLN_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i];
    return o == 0;
  }
  RLN_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i];
    return o == 0;
  }
  RLN_HD bool operator!=(const Fp& b) const { return !(*this == b); }

  // r = t - MOD if t >= MOD else t   (t < 2*MOD)
  static RLN_HD void reduce_once(uint32_t* t) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d[8], keep;
    asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
        "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
        "v_subb_co_u32 %8, vcc, 0, 0, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]),
          "=&v"(keep)
        : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]), "v"(P::MOD[0]),
          "v"(P::MOD[1]), "v"(P::MOD[2]), "v"(P::MOD[3]), "v"(P::MOD[4]), "v"(P::MOD[5]), "v"(P::MOD[6]),
          "v"(P::MOD[7])
        : "vcc");
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = keep ? t[i] : d[i];
#else
    uint32_t d[8];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t s = (uint64_t)t[i] - P::MOD[i] - borrow;
      d[i] = (uint32_t)s;
      borrow = (uint32_t)(s >> 63);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = borrow ? t[i] : d[i];
#endif
  }

  friend RLN_HD Fp operator+(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one carry chain for a+b, one borrow chain for (a+b)-p, then a select: 24 full-rate VALU ops
    Fp r, d;
    asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
        "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %15, %23, vcc"
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]),
          "=&v"(r.v[7])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
        : "vcc");
    uint32_t keep;  // all-ones when r < p (borrow out of the subtraction)
    asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
        "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
        "v_subb_co_u32 %8, vcc, 0, 0, vcc"
        : "=&v"(d.v[0]), "=&v"(d.v[1]), "=&v"(d.v[2]), "=&v"(d.v[3]), "=&v"(d.v[4]), "=&v"(d.v[5]), "=&v"(d.v[6]),
          "=&v"(d.v[7]), "=&v"(keep)
        : "v"(r.v[0]), "v"(r.v[1]), "v"(r.v[2]), "v"(r.v[3]), "v"(r.v[4]), "v"(r.v[5]), "v"(r.v[6]), "v"(r.v[7]),
          "v"(P::MOD[0]), "v"(P::MOD[1]), "v"(P::MOD[2]), "v"(P::MOD[3]), "v"(P::MOD[4]), "v"(P::MOD[5]),
          "v"(P::MOD[6]), "v"(P::MOD[7])
        : "vcc");
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = keep ? r.v[i] : d.v[i];
    return r;
#else
    Fp r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t s = (uint64_t)a.v[i] + b.v[i] + c;
      r.v[i] = (uint32_t)s;
      c = (uint32_t)(s >> 32);
    }
    reduce_once(r.v);  // a,b < MOD < 2^254 so no carry out of limb 7
    return r;
#endif
  }
  friend RLN_HD Fp operator-(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    Fp r;
    uint32_t borrow;  // all-ones when a < b
    asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
        "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
        "v_subb_co_u32 %8, vcc, 0, 0, vcc"
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]),
          "=&v"(r.v[7]), "=&v"(borrow)
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
        : "vcc");
    uint32_t m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = P::MOD[i] & borrow;
    asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %15, vcc"
        : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]),
          "+v"(r.v[7])
        : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7])
        : "vcc");
    return r;
#else
    Fp r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t s = (uint64_t)a.v[i] - b.v[i] - borrow;
      r.v[i] = (uint32_t)s;
      borrow = (uint32_t)(s >> 63);
    }
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t s = (uint64_t)r.v[i] + (borrow ? P::MOD[i] : 0u) + c;
      r.v[i] = (uint32_t)s;
      c = (uint32_t)(s >> 32);
    }
    return r;
#endif
  }
  RLN_HD Fp neg() const { return is_zero() ? *this : (zero() - *this); }
  RLN_HD Fp dbl() const { return *this + *this; }

  // Montgomery product a*b*R^-1 mod p.
  //
  // Device: product-scanning (column-wise) form.  Each column accumulates its a_i*b_j and m_i*p_j terms in
  // a 64-bit register pair with v_mad_u64_u32, whose carry-out is counted into a third word by one
  // v_addc_co_u32: two instructions per 32x32 product, no per-product zero-extension or 64-bit adds
  // (the operand-scanning C form costs ~6).  The modulus limbs ride in SGPRs (wave-uniform).
  // Host: operand-scanning CIOS in portable C (same value, used by tests and one-time table setup).
#if defined(__HIP_DEVICE_COMPILE__)
#include "mont_mac.inc"
  static __device__ __forceinline__ Fp mont_mul(const Fp& a, const Fp& b) { return mont_dot1(a, b); }
#else
  // Host: CIOS on 4 x 64-bit limbs (the same Montgomery residues: 8 x 32 little-endian words ARE 4 x 64-bit words)
  // with a 128-bit accumulator -- 32 wide products instead of 128 narrow ones.  The host side verifies proofs
  // (pairing.h, ffi_verify_rln_proof), decompresses points and derives tables; ark-ff does the same with mulx / adx.
  static constexpr uint64_t inv64() {   // -p^-1 mod 2^64 by Newton from -p^-1 mod 2^32
    const uint64_t p0 = (uint64_t)P::MOD[0] | ((uint64_t)P::MOD[1] << 32);
    uint64_t x = (uint64_t)(0u - P::INV32);   // p^-1 mod 2^32
    x = x * (2 - p0 * x);                     // mod 2^64
    return 0 - x;
  }
  static RLN_HD Fp mont_mul(const Fp& a, const Fp& b) {
    typedef unsigned __int128 u128;
#if defined(RLN_COUNT_HOST_MUL)
    rln_host_mul_count++;   // profiling builds of the host verifier (tests/host): products per stage
#endif
    uint64_t A[4], Bv[4], M[4], t[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
      Bv[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
      M[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
    }
    constexpr uint64_t INV = inv64();
#if defined(RLN_HOST_ADX)
    if (rln_host_has_adx()) {
      uint64_t o[4];
      rln_mont_mul_adx(A, Bv, M, INV, o);
      Fp r;
      for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (uint32_t)o[i];
        r.v[2 * i + 1] = (uint32_t)(o[i] >> 32);
      }
      return r;
    }
#endif
#pragma unroll
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        c += (u128)A[j] * Bv[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;   // p < 2^254: no carry out of five words
      const uint64_t m = t[0] * INV;
      c = (u128)m * M[0] + t[0];
      c >>= 64;
#pragma unroll
      for (int j = 1; j < 4; j++) {
        c += (u128)m * M[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = (uint64_t)(c >> 64);
    }
    uint64_t d[4], borrow = 0;   // t < 2 p: one conditional subtraction
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u128 x = (u128)t[i] - M[i] - borrow;
      d[i] = (uint64_t)x;
      borrow = (uint64_t)(x >> 64) & 1;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint64_t w = borrow ? t[i] : d[i];
      r.v[2 * i] = (uint32_t)w;
      r.v[2 * i + 1] = (uint32_t)(w >> 32);
    }
    return r;
  }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(RLN_NOINLINE_MUL)
  // Out-of-line multiply: one 3 KB body shared by every call site keeps point-arithmetic kernels
  // (a G2 mixed add is 30 base-field products) inside the instruction cache.  Operands travel in VGPRs.
  static __device__ __noinline__ Fp mont_mul_call(Fp a, Fp b) { return mont_mul(a, b); }
#endif
  friend RLN_HD Fp operator*(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(RLN_NOINLINE_MUL)
    return mont_mul_call(a, b);
#else
    return mont_mul(a, b);
#endif
  }
  RLN_HD Fp sqr() const { return (*this) * (*this); }

  // a*b + c*d (and 3-, 4-term forms) with ONE Montgomery reduction on the device; plain sums on the host.
  static RLN_HD Fp dot2(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RLN_NOINLINE_MUL)
    return mont_dot2(a, b, c, d);
#else
    return a * b + c * d;
#endif
  }
  static RLN_HD Fp dot3(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, const Fp& a2, const Fp& b2) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RLN_NOINLINE_MUL)
    return mont_dot3(a0, b0, a1, b1, a2, b2);
#else
    return a0 * b0 + a1 * b1 + a2 * b2;
#endif
  }
  static RLN_HD Fp dot4(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, const Fp& a2, const Fp& b2,
                        const Fp& a3, const Fp& b3) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RLN_NOINLINE_MUL)
    return mont_dot4(a0, b0, a1, b1, a2, b2, a3, b3);
#else
    return a0 * b0 + a1 * b1 + a2 * b2 + a3 * b3;
#endif
  }
  // a*b - c*d
  static RLN_HD Fp dot2_sub(const Fp& a, const Fp& b, const Fp& c, const Fp& d) { return dot2(a, b, c.neg(), d); }

  // canonical little-endian limbs <-> Montgomery
  static RLN_HD Fp from_canonical(const uint32_t* c) {
    Fp x, r2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      x.v[i] = c[i];
      r2.v[i] = P::R2[i];
    }
    return x * r2;
  }
  RLN_HD void to_canonical(uint32_t* c) const {
    Fp o = zero();
    o.v[0] = 1;
    Fp r = (*this) * o;
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = r.v[i];
  }
  static RLN_HD Fp from_u32(uint32_t x) {
    uint32_t c[8] = {x, 0, 0, 0, 0, 0, 0, 0};
    return from_canonical(c);
  }

  // generic exponentiation by a canonical 256-bit exponent (square-and-multiply, MSB first)
  RLN_HD Fp pow(const uint32_t* e) const {
    Fp r = one();
    for (int i = 255; i >= 0; i--) {
      r = r.sqr();
      if ((e[i >> 5] >> (i & 31)) & 1) r = r * (*this);
    }
    return r;
  }
  // Inverse; 0 -> 0.  Extended Euclid on the stored integer x = a R instead of the 380 field products of a^(p-2) (the
  // verifier's affine line steps, to_affine and the Fq12 inverse all end here; on the device an inversion is a lone
  // dependent chain wherever it occurs: the to_affine of a proof's back end, one per table-build level), then
  // x^-1 = a^-1 R^-1 is carried back to a^-1 R by two products with R^2.
  //   inv()         batched division steps, 30 per batch, branch-free inside a batch (modinv30.h): ~13 k instructions
  //   inv_binary()  the bit-by-bit binary form it replaced in round 4 (~760 halvings with a branch each; kept as the
  //                 cross-check of tests/host/hostmath.cpp).  -DRLN_DEVICE_FERMAT restores a^(p-2) on the device.
  RLN_HD Fp inv() const {
#if defined(__HIP_DEVICE_COMPILE__) && defined(RLN_DEVICE_FERMAT)
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::PM2[i];
    return pow(e);
#else
    if (is_zero()) return zero();
    uint32_t m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = P::MOD[i];
    Fp y, r2;
    modinv30(v, m, P::INV32, y.v);
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = P::R2[i];
    return y * r2 * r2;
#endif
  }
  RLN_HD Fp inv_binary() const {
#if defined(__HIP_DEVICE_COMPILE__) && defined(RLN_DEVICE_FERMAT)
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::PM2[i];
    return pow(e);
#else
    if (is_zero()) return zero();
    uint64_t u[4], w[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0}, M[4];
    for (int i = 0; i < 4; i++) {
      u[i] = (uint64_t)v[2 * i] | ((uint64_t)v[2 * i + 1] << 32);
      M[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
      w[i] = M[i];
    }
    auto is_one = [](const uint64_t* a) { return a[0] == 1 && (a[1] | a[2] | a[3]) == 0; };
    auto geq = [](const uint64_t* a, const uint64_t* b) {
      for (int i = 3; i >= 0; i--)
        if (a[i] != b[i]) return a[i] > b[i];
      return true;
    };
    auto sub = [](uint64_t* a, const uint64_t* b) {   // a -= b, returns the borrow (no 128-bit type: device code too)
      uint64_t br = 0;
      for (int i = 0; i < 4; i++) {
        const uint64_t t = a[i] - b[i], b1 = a[i] < b[i] ? 1u : 0u;
        const uint64_t t2 = t - br, b2 = t < br ? 1u : 0u;
        a[i] = t2;
        br = b1 | b2;
      }
      return br;
    };
    auto add = [](uint64_t* a, const uint64_t* b) {   // a += b, returns the carry
      uint64_t c = 0;
      for (int i = 0; i < 4; i++) {
        const uint64_t t = a[i] + b[i], c1 = t < a[i] ? 1u : 0u;
        const uint64_t t2 = t + c, c2 = t2 < t ? 1u : 0u;
        a[i] = t2;
        c = c1 | c2;
      }
      return c;
    };
    auto shr1 = [](uint64_t* a, uint64_t top) {
      for (int i = 0; i < 3; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63);
      a[3] = (a[3] >> 1) | (top << 63);
    };
    auto halve = [&](uint64_t* a) {   // a / 2 mod p for a < p
      uint64_t c = 0;
      if (a[0] & 1) c = add(a, M);
      shr1(a, c);
    };
    auto submod = [&](uint64_t* a, const uint64_t* b) {   // a - b mod p, a, b < p
      if (sub(a, b)) add(a, M);
    };
    while (!is_one(u) && !is_one(w)) {
      while (!(u[0] & 1)) {
        shr1(u, 0);
        halve(x1);
      }
      while (!(w[0] & 1)) {
        shr1(w, 0);
        halve(x2);
      }
      if (geq(u, w)) {
        sub(u, w);
        submod(x1, x2);
      } else {
        sub(w, u);
        submod(x2, x1);
      }
    }
    const bool first = is_one(u);
    Fp y, r2;
    for (int i = 0; i < 4; i++) {
      const uint64_t res = first ? x1[i] : x2[i];   // (no pointer to the local arrays: they stay in registers)
      y.v[2 * i] = (uint32_t)res;
      y.v[2 * i + 1] = (uint32_t)(res >> 32);
    }
    for (int i = 0; i < 8; i++) r2.v[i] = P::R2[i];
    return y * r2 * r2;
#endif
  }
};

// value (canonical limbs) > bound ?  used for the arkworks "y is negative" flag: y > (p-1)/2
RLN_HD bool limbs_gt(const uint32_t* a, const uint32_t* b) {
  for (int i = 7; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return false;
}
RLN_HD bool limbs_geq(const uint32_t* a, const uint32_t* b) { return !limbs_gt(b, a); }

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// Fq2 = Fq[u]/(u^2 + 1)
struct Fq2 {
  Fq c0, c1;
  static RLN_HD Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
  static RLN_HD Fq2 one() { return {Fq::one(), Fq::zero()}; }
  RLN_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  RLN_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
  RLN_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
  friend RLN_HD Fq2 operator+(const Fq2& a, const Fq2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  friend RLN_HD Fq2 operator-(const Fq2& a, const Fq2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  RLN_HD Fq2 neg() const { return {c0.neg(), c1.neg()}; }
  RLN_HD Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  // Device: c0 = REDC(a0 b0 + (-a1) b1), c1 = REDC(a0 b1 + a1 b0): 4 products but only 2 reductions and no
  // Karatsuba additions (3 full multiplications = 3 products + 3 reductions + 5 add/sub).
  // Host: Karatsuba, 3 base multiplications.
  friend RLN_HD Fq2 operator*(const Fq2& a, const Fq2& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RLN_NOINLINE_MUL)
    return {Fq::dot2(a.c0, b.c0, a.c1.neg(), b.c1), Fq::dot2(a.c0, b.c1, a.c1, b.c0)};
#else
    Fq v0 = a.c0 * b.c0;
    Fq v1 = a.c1 * b.c1;
    Fq s = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {v0 - v1, s - v0 - v1};
#endif
  }
  RLN_HD Fq2 sqr() const {
    Fq p = c0 * c1;
    return {(c0 + c1) * (c0 - c1), p.dbl()};
  }
  RLN_HD Fq2 mul_fq(const Fq& s) const { return {c0 * s, c1 * s}; }
  RLN_HD Fq2 conj() const { return {c0, c1.neg()}; }
  RLN_HD Fq2 inv() const {
    Fq n = (c0.sqr() + c1.sqr()).inv();
    return {c0 * n, (c1 * n).neg()};
  }
};

}  // namespace rlnamd
