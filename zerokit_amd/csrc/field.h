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

namespace rlnamd {

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
  }

  friend RLN_HD Fp operator+(const Fp& a, const Fp& b) {
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
  }
  friend RLN_HD Fp operator-(const Fp& a, const Fp& b) {
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
  }
  RLN_HD Fp neg() const { return is_zero() ? *this : (zero() - *this); }
  RLN_HD Fp dbl() const { return *this + *this; }

  // Montgomery product a*b*R^-1 mod p.  Operand-scanning CIOS, multiply and reduce interleaved so each
  // inner step is two v_mad_u64_u32 with 32-bit carries; "no-carry" form (top word fits, MOD < 2^254).
  static RLN_HD Fp mont_mul(const Fp& a, const Fp& b) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t bi = b.v[i];
      uint64_t acc = (uint64_t)a.v[0] * bi + t[0];
      const uint32_t m = (uint32_t)acc * P::INV32;
      uint64_t red = (uint64_t)m * P::MOD[0] + (uint32_t)acc;
      uint32_t c1 = (uint32_t)(acc >> 32), c2 = (uint32_t)(red >> 32);
#pragma unroll
      for (int j = 1; j < 8; j++) {
        acc = (uint64_t)a.v[j] * bi + t[j] + c1;
        c1 = (uint32_t)(acc >> 32);
        red = (uint64_t)m * P::MOD[j] + (uint32_t)acc + c2;
        c2 = (uint32_t)(red >> 32);
        t[j - 1] = (uint32_t)red;
      }
      t[7] = c1 + c2;
    }
    reduce_once(t);
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return r;
  }
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
  // Fermat inverse; 0 -> 0
  RLN_HD Fp inv() const {
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::PM2[i];
    return pow(e);
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
  // Karatsuba: 3 base multiplications
  friend RLN_HD Fq2 operator*(const Fq2& a, const Fq2& b) {
    Fq v0 = a.c0 * b.c0;
    Fq v1 = a.c1 * b.c1;
    Fq s = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {v0 - v1, s - v0 - v1};
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
