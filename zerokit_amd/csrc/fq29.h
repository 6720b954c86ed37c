// fq29.h -- BN254 Fq in 9 x 29-bit unsaturated limbs (Montgomery radix R' = 2^261), device only.
//
// Why a second representation: on gfx950 the 8 x 32-bit product-scanning multiply of field.h spends one v_addc per
// v_mad_u64_u32 to capture the carry out of the 64-bit column sum (128 mad + 131 addc + moves + the conditional
// subtraction = ~375 instructions, 134 G mul/s measured).  With 29-bit limbs a column of 9 products (2^58 each)
// plus the reduction terms stays below 2^64, so the accumulate needs no carry capture, and the 7 spare bits of
// 9 x 29 = 261 > 254 remove the final conditional subtraction: ~235 instructions, 173 G mul/s measured
// (tools/microbench29.hip).  The fixed-base MSM (prover.hip k_msm29) walks its tables and keeps its accumulators in
// this form; everything outside that kernel stays in the 8 x 32 form of field.h, bit-exact by construction
// (values are the same residues mod q; only their integer representatives differ).
//
// Representation classes (value = sum v[j] 2^(29 j), congruent to x 2^261 mod q):
//   N      limbs v[0..7] < 2^29, v[8] = the rest ("normalised"); the integer may be any multiple-of-q offset
//   lazy   limbs < 2^30 (one borrow-free subtraction K - b or one addition of two N values)
// mul/dot2 accept N x N, N x lazy (column bound: 9 (2^58 + 2^59 + 2^58) < 2^64) and return N with value
// < q + a b / 2^261 (q / 2^261 = 0.0059: inputs up to 10 q keep the output below 1.7 q).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve.h"
#include "fq29_constants.h"

namespace rlnamd {

template <class C, class Mont>
struct F29 {
  uint32_t v[9];

  static constexpr uint32_t M = (1u << 29) - 1;

  static __device__ __forceinline__ F29 from_const(const uint32_t (&c)[9]) {
    F29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.v[j] = c[j];
    return r;
  }
  static __device__ __forceinline__ F29 zero() {
    F29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.v[j] = 0;
    return r;
  }
  __device__ __forceinline__ bool limbs_all_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) o |= v[j];
    return o == 0;
  }
  __device__ __forceinline__ void normalize() {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      v[j + 1] += v[j] >> 29;
      v[j] &= M;
    }
  }

  // ---- Montgomery reduction rounds ------------------------------------------------------------------------------
  // Opaque scalar constants: with an unknown multiplier the compiler keeps "x * c + t" as ONE v_mad_u64_u32 (it would
  // turn x * 8 + t into a zero-extension, a 64-bit shift and a 64-bit add, and x * 1 + t into a zero-extension and an
  // add).  s_mov_b32 is scalar work, off the VALU issue slots that bound the walks.
  static __device__ __forceinline__ uint32_t sc8() {
    uint32_t r;
    asm("s_mov_b32 %0, 8" : "=s"(r));
    return r;
  }
  static __device__ __forceinline__ uint32_t sc1() {
    uint32_t r;
    asm("s_mov_b32 %0, 1" : "=s"(r));
    return r;
  }
  // One round retires column t[0] (weight 2^0) and renumbers t[1..9] to t[0..8].
  //   masked: m = (t[0] INV) mod 2^29, t += m p, carry = t[0] >> 29            -- 4 instructions beside the 9 products
  //   wide:   m' = (t[0] INV32) mod 2^32 with INV32 = -p[0]^-1 mod 2^32.  m' = m mod 2^29, so t[0] + m' p[0] is
  //           0 mod 2^29 as before (the extra bits of m' only add a multiple of p 2^29), and it is 0 mod 2^32 as well,
  //           so the carry t[0] / 2^29 is exactly 8 * hi32(t[0]): one multiply-add, no mask, no 64-bit shift
  //           -- 2 instructions beside the 9 products.
  // The price of the wide round is the column bound: m' p[j] < 2^32 p[j] instead of 2^29 p[j].  A column collects at
  // most m' (p[0] + ... + p[8]) < 2^32 * 1.51 2^30 = 6.1 2^60 (Fq; Fr 6.8 2^60) from the reduction, which leaves
  // 16 2^60 - 6.8 2^60 for the products: tools/check_fq29_bounds.py replays every call site of this file and of
  // poseidon.h with all operand limbs at their class maximum and asserts that no column reaches 2^64.  The last round
  // is always masked: an unmasked m_8 would add up to 7 p to the RESULT (the earlier rounds' excess is divided by
  // 2^29 at least once: below 2^-26 p in total), and the bounds of the callers assume result < p + a b / 2^261.
  template <bool WIDE>
  static __device__ __forceinline__ void redc_round(uint64_t (&t)[10]) {
    if (WIDE) {
      const uint32_t m = (uint32_t)t[0] * C::INV32;
#pragma unroll
      for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * C::P[j];
      t[1] += (uint64_t)(uint32_t)(t[0] >> 32) * sc8();
#pragma unroll
      for (int j = 0; j < 9; j++) t[j] = t[j + 1];
    } else {
      const uint32_t m = ((uint32_t)t[0] * C::INV) & M;
#pragma unroll
      for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * C::P[j];
      const uint64_t carry = t[0] >> 29;
#pragma unroll
      for (int j = 0; j < 9; j++) t[j] = t[j + 1];
      t[0] += carry;
    }
    t[9] = 0;
  }
  // carry chain over the nine result columns; `add` (limbs < 2^32, may be null) is added to the columns first, which
  // saves the separate normalisation of "product + K - b" (24 instructions)
  static __device__ __forceinline__ F29 finish(uint64_t (&t)[10], const F29* add) {
    if (add) {
#pragma unroll
      for (int j = 0; j < 9; j++) t[j] += (uint64_t)add->v[j] * sc1();
    }
    F29 r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      r.v[j] = (uint32_t)t[j] & M;
      t[j + 1] += t[j] >> 29;
    }
    r.v[8] = (uint32_t)t[8];
    return r;
  }

  // REDC(sum of N products) [+ add]: t accumulates columns; one limb is retired per round
  template <int NP, bool WIDE = false>
  static __device__ __forceinline__ F29 redc_dot(const F29* const (&a)[NP], const F29* const (&b)[NP],
                                                 const F29* add = nullptr) {
    uint64_t t[10];
#pragma unroll
    for (int j = 0; j < 10; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
      for (int k = 0; k < NP; k++) {
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)a[k]->v[j] * b[k]->v[i];
      }
      if (i < 8) redc_round<WIDE>(t); else redc_round<false>(t);
    }
    return finish(t, add);
  }
  // wide rounds: N x N, N x lazy, lazy x lazy (check_fq29_bounds.py: "mul")
  static __device__ __forceinline__ F29 mul(const F29& a, const F29& b) {
    const F29* const aa[1] = {&a};
    const F29* const bb[1] = {&b};
    return redc_dot<1, true>(aa, bb);
  }
  // a b / 2^261 + add, normalised (add: limbs < 2^32, e.g. K - x for a normalised x)
  static __device__ __forceinline__ F29 mul_add(const F29& a, const F29& b, const F29& add) {
    const F29* const aa[1] = {&a};
    const F29* const bb[1] = {&b};
    return redc_dot<1, true>(aa, bb, &add);
  }
  // REDC(a^2) [+ add]: row i adds a_i^2 into column 2 i and (2 a_i) a_l, l > i, into column i + l -- 45 products
  // instead of 81.  Column i is complete when round i retires it (its terms come from rows k <= i / 2).  The column
  // sums are the same integers as in mul(a, a), so the bounds of mul hold; a may be N or lazy.
  static __device__ __forceinline__ F29 sqr_add(const F29& a, const F29* add) {
    uint64_t t[10];
#pragma unroll
    for (int j = 0; j < 10; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      t[i] += (uint64_t)a.v[i] * a.v[i];
      const uint32_t d = 2 * a.v[i];
#pragma unroll
      for (int l = i + 1; l < 9; l++) t[l] += (uint64_t)d * a.v[l];
      if (i < 8) redc_round<true>(t); else redc_round<false>(t);
    }
    return finish(t, add);
  }
  static __device__ __forceinline__ F29 sqr(const F29& a) { return sqr_add(a, nullptr); }
  // a0 b0 + a1 b1 with one reduction, wide rounds (check_fq29_bounds.py lists the operand classes of every caller)
  static __device__ __forceinline__ F29 dot2(const F29& a0, const F29& b0, const F29& a1, const F29& b1) {
    const F29* const aa[2] = {&a0, &a1};
    const F29* const bb[2] = {&b0, &b1};
    return redc_dot<2, true>(aa, bb);
  }
  static __device__ __forceinline__ F29 dot2_add(const F29& a0, const F29& b0, const F29& a1, const F29& b1,
                                                  const F29& add) {
    const F29* const aa[2] = {&a0, &a1};
    const F29* const bb[2] = {&b0, &b1};
    return redc_dot<2, true>(aa, bb, &add);
  }
  // three / four products with one reduction.  WIDE rounds only where check_fq29_bounds.py has the call site: all
  // operands normalised (dot3, dot4: Poseidon's MDS rows), or dot4 with two lazy operands K8 - x, K4 - x (G2's Y3:
  // 15.53 2^60 of 16 2^60).  Masked rounds take at most two lazy operands (limbs < 2^30):
  // 9 (2 2^59 + 2 2^58 + 2^58) = 15.75 2^60 < 2^64
  template <bool WIDE = false>
  static __device__ __forceinline__ F29 dot3(const F29& a0, const F29& b0, const F29& a1, const F29& b1, const F29& a2,
                                             const F29& b2) {
    const F29* const aa[3] = {&a0, &a1, &a2};
    const F29* const bb[3] = {&b0, &b1, &b2};
    return redc_dot<3, WIDE>(aa, bb);
  }
  template <bool WIDE = false>
  static __device__ __forceinline__ F29 dot4(const F29& a0, const F29& b0, const F29& a1, const F29& b1,
                                              const F29& a2, const F29& b2, const F29& a3, const F29& b3) {
    const F29* const aa[4] = {&a0, &a1, &a2, &a3};
    const F29* const bb[4] = {&b0, &b1, &b2, &b3};
    return redc_dot<4, WIDE>(aa, bb);
  }
  // K - b limb by limb: no borrows because every limb of the biased constant K dominates a normalised limb
  static __device__ __forceinline__ F29 neg_lazy(const uint32_t (&K)[9], const F29& b) {
    F29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.v[j] = K[j] - b.v[j];
    return r;
  }
  // a + K - b, normalised
  static __device__ __forceinline__ F29 sub(const F29& a, const uint32_t (&K)[9], const F29& b) {
    F29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.v[j] = a.v[j] + K[j] - b.v[j];
    r.normalize();
    return r;
  }

  // exact: is the (normalised) value one of 0, q, 2q, ..., 7q ?
  __device__ __forceinline__ bool is_zero_mod_q() const {
    // the integer k q has low limb (k q) mod 2^29, so (v[0] q^-1) mod 2^29 = k: three instructions filter all but
    // 8 / 2^29 of the values (eight compares of v[0] against the low limbs before)
    if (((v[0] * C::QINV) & M) >= 8u) return false;
    for (int k = 0; k < 8; k++) {
      bool eq = true;
      for (int j = 0; j < 9; j++) eq &= v[j] == C::KP[k][j];
      if (eq) return true;
    }
    return false;
  }

  // ---- conversions to / from the 8 x 32 Montgomery form of field.h (same residue, radix 2^256)
  // the 256-bit integer of an 8 x 32 value cut into 29-bit limbs (no arithmetic)
  static __device__ __forceinline__ F29 slice(const Mont& a) {
    // written out limb by limb: an index computed from a loop variable kept callers' arrays of Mont values in scratch
    // memory (the NTT pass's eight-element block)
    const uint32_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], a4 = a.v[4], a5 = a.v[5], a6 = a.v[6], a7 = a.v[7];
    F29 u;
    u.v[0] = a0 & M;
    u.v[1] = ((a0 >> 29) | (a1 << 3)) & M;
    u.v[2] = ((a1 >> 26) | (a2 << 6)) & M;
    u.v[3] = ((a2 >> 23) | (a3 << 9)) & M;
    u.v[4] = ((a3 >> 20) | (a4 << 12)) & M;
    u.v[5] = ((a4 >> 17) | (a5 << 15)) & M;
    u.v[6] = ((a5 >> 14) | (a6 << 18)) & M;
    u.v[7] = ((a6 >> 11) | (a7 << 21)) & M;
    u.v[8] = a7 >> 8;
    return u;
  }
  // exact reduction of a normalised value < 2 q into [0, q), as 8 x 32 words
  __device__ __forceinline__ Mont pack_reduced() const {
    uint32_t d[9];
    int64_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      int64_t x = (int64_t)v[j] - (int64_t)C::P[j] + borrow;
      d[j] = (uint32_t)x & (j < 8 ? M : 0xFFFFFFFFu);
      borrow = x >> (j < 8 ? 29 : 63);
    }
    const bool ge = borrow == 0;
    const uint32_t r0 = ge ? d[0] : v[0], r1 = ge ? d[1] : v[1], r2 = ge ? d[2] : v[2], r3 = ge ? d[3] : v[3],
                   r4 = ge ? d[4] : v[4], r5 = ge ? d[5] : v[5], r6 = ge ? d[6] : v[6], r7 = ge ? d[7] : v[7],
                   r8 = ge ? d[8] : v[8];
    Mont o;  // bits [32 w, 32 w + 32) of sum r_j 2^(29 j), written out word by word (see slice)
    o.v[0] = r0 | (r1 << 29);
    o.v[1] = (r1 >> 3) | (r2 << 26);
    o.v[2] = (r2 >> 6) | (r3 << 23);
    o.v[3] = (r3 >> 9) | (r4 << 20);
    o.v[4] = (r4 >> 12) | (r5 << 17);
    o.v[5] = (r5 >> 15) | (r6 << 14);
    o.v[6] = (r6 >> 18) | (r7 << 11);
    o.v[7] = (r7 >> 21) | (r8 << 8);
    return o;
  }
  static __device__ __forceinline__ F29 from_fq(const Mont& a) {
    return mul(slice(a), from_const(C::FROM_FQ));  // x 2^256 2^266 / 2^261 = x 2^261
  }
  __device__ __forceinline__ Mont to_fq() const {
    return mul(*this, from_const(C::TO_FQ)).pack_reduced();  // x 2^256 + (0 or 1) q, limbs normalised
  }
  // a w for an 8 x 32 Montgomery value a (a 2^256, canonical) and a constant held in this form (w 2^261): the 8 x 32
  // integer re-sliced into 29-bit limbs times w 2^261, divided by 2^261, is (a w) 2^256 + (0 or 1) q -- the product in
  // the 8 x 32 form without a conversion multiplication (~290 instead of ~375 instructions per product)
  static __device__ __forceinline__ Mont mul_mont(const Mont& a, const F29& w29) { return mul(slice(a), w29).pack_reduced(); }
};

using Fq29 = F29<Fq29C, Fq>;
using Fr29 = F29<Fr29C, Fr>;  // same limb form for the scalar field (Poseidon): r / 2^261 = q / 2^261 to 4 digits

// Table entry: the two coordinates as 256-bit integers (the Montgomery-2^261 residues, fully reduced), 64 bytes so an
// entry never straddles a cache line; never the point at infinity (dropped when the table is built).
struct G1Affine29 {
  uint32_t x[8], y[8];
  __device__ __forceinline__ bool is_inf() const {  // (0, 0), as Affine<F>::inf(); never present in the prover tables
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) o |= x[j] | y[j];
    return o == 0;
  }
};
__device__ __forceinline__ Fq29 unpack29(const uint32_t (&w)[8]) {
  Fq29 u;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    const int bit = 29 * j, k = bit >> 5, s = bit & 31;
    uint64_t lo = w[k], hi = k + 1 < 8 ? w[k + 1] : 0;
    u.v[j] = (uint32_t)(((lo | (hi << 32)) >> s) & Fq29::M);
  }
  return u;
}
// exact reduction of a normalised value < 2 q into [0, q), packed as 8 x 32
__device__ __forceinline__ void pack29_reduced(const Fq29& t, uint32_t (&o)[8]) {
  uint32_t d[9];
  int64_t borrow = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    int64_t x = (int64_t)t.v[j] - (int64_t)Fq29C::P[j] + borrow;
    d[j] = (uint32_t)x & (j < 8 ? Fq29::M : 0xFFFFFFFFu);
    borrow = x >> (j < 8 ? 29 : 63);
  }
  const bool ge = borrow == 0;
  uint32_t r[9];
#pragma unroll
  for (int j = 0; j < 9; j++) r[j] = ge ? d[j] : t.v[j];
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const int j0 = (32 * w) / 29, s = 32 * w - 29 * j0;
    uint64_t acc = (uint64_t)r[j0] >> s;
    acc |= (uint64_t)r[j0 + 1] << (29 - s);
    if (j0 + 2 < 9) acc |= (uint64_t)r[j0 + 2] << (58 - s);
    o[w] = (uint32_t)acc;
  }
}
__device__ __forceinline__ G1Affine29 to_table29(const G1Affine& a) {
  G1Affine29 e;
  pack29_reduced(Fq29::from_fq(a.x), e.x);
  pack29_reduced(Fq29::from_fq(a.y), e.y);
  return e;
}

// XYZZ accumulator over Fq29.  Invariants between additions: X, Y, ZZ, ZZZ normalised; X < 5.2 q, Y < 2.1 q,
// ZZ, ZZZ < 1.7 q; infinity <=> ZZ has all limbs zero (a finite point never has ZZ = 0 mod q, and mul never returns
// the all-zero limb pattern for non-zero operands... it can only return 0 for a zero product).
struct G1Acc29 {
  Fq29 X, Y, ZZ, ZZZ;
  static constexpr uint32_t LPP = 1;   // lanes per point (G2AccPair29: 2)
  __device__ __forceinline__ void store_xyzz(G1XYZZ* dst) const { *dst = to_xyzz(); }
  static __device__ __forceinline__ G1Acc29 load_xyzz(const G1XYZZ* src) { return from_xyzz(*src); }
  __device__ __forceinline__ void store_lds(G1Acc29* dst) const { *dst = *this; }

  static __device__ __forceinline__ G1Acc29 inf() { return {Fq29::zero(), Fq29::zero(), Fq29::zero(), Fq29::zero()}; }
  __device__ __forceinline__ bool is_inf() const { return ZZ.limbs_all_zero(); }

  __device__ __forceinline__ G1XYZZ to_xyzz() const {
    if (is_inf()) return G1XYZZ::inf();
    return {X.to_fq(), Y.to_fq(), ZZ.to_fq(), ZZZ.to_fq()};
  }
  static __device__ __forceinline__ G1Acc29 from_xyzz(const G1XYZZ& p) {
    if (p.is_inf()) return inf();
    return {Fq29::from_fq(p.X), Fq29::from_fq(p.Y), Fq29::from_fq(p.ZZ), Fq29::from_fq(p.ZZZ)};
  }

  // 2 (x, py) for an affine point (mdbl-2008-s-1); py normalised, < 2 q
  static __device__ __forceinline__ G1Acc29 dbl_affine(const Fq29& x, const Fq29& py) {
    Fq29 U;
#pragma unroll
    for (int j = 0; j < 9; j++) U.v[j] = 2 * py.v[j];          // lazy, < 4 q
    Fq29 V = Fq29::sqr(U);
    Fq29 W = Fq29::mul(U, V);
    Fq29 S = Fq29::mul(x, V);
    Fq29 x2 = Fq29::sqr(x);
    Fq29 Mm;
#pragma unroll
    for (int j = 0; j < 9; j++) Mm.v[j] = 3 * x2.v[j];
    Mm.normalize();                                              // < 3.3 q
    Fq29 M2 = Fq29::sqr(Mm);
    Fq29 X3;
#pragma unroll
    for (int j = 0; j < 9; j++) X3.v[j] = M2.v[j] + Fq29C::K4T[j] - 2 * S.v[j];
    X3.normalize();                                              // < 5.1 q
    Fq29 D = Fq29::sub(S, Fq29C::K6, X3);
    Fq29 nY = Fq29::neg_lazy(Fq29C::K4, py);
    return {X3, Fq29::dot2(Mm, D, nY, W), V, W};
  }

  // this += (x, +-y)   madd-2008-s in the lazy representation (bounds: header comment).  Statement order keeps few
  // values alive at a time; the equal-x case (doubling / cancellation) stays in this representation too, so the
  // rare path does not raise the kernel's register count.
  __device__ __forceinline__ void madd(const G1Affine29& e, bool negate) {
    const Fq29 px = unpack29(e.x);
    Fq29 py = unpack29(e.y);
    if (negate) py = Fq29::neg_lazy(Fq29C::K2, py);    // lazy, value < 2 q (table y is canonical, < q)
    if (is_inf()) {
      X = px;
      Y = py;
      Y.normalize();
      ZZ = Fq29::from_const(Fq29C::ONE);
      ZZZ = ZZ;
      return;
    }
    // "product + K - b" leaves the reduction's own carry chain normalised (Fq29::mul_add): no second normalisation
    const Fq29 kX = Fq29::neg_lazy(Fq29C::K6, X);
    const Fq29 nY = Fq29::neg_lazy(Fq29C::K4, Y);            // lazy, < 4 q; also the -Y of Y3 below
    Fq29 P = Fq29::mul_add(px, ZZ, kX);                      // U2 - X   in (0.8 q, 7.2 q)
    Fq29 R = Fq29::mul_add(py, ZZZ, nY);                     // S2 - Y   in (1.9 q, 5.2 q)
    if (P.is_zero_mod_q()) {                                // same x
      if (R.is_zero_mod_q()) {
        py.normalize();
        *this = dbl_affine(px, py);
      } else {
        *this = inf();
      }
      return;
    }
    Fq29 PP = Fq29::sqr(P);                           // < 1.4 q
    ZZ = Fq29::mul(ZZ, PP);
    Fq29 Q = Fq29::mul(X, PP);                        // < 1.1 q
    Fq29 PPP = Fq29::mul(P, PP);                      // < 1.1 q
    ZZZ = Fq29::mul(ZZZ, PPP);
    Fq29 kT;
#pragma unroll
    for (int j = 0; j < 9; j++) kT.v[j] = Fq29C::K4T[j] - (PPP.v[j] + 2 * Q.v[j]);   // limbs in (0, 2^31)
    X = Fq29::sqr_add(R, &kT);                        // X3 = R^2 - PPP - 2 Q   in (0.7 q, 5.2 q)
    // Q - X3 < 7.1 q, left un-normalised: limbs < 2^29 + K6[j].  In the dot product below its partner R is normalised,
    // nY is lazy (K4[j] at most) and PPP normalised; with the wide rounds a column is at most
    // 2^29 (sum_j (K6[j] + 2^29) + sum_j K4[j]) + 2^32 sum_j p[j] = 13.8 2^60 < 2^64 (check_fq29_bounds.py "g1.Y3")
    Fq29 D;
#pragma unroll
    for (int j = 0; j < 9; j++) D.v[j] = Q.v[j] + Fq29C::K6[j] - X.v[j];
    Y = Fq29::dot2(R, D, nY, PPP);                    // R (Q - X3) - Y PPP  < 1.3 q
  }

  // this += o, both in this form (add-2008-s): the tree reductions of the small batches (k_sum_tree / k_sum_blocks), where
  // a general addition is a lone lane's dependent chain and the 8 x 32 form costs 1.9 x the instructions.  The tail is
  // madd's with (X, Y) -> (U1, S1): every product and the Y3 dot product take the operand classes madd's call sites
  // take (check_fq29_bounds.py: "mul", "g1.Y3"), P and R are normalised differences < 3.1 q, and the results keep the
  // accumulator invariants (X < 5.2 q, Y < 2.1 q, ZZ, ZZZ < 1.7 q).  Doubling / cancellation: through the 8 x 32 law.
  __device__ __forceinline__ void add(const G1Acc29& o) {
    if (o.is_inf()) return;
    if (is_inf()) {
      *this = o;
      return;
    }
    const Fq29 U1 = Fq29::mul(X, o.ZZ), U2 = Fq29::mul(o.X, ZZ);       // < 1.06 q
    const Fq29 S1 = Fq29::mul(Y, o.ZZZ), S2 = Fq29::mul(o.Y, ZZZ);     // < 1.03 q
    const Fq29 P = Fq29::sub(U2, Fq29C::K2, U1), R = Fq29::sub(S2, Fq29C::K2, S1);   // in (0.9 q, 3.1 q), normalised
    if (P.is_zero_mod_q()) {   // out of line: the rare path's registers are its own
      add_same_x(this, &o);
      return;
    }
    const Fq29 PP = Fq29::sqr(P);
    ZZ = Fq29::mul(Fq29::mul(ZZ, o.ZZ), PP);
    const Fq29 Q = Fq29::mul(U1, PP);
    const Fq29 PPP = Fq29::mul(P, PP);
    ZZZ = Fq29::mul(Fq29::mul(ZZZ, o.ZZZ), PPP);
    Fq29 kT;
#pragma unroll
    for (int j = 0; j < 9; j++) kT.v[j] = Fq29C::K4T[j] - (PPP.v[j] + 2 * Q.v[j]);
    X = Fq29::sqr_add(R, &kT);                        // X3 = R^2 - PPP - 2 Q
    Fq29 D;
#pragma unroll
    for (int j = 0; j < 9; j++) D.v[j] = Q.v[j] + Fq29C::K6[j] - X.v[j];
    const Fq29 nS1 = Fq29::neg_lazy(Fq29C::K4, S1);
    Y = Fq29::dot2(R, D, nS1, PPP);                   // R (Q - X3) - S1 PPP
  }
  static __device__ __noinline__ void add_same_x(G1Acc29* self, const G1Acc29* o) {
    G1XYZZ a = self->to_xyzz();
    a.add(o->to_xyzz());
    *self = from_xyzz(a);
  }
};


// ---------------------------------------------------------------------------------------------------------------
// G1 general addition by a lane PAIR (the sum trees of a lone proof: a level is one addition of a lone lane, all issue
// latency).  Both lanes hold the whole point; the fourteen products of add-2008-s are dealt to the two lanes level by
// level (lane 2 t the U / P / PPP / ZZZ side, lane 2 t + 1 the S / R / Q / ZZ / X3 / Y3 side), six product levels and the
// Y3 dot product instead of fourteen products in a row; what the other side needs crosses by DPP (quad_perm [1, 0, 3, 2]).
// Every product takes the operands G1Acc29::add gives it (R^2 + kT as mul_add(R, R, kT): the column sums of sqr_add), so
// the bounds and the field elements are the same; a lane's idle slots compute on operands of the same classes.
// Both lanes of a pair must take the same branches: they test the same values.
struct G1AccPair29 {
  Fq29 X, Y, ZZ, ZZZ;
  static constexpr uint32_t LPP = 2;
  static __device__ __forceinline__ bool odd() { return (threadIdx.x & 1u) != 0; }
  static __device__ __forceinline__ Fq29 swap(const Fq29& v) {
    Fq29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.v[j], 0xB1, 0xF, 0xF, true);
      asm volatile("" : "+v"(t));   // (see Fq2PairOps::join)
      r.v[j] = t;
    }
    return r;
  }
  static __device__ __forceinline__ Fq29 pick(bool o, const Fq29& if_odd, const Fq29& if_even) {
    Fq29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.v[j] = o ? if_odd.v[j] : if_even.v[j];
    return r;
  }
  static __device__ __forceinline__ G1AccPair29 inf() { return {Fq29::zero(), Fq29::zero(), Fq29::zero(), Fq29::zero()}; }
  __device__ __forceinline__ bool is_inf() const { return ZZ.limbs_all_zero(); }
  __device__ __forceinline__ G1XYZZ to_xyzz() const {
    if (is_inf()) return G1XYZZ::inf();
    return {X.to_fq(), Y.to_fq(), ZZ.to_fq(), ZZZ.to_fq()};
  }
  static __device__ __forceinline__ G1AccPair29 from_xyzz(const G1XYZZ& p) {
    if (p.is_inf()) return inf();
    return {Fq29::from_fq(p.X), Fq29::from_fq(p.Y), Fq29::from_fq(p.ZZ), Fq29::from_fq(p.ZZZ)};
  }
  // in and out of the common form: lane 2 t converts X and Y, lane 2 t + 1 ZZ and ZZZ
  static __device__ __forceinline__ G1AccPair29 load_xyzz(const G1XYZZ* src) {
    const bool o = odd();
    const Fq29 a = Fq29::from_fq(o ? src->ZZ : src->X), b = Fq29::from_fq(o ? src->ZZZ : src->Y);
    const Fq29 xa = swap(a), xb = swap(b);
    return {pick(o, xa, a), pick(o, xb, b), pick(o, a, xa), pick(o, b, xb)};   // (infinity: ZZ = 0 converts to all-zero limbs)
  }
  __device__ __forceinline__ void store_xyzz(G1XYZZ* dst) const {
    const bool o = odd(), z = is_inf();
    const Fq a = z ? Fq::zero() : pick(o, ZZ, X).to_fq(), b = z ? Fq::zero() : pick(o, ZZZ, Y).to_fq();
    *(o ? &dst->ZZ : &dst->X) = a;
    *(o ? &dst->ZZZ : &dst->Y) = b;
  }
  __device__ __forceinline__ void store_lds(G1AccPair29* dst) const {
    if (odd()) {
      dst->ZZ = ZZ;
      dst->ZZZ = ZZZ;
    } else {
      dst->X = X;
      dst->Y = Y;
    }
  }
  __device__ __forceinline__ void add(const G1AccPair29& o) {
    if (o.is_inf()) return;
    if (is_inf()) {
      *this = o;
      return;
    }
    const bool r = odd();
    // level 1, 2:  U1 = X oZZ | S1 = Y oZZZ ;  U2 = oX ZZ | S2 = oY ZZZ ;  d = P | R
    const Fq29 t1 = Fq29::mul(pick(r, Y, X), pick(r, o.ZZZ, o.ZZ));
    const Fq29 t2 = Fq29::mul(pick(r, o.Y, o.X), pick(r, ZZZ, ZZ));
    const Fq29 d = Fq29::sub(t2, Fq29C::K2, t1);          // P | R, in (0.9 q, 3.1 q), normalised
    {
      const Fq29 xd = swap(d);
      const Fq29 P = pick(r, xd, d);
      if (P.is_zero_mod_q()) {   // same x: doubling / cancellation, through the 8 x 32 law (both lanes, in full)
        add_same_x(this, &o);
        return;
      }
    }
    // level 3:  PP = P P | ZZt = ZZ oZZ
    const Fq29 t3 = Fq29::mul(pick(r, ZZ, d), pick(r, o.ZZ, d));
    const Fq29 xU1 = swap(t1), xPP = swap(t3);
    // level 4:  PPP = P PP | Q = U1 PP
    const Fq29 t4 = Fq29::mul(pick(r, xU1, d), pick(r, xPP, t3));
    // level 5:  ZZZt = ZZZ oZZZ | ZZ3 = ZZt PP
    const Fq29 t5 = Fq29::mul(pick(r, t3, ZZZ), pick(r, xPP, o.ZZZ));
    const Fq29 xPPP = swap(t4);
    // level 6:  ZZZ3 = ZZZt PPP | X3 = R R + K4T - PPP - 2 Q
    Fq29 kT;
#pragma unroll
    for (int j = 0; j < 9; j++) kT.v[j] = r ? Fq29C::K4T[j] - (xPPP.v[j] + 2 * t4.v[j]) : 0u;
    const Fq29 t6 = Fq29::mul_add(pick(r, d, t5), pick(r, d, t4), kT);
    // level 7 (the odd lane's; the even lane computes on operands of the same classes and drops the result):
    //   Y3 = R (Q + K6 - X3) + (K4 - S1) PPP
    Fq29 D;
#pragma unroll
    for (int j = 0; j < 9; j++) D.v[j] = t4.v[j] + Fq29C::K6[j] - t6.v[j];
    const Fq29 nS1 = Fq29::neg_lazy(Fq29C::K4, t1);
    const Fq29 t7 = Fq29::dot2(d, D, nS1, xPPP);
    const Fq29 x5 = swap(t5), x6 = swap(t6), x7 = swap(t7);
    X = pick(r, t6, x6);
    Y = pick(r, t7, x7);
    ZZ = pick(r, t5, x5);
    ZZZ = pick(r, x6, t6);
  }
  static __device__ __noinline__ void add_same_x(G1AccPair29* self, const G1AccPair29* o) {
    G1XYZZ a = self->to_xyzz();
    a.add(o->to_xyzz());
    *self = from_xyzz(a);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// G2: Fq2 = Fq[u]/(u^2 + 1) over Fq29.  Same lazy bounds as G1, component-wise.
struct Fq2_29 {
  Fq29 c0, c1;
  static __device__ __forceinline__ Fq2_29 mul(const Fq2_29& a, const Fq2_29& b) {  // a, b normalised
    Fq29 n1 = Fq29::neg_lazy(Fq29C::K8, a.c1);   // a.c1 < 7.9 q
    return {Fq29::dot2(a.c0, b.c0, n1, b.c1), Fq29::dot2(a.c0, b.c1, a.c1, b.c0)};
  }
  static __device__ __forceinline__ Fq2_29 sqr(const Fq2_29& a) {                   // (a0 + a1)(a0 - a1), 2 a0 a1
    Fq29 s, d, t;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      s.v[j] = a.c0.v[j] + a.c1.v[j];   // lazy
      t.v[j] = 2 * a.c0.v[j];           // lazy
    }
    d = Fq29::sub(a.c0, Fq29C::K8, a.c1);
    return {Fq29::mul(s, d), Fq29::mul(t, a.c1)};
  }
  // a b + add, add = (K - x.c0, K - x.c1) limb by limb: normalised by the reductions' own carry chains
  static __device__ __forceinline__ Fq2_29 mul_add(const Fq2_29& a, const Fq2_29& b, const Fq2_29& add) {
    Fq29 n1 = Fq29::neg_lazy(Fq29C::K8, a.c1);
    return {Fq29::dot2_add(a.c0, b.c0, n1, b.c1, add.c0), Fq29::dot2_add(a.c0, b.c1, a.c1, b.c0, add.c1)};
  }
  static __device__ __forceinline__ Fq2_29 sqr_add(const Fq2_29& a, const Fq2_29& add) {
    Fq29 s, d, t;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      s.v[j] = a.c0.v[j] + a.c1.v[j];   // lazy
      t.v[j] = 2 * a.c0.v[j];           // lazy
    }
    d = Fq29::sub(a.c0, Fq29C::K8, a.c1);
    return {Fq29::mul_add(s, d, add.c0), Fq29::mul_add(t, a.c1, add.c1)};
  }
  static __device__ __forceinline__ Fq2_29 neg_lazy(const uint32_t (&K)[9], const Fq2_29& b) {
    return {Fq29::neg_lazy(K, b.c0), Fq29::neg_lazy(K, b.c1)};
  }
  static __device__ __forceinline__ Fq2_29 sub(const Fq2_29& a, const uint32_t (&K)[9], const Fq2_29& b) {
    return {Fq29::sub(a.c0, K, b.c0), Fq29::sub(a.c1, K, b.c1)};
  }
  __device__ __forceinline__ bool is_zero_mod_q() const { return c0.is_zero_mod_q() && c1.is_zero_mod_q(); }
  static __device__ __forceinline__ Fq2_29 from_fq2(const Fq2& a) { return {Fq29::from_fq(a.c0), Fq29::from_fq(a.c1)}; }
  __device__ __forceinline__ Fq2 to_fq2() const { return {c0.to_fq(), c1.to_fq()}; }
};

struct G2Affine29 {  // 128 bytes: x.c0, x.c1, y.c0, y.c1 packed as in G1Affine29
  uint32_t x0[8], x1[8], y0[8], y1[8];
};
__device__ __forceinline__ G2Affine29 to_table29(const G2Affine& a) {
  G2Affine29 e;
  pack29_reduced(Fq29::from_fq(a.x.c0), e.x0);
  pack29_reduced(Fq29::from_fq(a.x.c1), e.x1);
  pack29_reduced(Fq29::from_fq(a.y.c0), e.y0);
  pack29_reduced(Fq29::from_fq(a.y.c1), e.y1);
  return e;
}

// ---------------------------------------------------------------------------------------------------------------
// Who computes an Fq2 product.  Fq2LaneOps: the lane that holds the operands, both components (the throughput walks: a
// lane = a proof).  Fq2PairOps: a PAIR of adjacent lanes holds the same operands and each computes ONE component, then
// they swap (nine DPP moves) -- for the lone dependent chains of a single proof (tiny G2 walk, G2 sum trees), where a
// lane's issue rate is the latency: an Fq2 product costs a lane one dot product instead of two (1.65 x fewer instructions
// per addition, the selects and swaps included).  Lane 2 t computes c0 with exactly the operands Fq2LaneOps gives the c0
// product, lane 2 t + 1 c1 with the c1 operands: the same call sites for check_fq29_bounds.py, the same field elements.
// Both lanes of a pair must take the same branches (they hold the same values, so they do).
struct Fq2LaneOps {
  static constexpr uint32_t LPP = 1;   // lanes per point
  static __device__ __forceinline__ Fq2_29 mul(const Fq2_29& a, const Fq2_29& b) { return Fq2_29::mul(a, b); }
  static __device__ __forceinline__ Fq2_29 sqr(const Fq2_29& a) { return Fq2_29::sqr(a); }
  static __device__ __forceinline__ Fq2_29 mul_add(const Fq2_29& a, const Fq2_29& b, const Fq2_29& add) { return Fq2_29::mul_add(a, b, add); }
  static __device__ __forceinline__ Fq2_29 sqr_add(const Fq2_29& a, const Fq2_29& add) { return Fq2_29::sqr_add(a, add); }
  // R D - S T with nS = K4 - S (lazy): four base products per component, one reduction each
  static __device__ __forceinline__ Fq2_29 rd_minus_st(const Fq2_29& R, const Fq2_29& D, const Fq2_29& nS, const Fq2_29& S,
                                                       const Fq2_29& T) {
    const Fq29 nR1 = Fq29::neg_lazy(Fq29C::K8, R.c1);
    return {Fq29::dot4<true>(R.c0, D.c0, nR1, D.c1, nS.c0, T.c0, S.c1, T.c1),
            Fq29::dot4<true>(R.c0, D.c1, R.c1, D.c0, nS.c0, T.c1, nS.c1, T.c0)};
  }
};
struct Fq2PairOps {
  static constexpr uint32_t LPP = 2;
  static __device__ __forceinline__ bool odd() { return (threadIdx.x & 1u) != 0; }
  static __device__ __forceinline__ Fq29 pick(bool o, const Fq29& if_odd, const Fq29& if_even) {
    Fq29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.v[j] = o ? if_odd.v[j] : if_even.v[j];
    return r;
  }
  // own component computed, the partner's fetched: quad_perm [1, 0, 3, 2]
  static __device__ __forceinline__ Fq2_29 join(bool o, const Fq29& own) {
    Fq29 other;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own.v[j], 0xB1, 0xF, 0xF, true);
      asm volatile("" : "+v"(t));   // pins the move in front of the selects (a move sunk into a one-sided branch reads a switched-off lane)
      other.v[j] = t;
    }
    return {pick(o, other, own), pick(o, own, other)};
  }
  static __device__ __forceinline__ Fq2_29 mul(const Fq2_29& a, const Fq2_29& b) {
    const bool o = odd();
    const Fq29 n1 = Fq29::neg_lazy(Fq29C::K8, a.c1);
    return join(o, Fq29::dot2(a.c0, pick(o, b.c1, b.c0), pick(o, a.c1, n1), pick(o, b.c0, b.c1)));
  }
  static __device__ __forceinline__ Fq2_29 mul_add(const Fq2_29& a, const Fq2_29& b, const Fq2_29& add) {
    const bool o = odd();
    const Fq29 n1 = Fq29::neg_lazy(Fq29C::K8, a.c1);
    return join(o, Fq29::dot2_add(a.c0, pick(o, b.c1, b.c0), pick(o, a.c1, n1), pick(o, b.c0, b.c1), pick(o, add.c1, add.c0)));
  }
  // c0 = (a0 + a1)(a0 - a1), c1 = (2 a0) a1
  static __device__ __forceinline__ void sqr_operands(bool o, const Fq2_29& a, Fq29* x, Fq29* y) {
    const Fq29 d = Fq29::sub(a.c0, Fq29C::K8, a.c1);
#pragma unroll
    for (int j = 0; j < 9; j++) x->v[j] = a.c0.v[j] + (o ? a.c0.v[j] : a.c1.v[j]);   // lazy
    *y = pick(o, a.c1, d);
  }
  static __device__ __forceinline__ Fq2_29 sqr(const Fq2_29& a) {
    const bool o = odd();
    Fq29 x, y;
    sqr_operands(o, a, &x, &y);
    return join(o, Fq29::mul(x, y));
  }
  static __device__ __forceinline__ Fq2_29 sqr_add(const Fq2_29& a, const Fq2_29& add) {
    const bool o = odd();
    Fq29 x, y;
    sqr_operands(o, a, &x, &y);
    return join(o, Fq29::mul_add(x, y, pick(o, add.c1, add.c0)));
  }
  static __device__ __forceinline__ Fq2_29 rd_minus_st(const Fq2_29& R, const Fq2_29& D, const Fq2_29& nS, const Fq2_29& S,
                                                       const Fq2_29& T) {
    const bool o = odd();
    const Fq29 nR1 = Fq29::neg_lazy(Fq29C::K8, R.c1);
    return join(o, Fq29::dot4<true>(R.c0, pick(o, D.c1, D.c0), pick(o, R.c1, nR1), pick(o, D.c0, D.c1), nS.c0,
                                    pick(o, T.c1, T.c0), pick(o, nS.c1, S.c1), pick(o, T.c0, T.c1)));
  }
};

// Bounds as in G1Acc29 (X < 5.2 q, Y < 2.1 q, ZZ, ZZZ < 1.7 q per component); products of two Fq2 values add two
// base products per component, which the 0.0059 factor absorbs (e.g. P P: 2 * 7.2^2 * 0.0059 + 1 < 1.7).
// O = Fq2LaneOps (G2Acc29: a lane per point) or Fq2PairOps (G2AccPair29: a lane pair per point, both holding all of it).
template <class O>
struct G2AccT {
  Fq2_29 X, Y, ZZ, ZZZ;
  static constexpr uint32_t LPP = O::LPP;
  static __device__ __forceinline__ G2AccT inf() {
    Fq2_29 z{Fq29::zero(), Fq29::zero()};
    return {z, z, z, z};
  }
  __device__ __forceinline__ bool is_inf() const { return ZZ.c0.limbs_all_zero() && ZZ.c1.limbs_all_zero(); }
  __device__ __forceinline__ G2XYZZ to_xyzz() const {
    if (is_inf()) return G2XYZZ::inf();
    return {X.to_fq2(), Y.to_fq2(), ZZ.to_fq2(), ZZZ.to_fq2()};
  }
  static __device__ __forceinline__ G2AccT from_xyzz(const G2XYZZ& p) {
    if (p.is_inf()) return inf();
    return {Fq2_29::from_fq2(p.X), Fq2_29::from_fq2(p.Y), Fq2_29::from_fq2(p.ZZ), Fq2_29::from_fq2(p.ZZZ)};
  }
  // *dst = this point in the common form / this = *src.  A lane pair converts and moves one component per lane.
  __device__ __forceinline__ void store_xyzz(G2XYZZ* dst) const {
    if (LPP == 1) {
      *dst = to_xyzz();
      return;
    }
    const bool o = Fq2PairOps::odd(), z = is_inf();
    const Fq2_29* const c[4] = {&X, &Y, &ZZ, &ZZZ};
    Fq2* const d[4] = {&dst->X, &dst->Y, &dst->ZZ, &dst->ZZZ};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const Fq v = z ? Fq::zero() : Fq2PairOps::pick(o, c[k]->c1, c[k]->c0).to_fq();
      *(o ? &d[k]->c1 : &d[k]->c0) = v;
    }
  }
  // the point into LDS for another pair / lane to add: a pair writes two coordinates per lane
  __device__ __forceinline__ void store_lds(G2AccT* dst) const {
    if (LPP == 1) {
      *dst = *this;
    } else if (Fq2PairOps::odd()) {
      dst->ZZ = ZZ;
      dst->ZZZ = ZZZ;
    } else {
      dst->X = X;
      dst->Y = Y;
    }
  }
  static __device__ __forceinline__ G2AccT load_xyzz(const G2XYZZ* src) {
    if (LPP == 1) return from_xyzz(*src);
    const bool o = Fq2PairOps::odd();
    const Fq2* const c[4] = {&src->X, &src->Y, &src->ZZ, &src->ZZZ};
    G2AccT r;
    Fq2_29* const d[4] = {&r.X, &r.Y, &r.ZZ, &r.ZZZ};
#pragma unroll
    for (int k = 0; k < 4; k++) *d[k] = Fq2PairOps::join(o, Fq29::from_fq(o ? c[k]->c1 : c[k]->c0));
    // (infinity in the common form is ZZ = 0: from_fq(0) has all limbs zero, so is_inf() holds)
    return r;
  }

  // 2 (x, y), y normalised < 2 q per component (mdbl-2008-s-1).  Rare (a walk meets its own point): computed in full by
  // the lane, and by both lanes of a pair alike.
  static __device__ __forceinline__ G2AccT dbl_affine(const Fq2_29& x, const Fq2_29& y) {
    Fq2_29 U;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      U.c0.v[j] = 2 * y.c0.v[j];
      U.c1.v[j] = 2 * y.c1.v[j];
    }
    U.c0.normalize();
    U.c1.normalize();                           // < 4 q
    Fq2_29 V = Fq2_29::sqr(U);
    Fq2_29 W = Fq2_29::mul(U, V);
    Fq2_29 S = Fq2_29::mul(x, V);
    Fq2_29 x2 = Fq2_29::sqr(x);
    Fq2_29 Mm;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      Mm.c0.v[j] = 3 * x2.c0.v[j];
      Mm.c1.v[j] = 3 * x2.c1.v[j];
    }
    Mm.c0.normalize();
    Mm.c1.normalize();                          // < 3.1 q
    Fq2_29 M2 = Fq2_29::sqr(Mm);
    Fq2_29 X3;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      X3.c0.v[j] = M2.c0.v[j] + Fq29C::K4T[j] - 2 * S.c0.v[j];
      X3.c1.v[j] = M2.c1.v[j] + Fq29C::K4T[j] - 2 * S.c1.v[j];
    }
    X3.c0.normalize();
    X3.c1.normalize();                          // < 5.5 q
    Fq2_29 D = Fq2_29::sub(S, Fq29C::K6, X3);
    Fq29 nM1 = Fq29::neg_lazy(Fq29C::K8, Mm.c1);
    Fq29 ny0 = Fq29::neg_lazy(Fq29C::K4, y.c0);
    Fq29 ny1 = Fq29::neg_lazy(Fq29C::K4, y.c1);
    Fq29 y0 = Fq29::dot4<true>(Mm.c0, D.c0, nM1, D.c1, W.c0, ny0, W.c1, y.c1);   // M D - W y
    Fq29 y1 = Fq29::dot4<true>(Mm.c0, D.c1, Mm.c1, D.c0, W.c0, ny1, W.c1, ny0);
    return {X3, {y0, y1}, V, W};
  }

  __device__ __forceinline__ void madd(const G2Affine29& e, bool negate) {
    const Fq2_29 px{unpack29(e.x0), unpack29(e.x1)};
    Fq2_29 py{unpack29(e.y0), unpack29(e.y1)};
    if (negate) {
      py.c0 = Fq29::neg_lazy(Fq29C::K2, py.c0);
      py.c1 = Fq29::neg_lazy(Fq29C::K2, py.c1);
      py.c0.normalize();
      py.c1.normalize();
    }
    if (is_inf()) {
      X = px;
      Y = py;
      ZZ = {Fq29::from_const(Fq29C::ONE), Fq29::zero()};
      ZZZ = ZZ;
      return;
    }
    const Fq2_29 kX = Fq2_29::neg_lazy(Fq29C::K6, X);
    const Fq2_29 nY = Fq2_29::neg_lazy(Fq29C::K4, Y);      // also the -Y of Y3 below
    Fq2_29 P = O::mul_add(px, ZZ, kX);
    Fq2_29 R = O::mul_add(py, ZZZ, nY);
    if (P.is_zero_mod_q()) {  // same x: doubling or cancellation (rare), kept in this form: no extra registers
      if (R.is_zero_mod_q()) *this = dbl_affine(px, py); else *this = inf();
      return;
    }
    Fq2_29 PP = O::sqr(P);
    ZZ = O::mul(ZZ, PP);
    Fq2_29 Q = O::mul(X, PP);
    Fq2_29 PPP = O::mul(P, PP);
    ZZZ = O::mul(ZZZ, PPP);
    Fq2_29 kT;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      kT.c0.v[j] = Fq29C::K4T[j] - (PPP.c0.v[j] + 2 * Q.c0.v[j]);
      kT.c1.v[j] = Fq29C::K4T[j] - (PPP.c1.v[j] + 2 * Q.c1.v[j]);
    }
    X = O::sqr_add(R, kT);                                  // X3 = R^2 - PPP - 2 Q
    Fq2_29 D = Fq2_29::sub(Q, Fq29C::K6, X);
    Y = O::rd_minus_st(R, D, nY, Y, PPP);                   // Y3 = R D - Y PPP
  }

  // this += *o (see G1Acc29::add): madd's tail with (X, Y) -> (U1, S1), the same operand classes at every call site.
  // The operand is read from memory (LDS in the sum trees) coordinate by coordinate and in the order that lets the
  // operands die early -- a G2 point is 72 words, and with both points, U1, U2, S1, S2 and a product's accumulators alive
  // at once the kernel spilled 283 registers; the compiler barriers keep the loads where they are written.
  __device__ __forceinline__ void add(const G2AccT* o) {
    if (o->is_inf()) return;
    if (is_inf()) {
      *this = *o;
      return;
    }
    Fq2_29 U1, P, S1, R;
    {
      const Fq2_29 oZZ = o->ZZ, oX = o->X;
      U1 = O::mul(X, oZZ);
      const Fq2_29 U2 = O::mul(oX, ZZ);
      P = Fq2_29::sub(U2, Fq29C::K2, U1);
    }
    asm volatile("" ::: "memory");
    {
      const Fq2_29 oZZZ = o->ZZZ, oY = o->Y;
      S1 = O::mul(Y, oZZZ);
      const Fq2_29 S2 = O::mul(oY, ZZZ);
      R = Fq2_29::sub(S2, Fq29C::K2, S1);
    }
    if (P.is_zero_mod_q()) {   // doubling / cancellation: through the 8 x 32 law, out of line (its registers are its own)
      add_same_x(this, o);
      return;
    }
    asm volatile("" ::: "memory");
    {
      const Fq2_29 oZZ = o->ZZ;
      ZZ = O::mul(ZZ, oZZ);
    }
    {
      const Fq2_29 oZZZ = o->ZZZ;
      ZZZ = O::mul(ZZZ, oZZZ);
    }
    asm volatile("" ::: "memory");
    const Fq2_29 PP = O::sqr(P);
    ZZ = O::mul(ZZ, PP);
    const Fq2_29 Q = O::mul(U1, PP);
    const Fq2_29 PPP = O::mul(P, PP);
    ZZZ = O::mul(ZZZ, PPP);
    Fq2_29 kT;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      kT.c0.v[j] = Fq29C::K4T[j] - (PPP.c0.v[j] + 2 * Q.c0.v[j]);
      kT.c1.v[j] = Fq29C::K4T[j] - (PPP.c1.v[j] + 2 * Q.c1.v[j]);
    }
    X = O::sqr_add(R, kT);
    const Fq2_29 D = Fq2_29::sub(Q, Fq29C::K6, X);
    const Fq2_29 nS = Fq2_29::neg_lazy(Fq29C::K4, S1);
    Y = O::rd_minus_st(R, D, nS, S1, PPP);                  // Y3 = R D - S1 PPP
  }
  __device__ __forceinline__ void add(const G2AccT& o) { add(&o); }
  static __device__ __noinline__ void add_same_x(G2AccT* self, const G2AccT* o) {
    G2XYZZ a = self->to_xyzz();
    a.add(o->to_xyzz());
    *self = from_xyzz(a);
  }
};
typedef G2AccT<Fq2LaneOps> G2Acc29;
typedef G2AccT<Fq2PairOps> G2AccPair29;

}  // namespace rlnamd
