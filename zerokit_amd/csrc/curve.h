// BN254 G1 / G2 group arithmetic (short Weierstrass, a = 0) templated on the coordinate field
// (Fq for G1, Fq2 for the sextic twist G2).
//
// Replaces ark-ec 0.5.0 `short_weierstrass::{Affine, Projective}` (third party, pinned in
// /root/reference/Cargo.lock) as used by the Groth16 assembly in
// /root/reference/rln/src/partial_proof.rs:182-274.  arkworks accumulates in Jacobian coordinates; here
// the accumulator is XYZZ (X, Y, ZZ = Z^2, ZZZ = Z^3): a mixed add is 8M + 2S with no per-add squaring of
// Z, which is the cheapest complete-enough form for the table-driven MSM (all addends are affine table
// entries).  Group elements are canonical, so the choice of coordinates cannot change any output bit.
#pragma once
#include "field.h"

namespace rlnamd {

// Affine point; the all-zero pair (0,0) is not on either curve (b != 0) and encodes infinity.
template <class F>
struct Affine {
  F x, y;
  static RLN_HD Affine inf() { return {F::zero(), F::zero()}; }
  RLN_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  RLN_HD Affine neg() const { return {x, y.neg()}; }
};

// a*b - c*d; for prime fields the two products share one Montgomery reduction on the device
template <class P>
RLN_HD Fp<P> fused_sub(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
  return Fp<P>::dot2_sub(a, b, c, d);
}
// Fq2: both components of a*b - c*d are 4-term dot products over Fq (4 p^2 < p 2^256): 8 products, 2 reductions
RLN_HD Fq2 fused_sub(const Fq2& a, const Fq2& b, const Fq2& c, const Fq2& d) {
  Fq nc0 = c.c0.neg(), nc1 = c.c1.neg();
  return {Fq::dot4(a.c0, b.c0, a.c1.neg(), b.c1, nc0, d.c0, c.c1, d.c1),
          Fq::dot4(a.c0, b.c1, a.c1, b.c0, nc0, d.c1, nc1, d.c0)};
}

template <class F>
struct XYZZ {
  F X, Y, ZZ, ZZZ;
  static RLN_HD XYZZ inf() { return {F::one(), F::one(), F::zero(), F::zero()}; }
  RLN_HD bool is_inf() const { return ZZ.is_zero(); }
  static RLN_HD XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    return {p.x, p.y, F::one(), F::one()};
  }
  RLN_HD XYZZ neg() const { return {X, Y.neg(), ZZ, ZZZ}; }

  // 2*P for affine P (mdbl-2008-s-1)
  static RLN_HD XYZZ dbl_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    F U = p.y.dbl();
    F V = U.sqr();
    F W = U * V;
    F S = p.x * V;
    F x2 = p.x.sqr();
    F M = x2.dbl() + x2;
    F X3 = M.sqr() - S.dbl();
    F Y3 = M * (S - X3) - W * p.y;
    return {X3, Y3, V, W};
  }
  // dbl-2008-s-1
  RLN_HD XYZZ dbl() const {
    if (is_inf()) return *this;
    F U = Y.dbl();
    F V = U.sqr();
    F W = U * V;
    F S = X * V;
    F x2 = X.sqr();
    F M = x2.dbl() + x2;
    F X3 = M.sqr() - S.dbl();
    F Y3 = M * (S - X3) - W * Y;
    return {X3, Y3, V * ZZ, W * ZZZ};
  }
  // this += affine p   (madd-2008-s, with the doubling / cancellation cases handled)
  RLN_HD void madd(const Affine<F>& p) {
    if (p.is_inf()) return;
    if (is_inf()) {
      X = p.x;
      Y = p.y;
      ZZ = F::one();
      ZZZ = F::one();
      return;
    }
    F U2 = p.x * ZZ;
    F S2 = p.y * ZZZ;
    F P = U2 - X;
    F R = S2 - Y;
    if (P.is_zero()) {
      if (R.is_zero())
        *this = dbl_affine(p);
      else
        *this = inf();
      return;
    }
    F PP = P.sqr();
    F PPP = P * PP;
    F Q = X * PP;
    F X3 = R.sqr() - PPP - Q.dbl();
    Y = fused_sub(R, Q - X3, Y, PPP);  // R (Q - X3) - Y PPP, one reduction
    X = X3;
    ZZ = ZZ * PP;
    ZZZ = ZZZ * PPP;
  }
  // this += o   (add-2008-s)
  RLN_HD void add(const XYZZ& o) {
    if (o.is_inf()) return;
    if (is_inf()) {
      *this = o;
      return;
    }
    F U1 = X * o.ZZ;
    F U2 = o.X * ZZ;
    F S1 = Y * o.ZZZ;
    F S2 = o.Y * ZZZ;
    F P = U2 - U1;
    F R = S2 - S1;
    if (P.is_zero()) {
      if (R.is_zero())
        *this = dbl();
      else
        *this = inf();
      return;
    }
    F PP = P.sqr();
    F PPP = P * PP;
    F Q = U1 * PP;
    F X3 = R.sqr() - PPP - Q.dbl();
    Y = fused_sub(R, Q - X3, S1, PPP);
    X = X3;
    ZZ = ZZ * o.ZZ * PP;
    ZZZ = ZZZ * o.ZZZ * PPP;
  }
  // one inversion: 1/Z = ZZ/ZZZ
  RLN_HD Affine<F> to_affine() const {
    if (is_inf()) return Affine<F>::inf();
    F izzz = ZZZ.inv();
    F iz = ZZ * izzz;
    F izz = iz.sqr();
    return {X * izz, Y * izzz};
  }
};

// k * P, k canonical 256-bit little-endian limbs (double-and-add, MSB first)
template <class F>
RLN_HD XYZZ<F> scalar_mul(const Affine<F>& p, const uint32_t* k) {
  XYZZ<F> acc = XYZZ<F>::inf();
  bool started = false;
  for (int i = 255; i >= 0; i--) {
    if (started) acc = acc.dbl();
    if ((k[i >> 5] >> (i & 31)) & 1) {
      acc.madd(p);
      started = true;
    }
  }
  return acc;
}

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

}  // namespace rlnamd
