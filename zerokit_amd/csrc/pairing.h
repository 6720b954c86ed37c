// Host-side Groth16 verifier for BN254 (optimal-ate pairing product check).
//
// Replaces ark-groth16 0.5.0 `Groth16::verify_proof` + `prepare_verifying_key` (third party) as called by
// verify_zk_proof, /root/reference/rln/src/protocol/proof.rs:856-894.  Verification stays on the CPU in
// the reference and here: it is the oracle of the proving path, not part of the metric.
// Plain textbook structure: Fq12 = Fq2[w]/(w^6 - xi), affine twist arithmetic, line functions evaluated
// sparsely, one shared final exponentiation by (q^12 - 1)/r.
#pragma once
#include <vector>

#include "curve.h"
#include "zkey.h"

namespace rlnamd {

struct Fq12 {
  Fq2 c[6];
  static Fq12 one() {
    Fq12 r;
    r.c[0] = Fq2::one();
    for (int i = 1; i < 6; i++) r.c[i] = Fq2::zero();
    return r;
  }
  bool is_one() const {
    if (c[0] != Fq2::one()) return false;
    for (int i = 1; i < 6; i++)
      if (!c[i].is_zero()) return false;
    return true;
  }
};

inline Fq2 fq2_from_limbs(const uint32_t v[2][8]) { return {Fq::from_canonical(v[0]), Fq::from_canonical(v[1])}; }

inline Fq12 f12_mul_schoolbook(const Fq12& a, const Fq12& b) {
  static const Fq2 xi{Fq::from_u32(9), Fq::one()};
  Fq2 t[11];
  for (auto& x : t) x = Fq2::zero();
  for (int i = 0; i < 6; i++) {
    if (a.c[i].is_zero()) continue;
    for (int j = 0; j < 6; j++) {
      if (b.c[j].is_zero()) continue;
      t[i + j] = t[i + j] + a.c[i] * b.c[j];
    }
  }
  Fq12 r;
  for (int k = 10; k >= 6; k--) t[k - 6] = t[k - 6] + t[k] * xi;
  for (int k = 0; k < 6; k++) r.c[k] = t[k];
  return r;
}

// ---- tower arithmetic: Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u.  The flat coefficient vector
// c[0..5] of w^0..w^5 is (A, B) with A = (c0, c2, c4), B = (c1, c3, c5) over v = w^2.  Karatsuba at both levels: a
// product is 18 Fq2 products and a squaring 12, against 36 for the schoolbook form above (kept as the cross-check of
// the host test); the verifier spends most of its time here.
struct Fq6 {
  Fq2 a, b, c;  // a + b v + c v^2
};
inline Fq2 mul_xi(const Fq2& x) {  // (9 + u) x with additions only: 9 x = 8 x + x
  const Fq2 x2 = x.dbl(), x4 = x2.dbl(), x9 = x4.dbl() + x;
  return {x9.c0 - x.c1, x9.c1 + x.c0};
}
inline Fq6 f6_add(const Fq6& x, const Fq6& y) { return {x.a + y.a, x.b + y.b, x.c + y.c}; }
inline Fq6 f6_sub(const Fq6& x, const Fq6& y) { return {x.a - y.a, x.b - y.b, x.c - y.c}; }
inline Fq6 f6_mul_v(const Fq6& x) { return {mul_xi(x.c), x.a, x.b}; }
inline Fq6 f6_mul(const Fq6& x, const Fq6& y) {  // 6 Fq2 products
  const Fq2 v0 = x.a * y.a, v1 = x.b * y.b, v2 = x.c * y.c;
  const Fq2 t12 = (x.b + x.c) * (y.b + y.c) - v1 - v2;
  const Fq2 t01 = (x.a + x.b) * (y.a + y.b) - v0 - v1;
  const Fq2 t02 = (x.a + x.c) * (y.a + y.c) - v0 - v2;
  return {v0 + mul_xi(t12), t01 + mul_xi(v2), t02 + v1};
}
// x * (y0 + y1 v): 5 Fq2 products
inline Fq6 f6_mul_01(const Fq6& x, const Fq2& y0, const Fq2& y1) {
  const Fq2 v0 = x.a * y0, v1 = x.b * y1;
  const Fq2 t01 = (x.a + x.b) * (y0 + y1) - v0 - v1;
  return {v0 + mul_xi(x.c * y1), t01, x.c * y0 + v1};
}
inline Fq6 f6_even(const Fq12& f) { return {f.c[0], f.c[2], f.c[4]}; }
inline Fq6 f6_odd(const Fq12& f) { return {f.c[1], f.c[3], f.c[5]}; }
inline Fq12 f12_from(const Fq6& A, const Fq6& B) {
  Fq12 r;
  r.c[0] = A.a; r.c[2] = A.b; r.c[4] = A.c;
  r.c[1] = B.a; r.c[3] = B.b; r.c[5] = B.c;
  return r;
}
inline Fq12 f12_mul(const Fq12& x, const Fq12& y) {  // 3 Fq6 products
  const Fq6 A = f6_even(x), B = f6_odd(x), C = f6_even(y), D = f6_odd(y);
  const Fq6 ac = f6_mul(A, C), bd = f6_mul(B, D);
  const Fq6 cross = f6_sub(f6_sub(f6_mul(f6_add(A, B), f6_add(C, D)), ac), bd);
  return f12_from(f6_add(ac, f6_mul_v(bd)), cross);
}
inline Fq12 f12_sqr(const Fq12& x) {  // complex squaring: 2 Fq6 products
  const Fq6 A = f6_even(x), B = f6_odd(x);
  const Fq6 t = f6_mul(A, B);
  const Fq6 s = f6_mul(f6_add(A, B), f6_add(A, f6_mul_v(B)));   // A^2 + v B^2 + (1 + v) A B
  return f12_from(f6_sub(f6_sub(s, t), f6_mul_v(t)), f6_add(t, t));
}
// f * (l0 + l1 w + l3 w^3) with l0 in Fq: the value of a line at a G1 point (A' = (l0, 0, 0), B' = (l1, l3, 0))
inline Fq12 f12_mul_line(const Fq12& f, const Fq& l0, const Fq2& l1, const Fq2& l3) {
  const Fq6 A = f6_even(f), B = f6_odd(f);
  const Fq6 aa{A.a.mul_fq(l0), A.b.mul_fq(l0), A.c.mul_fq(l0)};
  const Fq6 bb = f6_mul_01(B, l1, l3);
  const Fq2 s0{l1.c0 + l0, l1.c1};
  const Fq6 cross = f6_sub(f6_sub(f6_mul_01(f6_add(A, B), s0, l3), aa), bb);
  return f12_from(f6_add(aa, f6_mul_v(bb)), cross);
}

// line through twist points T, Q evaluated at P after the untwist (x', y') -> (x' w^2, y' w^3);
// advances T to T + Q (or 2T when T == Q)
inline Fq12 line_and_add(G2Affine* T, const G2Affine& Qp, const G1Affine& P) {
  Fq2 lam;
  if (T->x == Qp.x && T->y == Qp.y) {
    Fq2 x2 = T->x.sqr();
    lam = (x2.dbl() + x2) * T->y.dbl().inv();
  } else {
    lam = (Qp.y - T->y) * (Qp.x - T->x).inv();
  }
  Fq2 x3 = lam.sqr() - T->x - Qp.x;
  Fq2 y3 = lam * (T->x - x3) - T->y;
  Fq12 l;
  for (auto& x : l.c) x = Fq2::zero();
  l.c[0] = {P.y, Fq::zero()};
  l.c[1] = lam.mul_fq(P.x).neg();
  l.c[3] = lam * T->x - T->y;
  *T = {x3, y3};
  return l;
}

inline Fq12 miller_loop(const G1Affine& P, const G2Affine& Qp) {
  Fq12 f = Fq12::one();
  if (P.is_inf() || Qp.is_inf()) return f;
  G2Affine T = Qp;
  for (int i = ATE_LOOP_BITS - 2; i >= 0; i--) {
    Fq12 l = line_and_add(&T, T, P);
    f = f12_mul(f12_sqr(f), l);
    if ((ATE_LOOP[i >> 5] >> (i & 31)) & 1) {
      l = line_and_add(&T, Qp, P);
      f = f12_mul(f, l);
    }
  }
  static const Fq2 g12 = fq2_from_limbs(FROB_G12), g13 = fq2_from_limbs(FROB_G13);
  static const Fq2 g22 = fq2_from_limbs(FROB_G22), g23 = fq2_from_limbs(FROB_G23);
  G2Affine Q1{Qp.x.conj() * g12, Qp.y.conj() * g13};
  G2Affine Q2{Qp.x * g22, (Qp.y * g23).neg()};  // -pi^2(Q)
  f = f12_mul(f, line_and_add(&T, Q1, P));
  f = f12_mul(f, line_and_add(&T, Q2, P));
  return f;
}

// f^((q^12 - 1) / r) by plain square-and-multiply over the 2 790-bit exponent: the definition, kept as the cross-check
// of the structured form below (tests/host/hostmath.cpp) -- 4 200 Fq12 products, 27 ms.
inline Fq12 final_exponentiation_generic(const Fq12& f) {
  Fq12 r = Fq12::one();
  bool started = false;
  for (int i = FINAL_EXP_LIMBS * 32 - 1; i >= 0; i--) {
    if (started) r = f12_mul_schoolbook(r, r);
    if ((FINAL_EXP[i >> 5] >> (i & 31)) & 1) {
      r = f12_mul_schoolbook(r, f);
      started = true;
    }
  }
  return r;
}

// ---- structured final exponentiation: (q^6 - 1)(q^2 + 1) by conjugation / Frobenius / one inversion, then the hard
// part (q^4 - q^2 + 1) / r with three exponentiations by the BN parameter u = 4965661367192848881 (Scott et al.,
// "On the final exponentiation for calculating pairings on ordinary elliptic curves", the y0..y6 chain for u > 0).
// Fq12 = Fq2[w]/(w^6 - xi): Frobenius maps w^i to gamma_i w^i with gamma_i = xi^(i (q - 1) / 6), all derived from the
// two constants the Miller loop already uses (gamma_2 = FROB_G12, gamma_3 = FROB_G13).
struct FrobeniusCoeffs {
  Fq2 g1[6];  // q-power:   c_i -> conj(c_i) * g1[i]
  Fq2 g2[6];  // q^2-power: c_i -> c_i * g2[i]   (g2[i] = g1[i] * conj(g1[i]), in Fq)
  FrobeniusCoeffs() {
    Fq2 gam2 = fq2_from_limbs(FROB_G12), gam3 = fq2_from_limbs(FROB_G13);
    g1[0] = Fq2::one();
    g1[1] = gam3 * gam2.inv();
    g1[2] = gam2;
    g1[3] = gam3;
    g1[4] = gam2 * gam2;
    g1[5] = gam2 * gam3;
    for (int i = 0; i < 6; i++) g2[i] = g1[i] * g1[i].conj();
  }
};
inline const FrobeniusCoeffs& frob_coeffs() {
  static const FrobeniusCoeffs k;
  return k;
}
inline Fq12 f12_frob(const Fq12& a) {
  const FrobeniusCoeffs& k = frob_coeffs();
  Fq12 r;
  for (int i = 0; i < 6; i++) r.c[i] = a.c[i].conj() * k.g1[i];
  return r;
}
inline Fq12 f12_frob2(const Fq12& a) {
  const FrobeniusCoeffs& k = frob_coeffs();
  Fq12 r;
  for (int i = 0; i < 6; i++) r.c[i] = a.c[i] * k.g2[i];
  return r;
}
inline Fq12 f12_conj(const Fq12& a) {  // the q^6-power: w -> -w; the inverse inside the cyclotomic subgroup
  Fq12 r = a;
  for (int i = 1; i < 6; i += 2) r.c[i] = a.c[i].neg();
  return r;
}
// 1/a = (A - B w) / (A^2 - v B^2)
inline Fq6 f6_inv(const Fq6& x) {
  Fq2 t0 = x.a.sqr() - mul_xi(x.b * x.c);
  Fq2 t1 = mul_xi(x.c.sqr()) - x.a * x.b;
  Fq2 t2 = x.b.sqr() - x.a * x.c;
  Fq2 d = (x.a * t0 + mul_xi(x.c * t1 + x.b * t2)).inv();
  return {t0 * d, t1 * d, t2 * d};
}
inline Fq12 f12_inv(const Fq12& f) {
  Fq6 A{f.c[0], f.c[2], f.c[4]}, B{f.c[1], f.c[3], f.c[5]};
  Fq6 d = f6_inv(f6_sub(f6_mul(A, A), f6_mul_v(f6_mul(B, B))));
  Fq6 ra = f6_mul(A, d), rb = f6_mul(B, d);
  Fq12 r;
  r.c[0] = ra.a; r.c[2] = ra.b; r.c[4] = ra.c;
  r.c[1] = rb.a.neg(); r.c[3] = rb.b.neg(); r.c[5] = rb.c.neg();
  return r;
}
// Squaring inside the cyclotomic subgroup (Granger, Scott, "Faster squaring in the cyclotomic subgroup of sixth degree
// extensions"): three Fq4 squarings = 9 Fq2 squarings instead of the 12 Fq2 products of f12_sqr.  Only valid after
// the easy part of the final exponentiation; the host test compares the whole chain with the plain exponentiation.
inline Fq12 f12_cyclotomic_sqr(const Fq12& x) {
  auto fp4_sqr = [](const Fq2& a, const Fq2& b, Fq2* c0, Fq2* c1) {
    const Fq2 t0 = a.sqr(), t1 = b.sqr();
    *c0 = mul_xi(t1) + t0;
    *c1 = (a + b).sqr() - t0 - t1;
  };
  // even part (z0, z4, z3) = c[0], c[2], c[4]; odd part (z2, z1, z5) = c[1], c[3], c[5]
  const Fq2 &z0 = x.c[0], &z4 = x.c[2], &z3 = x.c[4], &z2 = x.c[1], &z1 = x.c[3], &z5 = x.c[5];
  Fq2 t0, t1, t2, t3;
  Fq12 r;
  auto m3sub2 = [](const Fq2& t, const Fq2& z) { return (t - z).dbl() + t; };   // 3 t - 2 z
  auto m3add2 = [](const Fq2& t, const Fq2& z) { return (t + z).dbl() + t; };   // 3 t + 2 z
  fp4_sqr(z0, z1, &t0, &t1);
  r.c[0] = m3sub2(t0, z0);
  r.c[3] = m3add2(t1, z1);
  fp4_sqr(z2, z3, &t0, &t1);
  fp4_sqr(z4, z5, &t2, &t3);
  r.c[2] = m3sub2(t0, z4);
  r.c[5] = m3add2(t1, z5);
  r.c[1] = m3add2(mul_xi(t3), z2);
  r.c[4] = m3sub2(t2, z3);
  return r;
}
inline Fq12 f12_pow_u(const Fq12& f) {   // f in the cyclotomic subgroup
  const uint64_t u = 4965661367192848881ULL;
  Fq12 r = f;
  for (int i = 61; i >= 0; i--) {  // u has 63 bits; the top bit is the initial value
    r = f12_cyclotomic_sqr(r);
    if ((u >> i) & 1) r = f12_mul(r, f);
  }
  return r;
}
inline Fq12 final_exponentiation(const Fq12& f0) {
  // easy part
  Fq12 f = f12_mul(f12_conj(f0), f12_inv(f0));  // f0^(q^6 - 1)
  f = f12_mul(f12_frob2(f), f);                  // ^(q^2 + 1): now in the cyclotomic subgroup
  // hard part
  Fq12 fx = f12_pow_u(f), fx2 = f12_pow_u(fx), fx3 = f12_pow_u(fx2);
  Fq12 fp = f12_frob(f), fp2 = f12_frob2(f), fp3 = f12_frob(fp2);
  Fq12 y0 = f12_mul(f12_mul(fp, fp2), fp3);
  Fq12 y1 = f12_conj(f);
  Fq12 y2 = f12_frob2(fx2);
  Fq12 y3 = f12_conj(f12_frob(fx));
  Fq12 y4 = f12_conj(f12_mul(fx, f12_frob(fx2)));
  Fq12 y5 = f12_conj(fx2);
  Fq12 y6 = f12_conj(f12_mul(fx3, f12_frob(fx3)));
  Fq12 t0 = f12_mul(f12_mul(f12_sqr(y6), y4), y5);
  Fq12 t1 = f12_mul(f12_mul(y3, y5), t0);
  t0 = f12_mul(t0, y2);
  t1 = f12_mul(f12_sqr(t1), t0);
  t1 = f12_sqr(t1);
  t0 = f12_mul(t1, y1);
  t1 = f12_mul(t1, y0);
  t0 = f12_sqr(t0);
  return f12_mul(t0, t1);
}

inline bool g1_on_curve(const G1Affine& p) {
  if (p.is_inf()) return true;
  return p.y.sqr() == p.x.sqr() * p.x + Fq::from_u32(3);
}
inline bool g2_on_curve(const G2Affine& p) {
  if (p.is_inf()) return true;
  static const Fq2 b2 = Fq2{Fq::from_u32(9), Fq::one()}.inv().mul_fq(Fq::from_u32(3));
  return p.y.sqr() == p.x.sqr() * p.x + b2;
}

// ---- prepared verifying key (ark-groth16 `prepare_verifying_key`): the Miller value of (-alpha, beta) and, for the
// fixed G2 points gamma and delta, the slope / intercept of every line of the Miller loop (they do not depend on the
// G1 argument), so a verification runs one variable Miller loop (A, B) with the two fixed ones riding on its squarings.
struct LineCoef {
  Fq2 lam, c;  // l(P) = P.y - lam P.x w + c w^3   with c = lam T.x - T.y
};
inline LineCoef line_coef_and_add(G2Affine* T, const G2Affine& Qp) {
  Fq2 lam;
  if (T->x == Qp.x && T->y == Qp.y) {
    Fq2 x2 = T->x.sqr();
    lam = (x2.dbl() + x2) * T->y.dbl().inv();
  } else {
    lam = (Qp.y - T->y) * (Qp.x - T->x).inv();
  }
  Fq2 x3 = lam.sqr() - T->x - Qp.x;
  Fq2 y3 = lam * (T->x - x3) - T->y;
  LineCoef l{lam, lam * T->x - T->y};
  *T = {x3, y3};
  return l;
}
inline Fq12 line_eval(const LineCoef& l, const G1Affine& P) {
  Fq12 r;
  for (auto& x : r.c) x = Fq2::zero();
  r.c[0] = {P.y, Fq::zero()};
  r.c[1] = l.lam.mul_fq(P.x).neg();
  r.c[3] = l.c;
  return r;
}
// the lines of miller_loop(., Q) in the order the loop consumes them
inline std::vector<LineCoef> prepare_g2(const G2Affine& Qp) {
  std::vector<LineCoef> out;
  G2Affine T = Qp;
  for (int i = ATE_LOOP_BITS - 2; i >= 0; i--) {
    out.push_back(line_coef_and_add(&T, T));
    if ((ATE_LOOP[i >> 5] >> (i & 31)) & 1) out.push_back(line_coef_and_add(&T, Qp));
  }
  static const Fq2 g12 = fq2_from_limbs(FROB_G12), g13 = fq2_from_limbs(FROB_G13);
  static const Fq2 g22 = fq2_from_limbs(FROB_G22), g23 = fq2_from_limbs(FROB_G23);
  G2Affine Q1{Qp.x.conj() * g12, Qp.y.conj() * g13};
  G2Affine Q2{Qp.x * g22, (Qp.y * g23).neg()};
  out.push_back(line_coef_and_add(&T, Q1));
  out.push_back(line_coef_and_add(&T, Q2));
  return out;
}
struct PreparedVk {
  Fq12 alpha_beta;  // miller_loop(-alpha, beta)
  std::vector<LineCoef> gamma, delta;
  // d * IC_i for d = 1..15, affine, i = 1..: the public-input combination sum x_i IC_i runs as ONE ladder of 63 x 4
  // doublings with a mixed addition per non-zero 4-bit digit (Straus) instead of one double-and-add per input
  std::vector<std::vector<G1Affine>> ic_mult;
};
inline G1XYZZ ic_combination(const Zkey& zk, const PreparedVk& pv, const std::vector<Fr>& x) {
  std::vector<uint32_t> k(x.size() * 8);
  for (size_t i = 0; i < x.size(); i++) x[i].to_canonical(&k[8 * i]);
  G1XYZZ acc = G1XYZZ::inf();
  for (int w = 63; w >= 0; w--) {
    if (w != 63) acc = acc.dbl().dbl().dbl().dbl();
    for (size_t i = 0; i < x.size(); i++) {
      const uint32_t d = (k[8 * i + (w >> 3)] >> ((w & 7) * 4)) & 15;
      if (d) acc.madd(pv.ic_mult[i][d - 1]);
    }
  }
  acc.madd(zk.gamma_abc_g1[0]);
  return acc;
}
inline const PreparedVk& prepared(const Zkey& zk) {
  if (!zk.prepared_vk) {
    auto pv = std::make_shared<PreparedVk>();
    pv->alpha_beta = miller_loop(zk.alpha_g1.neg(), zk.beta_g2);
    pv->gamma = prepare_g2(zk.gamma_g2);
    pv->delta = prepare_g2(zk.delta_g2);
    for (size_t i = 1; i < zk.gamma_abc_g1.size(); i++) {
      std::vector<G1Affine> t;
      G1XYZZ cur = G1XYZZ::from_affine(zk.gamma_abc_g1[i]);
      for (int d = 1; d <= 15; d++) {
        t.push_back(cur.to_affine());
        cur.madd(zk.gamma_abc_g1[i]);
      }
      pv->ic_mult.push_back(std::move(t));
    }
    zk.prepared_vk = pv;
  }
  return *static_cast<const PreparedVk*>(zk.prepared_vk.get());
}
// miller(A, B) * miller(Pg, gamma) * miller(Pd, delta): one pass, shared squarings
inline Fq12 miller_loop_3(const G1Affine& A, const G2Affine& B, const G1Affine& Pg, const std::vector<LineCoef>& lg,
                          const G1Affine& Pd, const std::vector<LineCoef>& ld) {
  Fq12 f = Fq12::one();
  const bool varying = !(A.is_inf() || B.is_inf());
  const bool useg = !Pg.is_inf(), used = !Pd.is_inf();
  G2Affine T = B;
  size_t k = 0;
  auto step = [&](const G2Affine& Qadd) {
    if (varying) {
      const LineCoef l = line_coef_and_add(&T, Qadd);
      f = f12_mul_line(f, A.y, l.lam.mul_fq(A.x).neg(), l.c);
    }
    if (useg) f = f12_mul_line(f, Pg.y, lg[k].lam.mul_fq(Pg.x).neg(), lg[k].c);
    if (used) f = f12_mul_line(f, Pd.y, ld[k].lam.mul_fq(Pd.x).neg(), ld[k].c);
    k++;
  };
  for (int i = ATE_LOOP_BITS - 2; i >= 0; i--) {
    f = f12_sqr(f);
    step(T);
    if ((ATE_LOOP[i >> 5] >> (i & 31)) & 1) step(B);
  }
  static const Fq2 g12 = fq2_from_limbs(FROB_G12), g13 = fq2_from_limbs(FROB_G13);
  static const Fq2 g22 = fq2_from_limbs(FROB_G22), g23 = fq2_from_limbs(FROB_G23);
  G2Affine Q1{B.x.conj() * g12, B.y.conj() * g13};
  G2Affine Q2{B.x * g22, (B.y * g23).neg()};
  step(Q1);
  step(Q2);
  return f;
}

// e(A,B) == e(alpha,beta) e(IC,gamma) e(C,delta), IC = ic[0] + sum x_i ic[i+1]; inputs are canonical limbs
inline bool groth16_verify(const Zkey& zk, const G1Affine& A, const G2Affine& B, const G1Affine& C,
                           const std::vector<Fr>& public_inputs) {
  if (public_inputs.size() + 1 != zk.gamma_abc_g1.size()) throw Error("MalformedVerifyingKey");
  if (!g1_on_curve(A) || !g2_on_curve(B) || !g1_on_curve(C)) return false;
  const PreparedVk& pv = prepared(zk);
  G1Affine icA = ic_combination(zk, pv, public_inputs).to_affine();
  Fq12 f = miller_loop_3(A, B, icA.neg(), pv.gamma, C.neg(), pv.delta);
  f = f12_mul(f, pv.alpha_beta);
  return final_exponentiation(f).is_one();
}

}  // namespace rlnamd
