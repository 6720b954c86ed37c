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

inline Fq12 f12_mul(const Fq12& a, const Fq12& b) {
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
    f = f12_mul(f12_mul(f, f), l);
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

inline Fq12 final_exponentiation(const Fq12& f) {
  Fq12 r = Fq12::one();
  bool started = false;
  for (int i = FINAL_EXP_LIMBS * 32 - 1; i >= 0; i--) {
    if (started) r = f12_mul(r, r);
    if ((FINAL_EXP[i >> 5] >> (i & 31)) & 1) {
      r = f12_mul(r, f);
      started = true;
    }
  }
  return r;
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

// e(A,B) == e(alpha,beta) e(IC,gamma) e(C,delta), IC = ic[0] + sum x_i ic[i+1]; inputs are canonical limbs
inline bool groth16_verify(const Zkey& zk, const G1Affine& A, const G2Affine& B, const G1Affine& C,
                           const std::vector<Fr>& public_inputs) {
  if (public_inputs.size() + 1 != zk.gamma_abc_g1.size()) throw Error("MalformedVerifyingKey");
  if (!g1_on_curve(A) || !g2_on_curve(B) || !g1_on_curve(C)) return false;
  G1XYZZ ic = G1XYZZ::from_affine(zk.gamma_abc_g1[0]);
  for (size_t i = 0; i < public_inputs.size(); i++) {
    uint32_t k[8];
    public_inputs[i].to_canonical(k);
    ic.add(scalar_mul(zk.gamma_abc_g1[i + 1], k));
  }
  G1Affine icA = ic.to_affine();
  Fq12 f = miller_loop(A, B);
  f = f12_mul(f, miller_loop(zk.alpha_g1.neg(), zk.beta_g2));
  f = f12_mul(f, miller_loop(icA.neg(), zk.gamma_g2));
  f = f12_mul(f, miller_loop(C.neg(), zk.delta_g2));
  return final_exponentiation(f).is_one();
}

}  // namespace rlnamd
