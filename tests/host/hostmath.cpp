// CPU build of the product's __host__ __device__ math (zerokit_amd/csrc/{field,curve}.h) behind a tiny
// C interface so tests/test_host_math.py can compare it with the Python oracle without a GPU.
// Test infrastructure only: nothing in the product links this.
#include <string.h>

#include <stdexcept>
#include <string>
#include "curve.h"
namespace rlnamd {
struct Error : std::runtime_error {  // common.h's Error without the HIP headers
  explicit Error(const std::string& m) : std::runtime_error(m) {}
};
}  // namespace rlnamd
#include "pairing.h"
#include "glv.h"
using namespace rlnamd;

template <class F> static F ld(const uint8_t* p) { uint32_t c[8]; memcpy(c, p, 32); return F::from_canonical(c); }
template <class F> static void st(uint8_t* p, const F& x) { uint32_t c[8]; x.to_canonical(c); memcpy(p, c, 32); }
static Fq2 ld2(const uint8_t* p) { return {ld<Fq>(p), ld<Fq>(p + 32)}; }
static void st2(uint8_t* p, const Fq2& x) { st(p, x.c0); st(p + 32, x.c1); }

extern "C" {
// structured final exponentiation against the definition (square-and-multiply over (q^12 - 1) / r) on an arbitrary
// Fq12 element, plus the Fq12 inverse; returns a bit mask of the checks that hold (7 = all)
int hm_final_exp_check(uint32_t seed) {
  Fq12 f;
  for (int k = 0; k < 6; k++) f.c[k] = {Fq::from_u32(seed + 3 * k + 1), Fq::from_u32(seed * 7 + k + 2)};
  int ok = 0;
  Fq12 fi = f12_inv(f);
  if (f12_mul(f, fi).is_one()) ok |= 1;
  Fq12 slow = final_exponentiation_generic(f), fast = final_exponentiation(f);
  bool eq = true;
  for (int k = 0; k < 6; k++) eq &= slow.c[k] == fast.c[k];
  if (eq) ok |= 2;
  // the result lies in the order-r subgroup: frob(x) has the same order and x^(q^6) = x^-1
  if (f12_mul(fast, f12_conj(fast)).is_one()) ok |= 4;
  // tower arithmetic (Karatsuba product, complex squaring, sparse line product) against the schoolbook product
  Fq12 g;
  for (int k = 0; k < 6; k++) g.c[k] = {Fq::from_u32(seed * 11 + 5 * k + 3), Fq::from_u32(seed + 17 * k + 9)};
  auto same = [](const Fq12& a, const Fq12& b) {
    bool e = true;
    for (int k = 0; k < 6; k++) e &= a.c[k] == b.c[k];
    return e;
  };
  if (same(f12_mul(f, g), f12_mul_schoolbook(f, g)) && same(f12_sqr(f), f12_mul_schoolbook(f, f))) ok |= 8;
  LineCoef lc{g.c[2], g.c[4]};
  G1Affine P{Fq::from_u32(seed + 101), Fq::from_u32(seed * 3 + 7)};
  if (same(f12_mul_line(f, P.y, lc.lam.mul_fq(P.x).neg(), lc.c), f12_mul_schoolbook(f, line_eval(lc, P)))) ok |= 16;
  return ok;
}

// op: 0 add 1 sub 2 mul 3 inv 4 neg 5 sqr 6 inv by the bit-by-bit binary Euclid ; field: 0 Fr 1 Fq
void hm_fp_op(int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  if (field == 0) {
    Fr x = ld<Fr>(a), y = ld<Fr>(b), r;
    r = op == 0 ? x + y : op == 1 ? x - y : op == 2 ? x * y : op == 3 ? x.inv() : op == 4 ? x.neg() : op == 6 ? x.inv_binary() : x.sqr();
    st(out, r);
  } else {
    Fq x = ld<Fq>(a), y = ld<Fq>(b), r;
    r = op == 0 ? x + y : op == 1 ? x - y : op == 2 ? x * y : op == 3 ? x.inv() : op == 4 ? x.neg() : op == 6 ? x.inv_binary() : x.sqr();
    st(out, r);
  }
}
void hm_fq2_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Fq2 x = ld2(a), y = ld2(b), r;
  r = op == 0 ? x + y : op == 1 ? x - y : op == 2 ? x * y : op == 3 ? x.inv() : op == 4 ? x.neg() : x.sqr();
  st2(out, r);
}
// G1: points as x||y canonical (0,0 = infinity)
static G1Affine ldg1(const uint8_t* p) { return {ld<Fq>(p), ld<Fq>(p + 32)}; }
static void stg1(uint8_t* p, const G1Affine& a) { st(p, a.x); st(p + 32, a.y); }
static G2Affine ldg2(const uint8_t* p) { return {ld2(p), ld2(p + 64)}; }
static void stg2(uint8_t* p, const G2Affine& a) { st2(p, a.x); st2(p + 64, a.y); }

void hm_g1_add(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  G1XYZZ acc = G1XYZZ::from_affine(ldg1(a));
  acc.madd(ldg1(b));
  stg1(out, acc.to_affine());
}
void hm_g1_add_full(const uint8_t* a, const uint8_t* b, uint8_t* out) {  // exercises add-2008-s via 3P forms
  G1XYZZ x = G1XYZZ::dbl_affine(ldg1(a));  // 2a (non-trivial ZZ)
  G1XYZZ y = G1XYZZ::dbl_affine(ldg1(b));
  y = y.dbl();                               // 4b
  x.add(y);
  stg1(out, x.to_affine());                  // 2a + 4b
}
void hm_g1_mul(const uint8_t* a, const uint8_t* k, uint8_t* out) {
  uint32_t kk[8];
  memcpy(kk, k, 32);
  stg1(out, scalar_mul(ldg1(a), kk).to_affine());
}
void hm_g2_add(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  G2XYZZ acc = G2XYZZ::from_affine(ldg2(a));
  acc.madd(ldg2(b));
  stg2(out, acc.to_affine());
}
void hm_g2_mul(const uint8_t* a, const uint8_t* k, uint8_t* out) {
  uint32_t kk[8];
  memcpy(kk, k, 32);
  stg2(out, scalar_mul(ldg2(a), kk).to_affine());
}
// GLV split (glv.h): out = k1 (16 B LE) | neg1 (1 B) | k2 (16 B LE) | neg2 (1 B)
void hm_glv_split(const uint8_t* k, uint8_t* out) {
  uint32_t kk[8], k1[4], k2[4], n1, n2;
  memcpy(kk, k, 32);
  glv_split(kk, k1, &n1, k2, &n2);
  memcpy(out, k1, 16);
  out[16] = (uint8_t)n1;
  memcpy(out + 17, k2, 16);
  out[33] = (uint8_t)n2;
}
// the endomorphism itself on affine points: (beta x, y) with the constants the kernels use
void hm_glv_phi_g1(const uint8_t* a, uint8_t* out) {
  G1Affine p = ldg1(a);
  if (!p.is_inf()) p.x = p.x * Fq::from_canonical(GlvParams::BETA_G1);
  stg1(out, p);
}
void hm_glv_phi_g2(const uint8_t* a, uint8_t* out) {
  G2Affine p = ldg2(a);
  if (!p.is_inf()) p.x = p.x.mul_fq(Fq::from_canonical(GlvParams::BETA_G2));
  stg2(out, p);
}
}
