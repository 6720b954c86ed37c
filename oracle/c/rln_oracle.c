/* ORACLE -- test infrastructure, NOT product code.
 *
 * CPU restatement, in plain C, of the reference's RLN proving path: the shipped single-message circuits (depth 20, depth
 * 10) and the multi-message-id circuit (depth 20, max_out 4) -- the prover is generic over (arkzkey, graph), the proof
 * values follow witness.rs:759-802 for both variants.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (zerokit_amd/) never links or calls it.
 *
 * What it restates (reference file:line under /root/reference):
 *   field arithmetic        ark-ff 0.5.0 Fp256<MontBackend<.., 4>> (third party, Cargo.lock): 4 x 64-bit
 *                           Montgomery limbs, the representation arkworks itself uses
 *   G1/G2 arithmetic        ark-ec 0.5.0 short_weierstrass Jacobian formulas (third party)
 *   MSM                     ark-ec 0.5.0 VariableBaseMSM::msm_bigint = windowed Pippenger; call sites
 *                           rln/src/partial_proof.rs:98-104,255-256
 *   NTT                     ark-poly 0.5.0 Radix2EvaluationDomain fft/ifft; call sites rln/src/circuit/qap.rs:69-90
 *   QAP witness map         rln/src/circuit/qap.rs:30-98
 *   Groth16 assembly        rln/src/partial_proof.rs:182-274 (== ark-groth16 create_proof_with_reduction_and_matrices,
 *                           rln/tests/partial_proof.rs:110-180)
 *   witness graph           rln/src/circuit/iden3calc/{storage.rs:265-302, proto.rs:7-117, graph.rs:72-143,246-272,314-466}
 *   arkzkey parser          rln/src/circuit/mod.rs:256-305
 *   Poseidon                utils/src/poseidon/poseidon_hash.rs:97-135, poseidon_constants.rs:15-261, rln/src/hashers.rs:14-23
 *   proof values            rln/src/protocol/witness.rs:759-828
 *   compressed proof        ark-serialize 0.5.0 (flags 0x80 / 0x40 in the top byte)
 *
 * Pinning: tests/test_oracle_c.py checks this file against the Python oracle (oracle/pyref, itself pinned
 * to the reference's Poseidon / tree / snarkjs-proof KATs by tests/test_oracle_kats.py) and against the
 * committed golden vectors.  Proof BYTES have no golden in the reference ("parity unpinned" for bytes);
 * they are pinned by uniqueness: A, B are closed forms and C is the unique solution of the pairing check.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------ fields */
typedef struct { u64 v[4]; } fe;                       /* Montgomery residue */
typedef struct { u64 p[4], r1[4], r2[4], inv; } field; /* modulus, R mod p, R^2 mod p, -p^-1 mod 2^64 */

static const field FR = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
                         {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
                         {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
                         0xc2e1f593efffffffULL};
static const field FQ = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
                         {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
                         {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
                         0x87d20782e4866389ULL};

static inline int ge4(const u64 a[4], const u64 b[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; }
  return 1;
}
static inline void sub4(u64 r[4], const u64 a[4], const u64 b[4]) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - b[i] - (u64)br; r[i] = (u64)t; br = (t >> 64) & 1; }
}
static inline int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) { return memcmp(a, b, sizeof(fe)) == 0; }
static inline void fe_add(const field* F, fe* r, const fe* a, const fe* b) {   /* a, b < p < 2^254: no carry out of limb 3 */
  u128 c = 0; u64 t[4], d[4];
  for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; t[i] = (u64)c; c >>= 64; }
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 x = (u128)t[i] - F->p[i] - (u64)br; d[i] = (u64)x; br = (x >> 64) & 1; }
  const u64 keep = (u64)0 - (u64)br;   /* t < p: keep t */
  for (int i = 0; i < 4; i++) r->v[i] = (t[i] & keep) | (d[i] & ~keep);
}
static inline void fe_sub(const field* F, fe* r, const fe* a, const fe* b) {
  u64 t[4]; u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 x = (u128)a->v[i] - b->v[i] - (u64)br; t[i] = (u64)x; br = (x >> 64) & 1; }
  const u64 m = (u64)0 - (u64)br;      /* borrowed: add p back */
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)t[i] + (F->p[i] & m); r->v[i] = (u64)c; c >>= 64; }
}
static inline void fe_neg(const field* F, fe* r, const fe* a) {
  if (fe_is_zero(a)) { *r = *a; return; }
  fe z = {{0, 0, 0, 0}}; fe_sub(F, r, &z, a);
}
static inline void fe_dbl(const field* F, fe* r, const fe* a) { fe_add(F, r, a, a); }
/* CIOS Montgomery product, portable form (the form every round until 6 timed; kept as the cross-check of the one below) */
static inline void fe_mul_portable(const field* F, fe* r, const fe* a, const fe* b) {
  u64 t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (u64)c; c >>= 64; }
    c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
    u64 m = t[0] * F->inv;
    c = ((u128)m * F->p[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * F->p[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
    c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
  }
  if (t[4] || ge4(t, F->p)) sub4(t, t, F->p);
  memcpy(r->v, t, 32);
}
#if defined(__x86_64__) && defined(__ADX__) && defined(__BMI2__)
#define ORACLE_MUL_KIND "mulx/adcx/adox"
/* What ark-ff 0.5.0 runs on x86-64 with the `asm` feature (ark-ff-asm 0.5.0, Cargo.lock; the reference's default build
 * enables it through ark-ff's features): the "no-carry" CIOS product for moduli with a spare top bit (both BN254 moduli
 * are 254 bits) as one mulx row per limb of b with two independent carry chains (adcx: the row's high halves, adox:
 * the accumulation), the reduction row interleaved.  Checked against fe_mul_portable by oracle_selftest_mul. */
static inline void fe_mul(const field* F, fe* r, const fe* a, const fe* b) {
  u64 t0, t1, t2, t3, A, lo, hi;
  const u64 *pa = a->v, *pb = b->v, *pp = F->p; u64 inv = F->inv;
#define RED_ROW \
    "movq %[inv], %%rdx\n\t imulq %[t0], %%rdx\n\t" \
    "xorq %[lo], %[lo]\n\t" \
    "mulxq 0(%[p]), %[lo], %[hi]\n\t adcxq %[t0], %[lo]\n\t movq %[hi], %[t0]\n\t" \
    "adcxq %[t1], %[t0]\n\t mulxq 8(%[p]), %[lo], %[t1]\n\t adoxq %[lo], %[t0]\n\t" \
    "adcxq %[t2], %[t1]\n\t mulxq 16(%[p]), %[lo], %[t2]\n\t adoxq %[lo], %[t1]\n\t" \
    "adcxq %[t3], %[t2]\n\t mulxq 24(%[p]), %[lo], %[t3]\n\t adoxq %[lo], %[t2]\n\t" \
    "movl $0, %k[lo]\n\t adcxq %[lo], %[t3]\n\t adoxq %[A], %[t3]\n\t"
#define MUL_ROW(OFF) \
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
    RED_ROW MUL_ROW(8) RED_ROW MUL_ROW(16) RED_ROW MUL_ROW(24) RED_ROW
    : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [A] "=&r"(A), [lo] "=&r"(lo), [hi] "=&r"(hi)
    : [a] "r"(pa), [b] "r"(pb), [p] "r"(pp), [inv] "r"(inv), "m"(*a), "m"(*b)
    : "rdx", "cc");
#undef RED_ROW
#undef MUL_ROW
  const u64 t[4] = {t0, t1, t2, t3};
  u64 d[4]; u128 br = 0;   /* result < 2p: one conditional subtraction, without a branch */
  for (int i = 0; i < 4; i++) { u128 x = (u128)t[i] - F->p[i] - (u64)br; d[i] = (u64)x; br = (x >> 64) & 1; }
  const u64 keep = (u64)0 - (u64)br;
  for (int i = 0; i < 4; i++) r->v[i] = (t[i] & keep) | (d[i] & ~keep);
}
#else
#define ORACLE_MUL_KIND "portable (unsigned __int128)"
static inline void fe_mul(const field* F, fe* r, const fe* a, const fe* b) { fe_mul_portable(F, r, a, b); }
#endif
static inline void fe_sqr(const field* F, fe* r, const fe* a) { fe_mul(F, r, a, a); }
static void fe_from_u64x4(const field* F, fe* r, const u64 c[4]) { fe x; memcpy(x.v, c, 32); fe r2; memcpy(r2.v, F->r2, 32); fe_mul(F, r, &x, &r2); }
static void fe_to_u64x4(const field* F, u64 c[4], const fe* a) { fe one = {{1, 0, 0, 0}}, t; fe_mul(F, &t, a, &one); memcpy(c, t.v, 32); }
static void fe_from_bytes(const field* F, fe* r, const uint8_t* le) { u64 c[4]; memcpy(c, le, 32); fe_from_u64x4(F, r, c); }
static void fe_to_bytes(const field* F, uint8_t* le, const fe* a) { u64 c[4]; fe_to_u64x4(F, c, a); memcpy(le, c, 32); }
static void fe_one(const field* F, fe* r) { memcpy(r->v, F->r1, 32); }
static void fe_set_u64(const field* F, fe* r, u64 x) { u64 c[4] = {x, 0, 0, 0}; fe_from_u64x4(F, r, c); }
static void fe_pow(const field* F, fe* r, const fe* a, const u64 e[4]) {
  fe acc; fe_one(F, &acc);
  for (int i = 255; i >= 0; i--) { fe_sqr(F, &acc, &acc); if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, a); }
  *r = acc;
}
static void fe_inv(const field* F, fe* r, const fe* a) {
  u64 e[4]; u64 two[4] = {2, 0, 0, 0}; sub4(e, F->p, two); fe_pow(F, r, a, e);
}

/* Fq2 = Fq[u]/(u^2+1) */
typedef struct { fe c0, c1; } fe2;
static inline void f2_add(fe2* r, const fe2* a, const fe2* b) { fe_add(&FQ, &r->c0, &a->c0, &b->c0); fe_add(&FQ, &r->c1, &a->c1, &b->c1); }
static inline void f2_sub(fe2* r, const fe2* a, const fe2* b) { fe_sub(&FQ, &r->c0, &a->c0, &b->c0); fe_sub(&FQ, &r->c1, &a->c1, &b->c1); }
static inline void f2_neg(fe2* r, const fe2* a) { fe_neg(&FQ, &r->c0, &a->c0); fe_neg(&FQ, &r->c1, &a->c1); }
static inline void f2_mul(fe2* r, const fe2* a, const fe2* b) {
  fe v0, v1, s, t;
  fe_mul(&FQ, &v0, &a->c0, &b->c0); fe_mul(&FQ, &v1, &a->c1, &b->c1);
  fe_add(&FQ, &s, &a->c0, &a->c1); fe_add(&FQ, &t, &b->c0, &b->c1); fe_mul(&FQ, &s, &s, &t);
  fe_sub(&FQ, &r->c0, &v0, &v1); fe_sub(&FQ, &s, &s, &v0); fe_sub(&FQ, &r->c1, &s, &v1);
}
/* complex squaring, two base-field products: (c0 + c1)(c0 - c1) + 2 c0 c1 u  (what ark-ff's QuadExtField::square_in_place
 * does when the non-residue is -1) */
static inline void f2_sqr(fe2* r, const fe2* a) {
  fe s, d, m;
  fe_add(&FQ, &s, &a->c0, &a->c1); fe_sub(&FQ, &d, &a->c0, &a->c1); fe_mul(&FQ, &m, &a->c0, &a->c1);
  fe_mul(&FQ, &r->c0, &s, &d); fe_add(&FQ, &r->c1, &m, &m);
}
static inline int f2_is_zero(const fe2* a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
static inline int f2_eq(const fe2* a, const fe2* b) { return fe_eq(&a->c0, &b->c0) && fe_eq(&a->c1, &b->c1); }
static void f2_inv(fe2* r, const fe2* a) {
  fe n, t; fe_sqr(&FQ, &n, &a->c0); fe_sqr(&FQ, &t, &a->c1); fe_add(&FQ, &n, &n, &t); fe_inv(&FQ, &n, &n);
  fe_mul(&FQ, &r->c0, &a->c0, &n); fe_mul(&FQ, &t, &a->c1, &n); fe_neg(&FQ, &r->c1, &t);
}

/* ------------------------------------------------------------------------------------------ curves
 * One generic Jacobian implementation over an "element" of 1 (G1) or 2 (G2) base-field limbs groups. */
#define DEF_CURVE(NAME, EL, ADD, SUB, MUL, SQR, NEG, ISZ, EQ, INV, ONE)                                   \
  typedef struct { EL x, y; int inf; } NAME##_aff;                                                          \
  typedef struct { EL X, Y, Z; } NAME##_jac; /* Z == 0 : infinity */                                       \
  static void NAME##_jac_inf(NAME##_jac* r) { memset(r, 0, sizeof(*r)); ONE(&r->X); ONE(&r->Y); }           \
  static int NAME##_jac_is_inf(const NAME##_jac* p) { return ISZ(&p->Z); }                                  \
  static void NAME##_dbl(NAME##_jac* r, const NAME##_jac* p) {                                              \
    if (ISZ(&p->Z)) { *r = *p; return; }                                                                    \
    EL A, B, C, D, E, F, t, X3, Y3, Z3;                                                                     \
    SQR(&A, &p->X); SQR(&B, &p->Y); SQR(&C, &B);                                                            \
    ADD(&t, &p->X, &B); SQR(&t, &t); SUB(&t, &t, &A); SUB(&t, &t, &C); ADD(&D, &t, &t);                     \
    ADD(&E, &A, &A); ADD(&E, &E, &A); SQR(&F, &E);                                                          \
    ADD(&t, &D, &D); SUB(&X3, &F, &t);                                                                      \
    ADD(&C, &C, &C); ADD(&C, &C, &C); ADD(&C, &C, &C);                                                      \
    SUB(&t, &D, &X3); MUL(&t, &E, &t); SUB(&Y3, &t, &C);                                                    \
    MUL(&Z3, &p->Y, &p->Z); ADD(&Z3, &Z3, &Z3);                                                             \
    r->X = X3; r->Y = Y3; r->Z = Z3;                                                                        \
  }                                                                                                         \
  static void NAME##_add_mixed(NAME##_jac* r, const NAME##_jac* p, const NAME##_aff* q) {                   \
    if (q->inf) { *r = *p; return; }                                                                        \
    if (ISZ(&p->Z)) { r->X = q->x; r->Y = q->y; ONE(&r->Z); return; }                                       \
    EL Z1Z1, U2, S2, H, HH, HHH, rr, V, t, X3, Y3, Z3;                                                      \
    SQR(&Z1Z1, &p->Z); MUL(&U2, &q->x, &Z1Z1); MUL(&S2, &q->y, &p->Z); MUL(&S2, &S2, &Z1Z1);              \
    if (EQ(&U2, &p->X)) { if (EQ(&S2, &p->Y)) { NAME##_dbl(r, p); } else { NAME##_jac_inf(r); } return; }   \
    SUB(&H, &U2, &p->X); SQR(&HH, &H); MUL(&HHH, &H, &HH); SUB(&rr, &S2, &p->Y); MUL(&V, &p->X, &HH);      \
    SQR(&X3, &rr); SUB(&X3, &X3, &HHH); ADD(&t, &V, &V); SUB(&X3, &X3, &t);                                 \
    SUB(&t, &V, &X3); MUL(&Y3, &rr, &t); MUL(&t, &p->Y, &HHH); SUB(&Y3, &Y3, &t);                           \
    MUL(&Z3, &p->Z, &H);                                                                                    \
    r->X = X3; r->Y = Y3; r->Z = Z3;                                                                        \
  }                                                                                                         \
  static void NAME##_add(NAME##_jac* r, const NAME##_jac* p, const NAME##_jac* q) {                         \
    if (ISZ(&p->Z)) { *r = *q; return; }                                                                    \
    if (ISZ(&q->Z)) { *r = *p; return; }                                                                    \
    EL Z1Z1, Z2Z2, U1, U2, S1, S2, H, HH, HHH, rr, V, t, X3, Y3, Z3;                                        \
    SQR(&Z1Z1, &p->Z); SQR(&Z2Z2, &q->Z); MUL(&U1, &p->X, &Z2Z2); MUL(&U2, &q->X, &Z1Z1);                 \
    MUL(&S1, &p->Y, &q->Z); MUL(&S1, &S1, &Z2Z2); MUL(&S2, &q->Y, &p->Z); MUL(&S2, &S2, &Z1Z1);            \
    if (EQ(&U1, &U2)) { if (EQ(&S1, &S2)) { NAME##_dbl(r, p); } else { NAME##_jac_inf(r); } return; }       \
    SUB(&H, &U2, &U1); SQR(&HH, &H); MUL(&HHH, &H, &HH); SUB(&rr, &S2, &S1); MUL(&V, &U1, &HH);            \
    SQR(&X3, &rr); SUB(&X3, &X3, &HHH); ADD(&t, &V, &V); SUB(&X3, &X3, &t);                                 \
    SUB(&t, &V, &X3); MUL(&Y3, &rr, &t); MUL(&t, &S1, &HHH); SUB(&Y3, &Y3, &t);                             \
    MUL(&Z3, &p->Z, &q->Z); MUL(&Z3, &Z3, &H);                                                              \
    r->X = X3; r->Y = Y3; r->Z = Z3;                                                                        \
  }                                                                                                         \
  static void NAME##_to_aff(NAME##_aff* r, const NAME##_jac* p) {                                           \
    if (ISZ(&p->Z)) { memset(r, 0, sizeof(*r)); r->inf = 1; return; }                                       \
    EL zi, zi2; INV(&zi, &p->Z); SQR(&zi2, &zi); MUL(&r->x, &p->X, &zi2); MUL(&zi2, &zi2, &zi);             \
    MUL(&r->y, &p->Y, &zi2); r->inf = 0;                                                                    \
  }                                                                                                         \
  static void NAME##_mul(NAME##_jac* r, const NAME##_aff* p, const u64 k[4]) {                              \
    NAME##_jac acc; NAME##_jac_inf(&acc);                                                                   \
    for (int i = 255; i >= 0; i--) { NAME##_dbl(&acc, &acc); if ((k[i >> 6] >> (i & 63)) & 1) NAME##_add_mixed(&acc, &acc, p); } \
    *r = acc;                                                                                               \
  }                                                                                                         \
  /* VariableBaseMSM::msm_bigint as ark-ec 0.5.0 runs it (msm_bigint_wnaf): window c = 3 below 32 points, else      \
   * ln_without_floor(n) + 2 = floor(log2 n) * 69 / 100 + 2; every scalar cut into ceil(254 / c) SIGNED digits in       \
   * [-2^(c-1), 2^(c-1)] (make_digits: a digit above half the radix borrows from the next), so a window has 2^(c-1)      \
   * buckets and a negative digit subtracts the base; running-sum bucket reduction; windows folded high to low with c   \
   * doublings each.  (Rounds 1-5 restated the unsigned form of ark-ec 0.4's msm_bigint: twice the buckets.) */        \
  static void NAME##_msm(NAME##_jac* out, const NAME##_aff* pts, const u64 (*sc)[4], size_t n) {                        \
    int c = 0; { size_t m = n; while (m > 1) { m >>= 1; c++; } c = n < 32 ? 3 : c * 69 / 100 + 2; }                     \
    const int nw = (254 + c - 1) / c; const size_t nb = (size_t)1 << (c - 1);                                           \
    int32_t* dig = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1) * (size_t)nw);                                        \
    const u64 radix = (u64)1 << c, mask = radix - 1;                                                                    \
    for (size_t i = 0; i < n; i++) {                                                                                    \
      u64 carry = 0;                                                                                                    \
      for (int w = 0; w < nw; w++) {                                                                                    \
        const int bit = w * c, wi = bit >> 6, bi = bit & 63;                                                            \
        u64 buf = (bi < 64 - c || wi == 3) ? sc[i][wi] >> bi : (sc[i][wi] >> bi) | (sc[i][wi + 1] << (64 - bi));        \
        const u64 coef = carry + (buf & mask);                                                                          \
        carry = (coef + radix / 2) >> c;                                                                                \
        dig[i * nw + w] = (int32_t)((int64_t)coef - (int64_t)(carry << c));                                             \
      }                                                                                                                 \
      dig[i * nw + nw - 1] += (int32_t)(carry << c);                                                                    \
    }                                                                                                                   \
    NAME##_jac* buckets = (NAME##_jac*)malloc(nb * sizeof(NAME##_jac));                                                 \
    NAME##_jac total; NAME##_jac_inf(&total);                                                                           \
    for (int w = nw - 1; w >= 0; w--) {                                                                                 \
      for (int k = 0; k < c; k++) NAME##_dbl(&total, &total);                                                           \
      for (size_t b = 0; b < nb; b++) NAME##_jac_inf(&buckets[b]);                                                      \
      for (size_t i = 0; i < n; i++) {                                                                                  \
        if (pts[i].inf) continue;                                                                                       \
        const int32_t d = dig[i * nw + w];                                                                              \
        if (d > 0) NAME##_add_mixed(&buckets[d - 1], &buckets[d - 1], &pts[i]);                                         \
        else if (d < 0) { NAME##_aff m = pts[i]; NEG(&m.y, &m.y); NAME##_add_mixed(&buckets[-d - 1], &buckets[-d - 1], &m); } \
      }                                                                                                                 \
      NAME##_jac run, ws; NAME##_jac_inf(&run); NAME##_jac_inf(&ws);                                                    \
      for (size_t b = nb; b-- > 0;) { NAME##_add(&run, &run, &buckets[b]); NAME##_add(&ws, &ws, &run); }                \
      NAME##_add(&total, &total, &ws);                                                                                  \
    }                                                                                                                   \
    free(buckets); free(dig); *out = total;                                                                             \
  }

#define FQ_ADD(r, a, b) fe_add(&FQ, r, a, b)
#define FQ_SUB(r, a, b) fe_sub(&FQ, r, a, b)
#define FQ_MUL(r, a, b) fe_mul(&FQ, r, a, b)
#define FQ_SQR(r, a) fe_sqr(&FQ, r, a)
#define FQ_NEG(r, a) fe_neg(&FQ, r, a)
#define FQ_INV(r, a) fe_inv(&FQ, r, a)
#define FQ_ONE(r) fe_one(&FQ, r)
static void f2_one(fe2* r) { fe_one(&FQ, &r->c0); memset(&r->c1, 0, sizeof(fe)); }
DEF_CURVE(g1, fe, FQ_ADD, FQ_SUB, FQ_MUL, FQ_SQR, FQ_NEG, fe_is_zero, fe_eq, FQ_INV, FQ_ONE)
DEF_CURVE(g2, fe2, f2_add, f2_sub, f2_mul, f2_sqr, f2_neg, f2_is_zero, f2_eq, f2_inv, f2_one)

/* ------------------------------------------------------------------------------------------ Poseidon */
typedef struct { int t, rf, rp; fe* ark; fe* mds; } pparams;
static pparams PP[5]; /* index t = 2..4 */
static int pp_ready = 0;
typedef struct { uint8_t st[80]; int head; } grain;
static int grain_update(grain* g) {
  int h = g->head;
  int b = g->st[(h + 62) % 80] ^ g->st[(h + 51) % 80] ^ g->st[(h + 38) % 80] ^ g->st[(h + 23) % 80] ^ g->st[(h + 13) % 80] ^ g->st[h];
  g->st[h] = (uint8_t)b; g->head = (h + 1) % 80; return b;
}
static void grain_put(grain* g, int lo, int hi, u64 v) { for (int i = hi; i >= lo; i--) { g->st[i] = v & 1; v >>= 1; } }
static void grain_value(grain* g, u64 out[4]) { /* 254 bits, first generated bit most significant */
  memset(out, 0, 32);
  for (int k = 253; k >= 0; k--) {
    int b = grain_update(g);
    while (!b) { grain_update(g); b = grain_update(g); }
    if (grain_update(g)) out[k >> 6] |= (u64)1 << (k & 63);
  }
}
static void poseidon_init(void) {
  if (pp_ready) return;
  static const int tab[3][3] = {{2, 8, 56}, {3, 8, 57}, {4, 8, 56}}; /* hashers.rs:14-23, skip_matrices = 0 */
  for (int q = 0; q < 3; q++) {
    int t = tab[q][0], rf = tab[q][1], rp = tab[q][2];
    grain g; memset(&g, 0, sizeof g);
    g.st[1] = 1; grain_put(&g, 6, 17, 254); grain_put(&g, 18, 29, t); grain_put(&g, 30, 39, rf); grain_put(&g, 40, 49, rp);
    for (int i = 50; i < 80; i++) g.st[i] = 1;
    for (int i = 0; i < 160; i++) grain_update(&g);
    pparams* P = &PP[t]; P->t = t; P->rf = rf; P->rp = rp;
    P->ark = (fe*)malloc(sizeof(fe) * (rf + rp) * t); P->mds = (fe*)malloc(sizeof(fe) * t * t);
    for (int i = 0; i < (rf + rp) * t; i++) { u64 v[4]; do { grain_value(&g, v); } while (ge4(v, FR.p)); fe_from_u64x4(&FR, &P->ark[i], v); }
    fe xs[4], ys[4];
    for (int i = 0; i < t; i++) { u64 v[4]; grain_value(&g, v); if (ge4(v, FR.p)) sub4(v, v, FR.p); fe_from_u64x4(&FR, &xs[i], v); }
    for (int i = 0; i < t; i++) { u64 v[4]; grain_value(&g, v); if (ge4(v, FR.p)) sub4(v, v, FR.p); fe_from_u64x4(&FR, &ys[i], v); }
    for (int i = 0; i < t; i++) for (int j = 0; j < t; j++) { fe s; fe_add(&FR, &s, &xs[i], &ys[j]); fe_inv(&FR, &P->mds[i * t + j], &s); }
  }
  pp_ready = 1;
}
static void poseidon(fe* out, const fe* in, int arity) { /* poseidon_hash.rs:97-135 */
  int t = arity + 1; const pparams* P = &PP[t];
  fe st[4], nx[4]; memset(&st[0], 0, sizeof(fe));
  for (int j = 1; j < t; j++) st[j] = in[j - 1];
  for (int r = 0; r < P->rf + P->rp; r++) {
    for (int j = 0; j < t; j++) fe_add(&FR, &st[j], &st[j], &P->ark[r * t + j]);
    int full = r < P->rf / 2 || r >= P->rf / 2 + P->rp;
    for (int j = 0; j < (full ? t : 1); j++) { fe x2, x4; fe_sqr(&FR, &x2, &st[j]); fe_sqr(&FR, &x4, &x2); fe_mul(&FR, &st[j], &x4, &st[j]); }
    for (int i = 0; i < t; i++) { fe acc; memset(&acc, 0, sizeof acc); for (int j = 0; j < t; j++) { fe m; fe_mul(&FR, &m, &P->mds[i * t + j], &st[j]); fe_add(&FR, &acc, &acc, &m); } nx[i] = acc; }
    for (int j = 0; j < t; j++) st[j] = nx[j];
  }
  *out = st[0];
}

/* ------------------------------------------------------------------------------------------ circuit */
typedef struct { uint32_t op, a, b, c; } gnode; /* op: 0 input, 1 const, 2+duo (proto.rs:88-110), 22 neg, 23 id, 24 tern */
typedef struct {
  /* zkey */
  g1_aff alpha1, beta1, delta1; g2_aff beta2, gamma2, delta2;
  g1_aff *ic, *aq, *b1q, *hq, *lq; g2_aff* b2q;
  size_t n_ic, n_aq, n_b1, n_b2, n_h, n_l;
  u64 n_inst, n_wit, n_cons;
  size_t* a_ptr; uint32_t* a_col; fe* a_val; size_t* b_ptr; uint32_t* b_col; fe* b_val;
  /* graph */
  gnode* nodes; size_t n_nodes; fe* consts; uint32_t* signals; size_t n_signals; size_t n_inputs;
  uint32_t off_secret, off_limit, off_msg, off_path, off_idx, off_x, off_ext, depth;
  uint32_t off_sel, n_msg, has_sel; /* multi-message-id circuit: selectorUsed[max_out], messageId[max_out] */
  /* ntt */
  int logn; size_t n; fe *tw, *twi, *coset; fe ninv;
} circuit;

static const uint8_t* rd_g1(const uint8_t* p, g1_aff* o) {
  uint8_t yb[32]; memcpy(yb, p + 32, 32); int fl = yb[31] & 0xC0; yb[31] &= 0x3F;
  o->inf = (fl & 0x40) != 0;
  if (o->inf) { memset(&o->x, 0, sizeof(fe)); memset(&o->y, 0, sizeof(fe)); } else { fe_from_bytes(&FQ, &o->x, p); fe_from_bytes(&FQ, &o->y, yb); }
  return p + 64;
}
static const uint8_t* rd_g2(const uint8_t* p, g2_aff* o) {
  uint8_t yb[32]; memcpy(yb, p + 96, 32); int fl = yb[31] & 0xC0; yb[31] &= 0x3F;
  o->inf = (fl & 0x40) != 0;
  if (o->inf) { memset(&o->x, 0, sizeof(fe2)); memset(&o->y, 0, sizeof(fe2)); }
  else { fe_from_bytes(&FQ, &o->x.c0, p); fe_from_bytes(&FQ, &o->x.c1, p + 32); fe_from_bytes(&FQ, &o->y.c0, p + 64); fe_from_bytes(&FQ, &o->y.c1, yb); }
  return p + 128;
}
static u64 rd_u64(const uint8_t** p) { u64 v; memcpy(&v, *p, 8); *p += 8; return v; }
static u64 rd_varint(const uint8_t** p) { u64 v = 0; int s = 0; for (;;) { uint8_t c = *(*p)++; v |= (u64)(c & 0x7F) << s; if (!(c & 0x80)) return v; s += 7; } }

static int parse_zkey(circuit* C, const uint8_t* d, size_t len) { /* circuit/mod.rs:256-305 */
  const uint8_t* p = d; (void)len;
  p = rd_g1(p, &C->alpha1); p = rd_g2(p, &C->beta2); p = rd_g2(p, &C->gamma2); p = rd_g2(p, &C->delta2);
#define VEC1(arr, cnt) do { cnt = rd_u64(&p); arr = (g1_aff*)malloc(sizeof(g1_aff) * (cnt ? cnt : 1)); for (size_t i = 0; i < cnt; i++) p = rd_g1(p, &arr[i]); } while (0)
  VEC1(C->ic, C->n_ic);
  p = rd_g1(p, &C->beta1); p = rd_g1(p, &C->delta1);
  VEC1(C->aq, C->n_aq); VEC1(C->b1q, C->n_b1);
  C->n_b2 = rd_u64(&p); C->b2q = (g2_aff*)malloc(sizeof(g2_aff) * C->n_b2); for (size_t i = 0; i < C->n_b2; i++) p = rd_g2(p, &C->b2q[i]);
  VEC1(C->hq, C->n_h); VEC1(C->lq, C->n_l);
  C->n_inst = rd_u64(&p); C->n_wit = rd_u64(&p); C->n_cons = rd_u64(&p);
  u64 annz = rd_u64(&p), bnnz = rd_u64(&p); (void)rd_u64(&p);
  for (int m = 0; m < 3; m++) {
    u64 rows = rd_u64(&p);
    size_t* ptr = (size_t*)malloc(sizeof(size_t) * (rows + 1)); u64 cap = m == 0 ? annz : m == 1 ? bnnz : 1;
    uint32_t* col = (uint32_t*)malloc(sizeof(uint32_t) * (cap ? cap : 1)); fe* val = (fe*)malloc(sizeof(fe) * (cap ? cap : 1));
    size_t k = 0; ptr[0] = 0;
    for (u64 r = 0; r < rows; r++) {
      u64 e = rd_u64(&p);
      for (u64 j = 0; j < e; j++) { if (k >= cap) return -1; fe_from_bytes(&FR, &val[k], p); p += 32; col[k] = (uint32_t)rd_u64(&p); k++; }
      ptr[r + 1] = k;
    }
    if (m == 0) { C->a_ptr = ptr; C->a_col = col; C->a_val = val; } else if (m == 1) { C->b_ptr = ptr; C->b_col = col; C->b_val = val; } else { free(ptr); free(col); free(val); }
  }
  return (size_t)(p - d) == len ? 0 : -2;
}

static void pb_skip(const uint8_t** p, int wt) { if (wt == 0) rd_varint(p); else if (wt == 2) { u64 l = rd_varint(p); *p += l; } else if (wt == 1) *p += 8; else *p += 4; }
static int parse_graph(circuit* C, const uint8_t* d, size_t len) { /* storage.rs:265-302 */
  if (len < 22 || memcmp(d, "wtns.graph.001", 14) != 0) return -1;
  const uint8_t* p = d + 14; u64 nn = rd_u64(&p);
  C->nodes = (gnode*)calloc(nn, sizeof(gnode)); C->consts = (fe*)malloc(sizeof(fe) * nn); C->n_nodes = nn;
  size_t nconst = 0;
  for (u64 i = 0; i < nn; i++) {
    u64 ml = rd_varint(&p); const uint8_t* e = p + ml;
    while (p < e) {
      u64 key = rd_varint(&p); int tag = (int)(key >> 3), wt = (int)(key & 7);
      if (wt != 2 || tag < 1 || tag > 5) { pb_skip(&p, wt); continue; }
      u64 bl = rd_varint(&p); const uint8_t* be = p + bl; uint32_t f[5] = {0, 0, 0, 0, 0};
      if (tag == 2) {
        uint8_t raw[32]; memset(raw, 0, 32);
        while (p < be) { u64 k2 = rd_varint(&p); if ((k2 >> 3) == 1 && (k2 & 7) == 2) { u64 l2 = rd_varint(&p); const uint8_t* e2 = p + l2;
            while (p < e2) { u64 k3 = rd_varint(&p); if ((k3 >> 3) == 1 && (k3 & 7) == 2) { u64 l3 = rd_varint(&p); memcpy(raw, p, l3 > 32 ? 32 : l3); p += l3; } else pb_skip(&p, (int)(k3 & 7)); } }
          else pb_skip(&p, (int)(k2 & 7)); }
        u64 v[4]; memcpy(v, raw, 32); while (ge4(v, FR.p)) sub4(v, v, FR.p);
        fe_from_u64x4(&FR, &C->consts[nconst], v); C->nodes[i].op = 1; C->nodes[i].a = (uint32_t)nconst++;
      } else {
        while (p < be) { u64 k2 = rd_varint(&p); int t2 = (int)(k2 >> 3); if ((k2 & 7) == 0 && t2 >= 1 && t2 <= 4) f[t2] = (uint32_t)rd_varint(&p); else pb_skip(&p, (int)(k2 & 7)); }
        if (tag == 1) { C->nodes[i].op = 0; C->nodes[i].a = f[1]; }
        else if (tag == 3) { C->nodes[i].op = f[1] == 0 ? 22 : 23; C->nodes[i].a = f[2]; }
        else if (tag == 4) { C->nodes[i].op = 2 + f[1]; C->nodes[i].a = f[2]; C->nodes[i].b = f[3]; }
        else { C->nodes[i].op = 24; C->nodes[i].a = f[2]; C->nodes[i].b = f[3]; C->nodes[i].c = f[4]; }
      }
      p = be;
    }
    p = e;
  }
  u64 ml = rd_varint(&p); const uint8_t* e = p + ml;
  C->signals = (uint32_t*)malloc(sizeof(uint32_t) * nn); C->n_signals = 0;
  while (p < e) {
    u64 key = rd_varint(&p); int tag = (int)(key >> 3), wt = (int)(key & 7);
    if (tag == 1 && wt == 2) { u64 l = rd_varint(&p); const uint8_t* pe = p + l; while (p < pe) C->signals[C->n_signals++] = (uint32_t)rd_varint(&p); }
    else if (tag == 1 && wt == 0) C->signals[C->n_signals++] = (uint32_t)rd_varint(&p);
    else if (tag == 2 && wt == 2) {
      u64 l = rd_varint(&p); const uint8_t* ee = p + l; char name[64] = {0}; uint32_t off = 0, ln = 0;
      while (p < ee) { u64 k2 = rd_varint(&p);
        if ((k2 >> 3) == 1 && (k2 & 7) == 2) { u64 sl = rd_varint(&p); memcpy(name, p, sl < 63 ? sl : 63); p += sl; }
        else if ((k2 >> 3) == 2 && (k2 & 7) == 2) { u64 sl = rd_varint(&p); const uint8_t* se = p + sl; while (p < se) { u64 k3 = rd_varint(&p); u64 v = rd_varint(&p); if ((k3 >> 3) == 1) off = (uint32_t)v; else if ((k3 >> 3) == 2) ln = (uint32_t)v; } }
        else pb_skip(&p, (int)(k2 & 7)); }
      if (!strcmp(name, "identitySecret")) C->off_secret = off; else if (!strcmp(name, "userMessageLimit")) C->off_limit = off;
      else if (!strcmp(name, "messageId")) { C->off_msg = off; C->n_msg = ln; } else if (!strcmp(name, "selectorUsed")) { C->off_sel = off; C->has_sel = 1; } else if (!strcmp(name, "pathElements")) { C->off_path = off; C->depth = ln; }
      else if (!strcmp(name, "identityPathIndex")) C->off_idx = off; else if (!strcmp(name, "x")) C->off_x = off;
      else if (!strcmp(name, "externalNullifier")) C->off_ext = off;
    } else pb_skip(&p, wt);
  }
  uint32_t mx = 0; int started = 0;
  for (size_t i = 0; i < C->n_nodes; i++) { if (C->nodes[i].op == 0) { if (C->nodes[i].a > mx) mx = C->nodes[i].a; started = 1; } else if (started) break; }
  C->n_inputs = mx + 1;
  return 0;
}

static void u256_shr(u64 r[4], const u64 a[4], unsigned n) {
  unsigned w = n >> 6, b = n & 63;
  for (int i = 0; i < 4; i++) { u64 lo = i + w < 4 ? a[i + w] : 0, hi = i + w + 1 < 4 ? a[i + w + 1] : 0; r[i] = b ? (lo >> b) | (hi << (64 - b)) : lo; }
}
static int gt4(const u64 a[4], const u64 b[4]) { for (int i = 3; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i]; return 0; }

/* graph.rs:246-272; ops the RLN circuits use (others -> error code) */
static int eval_graph(const circuit* C, const uint8_t* inputs_le, fe* vals, fe* out) {
  for (size_t n = 0; n < C->n_nodes; n++) {
    const gnode* nd = &C->nodes[n]; fe* v = &vals[n];
    switch (nd->op) {
      case 0: { u64 c[4]; memcpy(c, inputs_le + 32 * (size_t)nd->a, 32); if (ge4(c, FR.p)) return -1; fe_from_u64x4(&FR, v, c); break; }
      case 1: *v = C->consts[nd->a]; break;
      case 2: fe_mul(&FR, v, &vals[nd->a], &vals[nd->b]); break;
      case 4: fe_add(&FR, v, &vals[nd->a], &vals[nd->b]); break;
      case 5: fe_sub(&FR, v, &vals[nd->a], &vals[nd->b]); break;
      case 3: if (fe_is_zero(&vals[nd->b])) memset(v, 0, sizeof *v); else { fe i; fe_inv(&FR, &i, &vals[nd->b]); fe_mul(&FR, v, &vals[nd->a], &i); } break;
      case 9: if (fe_eq(&vals[nd->a], &vals[nd->b])) fe_one(&FR, v); else memset(v, 0, sizeof *v); break;
      case 10: if (!fe_eq(&vals[nd->a], &vals[nd->b])) fe_one(&FR, v); else memset(v, 0, sizeof *v); break;
      case 18: { /* Shr graph.rs:328-363 */
        u64 a[4], b[4], r[4]; fe_to_u64x4(&FR, a, &vals[nd->a]); fe_to_u64x4(&FR, b, &vals[nd->b]);
        if (!(b[0] | b[1] | b[2] | b[3])) { *v = vals[nd->a]; break; }
        if (b[1] | b[2] | b[3] || b[0] >= 254) { memset(v, 0, sizeof *v); break; }
        u256_shr(r, a, (unsigned)b[0]); fe_from_u64x4(&FR, v, r); break; }
      case 20: { /* Band graph.rs:365-380 */
        u64 a[4], b[4], r[4]; fe_to_u64x4(&FR, a, &vals[nd->a]); fe_to_u64x4(&FR, b, &vals[nd->b]);
        for (int i = 0; i < 4; i++) r[i] = a[i] & b[i];
        if (gt4(r, FR.p)) sub4(r, r, FR.p);
        if (ge4(r, FR.p)) return -3;
        fe_from_u64x4(&FR, v, r); break; }
      case 22: fe_neg(&FR, v, &vals[nd->a]); break;
      case 24: *v = fe_is_zero(&vals[nd->a]) ? vals[nd->c] : vals[nd->b]; break;
      default: return -2;
    }
  }
  for (size_t i = 0; i < C->n_signals; i++) out[i] = vals[C->signals[i]];
  return 0;
}

static void bitrev_perm(fe* a, size_t n) {
  for (size_t i = 1, j = 0; i < n; i++) { size_t bit = n >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; } }
}
static void ntt(const circuit* C, fe* a, int inverse) { /* out[i] = sum a[j] w^(ij) */
  size_t n = C->n; const fe* tw = inverse ? C->twi : C->tw;
  bitrev_perm(a, n);
  for (size_t len = 2; len <= n; len <<= 1) {
    size_t half = len >> 1, step = n / len;
    for (size_t s = 0; s < n; s += len)
      for (size_t k = 0; k < half; k++) { fe u = a[s + k], v; fe_mul(&FR, &v, &a[s + k + half], &tw[k * step]); fe_add(&FR, &a[s + k], &u, &v); fe_sub(&FR, &a[s + k + half], &u, &v); }
  }
  if (inverse) for (size_t i = 0; i < n; i++) fe_mul(&FR, &a[i], &a[i], &C->ninv);
}
static void ntt_init(circuit* C) {
  size_t need = C->n_cons + C->n_inst; C->n = 1; C->logn = 0; while (C->n < need) { C->n <<= 1; C->logn++; }
  u64 wc[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL}; /* 5^((r-1)/2^28) */
  fe g; fe_from_u64x4(&FR, &g, wc);
  for (int i = 0; i < 28 - (C->logn + 1); i++) fe_sqr(&FR, &g, &g);
  fe w, wi; fe_sqr(&FR, &w, &g); fe_inv(&FR, &wi, &w);
  C->tw = (fe*)malloc(sizeof(fe) * C->n / 2); C->twi = (fe*)malloc(sizeof(fe) * C->n / 2); C->coset = (fe*)malloc(sizeof(fe) * C->n);
  fe a, b; fe_one(&FR, &a); fe_one(&FR, &b);
  for (size_t k = 0; k < C->n / 2; k++) { C->tw[k] = a; C->twi[k] = b; fe_mul(&FR, &a, &a, &w); fe_mul(&FR, &b, &b, &wi); }
  fe_one(&FR, &a); for (size_t i = 0; i < C->n; i++) { C->coset[i] = a; fe_mul(&FR, &a, &a, &g); }
  fe nn; fe_set_u64(&FR, &nn, C->n); fe_inv(&FR, &C->ninv, &nn);
}

static void g1_compress(const g1_aff* p, uint8_t out[32]) {
  if (p->inf) { memset(out, 0, 32); out[31] = 0x40; return; }
  fe_to_bytes(&FQ, out, &p->x); u64 y[4], h[4] = {0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};
  fe_to_u64x4(&FQ, y, &p->y); if (gt4(y, h)) out[31] |= 0x80;
}
static void g2_compress(const g2_aff* p, uint8_t out[64]) {
  if (p->inf) { memset(out, 0, 64); out[63] = 0x40; return; }
  fe_to_bytes(&FQ, out, &p->x.c0); fe_to_bytes(&FQ, out + 32, &p->x.c1);
  u64 y[4], h[4] = {0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};
  fe_to_u64x4(&FQ, y, fe_is_zero(&p->y.c1) ? &p->y.c0 : &p->y.c1); if (gt4(y, h)) out[63] |= 0x80;
}

/* ------------------------------------------------------------------------------------------ public API */
void* oracle_load(const uint8_t* zkey, size_t zlen, const uint8_t* graph, size_t glen) {
  poseidon_init();
  circuit* C = (circuit*)calloc(1, sizeof(circuit));
  if (parse_zkey(C, zkey, zlen) != 0 || parse_graph(C, graph, glen) != 0) { free(C); return NULL; }
  ntt_init(C);
  return C;
}
size_t oracle_num_inputs(void* h) { return ((circuit*)h)->n_inputs; }
size_t oracle_num_signals(void* h) { return ((circuit*)h)->n_signals; }
size_t oracle_domain(void* h) { return ((circuit*)h)->n; }

void oracle_poseidon(const uint8_t* in_le, size_t n, int arity, uint8_t* out_le) {
  poseidon_init();
  for (size_t i = 0; i < n; i++) { fe x[3], h; for (int j = 0; j < arity; j++) fe_from_bytes(&FR, &x[j], in_le + 32 * (i * arity + j)); poseidon(&h, x, arity); fe_to_bytes(&FR, out_le + 32 * i, &h); }
}

/* proof_values_from_witness (witness.rs:759-828) -> y, root, nullifier, x, ext */
static void proof_values(const circuit* C, const uint8_t* in, uint8_t* out160) {
  fe secret, limit, msg, x, ext, idc, root, a1, y, nul, t[3];
  fe_from_bytes(&FR, &secret, in + 32 * C->off_secret); fe_from_bytes(&FR, &limit, in + 32 * C->off_limit);
  fe_from_bytes(&FR, &msg, in + 32 * C->off_msg); fe_from_bytes(&FR, &x, in + 32 * C->off_x); fe_from_bytes(&FR, &ext, in + 32 * C->off_ext);
  poseidon(&idc, &secret, 1); t[0] = idc; t[1] = limit; poseidon(&root, t, 2);
  for (uint32_t i = 0; i < C->depth; i++) {
    fe e; fe_from_bytes(&FR, &e, in + 32 * (C->off_path + i));
    const uint8_t* b = in + 32 * (C->off_idx + i); int nz = 0; for (int k = 0; k < 32; k++) nz |= b[k];
    if (!nz) { t[0] = root; t[1] = e; } else { t[0] = e; t[1] = root; }
    poseidon(&root, t, 2);
  }
  t[0] = secret; t[1] = ext; t[2] = msg; poseidon(&a1, t, 3);
  fe_mul(&FR, &y, &x, &a1); fe_add(&FR, &y, &y, &secret); poseidon(&nul, &a1, 1);
  fe_to_bytes(&FR, out160, &y); fe_to_bytes(&FR, out160 + 32, &root); fe_to_bytes(&FR, out160 + 64, &nul);
  fe_to_bytes(&FR, out160 + 96, &x); fe_to_bytes(&FR, out160 + 128, &ext);
}

/* Public inputs in the verifier's order for either variant: single (proof.rs:863-869) y, root, nullifier, x, ext = the 160
 * bytes of proof_values; multi (witness.rs:777-802, proof.rs:870-885) ys[max_out], root, nullifiers[max_out], x, ext,
 * selector_used[max_out], where an unused slot contributes y = 0 and nullifier = 0. */
size_t oracle_num_public(void* h) { return (size_t)((circuit*)h)->n_inst - 1; }
int oracle_input_slot(void* h, const char* name, uint32_t* off, uint32_t* len) {
  const circuit* C = (const circuit*)h;
  if (!strcmp(name, "identitySecret")) { *off = C->off_secret; *len = 1; } else if (!strcmp(name, "userMessageLimit")) { *off = C->off_limit; *len = 1; }
  else if (!strcmp(name, "messageId")) { *off = C->off_msg; *len = C->n_msg; } else if (!strcmp(name, "pathElements")) { *off = C->off_path; *len = C->depth; }
  else if (!strcmp(name, "identityPathIndex")) { *off = C->off_idx; *len = C->depth; } else if (!strcmp(name, "x")) { *off = C->off_x; *len = 1; }
  else if (!strcmp(name, "externalNullifier")) { *off = C->off_ext; *len = 1; }
  else if (!strcmp(name, "selectorUsed") && C->has_sel) { *off = C->off_sel; *len = C->n_msg; } else return -1;
  return 0;
}
static void public_values(const circuit* C, const uint8_t* in, uint8_t* out) {
  if (!C->has_sel) { proof_values(C, in, out); return; }
  fe secret, limit, x, ext, idc, root, t[3], zero; memset(&zero, 0, sizeof zero);
  fe_from_bytes(&FR, &secret, in + 32 * C->off_secret); fe_from_bytes(&FR, &limit, in + 32 * C->off_limit);
  fe_from_bytes(&FR, &x, in + 32 * C->off_x); fe_from_bytes(&FR, &ext, in + 32 * C->off_ext);
  poseidon(&idc, &secret, 1); t[0] = idc; t[1] = limit; poseidon(&root, t, 2);
  for (uint32_t i = 0; i < C->depth; i++) {
    fe e; fe_from_bytes(&FR, &e, in + 32 * (C->off_path + i));
    const uint8_t* b = in + 32 * (C->off_idx + i); int nz = 0; for (int k = 0; k < 32; k++) nz |= b[k];
    if (!nz) { t[0] = root; t[1] = e; } else { t[0] = e; t[1] = root; }
    poseidon(&root, t, 2);
  }
  const uint32_t mo = C->n_msg;
  for (uint32_t k = 0; k < mo; k++) {
    const uint8_t* sb = in + 32 * (C->off_sel + k); int used = 0; for (int q = 0; q < 32; q++) used |= sb[q];
    fe msg, a1, y, nul; fe_from_bytes(&FR, &msg, in + 32 * (C->off_msg + k));
    t[0] = secret; t[1] = ext; t[2] = msg; poseidon(&a1, t, 3);
    fe_mul(&FR, &y, &x, &a1); fe_add(&FR, &y, &y, &secret); poseidon(&nul, &a1, 1);
    fe_to_bytes(&FR, out + 32 * k, used ? &y : &zero); fe_to_bytes(&FR, out + 32 * (mo + 1 + k), used ? &nul : &zero);
    memset(out + 32 * (2 * mo + 3 + k), 0, 32); out[32 * (2 * mo + 3 + k)] = used ? 1 : 0;
  }
  fe_to_bytes(&FR, out + 32 * mo, &root); fe_to_bytes(&FR, out + 32 * (2 * mo + 1), &x); fe_to_bytes(&FR, out + 32 * (2 * mo + 2), &ext);
}
/* the public values alone (num_public x 32 B): what a verifier is handed beside the proof */
void oracle_public_values(void* h, const uint8_t* inputs_le, uint8_t* out) { public_values((const circuit*)h, inputs_le, out); }

/* One proof.  inputs: n_inputs x 32 B (slot 0 = 1); rs: r|s; outputs optional. Returns 0 on success. */
int oracle_prove(void* hnd, const uint8_t* inputs_le, const uint8_t* rs_le, uint8_t* proof128, uint8_t* coords256,
                 uint8_t* values160, uint8_t* witness_le, uint8_t* h_le) {
  const circuit* C = (const circuit*)hnd;
  size_t ns = C->n_signals, n = C->n, nc = C->n_cons, ni = C->n_inst;
  fe* vals = (fe*)malloc(sizeof(fe) * C->n_nodes); fe* w = (fe*)malloc(sizeof(fe) * ns);
  int rc = eval_graph(C, inputs_le, vals, w); free(vals);
  if (rc) { free(w); return rc; }
  if (witness_le) for (size_t i = 0; i < ns; i++) fe_to_bytes(&FR, witness_le + 32 * i, &w[i]);
  /* qap.rs:30-98 */
  fe* a = (fe*)calloc(n, sizeof(fe)); fe* b = (fe*)calloc(n, sizeof(fe)); fe* c = (fe*)calloc(n, sizeof(fe));
  for (size_t r = 0; r < nc; r++) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t k = C->a_ptr[r]; k < C->a_ptr[r + 1]; k++) { fe m; fe_mul(&FR, &m, &C->a_val[k], &w[C->a_col[k]]); fe_add(&FR, &acc, &acc, &m); }
    a[r] = acc; memset(&acc, 0, sizeof acc);
    for (size_t k = C->b_ptr[r]; k < C->b_ptr[r + 1]; k++) { fe m; fe_mul(&FR, &m, &C->b_val[k], &w[C->b_col[k]]); fe_add(&FR, &acc, &acc, &m); }
    b[r] = acc; fe_mul(&FR, &c[r], &a[r], &b[r]);
  }
  for (size_t i = 0; i < ni; i++) a[nc + i] = w[i];
  fe* v3[3] = {a, b, c};
  for (int q = 0; q < 3; q++) { ntt(C, v3[q], 1); for (size_t i = 0; i < n; i++) fe_mul(&FR, &v3[q][i], &v3[q][i], &C->coset[i]); ntt(C, v3[q], 0); }
  for (size_t i = 0; i < n; i++) { fe t; fe_mul(&FR, &t, &a[i], &b[i]); fe_sub(&FR, &a[i], &t, &c[i]); } /* h in a */
  if (h_le) for (size_t i = 0; i < n; i++) fe_to_bytes(&FR, h_le + 32 * i, &a[i]);
  /* partial_proof.rs:182-274 */
  u64(*ws)[4] = (u64(*)[4])malloc(sizeof(u64[4]) * ns); u64(*hs)[4] = (u64(*)[4])malloc(sizeof(u64[4]) * n);
  for (size_t i = 0; i < ns; i++) fe_to_u64x4(&FR, ws[i], &w[i]);
  for (size_t i = 0; i < n; i++) fe_to_u64x4(&FR, hs[i], &a[i]);
  u64 r[4], s[4], rsv[4]; memcpy(r, rs_le, 32); memcpy(s, rs_le + 32, 32);
  { fe fr, fs, frs; fe_from_u64x4(&FR, &fr, r); fe_from_u64x4(&FR, &fs, s); fe_mul(&FR, &frs, &fr, &fs); fe_to_u64x4(&FR, rsv, &frs); }
  g1_jac ga, gb1, gc, t1; g2_jac gb2, t2;
  g1_msm(&ga, C->aq + 1, (const u64(*)[4])(ws + 1), ns - 1);
  g1_add_mixed(&ga, &ga, &C->alpha1); g1_add_mixed(&ga, &ga, &C->aq[0]); g1_mul(&t1, &C->delta1, r); g1_add(&ga, &ga, &t1);
  int rz = !(r[0] | r[1] | r[2] | r[3]);
  if (!rz) { g1_msm(&gb1, C->b1q + 1, (const u64(*)[4])(ws + 1), ns - 1); g1_add_mixed(&gb1, &gb1, &C->beta1); g1_add_mixed(&gb1, &gb1, &C->b1q[0]); g1_mul(&t1, &C->delta1, s); g1_add(&gb1, &gb1, &t1); }
  else g1_jac_inf(&gb1);
  g2_msm(&gb2, C->b2q + 1, (const u64(*)[4])(ws + 1), ns - 1);
  g2_add_mixed(&gb2, &gb2, &C->beta2); g2_add_mixed(&gb2, &gb2, &C->b2q[0]); g2_mul(&t2, &C->delta2, s); g2_add(&gb2, &gb2, &t2);
  g1_jac lacc, hacc; g1_msm(&lacc, C->lq, (const u64(*)[4])(ws + ni), C->n_l); g1_msm(&hacc, C->hq, (const u64(*)[4])hs, n);
  g1_aff A, B1; g2_aff B2; g1_to_aff(&A, &ga); g1_to_aff(&B1, &gb1); g2_to_aff(&B2, &gb2);
  g1_mul(&gc, &A, s); g1_mul(&t1, &B1, r); g1_add(&gc, &gc, &t1);
  g1_mul(&t1, &C->delta1, rsv); fe_neg(&FQ, &t1.Y, &t1.Y); g1_add(&gc, &gc, &t1);
  g1_add(&gc, &gc, &lacc); g1_add(&gc, &gc, &hacc);
  g1_aff Cc; g1_to_aff(&Cc, &gc);
  if (proof128) { g1_compress(&A, proof128); g2_compress(&B2, proof128 + 32); g1_compress(&Cc, proof128 + 96); }
  if (coords256) { fe_to_bytes(&FQ, coords256, &A.x); fe_to_bytes(&FQ, coords256 + 32, &A.y); fe_to_bytes(&FQ, coords256 + 64, &B2.x.c0); fe_to_bytes(&FQ, coords256 + 96, &B2.x.c1);
    fe_to_bytes(&FQ, coords256 + 128, &B2.y.c0); fe_to_bytes(&FQ, coords256 + 160, &B2.y.c1); fe_to_bytes(&FQ, coords256 + 192, &Cc.x); fe_to_bytes(&FQ, coords256 + 224, &Cc.y); }
  if (values160) proof_values(C, inputs_le, values160); /* single-message circuits; oracle_public_values serves both */
  free(w); free(a); free(b); free(c); free(ws); free(hs);
  return 0;
}

/* ---- partial proofs: generate_partial_zk_proof / finish_zk_proof_with_rs (protocol/proof.rs:783-849) over
 * Groth16Partial (partial_proof.rs:108-274) and evaluate_partial (iden3calc/graph.rs:274-312).
 * The partial witness of witness.rs:887-937 fixes identitySecret, userMessageLimit, pathElements, identityPathIndex;
 * messageId, x, externalNullifier (and selectorUsed on the multi-message-id circuit) are unknown. */
static void input_known(const circuit* C, uint8_t* k /* n_inputs */) {
  memset(k, 1, C->n_inputs);
  for (uint32_t i = 0; i < C->n_msg; i++) k[C->off_msg + i] = 0;
  k[C->off_x] = 0; k[C->off_ext] = 0;
  if (C->has_sel) for (uint32_t i = 0; i < C->n_msg; i++) k[C->off_sel + i] = 0;
}
/* graph.rs:274-312: a node is Some(..) iff every operand is; returns the per-NODE flags, mask_out = per witness signal */
static void known_nodes(const circuit* C, uint8_t* kn /* n_nodes */, uint8_t* mask_out /* n_signals or NULL */) {
  uint8_t* ik = (uint8_t*)malloc(C->n_inputs); input_known(C, ik);
  for (size_t n = 0; n < C->n_nodes; n++) {
    const gnode* nd = &C->nodes[n];
    switch (nd->op) {
      case 0: kn[n] = ik[nd->a]; break;
      case 1: kn[n] = 1; break;
      case 22: case 23: kn[n] = kn[nd->a]; break;
      case 24: kn[n] = kn[nd->a] && kn[nd->b] && kn[nd->c]; break;
      default: kn[n] = kn[nd->a] && kn[nd->b]; break;
    }
  }
  if (mask_out) for (size_t i = 0; i < C->n_signals; i++) mask_out[i] = kn[C->signals[i]];
  free(ik);
}
/* PartialProof::mask with entry 0 (the constant 1, always known) in front: n_signals bytes */
void oracle_known_mask(void* h, uint8_t* mask_out) {
  const circuit* C = (const circuit*)h; uint8_t* kn = (uint8_t*)malloc(C->n_nodes); known_nodes(C, kn, mask_out); free(kn);
}
static void g1_put(uint8_t* o, const g1_jac* p) { g1_aff a; g1_to_aff(&a, p); memset(o, 0, 64); if (!a.inf) { fe_to_bytes(&FQ, o, &a.x); fe_to_bytes(&FQ, o + 32, &a.y); } }
static void g1_get(g1_aff* a, const uint8_t* i) { int nz = 0; for (int k = 0; k < 64; k++) nz |= i[k]; a->inf = !nz; fe_from_bytes(&FQ, &a->x, i); fe_from_bytes(&FQ, &a->y, i + 32); }
/* the four MSMs of either half: rows whose mask equals `want` (signal i >= 1; L rows: i >= n_inst) */
static void masked_msms(const circuit* C, const uint8_t* mask, int want, const u64 (*ws)[4], g1_jac* a, g1_jac* b1, g2_jac* b2, g1_jac* l, int with_b1) {
  const size_t ns = C->n_signals, ni = C->n_inst;
  size_t m = 0, ml = 0; for (size_t i = 1; i < ns; i++) if (!!mask[i] == want) { m++; if (i >= ni) ml++; }
  g1_aff* pa = (g1_aff*)malloc(sizeof(g1_aff) * (m + 1)); g1_aff* pb = (g1_aff*)malloc(sizeof(g1_aff) * (m + 1));
  g2_aff* p2 = (g2_aff*)malloc(sizeof(g2_aff) * (m + 1)); g1_aff* pl = (g1_aff*)malloc(sizeof(g1_aff) * (ml + 1));
  u64(*sc)[4] = (u64(*)[4])malloc(sizeof(u64[4]) * (m + 1)); u64(*sl)[4] = (u64(*)[4])malloc(sizeof(u64[4]) * (ml + 1));
  size_t k = 0, kl = 0;
  for (size_t i = 1; i < ns; i++) if (!!mask[i] == want) {
    pa[k] = C->aq[i]; pb[k] = C->b1q[i]; p2[k] = C->b2q[i]; memcpy(sc[k], ws[i], 32); k++;
    if (i >= ni) { pl[kl] = C->lq[i - ni]; memcpy(sl[kl], ws[i], 32); kl++; }
  }
  if (m) g1_msm(a, pa, (const u64(*)[4])sc, m); else g1_jac_inf(a);
  if (with_b1 && m) g1_msm(b1, pb, (const u64(*)[4])sc, m); else g1_jac_inf(b1);
  if (m) g2_msm(b2, p2, (const u64(*)[4])sc, m); else g2_jac_inf(b2);
  if (ml) g1_msm(l, pl, (const u64(*)[4])sl, ml); else g1_jac_inf(l);
  free(pa); free(pb); free(p2); free(pl); free(sc); free(sl);
}
/* generate_partial_zk_proof (proof.rs:783-803 -> partial_proof.rs:108-179).  inputs: the full inputs buffer with the
 * unknown slots at any value (they are not read: evaluate_partial leaves every node they reach None).
 * out320 = pi_a | rho | pi_b | pi_c as affine canonical LE coordinates (64 + 64 + 128 + 64; infinity = zeros). */
int oracle_prove_partial(void* hnd, const uint8_t* inputs_le, uint8_t* out320, uint8_t* mask_out) {
  const circuit* C = (const circuit*)hnd; const size_t ns = C->n_signals;
  uint8_t* kn = (uint8_t*)malloc(C->n_nodes); uint8_t* mask = (uint8_t*)malloc(ns); known_nodes(C, kn, mask);
  /* evaluate_partial: known nodes take their values; unknown ones are never read by a known node */
  uint8_t* in = (uint8_t*)malloc(32 * C->n_inputs); memcpy(in, inputs_le, 32 * C->n_inputs);
  { uint8_t* ik = (uint8_t*)malloc(C->n_inputs); input_known(C, ik); for (size_t i = 0; i < C->n_inputs; i++) if (!ik[i]) memset(in + 32 * i, 0, 32); free(ik); }
  fe* vals = (fe*)malloc(sizeof(fe) * C->n_nodes); fe* w = (fe*)malloc(sizeof(fe) * ns);
  int rc = eval_graph(C, in, vals, w); free(vals); free(in); free(kn);
  if (rc) { free(w); free(mask); return rc; }
  u64(*ws)[4] = (u64(*)[4])malloc(sizeof(u64[4]) * ns);
  for (size_t i = 0; i < ns; i++) fe_to_u64x4(&FR, ws[i], &w[i]);
  g1_jac a, b1, l; g2_jac b2;
  masked_msms(C, mask, 1, (const u64(*)[4])ws, &a, &b1, &b2, &l, 1);
  g1_add_mixed(&a, &a, &C->alpha1); g1_add_mixed(&a, &a, &C->aq[0]);
  g1_add_mixed(&b1, &b1, &C->beta1); g1_add_mixed(&b1, &b1, &C->b1q[0]);
  g2_add_mixed(&b2, &b2, &C->beta2); g2_add_mixed(&b2, &b2, &C->b2q[0]);
  g1_put(out320, &a); g1_put(out320 + 64, &b1);
  { g2_aff q; g2_to_aff(&q, &b2); memset(out320 + 128, 0, 128);
    if (!q.inf) { fe_to_bytes(&FQ, out320 + 128, &q.x.c0); fe_to_bytes(&FQ, out320 + 160, &q.x.c1); fe_to_bytes(&FQ, out320 + 192, &q.y.c0); fe_to_bytes(&FQ, out320 + 224, &q.y.c1); } }
  g1_put(out320 + 256, &l);
  if (mask_out) memcpy(mask_out, mask, ns);
  free(w); free(ws); free(mask);
  return 0;
}
/* finish_zk_proof_with_rs (proof.rs:821-849 -> partial_proof.rs:276-305 witness map + :182-274): the FULL witness is
 * calculated again (calc_witness), h over all of it, the four MSMs over the rows the mask leaves unknown. */
int oracle_finish(void* hnd, const uint8_t* inputs_le, const uint8_t* rs_le, const uint8_t* partial320, uint8_t* proof128) {
  const circuit* C = (const circuit*)hnd;
  size_t ns = C->n_signals, n = C->n, nc = C->n_cons, ni = C->n_inst;
  fe* vals = (fe*)malloc(sizeof(fe) * C->n_nodes); fe* w = (fe*)malloc(sizeof(fe) * ns);
  int rc = eval_graph(C, inputs_le, vals, w); free(vals);
  if (rc) { free(w); return rc; }
  uint8_t* kn = (uint8_t*)malloc(C->n_nodes); uint8_t* mask = (uint8_t*)malloc(ns); known_nodes(C, kn, mask); free(kn);
  fe* a = (fe*)calloc(n, sizeof(fe)); fe* b = (fe*)calloc(n, sizeof(fe)); fe* c = (fe*)calloc(n, sizeof(fe));
  for (size_t r = 0; r < nc; r++) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t k = C->a_ptr[r]; k < C->a_ptr[r + 1]; k++) { fe m; fe_mul(&FR, &m, &C->a_val[k], &w[C->a_col[k]]); fe_add(&FR, &acc, &acc, &m); }
    a[r] = acc; memset(&acc, 0, sizeof acc);
    for (size_t k = C->b_ptr[r]; k < C->b_ptr[r + 1]; k++) { fe m; fe_mul(&FR, &m, &C->b_val[k], &w[C->b_col[k]]); fe_add(&FR, &acc, &acc, &m); }
    b[r] = acc; fe_mul(&FR, &c[r], &a[r], &b[r]);
  }
  for (size_t i = 0; i < ni; i++) a[nc + i] = w[i];
  fe* v3[3] = {a, b, c};
  for (int q = 0; q < 3; q++) { ntt(C, v3[q], 1); for (size_t i = 0; i < n; i++) fe_mul(&FR, &v3[q][i], &v3[q][i], &C->coset[i]); ntt(C, v3[q], 0); }
  for (size_t i = 0; i < n; i++) { fe t; fe_mul(&FR, &t, &a[i], &b[i]); fe_sub(&FR, &a[i], &t, &c[i]); }
  u64(*ws)[4] = (u64(*)[4])malloc(sizeof(u64[4]) * ns); u64(*hs)[4] = (u64(*)[4])malloc(sizeof(u64[4]) * n);
  for (size_t i = 0; i < ns; i++) fe_to_u64x4(&FR, ws[i], &w[i]);
  for (size_t i = 0; i < n; i++) fe_to_u64x4(&FR, hs[i], &a[i]);
  u64 r[4], s[4], rsv[4]; memcpy(r, rs_le, 32); memcpy(s, rs_le + 32, 32);
  { fe fr, fs, frs; fe_from_u64x4(&FR, &fr, r); fe_from_u64x4(&FR, &fs, s); fe_mul(&FR, &frs, &fr, &fs); fe_to_u64x4(&FR, rsv, &frs); }
  const int rz = !(r[0] | r[1] | r[2] | r[3]);
  g1_aff pi_a, rho, pi_c; g2_aff pi_b; g1_get(&pi_a, partial320); g1_get(&rho, partial320 + 64); g1_get(&pi_c, partial320 + 256);
  { int nz = 0; for (int k = 0; k < 128; k++) nz |= partial320[128 + k]; pi_b.inf = !nz;
    fe_from_bytes(&FQ, &pi_b.x.c0, partial320 + 128); fe_from_bytes(&FQ, &pi_b.x.c1, partial320 + 160);
    fe_from_bytes(&FQ, &pi_b.y.c0, partial320 + 192); fe_from_bytes(&FQ, &pi_b.y.c1, partial320 + 224); }
  g1_jac ga, gb1, lacc, hacc, gc, t1; g2_jac gb2, t2;
  masked_msms(C, mask, 0, (const u64(*)[4])ws, &ga, &gb1, &gb2, &lacc, !rz);
  g1_add_mixed(&ga, &ga, &pi_a); g1_mul(&t1, &C->delta1, r); g1_add(&ga, &ga, &t1);
  if (!rz) { g1_add_mixed(&gb1, &gb1, &rho); g1_mul(&t1, &C->delta1, s); g1_add(&gb1, &gb1, &t1); } else g1_jac_inf(&gb1);
  g2_add_mixed(&gb2, &gb2, &pi_b); g2_mul(&t2, &C->delta2, s); g2_add(&gb2, &gb2, &t2);
  g1_add_mixed(&lacc, &lacc, &pi_c);
  g1_msm(&hacc, C->hq, (const u64(*)[4])hs, n);
  g1_aff A, B1; g2_aff B2; g1_to_aff(&A, &ga); g1_to_aff(&B1, &gb1); g2_to_aff(&B2, &gb2);
  g1_mul(&gc, &A, s); g1_mul(&t1, &B1, r); g1_add(&gc, &gc, &t1);
  g1_mul(&t1, &C->delta1, rsv); fe_neg(&FQ, &t1.Y, &t1.Y); g1_add(&gc, &gc, &t1);
  g1_add(&gc, &gc, &lacc); g1_add(&gc, &gc, &hacc);
  g1_aff Cc; g1_to_aff(&Cc, &gc);
  if (proof128) { g1_compress(&A, proof128); g2_compress(&B2, proof128 + 32); g1_compress(&Cc, proof128 + 96); }
  free(w); free(a); free(b); free(c); free(ws); free(hs); free(mask);
  return 0;
}
/* 1 when fe_mul (whatever form this build took) equals the portable CIOS product on `iters` pseudo-random pairs and on
 * the edge operands 0, 1, p - 1, both fields */
int oracle_selftest_mul(u64 seed, size_t iters) {
  const field* Fs[2] = {&FR, &FQ}; u64 s = seed ? seed : 88172645463325252ULL;
  for (int f = 0; f < 2; f++) {
    const field* F = Fs[f]; fe e[3]; memset(e, 0, sizeof e); e[1].v[0] = 1; memcpy(e[2].v, F->p, 32); e[2].v[0] -= 1;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { fe x, y; fe_mul(F, &x, &e[i], &e[j]); fe_mul_portable(F, &y, &e[i], &e[j]); if (!fe_eq(&x, &y)) return 0; }
    for (size_t k = 0; k < iters; k++) {
      fe a, b, x, y;
      for (int i = 0; i < 4; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a.v[i] = s; s ^= s << 13; s ^= s >> 7; s ^= s << 17; b.v[i] = s; }
      a.v[3] &= 0x0fffffffffffffffULL; b.v[3] &= 0x0fffffffffffffffULL; /* below both moduli */
      fe_mul(F, &x, &a, &b); fe_mul_portable(F, &y, &a, &b);
      if (!fe_eq(&x, &y)) return 0;
    }
  }
  return 1;
}
const char* oracle_mul_kind(void) { return ORACLE_MUL_KIND; }

typedef struct { void* h; const uint8_t *in, *rs; uint8_t *proofs, *values; size_t n, next; pthread_mutex_t* mu; int rc; const uint8_t* partial; } job;
static void* worker(void* arg) {
  job* J = (job*)arg; size_t ni = oracle_num_inputs(J->h), np = oracle_num_public(J->h);
  for (;;) {
    pthread_mutex_lock(J->mu); size_t i = J->next++; pthread_mutex_unlock(J->mu);
    if (i >= J->n) break;
    /* values: num_public x 32 B per proof (160 B for the single-message circuits) */
    int rc = J->partial ? oracle_finish(J->h, J->in + i * ni * 32, J->rs + i * 64, J->partial + i * 320, J->proofs ? J->proofs + i * 128 : NULL)
                        : oracle_prove(J->h, J->in + i * ni * 32, J->rs + i * 64, J->proofs ? J->proofs + i * 128 : NULL, NULL, NULL, NULL, NULL);
    if (!rc && J->values) public_values((const circuit*)J->h, J->in + i * ni * 32, J->values + i * np * 32);
    if (rc) J->rc = rc;
  }
  return NULL;
}
/* n proofs, one proof per host thread (the deployment mode rln/README.md:324-332 recommends); returns seconds */
static double run_many(void* h, const uint8_t* inputs, const uint8_t* rs, const uint8_t* partial, size_t n, int threads, uint8_t* proofs, uint8_t* values, int* rc_out);
double oracle_prove_many(void* h, const uint8_t* inputs, const uint8_t* rs, size_t n, int threads, uint8_t* proofs, uint8_t* values, int* rc_out) {
  return run_many(h, inputs, rs, NULL, n, threads, proofs, values, rc_out);
}
/* n x finish_zk_proof_with_rs from n cached partial proofs (n x 320 B), one per host thread; returns seconds */
double oracle_finish_many(void* h, const uint8_t* inputs, const uint8_t* rs, const uint8_t* partial320, size_t n, int threads, uint8_t* proofs, int* rc_out) {
  return run_many(h, inputs, rs, partial320, n, threads, proofs, NULL, rc_out);
}
static double run_many(void* h, const uint8_t* inputs, const uint8_t* rs, const uint8_t* partial, size_t n, int threads, uint8_t* proofs, uint8_t* values, int* rc_out) {
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER; job J = {h, inputs, rs, proofs, values, n, 0, &mu, 0, partial};
  struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, &J);
  for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1); free(th);
  if (rc_out) *rc_out = J.rc;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* dense Poseidon tree: leaves (n x 32 B) at [0, n) of a depth-d tree with default leaf 0; root + optional nodes */
void oracle_tree_root(int depth, const uint8_t* leaves_le, size_t n, uint8_t root_le[32]) {
  poseidon_init();
  size_t cap = (size_t)1 << depth; fe* lvl = (fe*)calloc(cap, sizeof(fe));
  for (size_t i = 0; i < n && i < cap; i++) fe_from_bytes(&FR, &lvl[i], leaves_le + 32 * i);
  for (size_t w = cap; w > 1; w >>= 1) for (size_t i = 0; i < w / 2; i++) { fe in[2] = {lvl[2 * i], lvl[2 * i + 1]}; poseidon(&lvl[i], in, 2); }
  fe_to_bytes(&FR, root_le, &lvl[0]); free(lvl);
}

/* ================================================================================================================
 * Side configs of BASELINE.json (configs 3 and 5): closed forms and CPU baselines.  Test infrastructure as above.
 * ================================================================================================================ */

/* ------------------------------------------------------------------------------------------ config 5: one large MSM
 * Workload: point i is P_i = k_i * G, its scalar is s_i, where k_i and
 * s_i are the 253-bit values formed from words 8i..8i+3 and 8i+4..8i+7 of the SplitMix64(seed) stream (little-endian
 * word order, top word masked to 61 bits).  Distribution variants (`mode`, a bit mask):
 *   bit 0   every scalar equals s_0                 -> ONE bucket per window holds all n points
 *   bit 1   k_i = k_(i mod 4)                       -> four distinct bases, every bucket meets its own point again
 * What the reference computes on such an input is VariableBaseMSM::msm_bigint(bases, scalars) = sum s_i P_i
 * (ark-ec 0.5.0; call sites rln/src/partial_proof.rs:98-104).  Because P_i = k_i G the same group element is
 *   (sum_i k_i s_i mod r) * G
 * which needs no curve arithmetic per point: this is the independent full-size check of the device MSM (the product
 * carries no closed form of its own). */
static inline u64 sm64_at(u64 seed, u64 j) {
  u64 z = seed + (j + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
static void msm_wl_k(u64 seed, u64 i, int mode, u64 k[4]) {
  u64 j = (mode & 2) ? (i & 3) : i;
  for (int q = 0; q < 4; q++) k[q] = sm64_at(seed, 8 * j + q);
  k[3] &= 0x1FFFFFFFFFFFFFFFULL;
}
static void msm_wl_s(u64 seed, u64 i, int mode, u64 s[4]) {
  u64 j = (mode & 1) ? 0 : i;
  for (int q = 0; q < 4; q++) s[q] = sm64_at(seed, 8 * j + 4 + q);
  s[3] &= 0x1FFFFFFFFFFFFFFFULL;
}
static const g1_aff* g1_generator(void) {
  static g1_aff G; static int ready = 0;
  if (!ready) { fe_set_u64(&FQ, &G.x, 1); fe_set_u64(&FQ, &G.y, 2); G.inf = 0; ready = 1; }
  return &G;
}
typedef struct { u64 seed, lo, hi; int mode; fe acc; } cf_job;
static void* cf_worker(void* a) {
  cf_job* J = (cf_job*)a; fe acc; memset(&acc, 0, sizeof acc);
  for (u64 i = J->lo; i < J->hi; i++) {
    u64 k[4], s[4]; fe fk, fs, m; msm_wl_k(J->seed, i, J->mode, k); msm_wl_s(J->seed, i, J->mode, s);
    fe_from_u64x4(&FR, &fk, k); fe_from_u64x4(&FR, &fs, s); fe_mul(&FR, &m, &fk, &fs); fe_add(&FR, &acc, &acc, &m);
  }
  J->acc = acc; return NULL;
}
/* (sum_{i in [first, first+n)} k_i s_i mod r) * G  ->  affine x | y, 32-byte little-endian each; (0, 0) = infinity */
void oracle_msm_expected(u64 seed, u64 first, u64 n, int mode, int threads, uint8_t out_xy_le[64]) {
  if (threads < 1) threads = 1;
  if ((u64)threads > n) threads = n ? (int)n : 1;
  cf_job* J = (cf_job*)calloc(threads, sizeof(cf_job)); pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  for (int t = 0; t < threads; t++) {
    J[t].seed = seed; J[t].mode = mode; J[t].lo = first + n * t / threads; J[t].hi = first + n * (t + 1) / threads;
    pthread_create(&th[t], NULL, cf_worker, &J[t]);
  }
  fe acc; memset(&acc, 0, sizeof acc);
  for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); fe_add(&FR, &acc, &acc, &J[t].acc); }
  free(J); free(th);
  u64 e[4]; fe_to_u64x4(&FR, e, &acc);
  g1_jac r; g1_mul(&r, g1_generator(), e); g1_aff a; g1_to_aff(&a, &r);
  memset(out_xy_le, 0, 64);
  if (!a.inf) { fe_to_bytes(&FQ, out_xy_le, &a.x); fe_to_bytes(&FQ, out_xy_le + 32, &a.y); }
}
/* the workload itself, for spot checks of the device generator: point i (affine x | y) and scalar i */
void oracle_msm_workload_item(u64 seed, u64 i, int mode, uint8_t point_xy_le[64], uint8_t scalar_le[32]) {
  u64 k[4], s[4]; msm_wl_k(seed, i, mode, k); msm_wl_s(seed, i, mode, s);
  g1_jac r; g1_mul(&r, g1_generator(), k); g1_aff a; g1_to_aff(&a, &r);
  memset(point_xy_le, 0, 64);
  if (!a.inf) { fe_to_bytes(&FQ, point_xy_le, &a.x); fe_to_bytes(&FQ, point_xy_le + 32, &a.y); }
  memcpy(scalar_le, s, 32);
}

/* The same workload on G2 (north_star: "windowed Pippenger MSM on G1/G2"): P_i = k_i * G2, G2 = the generator of the
 * twist's order-r subgroup (ark-bn254 g2::G2_GENERATOR_X / _Y).  out: x.c0 | x.c1 | y.c0 | y.c1, all zero = infinity */
static const g2_aff* g2_generator(void) {
  static g2_aff G; static int ready = 0;
  if (!ready) {
    static const u64 w[4][4] = {
      {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL},
      {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL},
      {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL},
      {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL}};
    fe_from_u64x4(&FQ, &G.x.c0, w[0]); fe_from_u64x4(&FQ, &G.x.c1, w[1]); fe_from_u64x4(&FQ, &G.y.c0, w[2]); fe_from_u64x4(&FQ, &G.y.c1, w[3]);
    G.inf = 0; ready = 1;
  }
  return &G;
}
static void g2_aff_to_bytes(const g2_aff* a, uint8_t out[128]) {
  memset(out, 0, 128);
  if (a->inf) return;
  fe_to_bytes(&FQ, out, &a->x.c0); fe_to_bytes(&FQ, out + 32, &a->x.c1); fe_to_bytes(&FQ, out + 64, &a->y.c0); fe_to_bytes(&FQ, out + 96, &a->y.c1);
}
void oracle_msm_expected_g2(u64 seed, u64 first, u64 n, int mode, int threads, uint8_t out_le[128]) {
  if (threads < 1) threads = 1;
  if ((u64)threads > n) threads = n ? (int)n : 1;
  cf_job* J = (cf_job*)calloc(threads, sizeof(cf_job)); pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  for (int t = 0; t < threads; t++) {
    J[t].seed = seed; J[t].mode = mode; J[t].lo = first + n * t / threads; J[t].hi = first + n * (t + 1) / threads;
    pthread_create(&th[t], NULL, cf_worker, &J[t]);
  }
  fe acc; memset(&acc, 0, sizeof acc);
  for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); fe_add(&FR, &acc, &acc, &J[t].acc); }
  free(J); free(th);
  u64 e[4]; fe_to_u64x4(&FR, e, &acc);
  g2_jac r; g2_mul(&r, g2_generator(), e); g2_aff a; g2_to_aff(&a, &r);
  g2_aff_to_bytes(&a, out_le);
}
void oracle_msm_workload_item_g2(u64 seed, u64 i, int mode, uint8_t point_le[128], uint8_t scalar_le[32]) {
  u64 k[4], s[4]; msm_wl_k(seed, i, mode, k); msm_wl_s(seed, i, mode, s);
  g2_jac r; g2_mul(&r, g2_generator(), k); g2_aff a; g2_to_aff(&a, &r);
  g2_aff_to_bytes(&a, point_le);
  memcpy(scalar_le, s, 32);
}

/* CPU baseline of config 5: the points are materialised (fixed-base comb of G, batch-normalised -- untimed), then
 * msm_bigint's windowed Pippenger runs over them with the windows spread over the host threads, which is how ark-ec's
 * `parallel` feature spreads them (one rayon task per window, serial fold at the end). */
typedef struct { const g1_aff* pts; const u64 (*sc)[4]; size_t n; int c, nw; g1_jac* wsum; int next; pthread_mutex_t mu; } pip_job;
/* one window of msm_bigint_wnaf (see g1_msm): the signed digit of every scalar in window w, recomputed from the carry
 * rule (the carry into window w depends only on the bits below it: digit_{w-1} > half the radix, recursively -- here
 * evaluated by walking the scalar's windows up to w) */
static inline int32_t signed_digit(const u64 sc[4], int c, int nw, int w) {
  const u64 radix = (u64)1 << c, mask = radix - 1; u64 carry = 0; int32_t d = 0;
  for (int k = 0; k <= w; k++) {
    const int bit = k * c, wi = bit >> 6, bi = bit & 63;
    const u64 buf = (bi < 64 - c || wi == 3) ? sc[wi] >> bi : (sc[wi] >> bi) | (sc[wi + 1] << (64 - bi));
    const u64 coef = carry + (buf & mask);
    carry = (coef + radix / 2) >> c;
    d = (int32_t)((int64_t)coef - (int64_t)(carry << c));
  }
  if (w == nw - 1) d += (int32_t)(carry << c);
  return d;
}
static void pip_window(const pip_job* J, int w, g1_jac* out) {
  int c = J->c; size_t nb = (size_t)1 << (c - 1);
  g1_jac* buckets = (g1_jac*)malloc(nb * sizeof(g1_jac));
  for (size_t b = 0; b < nb; b++) g1_jac_inf(&buckets[b]);
  for (size_t i = 0; i < J->n; i++) {
    if (J->pts[i].inf) continue;
    const int32_t d = signed_digit(J->sc[i], c, J->nw, w);
    if (d > 0) g1_add_mixed(&buckets[d - 1], &buckets[d - 1], &J->pts[i]);
    else if (d < 0) { g1_aff m = J->pts[i]; fe_neg(&FQ, &m.y, &m.y); g1_add_mixed(&buckets[-d - 1], &buckets[-d - 1], &m); }
  }
  g1_jac run, ws; g1_jac_inf(&run); g1_jac_inf(&ws);
  for (size_t b = nb; b-- > 0;) { g1_add(&run, &run, &buckets[b]); g1_add(&ws, &ws, &run); }
  free(buckets); *out = ws;
}
static void* pip_worker(void* a) {
  pip_job* J = (pip_job*)a;
  for (;;) {
    pthread_mutex_lock(&J->mu); int w = J->next++; pthread_mutex_unlock(&J->mu);
    if (w >= J->nw) break;
    pip_window(J, w, &J->wsum[w]);
  }
  return NULL;
}
typedef struct { u64 seed, first; size_t lo, hi; int mode; const g1_aff* comb; g1_aff* pts; u64 (*sc)[4]; } gen_job;
static void* gen_worker(void* a) {
  gen_job* J = (gen_job*)a; size_t m = J->hi - J->lo; if (!m) return NULL;
  g1_jac* tmp = (g1_jac*)malloc(m * sizeof(g1_jac)); fe* pref = (fe*)malloc(m * sizeof(fe));
  for (size_t t = 0; t < m; t++) {
    u64 k[4]; msm_wl_k(J->seed, J->first + J->lo + t, J->mode, k); msm_wl_s(J->seed, J->first + J->lo + t, J->mode, J->sc[J->lo + t]);
    g1_jac acc; g1_jac_inf(&acc);
    for (int w = 0; w < 32; w++) { unsigned d = (unsigned)(k[w >> 3] >> ((w & 7) * 8)) & 255u; if (d) g1_add_mixed(&acc, &acc, &J->comb[w * 255 + d - 1]); }
    tmp[t] = acc;
  }
  /* batch normalisation (Montgomery's trick) */
  fe run; fe_one(&FQ, &run);
  for (size_t t = 0; t < m; t++) { pref[t] = run; if (!fe_is_zero(&tmp[t].Z)) fe_mul(&FQ, &run, &run, &tmp[t].Z); }
  fe inv; fe_inv(&FQ, &inv, &run);
  for (size_t t = m; t-- > 0;) {
    g1_aff* o = &J->pts[J->lo + t];
    if (fe_is_zero(&tmp[t].Z)) { memset(o, 0, sizeof *o); o->inf = 1; continue; }
    fe zi, zi2; fe_mul(&FQ, &zi, &inv, &pref[t]); fe_mul(&FQ, &inv, &inv, &tmp[t].Z);
    fe_sqr(&FQ, &zi2, &zi); fe_mul(&FQ, &o->x, &tmp[t].X, &zi2); fe_mul(&FQ, &zi2, &zi2, &zi); fe_mul(&FQ, &o->y, &tmp[t].Y, &zi2); o->inf = 0;
  }
  free(tmp); free(pref); return NULL;
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
/* returns the seconds of the Pippenger MSM alone; *gen_seconds = the untimed workload generation */
double oracle_msm_pippenger(u64 seed, u64 first, size_t n, int mode, int threads, uint8_t out_xy_le[64], double* gen_seconds) {
  if (threads < 1) threads = 1;
  double t0 = now_s();
  /* comb table of G: entry [w][d-1] = d * 2^(8w) * G */
  g1_aff* comb = (g1_aff*)malloc(32 * 255 * sizeof(g1_aff));
  g1_jac base; { g1_jac_inf(&base); g1_add_mixed(&base, &base, g1_generator()); }
  for (int w = 0; w < 32; w++) {
    g1_aff b; g1_to_aff(&b, &base); g1_jac e; g1_jac_inf(&e);
    for (int d = 1; d <= 255; d++) { g1_add_mixed(&e, &e, &b); g1_to_aff(&comb[w * 255 + d - 1], &e); }
    for (int k = 0; k < 8; k++) g1_dbl(&base, &base);
  }
  g1_aff* pts = (g1_aff*)malloc(n * sizeof(g1_aff)); u64(*sc)[4] = (u64(*)[4])malloc(n * sizeof(u64[4]));
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads); gen_job* G = (gen_job*)calloc(threads, sizeof(gen_job));
  for (int t = 0; t < threads; t++) {
    G[t] = (gen_job){seed, first, n * t / threads, n * (t + 1) / threads, mode, comb, pts, sc};
    pthread_create(&th[t], NULL, gen_worker, &G[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(G); free(comb);
  double t1 = now_s();
  if (gen_seconds) *gen_seconds = t1 - t0;
  /* ---- timed: msm_bigint */
  pip_job J; J.pts = pts; J.sc = (const u64(*)[4])sc; J.n = n; J.next = 0; pthread_mutex_init(&J.mu, NULL);
  { int c = 0; size_t m = n; while (m > 1) { m >>= 1; c++; } J.c = n < 32 ? 3 : c * 69 / 100 + 2; } /* ark-ec: ln(n) + 2 */
  J.nw = (254 + J.c - 1) / J.c; J.wsum = (g1_jac*)malloc(J.nw * sizeof(g1_jac));
  int nt = threads < J.nw ? threads : J.nw;
  for (int t = 0; t < nt; t++) pthread_create(&th[t], NULL, pip_worker, &J);
  for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
  g1_jac total; g1_jac_inf(&total);
  for (int w = J.nw - 1; w >= 0; w--) { for (int k = 0; k < J.c; k++) g1_dbl(&total, &total); g1_add(&total, &total, &J.wsum[w]); }
  g1_aff a; g1_to_aff(&a, &total);
  double t2 = now_s();
  memset(out_xy_le, 0, 64);
  if (!a.inf) { fe_to_bytes(&FQ, out_xy_le, &a.x); fe_to_bytes(&FQ, out_xy_le + 32, &a.y); }
  free(J.wsum); free(th); free(pts); free(sc); pthread_mutex_destroy(&J.mu);
  return t2 - t1;
}

/* ------------------------------------------------------------------------------------------ config 3: Merkle tree
 * FullMerkleTree restated (utils/src/merkle_tree/full_merkle_tree.rs): flat node array, node i has children 2i+1 and
 * 2i+2, leaves at capacity - 1 + index (:82-115); set_range writes the leaves and calls update_hashes, which recomputes
 * the parents of the touched range level by level, in parallel when a level has many of them (:197-223, :360-399);
 * set(index) is the one-leaf case: `depth` dependent hashes (:336-399). */
typedef struct { int depth; size_t cap; fe* nodes; } otree;
typedef struct { otree* T; size_t lo, hi; } lvl_job;
static void* lvl_worker(void* a) {
  lvl_job* J = (lvl_job*)a;
  for (size_t p = J->lo; p < J->hi; p++) { fe in[2] = {J->T->nodes[2 * p + 1], J->T->nodes[2 * p + 2]}; poseidon(&J->T->nodes[p], in, 2); }
  return NULL;
}
static void otree_update_hashes(otree* T, size_t start, size_t end, int threads) { /* inclusive node range on one level */
  while (start > 0) {
    size_t sp = ((start + 1) >> 1) - 1, ep = ((end + 1) >> 1) - 1, cnt = ep - sp + 1;
    int nt = (threads > 1 && cnt >= 64) ? threads : 1; /* reference: MIN_PARALLEL_NODES = 8 on a rayon pool (merkle_tree.rs:18); threads are spawned per level here, so the cut is 64 */
    if (nt == 1) { lvl_job J = {T, sp, ep + 1}; lvl_worker(&J); }
    else {
      pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nt); lvl_job* J = (lvl_job*)malloc(sizeof(lvl_job) * nt);
      for (int t = 0; t < nt; t++) { J[t] = (lvl_job){T, sp + cnt * t / nt, sp + cnt * (t + 1) / nt}; pthread_create(&th[t], NULL, lvl_worker, &J[t]); }
      for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
      free(th); free(J);
    }
    start = sp; end = ep;
  }
}
void* oracle_tree_new(int depth) {
  poseidon_init();
  otree* T = (otree*)malloc(sizeof(otree)); T->depth = depth; T->cap = (size_t)1 << depth;
  T->nodes = (fe*)calloc(2 * T->cap - 1, sizeof(fe));
  /* ::new(depth, default_leaf = 0): every level holds the hash of two default children (:82-115) */
  fe cur; memset(&cur, 0, sizeof cur);
  for (int l = depth; l >= 0; l--) {
    size_t lo = ((size_t)1 << l) - 1, n = (size_t)1 << l;
    for (size_t i = 0; i < n; i++) T->nodes[lo + i] = cur;
    fe in[2] = {cur, cur}; poseidon(&cur, in, 2);
  }
  return T;
}
void oracle_tree_free(void* h) { otree* T = (otree*)h; free(T->nodes); free(T); }
int oracle_tree_set_range(void* h, size_t start, const uint8_t* leaves_le, size_t n, int threads) {
  otree* T = (otree*)h; if (start + n > T->cap) return 1; if (!n) return 0;
  size_t idx = T->cap - 1 + start;
  for (size_t i = 0; i < n; i++) fe_from_bytes(&FR, &T->nodes[idx + i], leaves_le + 32 * i);
  otree_update_hashes(T, idx, idx + n - 1, threads); return 0;
}
void oracle_tree_get_root(void* h, uint8_t root_le[32]) { fe_to_bytes(&FR, root_le, &((otree*)h)->nodes[0]); }
void oracle_tree_proof(void* h, size_t leaf, uint8_t* elems_le, uint8_t* bits) { /* :288-304 */
  otree* T = (otree*)h; size_t idx = T->cap - 1 + leaf; int k = 0;
  while (idx > 0) { int right = !(idx & 1); fe_to_bytes(&FR, elems_le + 32 * k, &T->nodes[right ? idx - 1 : idx + 1]); bits[k++] = (uint8_t)right; idx = ((idx + 1) >> 1) - 1; }
}
/* CPU baseline of config 3 and of the tree-mutation calls: leaves i -> first_value + i (the product's config-3 workload).
 * out[0] = seconds of the full build (set_range of n leaves, all threads), out[1] = seconds of ONE set(index) incl. the
 * root read, averaged over `singles` calls, out[2] = seconds of `scattered` set() calls at pseudo-random indices followed
 * by one root read (the reference pays depth hashes per call), out[3] = seconds of n membership proofs (path copies). */
void oracle_tree_bench(int depth, size_t n, u64 first_value, int threads, int singles, int scattered, double out[4], uint8_t root_le[32],
                       uint8_t root_after_le[32]) {
  otree* T = (otree*)oracle_tree_new(depth);
  uint8_t* leaves = (uint8_t*)calloc(n, 32);
  for (size_t i = 0; i < n; i++) { u64 v = first_value + i; memcpy(leaves + 32 * i, &v, 8); }
  double t0 = now_s(); oracle_tree_set_range(T, 0, leaves, n, threads); double t1 = now_s();
  out[0] = t1 - t0; oracle_tree_get_root(T, root_le);
  uint8_t* el = (uint8_t*)malloc(32 * depth); uint8_t* bits = (uint8_t*)malloc(depth);
  t0 = now_s(); for (size_t i = 0; i < n; i++) oracle_tree_proof(T, i, el, bits); t1 = now_s(); out[3] = t1 - t0;
  uint8_t leaf[32]; uint8_t r[32];
  t0 = now_s();
  for (int k = 0; k < singles; k++) { memset(leaf, 0, 32); u64 v = 0x5157000000000000ULL + k; memcpy(leaf, &v, 8); oracle_tree_set_range(T, (sm64_at(0x7EE, k) % n), leaf, 1, 1); oracle_tree_get_root(T, r); }
  t1 = now_s(); out[1] = singles ? (t1 - t0) / singles : 0;
  t0 = now_s();
  for (int k = 0; k < scattered; k++) { memset(leaf, 0, 32); u64 v = 0x5CA7000000000000ULL + k; memcpy(leaf, &v, 8); oracle_tree_set_range(T, (sm64_at(0x5CA7, k) % n), leaf, 1, 1); }
  oracle_tree_get_root(T, root_after_le); t1 = now_s(); out[2] = t1 - t0;
  free(el); free(bits); free(leaves); oracle_tree_free(T);
}
