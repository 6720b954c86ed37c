// GLV split of a BN254 scalar for the fixed-base table walks.
//
// The reference's MSM (ark-ec 0.5.0 `VariableBaseMSM::msm_bigint`, call sites rln/src/partial_proof.rs:103,256) walks
// 254-bit scalars.  Here every base P carries a comb table, and phi(x, y) = (beta x, y) = [lambda] P is free on table
// entries: sum_k s_k P_k = sum_k k1_k P_k + phi(sum_k k2_k P_k) with s = k1 + lambda k2 (mod r), |k1|, |k2| < 2^126.
// The table then only has to cover 127 bits: the same HBM buys wider windows (G1: 9 + 9 additions per point instead of
// 19, G2: 8 + 8 instead of 19), and phi is applied ONCE per proof to the sum of the k2 halves (one Fq product).
// Group elements are canonical, so the split cannot change an output bit.
//
// Everything is __host__ __device__: the CPU build is checked against tools/gen_glv.py's model in
// tests/test_host_math.py.
#pragma once
#include "field.h"
#include "glv_constants.h"

namespace rlnamd {

// out[0..NO) = limbs [drop, drop + NO) of a * b + round, a: NA limbs, b: NB limbs (schoolbook, 64-bit columns)
template <int NA, int NB, int DROP, int NO>
RLN_HD void glv_mul_window(const uint32_t* a, const uint32_t* b, bool round_half, uint32_t* out) {
  uint32_t t[NA + NB];
#pragma unroll
  for (int i = 0; i < NA + NB; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      uint64_t v = (uint64_t)a[i] * b[j] + t[i + j] + carry;
      t[i + j] = (uint32_t)v;
      carry = v >> 32;
    }
    t[i + NB] = (uint32_t)carry;
  }
  if (round_half) {  // + 2^(32 DROP - 1): bit 31 of limb DROP - 1, carried upward
    uint64_t v = (uint64_t)t[DROP - 1] + 0x80000000u;
    uint32_t c = (uint32_t)(v >> 32);
#pragma unroll
    for (int i = DROP; i < NA + NB; i++) {
      uint64_t w = (uint64_t)t[i] + c;
      t[i] = (uint32_t)w;
      c = (uint32_t)(w >> 32);
    }
  }
#pragma unroll
  for (int i = 0; i < NO; i++) out[i] = (DROP + i < NA + NB) ? t[DROP + i] : 0;
}

// low four limbs of a * b (both four limbs)
RLN_HD void glv_mul_lo128(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  uint32_t t[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j + i < 4; j++) {
      uint64_t v = (uint64_t)a[i] * b[j] + t[i + j] + carry;
      t[i + j] = (uint32_t)v;
      carry = v >> 32;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = t[i];
}
RLN_HD void glv_sub128(uint32_t* a, const uint32_t* b) {  // a -= b mod 2^128
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint64_t v = (uint64_t)a[i] - b[i] - borrow;
    a[i] = (uint32_t)v;
    borrow = (uint32_t)(v >> 63);
  }
}
RLN_HD uint32_t glv_abs128(uint32_t* a) {  // two's complement -> magnitude, returns the sign
  const uint32_t neg = a[3] >> 31;
  if (neg) {
    uint32_t carry = 1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint64_t v = (uint64_t)(~a[i]) + carry;
      a[i] = (uint32_t)v;
      carry = (uint32_t)(v >> 32);
    }
  }
  return neg;
}

// k: canonical scalar (< r).  k = (-1)^neg1 k1 + lambda (-1)^neg2 k2 (mod r), k1, k2 < 2^126.
RLN_HD void glv_split(const uint32_t k[8], uint32_t k1[4], uint32_t* neg1, uint32_t k2[4], uint32_t* neg2) {
  // c1 = round(k b2 / r) < 2^64, c2 = round(k |b1| / r) < 2^127: products against g_i = round(2^288 |b_i| / r), top
  // limbs kept.  The result only matters mod 2^128 because |k1|, |k2| < 2^126 (tools/gen_glv.py).
  uint32_t c1[4], c2[4], t[4];
  glv_mul_window<8, 4, 9, 4>(k, GlvParams::G1, true, c1);
  glv_mul_window<8, 6, 9, 4>(k, GlvParams::G2, true, c2);
  // k1 = k - c1 a1 - c2 a2
#pragma unroll
  for (int i = 0; i < 4; i++) k1[i] = k[i];
  glv_mul_lo128(c1, GlvParams::A1, t);
  glv_sub128(k1, t);
  glv_mul_lo128(c2, GlvParams::A2, t);
  glv_sub128(k1, t);
  // k2 = -c1 b1 - c2 b2 = c1 |b1| - c2 b2
  glv_mul_lo128(c1, GlvParams::B1ABS, k2);
  glv_mul_lo128(c2, GlvParams::B2, t);
  glv_sub128(k2, t);
  *neg1 = glv_abs128(k1);
  *neg2 = glv_abs128(k2);
}

}  // namespace rlnamd
