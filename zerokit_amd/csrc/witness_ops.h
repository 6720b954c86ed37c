// Device-side graph operations shared by the witness interpreters (prover.hip: k_witness, k_witness29;
// witness_lanes.hip: k_witness_lanes): 256-bit integer helpers and every operation of
// /root/reference/rln/src/circuit/iden3calc/graph.rs:72-143, 314-466 that is not Mul / Add / Sub / Neg / TernCond.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RLN_WOP static __host__ __device__
#define RLN_WOP_NOINLINE static __host__ __device__ __noinline__
#else   // the CPU suite compiles the same operations into its host emulator of the interpreter (tests/host/witsched.cpp)
#define RLN_WOP static inline
#define RLN_WOP_NOINLINE static inline
#endif
#include <stdint.h>

#include "curve.h"
#include "zkey.h"

namespace rlnamd {

struct U256 {
  uint32_t v[8];
};
RLN_WOP bool u_is_zero(const U256& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}
RLN_WOP int u_cmp(const U256& a, const U256& b) {  // -1, 0, 1
  for (int i = 7; i >= 0; i--) {
    if (a.v[i] != b.v[i]) return a.v[i] > b.v[i] ? 1 : -1;
  }
  return 0;
}
RLN_WOP U256 u_from_limbs(const uint32_t* p) {
  U256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = p[i];
  return r;
}
RLN_WOP U256 u_sub(const U256& a, const U256& b) {
  U256 r;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t s = (uint64_t)a.v[i] - b.v[i] - borrow;
    r.v[i] = (uint32_t)s;
    borrow = (uint32_t)(s >> 63);
  }
  return r;
}
RLN_WOP U256 u_shr(const U256& a, unsigned n) {  // n < 256
  U256 r;
  unsigned w = n >> 5, b = n & 31;
  for (int i = 0; i < 8; i++) {
    uint32_t lo = (i + w < 8) ? a.v[i + w] : 0;
    uint32_t hi = (i + w + 1 < 8) ? a.v[i + w + 1] : 0;
    r.v[i] = b ? ((lo >> b) | (hi << (32 - b))) : lo;
  }
  return r;
}
RLN_WOP U256 u_shl(const U256& a, unsigned n) {  // n < 256, bits above 256 dropped (ark BigInt <<)
  U256 r;
  unsigned w = n >> 5, b = n & 31;
  for (int i = 7; i >= 0; i--) {
    uint32_t hi = (i >= (int)w) ? a.v[i - w] : 0;
    uint32_t lo = (i >= (int)w + 1) ? a.v[i - w - 1] : 0;
    r.v[i] = b ? ((hi << b) | (lo >> (32 - b))) : hi;
  }
  return r;
}
// a / b and a % b by shift-subtract (b != 0)
RLN_WOP void u_divmod(const U256& a, const U256& b, U256* q, U256* rem) {
  U256 Q, Rm;
  for (int i = 0; i < 8; i++) Q.v[i] = Rm.v[i] = 0;
  for (int bit = 255; bit >= 0; bit--) {
    uint32_t top = Rm.v[7] >> 31;
    Rm = u_shl(Rm, 1);
    Rm.v[0] |= (a.v[bit >> 5] >> (bit & 31)) & 1;
    if (top || u_cmp(Rm, b) >= 0) {
      Rm = u_sub(Rm, b);
      Q.v[bit >> 5] |= 1u << (bit & 31);
    }
  }
  *q = Q;
  *rem = Rm;
}

enum WitnessErr : uint32_t { WERR_NONE = 0, WERR_INPUT_RANGE = 1, WERR_SHIFT = 2, WERR_BITOP = 3, WERR_UNO_ID = 4,
                             WERR_HINT = 0x100 };   // a flag: a cut node's value differs from the hint its consumers were given (Prover::collect runs the batch again)

// Every operation that is not Mul/Add/Sub/Neg/TernCond/Const/Input: iden3calc/graph.rs:72-143, 314-466.
RLN_WOP_NOINLINE Fr witness_slow_op(uint32_t op, Fr fa, Fr fb, uint32_t* err) {
  U256 a, b, m, half;
  fa.to_canonical(a.v);
  fb.to_canonical(b.v);
  m = u_from_limbs(FrParams::MOD);
  half = u_from_limbs(FrParams::HALF);
  auto boolean = [](bool x) { return x ? Fr::one() : Fr::zero(); };
  switch (op) {
    case G_DIV:
      return fb.is_zero() ? Fr::zero() : fa * fb.inv();
    case G_POW:
      return fa.pow(b.v);
    case G_IDIV:
    case G_MOD: {
      if (u_is_zero(b)) return Fr::zero();
      U256 q, r;
      u_divmod(a, b, &q, &r);
      return Fr::from_canonical(op == G_IDIV ? q.v : r.v);
    }
    case G_EQ:
      return boolean(u_cmp(a, b) == 0);
    case G_NEQ:
      return boolean(u_cmp(a, b) != 0);
    case G_LT:
    case G_GT:
    case G_LEQ:
    case G_GEQ: {  // values above M/2 are negative (graph.rs:410-466)
      bool an = u_cmp(a, half) > 0, bn = u_cmp(b, half) > 0;
      int c = u_cmp(a, b);
      bool res;
      if (an == bn)
        res = op == G_LT ? c < 0 : op == G_GT ? c > 0 : op == G_LEQ ? c <= 0 : c >= 0;
      else
        res = (op == G_LT || op == G_LEQ) ? an : bn;
      return boolean(res);
    }
    case G_LAND:
      return boolean(!u_is_zero(a) && !u_is_zero(b));
    case G_LOR:
      return boolean(!u_is_zero(a) || !u_is_zero(b));
    case G_SHL: {  // graph.rs:314-326
      if (u_is_zero(b)) return fa;
      U256 lim = {{254, 0, 0, 0, 0, 0, 0, 0}};
      if (u_cmp(b, lim) >= 0) return Fr::zero();
      U256 r = u_shl(a, b.v[0]);
      if (u_cmp(r, m) >= 0) {
        *err = WERR_SHIFT;
        return Fr::zero();
      }
      return Fr::from_canonical(r.v);
    }
    case G_SHR: {  // graph.rs:328-363
      if (u_is_zero(b)) return fa;
      U256 lim = {{254, 0, 0, 0, 0, 0, 0, 0}};
      if (u_cmp(b, lim) >= 0) return Fr::zero();
      U256 r = u_shr(a, b.v[0] & 0xFF);
      return Fr::from_canonical(r.v);
    }
    case G_BOR:
    case G_BAND:
    case G_BXOR: {  // graph.rs:365-408: one subtraction when d > MODULUS, then from_bigint
      U256 d;
      for (int i = 0; i < 8; i++)
        d.v[i] = op == G_BOR ? (a.v[i] | b.v[i]) : op == G_BAND ? (a.v[i] & b.v[i]) : (a.v[i] ^ b.v[i]);
      if (u_cmp(d, m) > 0) d = u_sub(d, m);
      if (u_cmp(d, m) >= 0) {
        *err = WERR_BITOP;
        return Fr::zero();
      }
      return Fr::from_canonical(d.v);
    }
    case G_ID:
      *err = WERR_UNO_ID;  // "uno operator Id not implemented for Montgomery" (graph.rs:201-204)
      return Fr::zero();
    default:
      return Fr::zero();
  }
}

}  // namespace rlnamd
