// modinv30.h -- modular inversion by batched division steps (Bernstein-Yang "safegcd", 2019), 30 steps per batch on
// 32-bit words.  Replaces the bit-by-bit binary extended Euclid wherever an inversion is a lone lane's dependent chain
// (the to_affine of a proof's three points, the table build's batch inversions): that loop is ~760 data-dependent
// halvings of a 256-bit integer with a branch each; here the 30 steps of a batch run on ONE word of f and g without
// branches and produce a 2 x 2 transition matrix that is applied to the full integers with 36 + 60 multiply-adds.
//
// Division step on (delta, f, g), f odd:
//     delta > 0 and g odd:  (delta, f, g) <- (1 - delta, g, (g - f) / 2)
//     otherwise:            (delta, f, g) <- (1 + delta, f, (g + (g mod 2) f) / 2)
// Starting from (1, p, x) the sequence reaches g = 0 with f = +-gcd(p, x) within 741 steps for 256-bit inputs (Theorem
// 11.2 of the paper); beside it (d, e) start as (0, 1) and follow the same matrices divided by 2^30 MODULO p, so that
// d x = f and e x = g (mod p) throughout: at the end x^-1 = +-d.
//
// Representation: signed limbs of 30 bits (value = sum v[i] 2^(30 i), v[0..7] in [0, 2^30), v[8] signed).
#pragma once
#include <stdint.h>

namespace rlnamd {

struct S30 {
  int32_t v[9];
};

namespace modinv30_detail {

constexpr int32_t M30 = (1 << 30) - 1;

}  // namespace modinv30_detail

#if defined(__HIPCC__)
#define RLN_MI_HD __host__ __device__ __forceinline__
#else
#define RLN_MI_HD inline
#endif

// 8 x 32 -> 9 x 30 (non-negative)
RLN_MI_HD S30 s30_from_words(const uint32_t* w) {
  S30 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 30 * i, k = bit >> 5, sh = bit & 31;
    uint64_t t = (uint64_t)w[k] >> sh;
    if (sh > 2 && k + 1 < 8) t |= (uint64_t)w[k + 1] << (32 - sh);
    r.v[i] = (int32_t)((uint32_t)t & (uint32_t)modinv30_detail::M30);
  }
  return r;
}
// non-negative 9 x 30 below 2^256 -> 8 x 32
RLN_MI_HD void s30_to_words(const S30& a, uint32_t* w) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int bit = 32 * k, i = bit / 30, sh = bit % 30;
    uint64_t t = (uint64_t)(uint32_t)a.v[i] >> sh;
    if (i + 1 < 9) t |= (uint64_t)(uint32_t)a.v[i + 1] << (30 - sh);
    if (i + 2 < 9 && 60 - sh < 32) t |= (uint64_t)(uint32_t)a.v[i + 2] << (60 - sh);
    w[k] = (uint32_t)t;
  }
}

// 30 division steps on the low words of f and g: the transition matrix t = (u v; q r), scaled by 2^30
struct Trans30 {
  int32_t u, v, q, r;
};
RLN_MI_HD int32_t divsteps30(int32_t delta, uint32_t f0, uint32_t g0, Trans30* t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;   // two's complement; |entries| <= 2^30
#pragma unroll
  for (int i = 0; i < 30; i++) {
    const uint32_t odd = 0u - (g0 & 1u);                           // all ones when g is odd
    const uint32_t sw = odd & (0u - (uint32_t)(delta > 0));        // swap-and-negate case
    // (delta, f, g, u, v, q, r) <- (-delta, g, -f, q, r, -u, -v) when sw
    const uint32_t nf = (f0 ^ sw) - sw, nu = (u ^ sw) - sw, nv = (v ^ sw) - sw;   // conditional negations
    const uint32_t f1 = sw ? g0 : f0, u1 = sw ? q : u, v1 = sw ? r : v;
    const uint32_t g1 = sw ? nf : g0, q1 = sw ? nu : q, r1 = sw ? nv : r;
    delta = (sw ? -delta : delta) + 1;
    // g <- (g + odd f) / 2, (q, r) <- (q, r) + odd (u, v), (u, v) <- 2 (u, v)
    g0 = (g1 + (f1 & odd)) >> 1;
    q = q1 + (u1 & odd);
    r = r1 + (v1 & odd);
    f0 = f1;
    u = u1 << 1;
    v = v1 << 1;
  }
  t->u = (int32_t)u;
  t->v = (int32_t)v;
  t->q = (int32_t)q;
  t->r = (int32_t)r;
  return delta;
}

// (f, g) <- t (f, g) / 2^30 (exact)
RLN_MI_HD void update_fg30(S30* f, S30* g, const Trans30& t) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  int64_t cf = u * f->v[0] + v * g->v[0], cg = q * f->v[0] + r * g->v[0];
  cf >>= 30;   // the low 30 bits are zero by construction
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    const int64_t fi = f->v[i], gi = g->v[i];
    cf += u * fi + v * gi;
    cg += q * fi + r * gi;
    f->v[i - 1] = (int32_t)((uint32_t)cf & (uint32_t)modinv30_detail::M30);
    g->v[i - 1] = (int32_t)((uint32_t)cg & (uint32_t)modinv30_detail::M30);
    cf >>= 30;
    cg >>= 30;
  }
  f->v[8] = (int32_t)cf;
  g->v[8] = (int32_t)cg;
}

// x in (-p, 2 p) with limbs 0..7 in [0, 2^30), limb 8 signed  ->  [0, p)
RLN_MI_HD void normalize30(S30* x, const S30& p) {
  // + p when negative
  {
    const int32_t neg = x->v[8] >> 31;   // all ones when negative
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int32_t s = x->v[i] + (p.v[i] & neg) + c;
      if (i < 8) {
        x->v[i] = s & modinv30_detail::M30;
        c = s >> 30;
      } else {
        x->v[i] = s;
      }
    }
  }
  // - p when the difference is not negative
  {
    S30 d;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int32_t s = x->v[i] - p.v[i] + c;
      if (i < 8) {
        d.v[i] = s & modinv30_detail::M30;
        c = s >> 30;
      } else {
        d.v[i] = s;
      }
    }
    const bool take = d.v[8] >= 0;
#pragma unroll
    for (int i = 0; i < 9; i++) x->v[i] = take ? d.v[i] : x->v[i];
  }
}

// (d, e) <- t (d, e) / 2^30 mod p, inputs and outputs in [0, p); ninv30 = -p^-1 mod 2^30
RLN_MI_HD void update_de30(S30* d, S30* e, const Trans30& t, const S30& p, uint32_t ninv30) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  int64_t cd = u * d->v[0] + v * e->v[0], ce = q * d->v[0] + r * e->v[0];
  // multiples of p that clear the low 30 bits
  const int64_t kd = (int64_t)(((uint32_t)cd * ninv30) & (uint32_t)modinv30_detail::M30);
  const int64_t ke = (int64_t)(((uint32_t)ce * ninv30) & (uint32_t)modinv30_detail::M30);
  cd += kd * p.v[0];
  ce += ke * p.v[0];
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    const int64_t di = d->v[i], ei = e->v[i], pi = p.v[i];
    cd += u * di + v * ei + kd * pi;
    ce += q * di + r * ei + ke * pi;
    d->v[i - 1] = (int32_t)((uint32_t)cd & (uint32_t)modinv30_detail::M30);
    e->v[i - 1] = (int32_t)((uint32_t)ce & (uint32_t)modinv30_detail::M30);
    cd >>= 30;
    ce >>= 30;
  }
  d->v[8] = (int32_t)cd;
  e->v[8] = (int32_t)ce;
  normalize30(d, p);
  normalize30(e, p);
}

// x^-1 mod p for 0 < x < p (8 x 32 words in, 8 x 32 words out); p odd, below 2^255; ninv32 = -p^-1 mod 2^32
RLN_MI_HD void modinv30(const uint32_t* x, const uint32_t* pw, uint32_t ninv32, uint32_t* out) {
  const S30 p = s30_from_words(pw);
  S30 f = p, g = s30_from_words(x), d, e;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    d.v[i] = 0;
    e.v[i] = i == 0 ? 1 : 0;
  }
  const uint32_t ninv30 = ninv32 & (uint32_t)modinv30_detail::M30;
  int32_t delta = 1;
  // 741 steps bound the sequence: 25 batches; it stops as soon as g = 0 (typically after 17 - 19)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int it = 0; it < 25; it++) {
    Trans30 t;
    // the low 32 bits of f and g as two's complement words (limb 1 contributes its low two bits)
    const uint32_t f0 = (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), g0 = (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30);
    delta = divsteps30(delta, f0, g0, &t);
    update_fg30(&f, &g, t);
    update_de30(&d, &e, t, p, ninv30);
    int32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) nz |= g.v[i];
    if (nz == 0) break;
  }
  // f = +-1: x^-1 = +-d
  if (f.v[8] < 0) {
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int32_t s = p.v[i] - d.v[i] + c;
      if (i < 8) {
        d.v[i] = s & modinv30_detail::M30;
        c = s >> 30;
      } else {
        d.v[i] = s;
      }
    }
    // (d = 0 cannot occur for x != 0: p - 0 = p would not be reduced)
  }
  s30_to_words(d, out);
}

}  // namespace rlnamd
