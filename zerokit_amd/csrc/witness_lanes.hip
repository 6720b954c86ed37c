// See witness_lanes.h.  Device program: steps of WL_W micro-ops (one per lane, 16 bytes each), three step kinds:
//   FMA   every lane computes a * b + c in the 9 x 29 form (Fr29::mul_add): a Mul is a * b + ZERO, an Add that shares
//         the step with a product is x * ONE + y, a Sub a + b * MINUS_ONE -- no divergence inside a step;
//   ADD   every lane computes a + b (only when no product is ready: a fifth of the time of an FMA step);
//   MISC  inputs (canonical -> Montgomery) and the rare operations (comparisons, shifts, bit operations, division,
//         TernCond ...: graph.rs:72-143, 314-466), on canonical integers as in the other interpreters.
// Bounds: values are kept below WL_BMAX r (fq29.h: products take operands up to 10 r, K8 - x up to 7.9 r).  A product
// reduces its multiplicands (result < r + a b / 2^261 + c), so an Add riding in an FMA step takes the operand with the
// larger bound as multiplicand; when a result would still pass WL_BMAX it is followed by a reduction x * ONE + ZERO.
#include "witness_lanes.h"
#include "witness_sched.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fq29.h"
#include "prover_kernels.h"
#include "witness_ops.h"

namespace rlnamd {

constexpr uint32_t WL_STAGE = WL_SLOTS * 12 + 16;                       // word offset of the store staging area
constexpr uint32_t WL_LDS_BYTES = WL_SLOTS * 48 + 64 + WL_PF * WL_ROWS * 48;   // slots, error word, staging
static_assert(WL_PF * WL_ROWS == WL_W, "a staging entry per lane");

__device__ __forceinline__ void wl_read(Fr29& r, const uint32_t* lds, uint32_t slot) {
  const char* a = (const char*)lds + slot * 48;
  const uint4 x = *(const uint4*)a, y = *(const uint4*)(a + 16);
  r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w;
  r.v[4] = y.x; r.v[5] = y.y; r.v[6] = y.z; r.v[7] = y.w;
  r.v[8] = *(const uint32_t*)(a + 32);
}
__device__ __forceinline__ void wl_write(uint32_t* lds, uint32_t slot, const Fr29& v) {
  char* a = (char*)lds + slot * 48;
  *(uint4*)a = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
  *(uint4*)(a + 16) = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
  *(uint32_t*)(a + 32) = v.v[8];
}
__device__ __forceinline__ void wl_store(uint4* __restrict__ V29, uint32_t slot, uint32_t B, uint32_t p, const Fr29& v) {
  uint4* g = V29 + ((size_t)slot * B + p) * 3;
  g[0] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
  g[1] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
  g[2] = make_uint4(v.v[8], 0, 0, 0);
}

// inputs and rare operations of a MISC step; out of line: the hot loop must not inherit its register pressure
struct WlOut {
  Fr29 v;
  uint32_t e;
};
struct WlIn {   // where a step's inputs come from: the proof's inputs, and behind them (index >= n_inputs) its hints
  const uint32_t* inputs;
  const uint32_t* hints;
  uint32_t n_inputs, n_hints;
};
__device__ __noinline__ WlOut wl_misc(uint32_t dx, uint32_t dy, uint32_t dz, const uint32_t* lds, const WlIn& I, uint32_t p) {
  const uint32_t lop = dx & 0xFF, gop = (dx >> 16) & 0xFF;
  WlOut o;
  o.e = WERR_NONE;
  uint32_t* const e = &o.e;
  Fr29 v = Fr29::zero();
  if (lop == WO_INPUT) {
    const uint32_t idx = dy >> 16;   // the `a` field holds the input index
    const uint32_t* in = idx < I.n_inputs ? I.inputs + ((size_t)p * I.n_inputs + idx) * 8
                                          : I.hints + ((size_t)p * I.n_hints + (idx - I.n_inputs)) * 8;
    if (limbs_geq(in, FrParams::MOD)) *e = WERR_INPUT_RANGE;  // u256_to_fr fails (graph.rs:42-45)
    Fr x;
#pragma unroll
    for (int k = 0; k < 8; k++) x.v[k] = in[k];
    v = Fr29::mul(Fr29::slice(x), Fr29::from_const(Fr29C::FROM_CANON));
  } else if (lop == WO_RARE) {
    Fr29 va, vb, vc;
    wl_read(va, lds, dy >> 16);
    wl_read(vb, lds, dz & 0xFFFF);
    wl_read(vc, lds, dz >> 16);
    va.normalize();   // row-form results carry limbs up to 2^29 + 2: the exact zero test compares limb patterns
    vb.normalize();
    vc.normalize();
    if (gop == G_TERN) {
      const bool z = va.is_zero_mod_q();  // graph.rs:214-224
#pragma unroll
      for (int k = 0; k < 9; k++) v.v[k] = z ? vc.v[k] : vb.v[k];
    } else if (gop == G_ID) {
      (void)witness_slow_op(G_ID, Fr::zero(), Fr::zero(), e);
    } else {  // comparisons, shifts, bit operations, division ...: on canonical integers, in the 8 x 32 form
      v = Fr29::from_fq(witness_slow_op(gop, va.to_fq(), vb.to_fq(), e));
    }
  }
  o.v = v;
  return o;
}


// ---------------------------------------------------------------------------------------------------------------------
// Row form of a * b + c (WK_ROW).  A lane-form step spends ~290 instructions of ONE wave on products that occupy two of
// its 64 lanes on average.  Here a value's nine 29-bit limbs sit in lanes 0..8 of a 16-lane DPP row (a wave = four
// products), and every lane computes ONE column of each partial product:
//     T_c = sum_k a_k b_(c-k)          b_(c-k) = the neighbour's limb, fetched with v_mov_dpp row_shr:k;
//                                      a's limbs are row-uniform (every lane reads all nine from LDS)
// The reduction is the separated (not interleaved) Montgomery form, because an interleaved round would need lane 0's
// digit broadcast to its row nine times in a dependent chain:  n = T mod 2^261 as 29-bit limbs (one carry pass across the
// lanes), m = n p' mod 2^261, U = m p, result = (T + U) / 2^261 + c.  The low 261 bits of T + U are an exact multiple of
// 2^261; its quotient K (the carry into column 9) is read off the top three low columns:  W = S_8 + (S_7 >> 29) +
// (S_6 >> 58) misses less than 3 units of the exact 2^29 K, so K = (W + 16) >> 29.
// Limbs: inputs "near-normalised" (< 2^29 + 4 below the top limb), columns < 9 (2^29.01)^2 + 9 2^29 2^30.1 < 2^63.3.
// Values: m is taken mod 2^261 (top limb masked; the limbs below stay near-normalised: m < 2^261 (1 + 2^-27)), so the
// result is below a b / 2^261 + 1.0001 r + c  -- the host's bounds use 1.06 r.  (Until round 5 the top limb stayed lazy:
// m < 2^262, 2.05 r, and every third partial round of Poseidon paid a reduction step for it.)
template <int CTRL>
__device__ __forceinline__ uint32_t dppz(uint32_t v) {   // the neighbour's value; 0 where the source lane is outside the row
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
#define WL_SHR(k) (0x110 + (k))   // row_shr:k  lane i reads lane i - k
#define WL_SHL(k) (0x100 + (k))   // row_shl:k  lane i reads lane i + k
constexpr uint32_t WL_PINV[9] = {0x0fffffffu, 0x170fac9fu, 0x1a446cf0u, 0x0d0c9698u, 0x02391658u,
                                 0x0c144c83u, 0x06cb8e6au, 0x03a1b068u, 0x1273f82fu};   // -r^-1 mod 2^261, 29-bit limbs
// a 64-bit column -> its three 29-bit pieces, each delivered to the lane it belongs to
__device__ __forceinline__ uint32_t wl_carry3(uint64_t x) {
  constexpr uint32_t M = (1u << 29) - 1;
  const uint32_t l = (uint32_t)x & M, mid = (uint32_t)(x >> 29) & M, h = (uint32_t)(x >> 58);
  return l + dppz<WL_SHR(1)>(mid) + dppz<WL_SHR(2)>(h);
}
// The 17 columns are spread over the row's SIXTEEN lanes: b is zero in lanes 9..15, so the row_shr fetches alone put
// column c = 0..15 into lane c (lane c >= 9 sees b_(c-k) exactly where c - k <= 8); only column 16 = a_8 b_8 needs a
// fetch of its own (lane 0, row_shl:8).  29 multiply-adds and 26 fetches (a second accumulator for the high columns in
// lanes 0..8, the first version: 45 and 40 -- the same latency for the lone wave, more instructions issued).  The
// result's limb i = column 9 + i: limbs 0..6 leave in lanes 9..15, limbs 7 and 8 (column 16 and the carries above it)
// in lane 0, which closes the carry chain as a ring (row_ror).
#define WL_ROR(k) (0x120 + (k))   // row_ror:k  lane i reads lane (i - k) mod 16
// (two interleaved accumulators: a lone wave would otherwise wait out the latency of every multiply-add of the chain)
#define WL_MACS16(acc, top, cst, v)                                                                 \
  {                                                                                                 \
    uint64_t e_ = (uint64_t)cst[0] * v;                                                             \
    uint64_t o_ = (uint64_t)cst[1] * dppz<WL_SHR(1)>(v);                                            \
    e_ += (uint64_t)cst[2] * dppz<WL_SHR(2)>(v);                                                    \
    o_ += (uint64_t)cst[3] * dppz<WL_SHR(3)>(v);                                                    \
    e_ += (uint64_t)cst[4] * dppz<WL_SHR(4)>(v);                                                    \
    o_ += (uint64_t)cst[5] * dppz<WL_SHR(5)>(v);                                                    \
    e_ += (uint64_t)cst[6] * dppz<WL_SHR(6)>(v);                                                    \
    o_ += (uint64_t)cst[7] * dppz<WL_SHR(7)>(v);                                                    \
    e_ += (uint64_t)cst[8] * dppz<WL_SHR(8)>(v);                                                    \
    top = (uint64_t)cst[8] * dppz<WL_SHL(8)>(v);                                                    \
    acc = e_ + o_;                                                                                  \
  }
// a: the first factor's nine limbs (row-uniform); b: this lane's limb of the second factor (0 in lanes 9..15);
// cj: the addend's limb j - 9 in lanes 9..15 and limb 7 in lane 0; c8: the addend's limb 8 (read by lane 0).
// Returns limb j - 9 of a b / 2^261 + c in lanes 9..15, limb 7 in lane 0 and limb 8 in lane 1.
// PI, PP: -r^-1 mod 2^261 and r as 29-bit limbs, held in VECTOR registers by the caller (as literals they cost a scalar
// move each per use: 18 of the step's ~31 scalar instructions, and the lone wave pays for every instruction it issues).
__device__ __forceinline__ uint32_t wl_row_mul_add16(const uint32_t (&a)[9], uint32_t b, uint32_t cj, uint32_t c8,
                                                     uint32_t j, const uint32_t (&PI)[9], const uint32_t (&PP)[9]) {
  constexpr uint32_t M = (1u << 29) - 1;
  uint64_t t, t16;
  WL_MACS16(t, t16, a, b)
  const uint32_t n = wl_carry3(t);                        // lanes 0..8: T mod 2^261 (lanes above: not read)
  uint64_t u = (uint64_t)PI[0] * n;
  uint64_t u1 = (uint64_t)PI[1] * dppz<WL_SHR(1)>(n);
  u += (uint64_t)PI[2] * dppz<WL_SHR(2)>(n);
  u1 += (uint64_t)PI[3] * dppz<WL_SHR(3)>(n);
  u += (uint64_t)PI[4] * dppz<WL_SHR(4)>(n);
  u1 += (uint64_t)PI[5] * dppz<WL_SHR(5)>(n);
  u += (uint64_t)PI[6] * dppz<WL_SHR(6)>(n);
  u1 += (uint64_t)PI[7] * dppz<WL_SHR(7)>(n);
  u += (uint64_t)PI[8] * dppz<WL_SHR(8)>(n);
  u += u1;
  uint32_t m = wl_carry3(u);
  m = j < 8 ? m : (j == 8 ? (m & M) : 0);   // mod 2^261: the top limb's lazy bits are multiples of 2^261 (m < 2^261 (1 + 2^-27))
  uint64_t U, U16;
  WL_MACS16(U, U16, PP, m)
  const uint64_t s = t + U;                               // column j
  uint64_t s16 = t16 + U16;                               // lane 0: column 16
  asm volatile("" : "+v"(s16));   // computed by every lane: otherwise its two products sink into a lane-0-only branch
  // carry of the low half: exact in lane 8, handed to column 9 next door
  const uint64_t t1 = s >> 29;
  const uint32_t t2 = (uint32_t)(s >> 58);
  const uint64_t t1n = (uint64_t)dppz<WL_SHR(1)>((uint32_t)t1) | ((uint64_t)dppz<WL_SHR(1)>((uint32_t)(t1 >> 32)) << 32);
  const uint64_t w = s + t1n + dppz<WL_SHR(2)>(t2);
  const uint64_t K = (w + 16) >> 29;
  uint64_t K9 = (uint64_t)dppz<WL_SHR(1)>((uint32_t)K) | ((uint64_t)dppz<WL_SHR(1)>((uint32_t)(K >> 32)) << 32);
  asm volatile("" : "+v"(K9));   // (as for top1 below: fetched with the whole row active)
  // the ring of the high columns: lanes 9..15 = columns 9..15, lane 0 = column 16, lanes 1..8 hold zero
  uint64_t y = s + (j == 9 ? K9 : 0);
  y = j >= 9 ? y : (j == 0 ? s16 : 0);
  y += cj;
  const uint32_t l = (uint32_t)y & M, mid = (uint32_t)(y >> 29) & M, h = (uint32_t)(y >> 58);
  const uint32_t h1 = dppz<WL_ROR(1)>(h);                 // lane 0: column 15's top piece, which belongs to column 17
  const uint32_t r1 = l + dppz<WL_ROR(1)>(mid) + dppz<WL_ROR(2)>(h);
  const uint32_t keep = r1 & M, carry = r1 >> 29;
  const uint32_t r = keep + dppz<WL_ROR(1)>(carry);       // lane 9 receives lane 8's zero
  const uint32_t top = (uint32_t)(y >> 29) + h1 + c8 + carry;   // lane 0: column 17 and whatever is above it
  uint32_t top1 = dppz<WL_SHR(1)>(top);                         // limb 8 leaves in lane 1
  // the fetch must run with the whole row active: without the pin the compiler sinks it into the j == 1 side of the
  // select below (a branch), where lane 0 -- its source -- is switched off and the fetch returns 0
  asm volatile("" : "+v"(top1));
  return j == 1 ? top1 : r;
}

// The steps that are not row-form products (lane-form products when RLNAMD_WITROWS=0, plain additions, inputs and
// the rare operations): out of line, so that the loop below is one compare and one branch away from its product.
__device__ __noinline__ uint32_t wl_other_step(uint32_t kind, uint4 q, uint32_t* lds, const WlIn& I,
                                               uint4* __restrict__ V29, uint32_t B, uint32_t p) {
  const uint32_t dst = q.y & 0xFFFF, sa = q.y >> 16, sb = q.z & 0xFFFF, sc = q.z >> 16;
  uint32_t e = WERR_NONE;
  Fr29 v;
  if (kind == WK_FMA) {
    Fr29 va, vb, vc;
    wl_read(va, lds, sa);
    wl_read(vb, lds, sb);
    wl_read(vc, lds, sc);
    v = Fr29::mul_add(va, vb, vc);
  } else if (kind == WK_SQR) {   // 45 products instead of 81: two of the three products of Poseidon's x^5
    Fr29 va, vc;
    wl_read(va, lds, sa);
    wl_read(vc, lds, sc);
    v = Fr29::sqr_add(va, &vc);
  } else if (kind == WK_ADD) {
    Fr29 va, vb;
    wl_read(va, lds, sa);
    wl_read(vb, lds, sb);
#pragma unroll
    for (int j = 0; j < 9; j++) v.v[j] = va.v[j] + vb.v[j];
    v.normalize();
  } else {
    const WlOut o = wl_misc(q.x, q.y, q.z, lds, I, p);
    v = o.v;
    e = o.e;
  }
  wl_write(lds, dst, v);
  if (q.x & WL_STORE) wl_store(V29, q.w, B, p, v);
  return e;
}

// The wave is alone on its SIMD and every step reads what the step before wrote, so a step costs its latencies, not its
// instruction count (PMC, one wave, 6 122 steps: 1 355 cycles per step = 580 VALU issue + 160 scalar + 55 branch +
// 510 waiting at s_waitcnt).  The waiting was vmcnt(0) once per group of WL_PF steps: the prefetched descriptors were
// loaded into fresh registers while the step still read the old ones, and copied at the end of the group -- which
// drains every outstanding store (1 - 2 us each to HBM) of the steps just executed (vmcnt counts loads and stores in
// order).  Hence
//  * the descriptors of group g + 1 are all loaded at the START of group g (a second bank of registers): whatever the
//    compiler's waits at the group boundary drain is a whole group (~6 us) old;
//  * a row step does not store to V29: it leaves its result and the V29 slot in a staging entry in LDS, and the 64
//    entries of a group are flushed together at the start of the next group, a lane per entry, three 16-byte stores.
__device__ __forceinline__ void wl_flush(uint32_t* lds, uint4* __restrict__ V29, uint32_t B, uint32_t p, uint32_t lane) {
  uint32_t* se = lds + WL_STAGE + lane * 12;
  const uint4 a = *(const uint4*)se, b = *(const uint4*)(se + 4);
  const uint32_t v8 = se[8], slot = se[9];
  if (slot != 0xFFFFFFFFu) {
    uint4* g = V29 + ((size_t)slot * B + p) * 3;
    g[0] = a;
    g[1] = b;
    g[2] = make_uint4(v8, 0, 0, 0);
    se[9] = 0xFFFFFFFFu;
  }
}

__global__ void __launch_bounds__(64) k_witness_lanes(const uint4* __restrict__ prog, uint32_t nsteps,
                                                      const uint32_t* __restrict__ consts29, uint32_t n_consts,
                                                      const uint32_t* __restrict__ inputs, uint32_t n_inputs,
                                                      uint4* __restrict__ V29, uint32_t* __restrict__ err, uint32_t B,
                                                      const WlSegDesc* __restrict__ segs, const uint32_t* __restrict__ hints,
                                                      uint32_t n_hints) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __builtin_amdgcn_s_setprio(3);
  const uint32_t lane = threadIdx.x, p = blockIdx.x;
  if (segs) {   // one of the graph's independent segments (grid.y), a program of its own
    const WlSegDesc d0 = segs[blockIdx.y];
    prog += d0.prog_off;
    nsteps = d0.nsteps;
    consts29 += d0.const_off;
    n_consts = d0.n_consts;
  }
  const WlIn I{inputs, hints, n_inputs, n_hints};
  // constants, then ZERO, ONE, MINUS_ONE; the last slot is the write target of idle lanes
  for (uint32_t i = lane; i < n_consts * 9; i += 64) lds[(i / 9) * 12 + i % 9] = consts29[i];
  if (lane == 0) {
    wl_write(lds, n_consts, Fr29::zero());
    const Fr29 one = Fr29::from_const(Fr29C::ONE);
    wl_write(lds, n_consts + 1, one);
    wl_write(lds, n_consts + 2, Fr29::mul(Fr29::neg_lazy(Fr29C::K8, one), one));   // -1, reduced
    lds[WL_SLOTS * 12] = 0;   // error word
  }
  lds[WL_STAGE + lane * 12 + 9] = 0xFFFFFFFFu;   // staging entries: nothing to store
  __syncthreads();
  uint32_t e = WERR_NONE;
  uint4 d[WL_PF];
#pragma unroll
  for (int k = 0; k < (int)WL_PF; k++) d[k] = prog[(size_t)k * WL_W + lane];
  // a row's lanes: 0..8 read the second factor's limbs; the result's limbs 0..6 leave in lanes 9..15, limbs 7 and 8 in
  // lanes 0 and 1; lane 2 carries the V29 slot into the staging entry (word 9: padding in the value slots)
  const uint32_t j = lane & 15, row = lane >> 4;
  const uint32_t jb = j < 9 ? j : 0, jc = j >= 9 ? j - 9 : 7 + j;
  const bool writer = j >= 9 || j < 3;
  const uint32_t wj = writer ? jc : 10 + (j & 1);   // idle lanes: padding words 10, 11
  uint32_t PI[9], PP[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    PI[k] = WL_PINV[k];
    PP[k] = Fr29C::P[k];
    asm volatile("" : "+v"(PI[k]), "+v"(PP[k]));
  }
#pragma unroll 1
  for (uint32_t t0 = 0; t0 < nsteps; t0 += WL_PF) {
    wl_flush(lds, V29, B, p, lane);
    uint4 dn[WL_PF];   // the next group's descriptors, all issued here: a group old when the group ends
#pragma unroll
    for (int k = 0; k < (int)WL_PF; k++) dn[k] = prog[(size_t)(t0 + WL_PF + k) * WL_W + lane];   // (padded by 2 WL_PF steps)
    // one product per 16-lane row; the row's descriptor is replicated over its lanes
    auto row_step = [&](const uint4& q, int k) {
      const uint32_t dst = q.y & 0xFFFF, sa = q.y >> 16, sb = q.z & 0xFFFF, sc = q.z >> 16;
      Fr29 va;
      wl_read(va, lds, sa);
      uint32_t vb = lds[sb * 12 + jb], vc = lds[sc * 12 + (j >= 9 ? jc : 7)];
      const uint32_t c8 = lds[sc * 12 + 8];
      __builtin_amdgcn_sched_barrier(0);   // all six reads in flight before the first wait
      vb = j < 9 ? vb : 0;
      vc = (j >= 9 || j == 0) ? vc : 0;
      uint32_t r = wl_row_mul_add16(va.v, vb, vc, c8, j, PI, PP);
      r = j == 2 ? ((q.x & WL_STORE) ? q.w : 0xFFFFFFFFu) : r;
      // every lane writes (no exec-mask branch): the idle lanes 3..8 into the padding words of the dummy slot / of
      // their staging entry
      lds[(writer ? dst : WL_SLOTS - 1) * 12 + wj] = r;
      lds[WL_STAGE + (k * WL_ROWS + row) * 12 + wj] = r;
    };
    // nine groups in ten hold nothing but row steps (the host marks them): no kind test, no branch between their steps,
    // and the LDS addresses of step k + 1 are computed while step k's operand reads are in flight (eight instructions
    // off the write -> read chain; the lone wave has nothing else to issue during that wait)
    struct RowAddr {
      uint32_t a, b, c, c8, w, meta;
    };
    auto row_addr = [&](const uint4& q) {
      const uint32_t dst = q.y & 0xFFFF, sa = q.y >> 16, sb = q.z & 0xFFFF, sc = q.z >> 16;
      // lanes that must see 0 (b above limb 8, c outside the result's lanes) read a word of the ZERO slot: the masking
      // happens here, in the address phase, instead of behind the reads
      return RowAddr{sa * 12, j < 9 ? sb * 12 + j : n_consts * 12, (j >= 9 || j == 0) ? sc * 12 + (j >= 9 ? jc : 7) : n_consts * 12,
                     sc * 12 + 8, (writer ? dst : WL_SLOTS - 1) * 12 + wj, (q.x & WL_STORE) ? q.w : 0xFFFFFFFFu};
    };
    if (__builtin_expect((__builtin_amdgcn_readfirstlane(d[0].x) & WL_GROUP_ROWS) != 0, 1)) {
      RowAddr A = row_addr(d[0]);
#pragma unroll
      for (int k = 0; k < (int)WL_PF; k++) {
        Fr29 va;
        {
          const uint4 x = *(const uint4*)(lds + A.a), y = *(const uint4*)(lds + A.a + 4);
          va.v[0] = x.x; va.v[1] = x.y; va.v[2] = x.z; va.v[3] = x.w;
          va.v[4] = y.x; va.v[5] = y.y; va.v[6] = y.z; va.v[7] = y.w;
          va.v[8] = lds[A.a + 8];
        }
        uint32_t vb = lds[A.b], vc = lds[A.c];
        const uint32_t c8 = lds[A.c8];
        __builtin_amdgcn_sched_barrier(0);   // all six reads in flight ...
        const RowAddr An = row_addr(d[k + 1 < (int)WL_PF ? k + 1 : k]);
        __builtin_amdgcn_sched_barrier(0);   // ... and the next step's addresses computed before the first wait
        uint32_t r = wl_row_mul_add16(va.v, vb, vc, c8, j, PI, PP);
        r = j == 2 ? A.meta : r;
        lds[A.w] = r;
        lds[WL_STAGE + (k * WL_ROWS + row) * 12 + wj] = r;
        A = An;
      }
    } else {
#pragma unroll
      for (int k = 0; k < (int)WL_PF; k++) {
        const uint4 q = d[k];
        const uint32_t kind = (__builtin_amdgcn_readfirstlane(q.x) >> 12) & 7;
        if (kind == WK_ROW) {
          row_step(q, k);
        } else {
          const uint32_t e1 = wl_other_step(kind, q, lds, I, V29, B, p);
          if (e1 && !e) e = e1;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < (int)WL_PF; k++) d[k] = dn[k];
  }
  wl_flush(lds, V29, B, p, lane);
  if (e) atomicOr(&lds[WL_SLOTS * 12], e);
  __builtin_amdgcn_wave_barrier();   // one wave: LDS operations complete in program order
  if (lane == 0) {
    if (segs) {
      if (lds[WL_SLOTS * 12]) atomicOr(&err[p], lds[WL_SLOTS * 12]);
    } else {
      err[p] = lds[WL_SLOTS * 12];
    }
  }
}

// ======================================================================================================= host
void WitLanes::build(const Graph& graph, const std::vector<uint32_t>& store_slot, uint32_t trash_slot, hipStream_t s) {
  ok = false;
  const char* off = getenv("RLNAMD_WITLANES");
  if (off && off[0] == '0') return;
  const char* r = getenv("RLNAMD_WITROWS");
  const bool rows = !(r && r[0] == '0');
  WlProgram P = wl_schedule(graph, store_slot, trash_slot, rows);
  nsteps = P.nsteps; nrow = P.nrow; nfma = P.nfma; nsqr = P.nsqr; nadd = P.nadd; nmisc = P.nmisc;
  peak_slots = P.peak_slots; n_consts = P.n_consts;
  if (!P.ok) return;
  static_assert(sizeof(WlDesc) == sizeof(uint4), "descriptor size");
  prog.alloc(P.img.size());
  prog.upload(reinterpret_cast<const uint4*>(P.img.data()), P.img.size(), s);
  {
    DevBuf<Fr> c8;
    c8.alloc(std::max<size_t>(P.consts.size(), 1));
    consts29.alloc(std::max<size_t>(P.consts.size(), 1) * 9);
    if (!P.consts.empty()) {
      c8.upload(P.consts.data(), P.consts.size(), s);
      hipLaunchKernelGGL(k_consts_to29, dim3(div_up(P.consts.size(), 256)), dim3(256), 0, s, c8.p, consts29.p,
                         (uint32_t)P.consts.size());
      RLN_HIP(hipGetLastError());
    }
    RLN_HIP(hipStreamSynchronize(s));   // c8 goes out of scope
  }
  if (hipFuncSetAttribute((const void*)k_witness_lanes, hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES) !=
      hipSuccess) {   // a device with less LDS per workgroup: keep k_witness29
    (void)hipGetLastError();
    return;
  }
  ok = true;
}

void WitLanes::launch(hipStream_t s, const uint32_t* d_inputs, uint32_t n_inputs, uint4* V29, uint32_t* err, uint32_t B,
                      uint32_t nb) const {
  hipLaunchKernelGGL(k_witness_lanes, dim3(nb), dim3(64), WL_LDS_BYTES, s, prog.p, nsteps, consts29.p, n_consts, d_inputs,
                     n_inputs, V29, err, B, (const WlSegDesc*)nullptr, (const uint32_t*)nullptr, 0u);
}

void WitSegs::build(const WlSegments& S, const std::vector<uint32_t>& store_slot_full, uint32_t trash_slot, hipStream_t s) {
  ok = false;
  const char* off = getenv("RLNAMD_WITLANES");
  if (off && off[0] == '0') return;
  const char* r = getenv("RLNAMD_WITROWS");
  const bool rows = !(r && r[0] == '0');
  std::vector<WlDesc> img;
  std::vector<Fr> consts;
  std::vector<WlSegDesc> d;
  nseg = (uint32_t)S.graphs.size();
  n_hints = S.n_hints;
  max_steps = total_steps = 0;
  for (size_t k = 0; k < S.graphs.size(); k++) {
    WlProgram P = wl_schedule(S.graphs[k], wl_segment_store_slots(S, k, store_slot_full), trash_slot, rows);
    if (!P.ok) return;
    d.push_back({(uint32_t)img.size(), P.nsteps, (uint32_t)consts.size() * 9, P.n_consts});
    img.insert(img.end(), P.img.begin(), P.img.end());
    consts.insert(consts.end(), P.consts.begin(), P.consts.end());
    max_steps = std::max(max_steps, P.nsteps);
    total_steps += P.nsteps;
  }
  if (d.empty()) return;
  prog.alloc(img.size());
  prog.upload(reinterpret_cast<const uint4*>(img.data()), img.size(), s);
  descs.alloc(d.size());
  descs.upload(d.data(), d.size(), s);
  {
    DevBuf<Fr> c8;
    c8.alloc(std::max<size_t>(consts.size(), 1));
    consts29.alloc(std::max<size_t>(consts.size(), 1) * 9);
    if (!consts.empty()) {
      c8.upload(consts.data(), consts.size(), s);
      hipLaunchKernelGGL(k_consts_to29, dim3(div_up(consts.size(), 256)), dim3(256), 0, s, c8.p, consts29.p, (uint32_t)consts.size());
      RLN_HIP(hipGetLastError());
    }
    RLN_HIP(hipStreamSynchronize(s));
  }
  if (hipFuncSetAttribute((const void*)k_witness_lanes, hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  ok = true;
}

void WitSegs::launch(hipStream_t s, const uint32_t* d_inputs, uint32_t n_inputs, const uint32_t* hints, uint4* V29,
                     uint32_t* err, uint32_t B, uint32_t nb) const {
  hipLaunchKernelGGL(k_witness_lanes, dim3(nb, nseg), dim3(64), WL_LDS_BYTES, s, prog.p, 0u, consts29.p, 0u, d_inputs, n_inputs,
                     V29, err, B, descs.p, hints, n_hints);
}

}  // namespace rlnamd
