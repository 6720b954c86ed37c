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
#include "witness_ops.h"

namespace rlnamd {

constexpr uint32_t WL_LDS_BYTES = WL_SLOTS * 48 + 64;

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
__device__ __noinline__ WlOut wl_misc(uint32_t dx, uint32_t dy, uint32_t dz, const uint32_t* lds,
                                      const uint32_t* __restrict__ inputs, uint32_t n_inputs, uint32_t p) {
  const uint32_t lop = dx & 0xFF, gop = (dx >> 16) & 0xFF;
  WlOut o;
  o.e = WERR_NONE;
  uint32_t* const e = &o.e;
  Fr29 v = Fr29::zero();
  if (lop == WO_INPUT) {
    const uint32_t* in = inputs + ((size_t)p * n_inputs + (dy >> 16)) * 8;   // the `a` field holds the input index
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
//     T_c = sum_k a_k b_(c-k)          b_(c-k) = the neighbour's limb, fetched with v_mov_dpp row_shr:k
//     T_(c+9) = sum_k a_k b_(c+9-k)    row_shl:(9-k);  a's limbs are row-uniform (every lane reads all nine from LDS)
// The reduction is the separated (not interleaved) Montgomery form, because an interleaved round would need lane 0's
// digit broadcast to its row nine times in a dependent chain:  n = T mod 2^261 as 29-bit limbs (one carry pass across the
// lanes), m = n p' mod 2^261, U = m p, result = (T + U) / 2^261 + c.  The low 261 bits of T + U are an exact multiple of
// 2^261; its quotient K (the carry into column 9) is read off the top three low columns:  W = S_8 + (S_7 >> 29) +
// (S_6 >> 58) misses less than 3 units of the exact 2^29 K, so K = (W + 16) >> 29.  ~165 instructions per step.
// Limbs: inputs "near-normalised" (< 2^29 + 4 below the top limb), columns < 9 (2^29.01)^2 + 9 2^29 2^30.1 < 2^63.3.
// Values: m < 2^262 (lazy top limb), so the result is below a b / 2^261 + 2 r + c  -- the host's bounds use 2.4 r.
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
#define WL_MACS(lo, hi, cst, v)                                                                    \
  lo = (uint64_t)cst[0] * v;                                                                        \
  hi = 0;                                                                                           \
  lo += (uint64_t)cst[1] * dppz<WL_SHR(1)>(v); hi += (uint64_t)cst[1] * dppz<WL_SHL(8)>(v);         \
  lo += (uint64_t)cst[2] * dppz<WL_SHR(2)>(v); hi += (uint64_t)cst[2] * dppz<WL_SHL(7)>(v);         \
  lo += (uint64_t)cst[3] * dppz<WL_SHR(3)>(v); hi += (uint64_t)cst[3] * dppz<WL_SHL(6)>(v);         \
  lo += (uint64_t)cst[4] * dppz<WL_SHR(4)>(v); hi += (uint64_t)cst[4] * dppz<WL_SHL(5)>(v);         \
  lo += (uint64_t)cst[5] * dppz<WL_SHR(5)>(v); hi += (uint64_t)cst[5] * dppz<WL_SHL(4)>(v);         \
  lo += (uint64_t)cst[6] * dppz<WL_SHR(6)>(v); hi += (uint64_t)cst[6] * dppz<WL_SHL(3)>(v);         \
  lo += (uint64_t)cst[7] * dppz<WL_SHR(7)>(v); hi += (uint64_t)cst[7] * dppz<WL_SHL(2)>(v);         \
  lo += (uint64_t)cst[8] * dppz<WL_SHR(8)>(v); hi += (uint64_t)cst[8] * dppz<WL_SHL(1)>(v);
// a: the nine limbs of the first factor (row-uniform); b, c: this lane's limb of the second factor and of the addend
// (0 in lanes 9..15 of the row); j = lane & 15.  Returns this lane's limb of a b / 2^261 + c (valid in lanes 0..8).
__device__ __forceinline__ uint32_t wl_row_mul_add(const uint32_t (&a)[9], uint32_t b, uint32_t c, uint32_t j) {
  constexpr uint32_t M = (1u << 29) - 1;
  uint64_t tlo, thi;
  WL_MACS(tlo, thi, a, b)
  const uint32_t n = wl_carry3(tlo);                      // T mod 2^261, limbs < 2^30 + 2^6 (lanes 0..8)
  uint64_t u = (uint64_t)WL_PINV[0] * n;
  u += (uint64_t)WL_PINV[1] * dppz<WL_SHR(1)>(n);
  u += (uint64_t)WL_PINV[2] * dppz<WL_SHR(2)>(n);
  u += (uint64_t)WL_PINV[3] * dppz<WL_SHR(3)>(n);
  u += (uint64_t)WL_PINV[4] * dppz<WL_SHR(4)>(n);
  u += (uint64_t)WL_PINV[5] * dppz<WL_SHR(5)>(n);
  u += (uint64_t)WL_PINV[6] * dppz<WL_SHR(6)>(n);
  u += (uint64_t)WL_PINV[7] * dppz<WL_SHR(7)>(n);
  u += (uint64_t)WL_PINV[8] * dppz<WL_SHR(8)>(n);
  uint32_t m = wl_carry3(u);                              // n p' mod 2^261 (what leaves lane 8 is dropped)
  m = j < 9 ? m : 0;
  uint64_t ulo, uhi;
  WL_MACS(ulo, uhi, Fr29C::P, m)
  const uint64_t slo = tlo + ulo;
  uint64_t shi = thi + uhi;
  // carry of the low half: exact in lane 8, moved to lane 0 of the high half
  const uint64_t t1 = slo >> 29;
  const uint32_t t2 = (uint32_t)(slo >> 58);
  const uint64_t t1n = (uint64_t)dppz<WL_SHR(1)>((uint32_t)t1) | ((uint64_t)dppz<WL_SHR(1)>((uint32_t)(t1 >> 32)) << 32);
  const uint64_t w = slo + t1n + dppz<WL_SHR(2)>(t2);
  const uint64_t K = (w + 16) >> 29;
  const uint64_t K0 = (uint64_t)dppz<WL_SHL(8)>((uint32_t)K) | ((uint64_t)dppz<WL_SHL(8)>((uint32_t)(K >> 32)) << 32);
  shi += (j == 0 ? K0 : 0) + c;
  // two carry passes: limbs 0..7 below 2^29 + 2, limb 8 keeps the rest
  const uint32_t r1 = wl_carry3(shi);
  const uint32_t keep = j < 8 ? (r1 & M) : r1, carry = j < 8 ? (r1 >> 29) : 0;
  return keep + dppz<WL_SHR(1)>(carry);
}

__global__ void __launch_bounds__(64) k_witness_lanes(const uint4* __restrict__ prog, uint32_t nsteps,
                                                      const uint32_t* __restrict__ consts29, uint32_t n_consts,
                                                      const uint32_t* __restrict__ inputs, uint32_t n_inputs,
                                                      uint4* __restrict__ V29, uint32_t* __restrict__ err, uint32_t B) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __builtin_amdgcn_s_setprio(3);
  const uint32_t lane = threadIdx.x, p = blockIdx.x;
  // constants, then ZERO, ONE, MINUS_ONE; the last slot is the write target of idle lanes
  for (uint32_t i = lane; i < n_consts * 9; i += 64) lds[(i / 9) * 12 + i % 9] = consts29[i];
  if (lane == 0) {
    wl_write(lds, n_consts, Fr29::zero());
    const Fr29 one = Fr29::from_const(Fr29C::ONE);
    wl_write(lds, n_consts + 1, one);
    wl_write(lds, n_consts + 2, Fr29::mul(Fr29::neg_lazy(Fr29C::K8, one), one));   // -1, reduced
    lds[WL_SLOTS * 12] = 0;   // error word
  }
  __syncthreads();
  uint32_t e = WERR_NONE;
  uint4 d[WL_PF];
#pragma unroll
  for (int k = 0; k < (int)WL_PF; k++) d[k] = prog[(size_t)k * WL_W + lane];
#pragma unroll 1
  for (uint32_t t0 = 0; t0 < nsteps; t0 += WL_PF) {
#pragma unroll
    for (int k = 0; k < (int)WL_PF; k++) {
      const uint4 q = d[k];
      d[k] = prog[(size_t)(t0 + WL_PF + k) * WL_W + lane];   // the program is padded by WL_PF empty steps
      const uint32_t kind = (__builtin_amdgcn_readfirstlane(q.x) >> 12) & 7;
      const uint32_t dst = q.y & 0xFFFF, sa = q.y >> 16, sb = q.z & 0xFFFF, sc = q.z >> 16;
      if (__builtin_expect(kind == WK_ROW, 1)) {
        // one product per 16-lane row; the row's descriptor is replicated over its lanes
        const uint32_t j = lane & 15, jj = j < 9 ? j : 0;
        Fr29 va;
        wl_read(va, lds, sa);
        uint32_t vb = lds[sb * 12 + jj], vc = lds[sc * 12 + jj];
        vb = j < 9 ? vb : 0;
        vc = j < 9 ? vc : 0;
        const uint32_t r = wl_row_mul_add(va.v, vb, vc, j);
        if (j < 9) {
          lds[dst * 12 + j] = r;
          if (q.x & WL_STORE) ((uint32_t*)(V29 + ((size_t)q.w * B + p) * 3))[j] = r;
        }
        continue;
      }
      Fr29 v;
      // FMA steps fall through (a uniform branch hop costs a lone wave 25 - 70 cycles)
      if (__builtin_expect(kind == WK_FMA, 1)) {
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
        const WlOut o = wl_misc(q.x, q.y, q.z, lds, inputs, n_inputs, p);
        v = o.v;
        if (o.e && !e) e = o.e;
      }
      wl_write(lds, dst, v);
      // (storing unconditionally into a trash row to save this branch was measured: 5.0 -> 5.4 ms; three 16-byte
      // stores per lane and step cost a lone wave more than the skipped branch)
      if (q.x & WL_STORE) wl_store(V29, q.w, B, p, v);
    }
  }
  if (e) atomicOr(&lds[WL_SLOTS * 12], e);
  __builtin_amdgcn_wave_barrier();   // one wave: LDS operations complete in program order
  if (lane == 0) err[p] = lds[WL_SLOTS * 12];
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
  RLN_HIP(hipStreamSynchronize(s));
  if (hipFuncSetAttribute((const void*)k_witness_lanes, hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES) !=
      hipSuccess) {   // a device with less LDS per workgroup: keep k_witness29
    (void)hipGetLastError();
    return;
  }
  ok = true;
  const char* info = getenv("RLNAMD_WITLANES_INFO");
  if (info && info[0] == '1')
    fprintf(stderr, "witness lanes: %u steps (%u row, %u fma, %u sqr, %u add, %u misc), peak %u live values, %u constants\n",
            nsteps, nrow, nfma, nsqr, nadd, nmisc, peak_slots, n_consts);
}

void WitLanes::launch(hipStream_t s, const uint32_t* d_consts29, const uint32_t* d_inputs, uint32_t n_inputs, uint4* V29,
                      uint32_t* err, uint32_t B, uint32_t nb) const {
  hipLaunchKernelGGL(k_witness_lanes, dim3(nb), dim3(64), WL_LDS_BYTES, s, prog.p, nsteps, d_consts29, n_consts, d_inputs,
                     n_inputs, V29, err, B);
}

}  // namespace rlnamd
