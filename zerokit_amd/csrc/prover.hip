// The 256-bit multiply is inlined (measured on MI355X: G1 MSM 44.5 -> 41.7 ms, G2 MSM 30.7 -> 22.5 ms per 1024
// proofs against the out-of-line form, which costs call overhead and a VGPR-hungry calling convention);
// -DRLN_NOINLINE_MUL restores the shared 2.5 KB body.
#include "prover.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <deque>

#include "fq29.h"
#include "walk29.h"
#include "glv.h"
#include "pairing.h"
#include "poseidon.h"
#include "witness_ops.h"
#include "witness_lanes.h"
#include "fin29.h"

namespace rlnamd {

const char* const kProverStageNames[PROVER_STAGES] = {"witness", "matvec", "ntt",      "recode",
                                                      "msm_g1",  "msm_g2", "finalize", "values"};

// =====================================================================================================
// 256-bit integer helpers on canonical limbs (witness-graph ops that are not field ops)
// =====================================================================================================
// =====================================================================================================
// 1. witness: one lane per proof interprets the straight-line graph (graph.rs:246-272)
// =====================================================================================================
// Operand encoding of the device program (built once on the host, Prover::Prover): the top two bits of a / b / c say
// where the value lives -- RING: produced at most 63 nodes earlier, read from the LDS ring; CONST: index into the
// constant table, a wave-uniform scalar load; FAR: anything else, read from the value array in HBM.  In the shipped
// circuits every operand is a constant (23 %), the previous node (33 %, forwarded in registers) or within the last 16
// nodes; only the 124 reads of input nodes go to HBM.  The ring is 64 slots x 64 lanes x 32 B = 128 KiB of LDS -- one
// wave per CU is all this kernel ever has (16 waves per 1024 proofs).
constexpr uint32_t OPK_RING = 0u << 30, OPK_CONST = 1u << 30, OPK_FAR = 2u << 30, OPK_MASK = 3u << 30;
constexpr uint32_t G_STORE = 1u << 31;  // flag on GNode.op: this node's value must reach HBM (witness signal, input, far operand)
constexpr uint32_t WIT_RING = 32;        // node values kept in LDS (64 KiB)
constexpr uint32_t WIT_LDS_CONSTS = 2048;  // constants kept in LDS (64 KiB)
__device__ __forceinline__ Fr ring_load(const uint32_t* ring, uint32_t node, uint32_t lane) {
  Fr r;
  const uint32_t* s = ring + (node % WIT_RING) * 8 * 64 + lane;
#pragma unroll
  for (int k = 0; k < 8; k++) r.v[k] = s[k * 64];
  return r;
}
__device__ __forceinline__ Fr operand_load(uint32_t enc, const uint32_t* ring, const Fr* __restrict__ consts,
                                           const Fr* __restrict__ V, uint32_t B, uint32_t p, uint32_t lane) {
  uint32_t kind = enc & OPK_MASK, id = enc & ~OPK_MASK;
  if (kind == OPK_RING) return ring_load(ring, id, lane);
  if (kind == OPK_CONST) {
    if (id >= WIT_LDS_CONSTS) return consts[id];
    Fr r;
    const uint32_t* c = ring + WIT_RING * 8 * 64 + id * 8;  // broadcast read: every lane the same address
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = c[k];
    return r;
  }
  return V[(size_t)id * B + p];
}
__global__ void __launch_bounds__(64) k_witness(const GNode* __restrict__ nodes, uint32_t n_nodes,
                                                const Fr* __restrict__ consts, uint32_t n_consts,
                                                const uint32_t* __restrict__ inputs, uint32_t n_inputs,
                                                Fr* __restrict__ V, uint32_t* __restrict__ err, uint32_t B, uint32_t nb) {
  extern __shared__ uint32_t ring[];  // [WIT_RING][8][64] node values, then [WIT_LDS_CONSTS][8] constants
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  const uint32_t lane = threadIdx.x;
  uint32_t p = blockIdx.x * 64 + lane;
  if (p >= nb) return;
  uint32_t e = WERR_NONE;
  Fr last = Fr::zero();
  {
    uint32_t* lc = ring + WIT_RING * 8 * 64;
    const uint32_t* gc = (const uint32_t*)consts;
    const uint32_t words = (n_consts < WIT_LDS_CONSTS ? n_consts : WIT_LDS_CONSTS) * 8;
    for (uint32_t i = lane; i < words; i += 64) lc[i] = gc[i];
    __syncthreads();
  }
  GNode ahead = nodes[0];
#pragma unroll 1
  for (uint32_t n = 0; n < n_nodes; n++) {
    // the descriptor of the next node is fetched while this one executes (scalar load)
    GNode nd = ahead;
    if (n + 1 < n_nodes) ahead = nodes[n + 1];
    const bool store = (nd.op & G_STORE) != 0;
    nd.op &= ~G_STORE;
    Fr v;
    if (nd.op == G_INPUT) {
      const uint32_t* src = inputs + ((size_t)p * n_inputs + nd.a) * 8;
      if (limbs_geq(src, FrParams::MOD)) e = e ? e : WERR_INPUT_RANGE;  // u256_to_fr fails (graph.rs:42-45)
      v = Fr::from_canonical(src);
    } else if (nd.op == G_CONST) {
      v = consts[nd.a];
    } else {
      // operand forwarding: chains (x^5 s-boxes, MDS sums) read the value produced one node earlier
      // (reading the NEXT node's LDS operands ahead of time was tried: 33 ms instead of 19.5 -- register pressure)
      Fr va = (nd.a == (OPK_RING | (n - 1))) ? last : operand_load(nd.a, ring, consts, V, B, p, lane);
      if (nd.op == G_NEG) {
        v = va.neg();
      } else if (nd.op == G_ID) {
        v = witness_slow_op(G_ID, va, va, &e);
      } else {
        Fr vb = (nd.b == (OPK_RING | (n - 1))) ? last : operand_load(nd.b, ring, consts, V, B, p, lane);
        if (nd.op == G_MUL)
          v = va * vb;
        else if (nd.op == G_ADD)
          v = va + vb;
        else if (nd.op == G_SUB)
          v = va - vb;
        else if (nd.op == G_TERN) {
          Fr vc = operand_load(nd.c, ring, consts, V, B, p, lane);
          v = va.is_zero() ? vc : vb;  // graph.rs:214-224
        } else {
          uint32_t e2 = 0;
          v = witness_slow_op(nd.op, va, vb, &e2);
          if (e2 && !e) e = e2;
        }
      }
    }
    // only ~6 000 of the 23 414 node values are read outside this kernel (190 MB instead of 767 MB per batch)
    if (store) V[(size_t)n * B + p] = v;
    uint32_t* slot = ring + (n % WIT_RING) * 8 * 64 + lane;
#pragma unroll
    for (int k = 0; k < 8; k++) slot[k * 64] = v.v[k];
    last = v;
  }
  err[p] = e;
}

// ---- The same interpreter with node values in the 9 x 29-bit form of fq29.h (default; RLNAMD_WIT29=0 keeps the one
// above).  One wave per SIMD is all this kernel ever has, so its time is latency, and tools/microbench_lonewave.hip
// shows what a lone wave pays on gfx950: ~12 cycles per LDS instruction issued (a 9 x ds_read_b32 operand is a 118-cycle
// round trip, two of them 225), 25 - 70 cycles per uniform branch hop, ~5.7 cycles per dependent multiply-add.  The 8 x 32
// interpreter spends two thirds of its 1 950 cycles per node on exactly that (profiled: an ADD node of 30 instructions
// takes 1 250 - 1 750 cycles).  Hence:
//   * values live in LDS as [slot][lane][12 words]: an operand is ds_read_b128 x 2 + ds_read_b32, conflict-free at
//     the 48-byte lane stride; constants as [id][12 words] are the SAME address form with lane multiplier 0, so both
//     operands of a node are read without a branch and share one round trip;
//   * the descriptor is 16 bytes (one broadcast ds_read_b128, issued one node ahead, made wave-uniform when its turn
//     comes); the program reaches LDS through coalesced vector loads, a chunk ahead (a scalar load in flight would turn
//     every LDS wait into s_waitcnt lgkmcnt(0));
//   * no register forwarding (a node waits for an LDS read anyway; the previous node's value comes back from the ring);
//     the rare sources (a value further back than the ring, a constant beyond the LDS table), the reduction flag and the
//     slow operations hide behind ONE flag test, taken before any operand is read;
//   * a product is ~200 instructions against ~375 in the 8 x 32 form, and the ~6 000 witness signals leave in the limb
//     form ([slot][proof][12 words], three 16-byte stores) for a throughput kernel to convert (k_v29_to_fr).
//
// Value discipline: every node value is normalised (limbs < 2^29) with a STATIC bound, computed on the host when the
// device program is built (Prover::Prover): products < 1 + 0.006 a b (in units of r), sums a + b, differences
// a + 8 (K8 - b, b < 7.9 r), inputs / constants / slow operations ~ 1.  A node whose bound would pass WIT29_BMAX
// carries W29_RED: its value is multiplied by the Montgomery one (result < 1.1 r).  Every operand is therefore below
// 7.5 r, inside what fq29.h's products (check_fq29_bounds.py: N(10)), K8 - b and the exact zero test (k r, k < 8) take.
constexpr uint32_t W29_STORE = 1u << 8, W29_RED = 1u << 9, W29_RARE = 1u << 10;  // flags in descriptor word 0
constexpr uint32_t W29_FMA = 25;             // program-only operation: a * b + c (an Add fused with its single-use product)
constexpr uint32_t WIT29_RING = 32;          // node values kept in LDS: 32 x 64 x 48 B = 96 KiB
constexpr uint32_t WIT29_LDS_CONSTS = 1024;  // constants kept in LDS: 48 KiB
constexpr uint32_t WIT29_CH = 256;           // descriptors per program chunk: 64 lanes x 64 B; two chunks in LDS (8 KiB)
constexpr uint32_t WIT29_LDS_BYTES = WIT29_RING * 64 * 48 + WIT29_LDS_CONSTS * 48 + 2 * WIT29_CH * 16;
constexpr double WIT29_BMAX = 7.5;
struct GNode29 {
  uint32_t w0;       // op | flags | slot << 16 (slot: index into the compact array of stored values)
  uint32_t a, b, c;  // operands as in GNode: OPK_RING | node, OPK_CONST | index, OPK_FAR | slot
};
// LDS byte address of an operand for this lane, without a branch (a uniform branch hop costs a lone wave 25 - 70
// cycles): ring value (id % RING) * 64 * 48 + lane * 48, LDS constant RING * 64 * 48 + id * 48.  Only valid for the
// operands of the fast path (ring or LDS constant); the rare path re-reads what else it needs.
__device__ __forceinline__ uint32_t wit29_addr(uint32_t enc, uint32_t lane48) {
  const uint32_t id = enc & ~OPK_MASK;
  const uint32_t cm = 0u - ((enc >> 30) & 1u);   // all ones for OPK_CONST
  const uint32_t ring_a = (id % WIT29_RING) * (64 * 48), const_a = WIT29_RING * 64 * 48 + id * 48;
  return ((const_a & cm) | (ring_a & ~cm)) + (lane48 & ~cm);
}
__device__ __forceinline__ void wit29_read(Fr29& r, uint32_t addr, const uint32_t* ring) {
  const char* a = (const char*)ring + addr;
  const uint4 x = *(const uint4*)a, y = *(const uint4*)(a + 16);
  r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w;
  r.v[4] = y.x; r.v[5] = y.y; r.v[6] = y.z; r.v[7] = y.w;
  r.v[8] = *(const uint32_t*)(a + 32);
}
struct Wit29Out {
  Fr29 v;
  uint32_t e;
};
// Everything that is not Mul / Add on ring / LDS-constant / forwarded operands.  Out of line and by value on purpose:
// inlined, its slow operations (calls with stack arguments) made the compiler keep the hot path's operands in scratch.
__device__ __noinline__ Wit29Out wit29_rare(uint32_t w0, uint32_t ea, uint32_t eb, uint32_t ec, const uint32_t* ring,
                                            uint32_t lane,
                                            const uint32_t* __restrict__ consts29, const uint32_t* __restrict__ inputs,
                                            uint32_t n_inputs, const uint4* __restrict__ V29, uint32_t B, uint32_t p) {
  Wit29Out o;
  o.e = WERR_NONE;
  const uint32_t op = w0 & 0xFF;
  auto src = [&](Fr29& r, uint32_t enc) {   // any source, from scratch
    const uint32_t kind = enc >> 30, id = enc & ~OPK_MASK;
    if (kind == (OPK_FAR >> 30)) {
      const uint4* g = V29 + ((size_t)id * B + p) * 3;
      const uint4 x = g[0], y = g[1], z = g[2];
      r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w;
      r.v[4] = y.x; r.v[5] = y.y; r.v[6] = y.z; r.v[7] = y.w;
      r.v[8] = z.x;
    } else if (kind == (OPK_CONST >> 30) && id >= WIT29_LDS_CONSTS) {
      const uint32_t* c = consts29 + (size_t)id * 9;
#pragma unroll
      for (int k = 0; k < 9; k++) r.v[k] = c[k];
    } else {
      wit29_read(r, wit29_addr(enc, lane * 48), ring);
    }
  };
  Fr29 v, va, vb;
  if (op == G_CONST) {
    src(v, OPK_CONST | ea);
  } else if (op == G_INPUT) {
    const uint32_t* in = inputs + ((size_t)p * n_inputs + ea) * 8;
    if (limbs_geq(in, FrParams::MOD)) o.e = WERR_INPUT_RANGE;  // u256_to_fr fails (graph.rs:42-45)
    Fr x;
#pragma unroll
    for (int k = 0; k < 8; k++) x.v[k] = in[k];
    v = Fr29::mul(Fr29::slice(x), Fr29::from_const(Fr29C::FROM_CANON));
  } else {
    src(va, ea);
    if (op != G_NEG && op != G_ID) src(vb, eb);
    if (op == G_MUL) {
      v = Fr29::mul(va, vb);
    } else if (op == W29_FMA) {
      Fr29 vc;
      src(vc, ec);
      v = Fr29::mul_add(va, vb, vc);
    } else if (op == G_ADD) {
#pragma unroll
      for (int k = 0; k < 9; k++) v.v[k] = va.v[k] + vb.v[k];
      v.normalize();
    } else if (op == G_SUB) {
      v = Fr29::sub(va, Fr29C::K8, vb);
    } else if (op == G_NEG) {
      v = Fr29::neg_lazy(Fr29C::K8, va);
      v.normalize();
    } else if (op == G_ID) {
      (void)witness_slow_op(G_ID, Fr::zero(), Fr::zero(), &o.e);
      v = Fr29::zero();
    } else if (op == G_TERN) {
      Fr29 vc;
      src(vc, ec);
      const bool z = va.is_zero_mod_q();  // graph.rs:214-224
#pragma unroll
      for (int k = 0; k < 9; k++) v.v[k] = z ? vc.v[k] : vb.v[k];
    } else {  // comparisons, shifts, bit operations, division ...: on canonical integers, in the 8 x 32 form
      v = Fr29::from_fq(witness_slow_op(op, va.to_fq(), vb.to_fq(), &o.e));
    }
  }
  if (w0 & W29_RED) v = Fr29::mul(v, Fr29::from_const(Fr29C::ONE));
  o.v = v;
  return o;
}
template <bool PROF>
__global__ void __launch_bounds__(64) k_witness29(const GNode29* __restrict__ nodes, uint32_t n_nodes,
                                                  const uint32_t* __restrict__ consts29, uint32_t n_consts,
                                                  const uint32_t* __restrict__ inputs, uint32_t n_inputs,
                                                  uint4* __restrict__ V29, uint32_t* __restrict__ err, uint32_t B,
                                                  uint32_t nb, unsigned long long* __restrict__ prof) {
  unsigned long long pc[4] = {0, 0, 0, 0}, pn[4] = {0, 0, 0, 0}, pw0 = 0, pc0 = 0;
  if (PROF) { pc0 = clock64(); pw0 = wall_clock64(); }
  // LDS: [WIT29_RING][64][12] node values, [WIT29_LDS_CONSTS][12] constants, [2 WIT29_CH][4] program words
  extern __shared__ __attribute__((aligned(16))) uint32_t ring[];
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  const uint32_t lane = threadIdx.x, lane48 = lane * 48;
  uint32_t p = blockIdx.x * 64 + lane;
  if (p >= nb) return;
  uint32_t e = WERR_NONE;
  uint32_t* const lconsts = ring + WIT29_RING * 64 * 12;
  uint32_t* const prog = lconsts + WIT29_LDS_CONSTS * 12;
  {
    const uint32_t nc = n_consts < WIT29_LDS_CONSTS ? n_consts : WIT29_LDS_CONSTS;
    for (uint32_t i = lane; i < nc * 9; i += 64) lconsts[(i / 9) * 12 + i % 9] = consts29[i];
  }
  const uint4* const gsrc = (const uint4*)nodes;   // lane l of chunk k: descriptors [k CH + 4 l, + 4)
  uint4 pf[4];
#pragma unroll
  for (int k = 0; k < 4; k++) pf[k] = gsrc[lane * 4 + k];
#pragma unroll
  for (int k = 0; k < 4; k++) ((uint4*)prog)[lane * 4 + k] = pf[k];
#pragma unroll
  for (int k = 0; k < 4; k++) pf[k] = gsrc[(size_t)WIT29_CH + lane * 4 + k];   // chunk 1
  __syncthreads();
  uint4 d_next = ((const uint4*)prog)[0];
  const uint32_t n_chunks = (n_nodes + WIT29_CH - 1) / WIT29_CH;
#pragma unroll 1
  for (uint32_t ch = 0; ch < n_chunks; ch++) {
    {   // chunk ch is in LDS; park chunk + 1, start loading chunk + 2
#pragma unroll
      for (int k = 0; k < 4; k++) ((uint4*)prog)[((ch + 1) & 1) * WIT29_CH + lane * 4 + k] = pf[k];
#pragma unroll
      for (int k = 0; k < 4; k++) pf[k] = gsrc[(size_t)(ch + 2) * WIT29_CH + lane * 4 + k];
    }
    const uint32_t n_end = (ch + 1) * WIT29_CH < n_nodes ? (ch + 1) * WIT29_CH : n_nodes;
#pragma unroll 1
    for (uint32_t n = ch * WIT29_CH; n < n_end; n++) {
      const uint32_t w0 = __builtin_amdgcn_readfirstlane(d_next.x), ea = __builtin_amdgcn_readfirstlane(d_next.y),
                     eb = __builtin_amdgcn_readfirstlane(d_next.z), ec = __builtin_amdgcn_readfirstlane(d_next.w);
      unsigned long long tn = 0;
      if (PROF) tn = clock64();
      Fr29 v;
      if (!(w0 & W29_RARE)) {
        // Mul / Add on ring values and LDS constants: both operand reads and the next descriptor go out together and
        // cost one LDS round trip.  The previous node's value is read back from the ring like any other (its write
        // was issued a few instructions earlier and LDS is in order): forwarding it in registers cost 18 selects and
        // saved nothing, because a node waits for at least one LDS read anyway.
        Fr29 va, vb, vc;
        wit29_read(va, wit29_addr(ea, lane48), ring);
        wit29_read(vb, wit29_addr(eb, lane48), ring);
        wit29_read(vc, wit29_addr(ec, lane48), ring);   // the addend of a * b + c (a harmless ring slot otherwise)
        d_next = ((const uint4*)prog)[(n + 1) % (2 * WIT29_CH)];
        if ((w0 & 0xFF) == W29_FMA) {
          v = Fr29::mul_add(va, vb, vc);
        } else if ((w0 & 0xFF) == G_MUL) {
          v = Fr29::mul(va, vb);
        } else {  // G_ADD
#pragma unroll
          for (int k = 0; k < 9; k++) v.v[k] = va.v[k] + vb.v[k];
          v.normalize();
        }
      } else {
        d_next = ((const uint4*)prog)[(n + 1) % (2 * WIT29_CH)];
        const Wit29Out o = wit29_rare(w0, ea, eb, ec, ring, lane, consts29, inputs, n_inputs, V29, B, p);
        v = o.v;
        if (o.e && !e) e = o.e;
      }
      {  // every value goes to the ring (three LDS instructions: cheaper than asking whether anybody reads it)
        char* a = (char*)ring + (n % WIT29_RING) * 64 * 48 + lane48;
        *(uint4*)a = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
        *(uint4*)(a + 16) = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
        *(uint32_t*)(a + 32) = v.v[8];
      }
      if (w0 & W29_STORE) {
        uint4* g = V29 + ((size_t)(w0 >> 16) * B + p) * 3;
        g[0] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
        g[1] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
        g[2] = make_uint4(v.v[8], 0, 0, 0);
      }
      if (PROF) {
        const uint32_t op = w0 & 0xFF;
        const int cls = (op == G_MUL || op == W29_FMA) ? 0 : op == G_ADD ? 1 : (op == G_CONST || op == G_INPUT) ? 2 : 3;
        pc[cls] += clock64() - tn;
        pn[cls]++;
      }
    }
  }
  err[p] = e;
  if (PROF && blockIdx.x == 0 && lane == 0) {
    for (int k = 0; k < 4; k++) { prof[k] = pc[k]; prof[4 + k] = pn[k]; }
    prof[8] = clock64() - pc0;
    prof[9] = wall_clock64() - pw0;
  }
}
// stored node values of the Fr29 interpreter -> the 8 x 32 Montgomery values every later kernel reads (V[node][proof])
// lg (small batches): lanes = stored values of ONE proof (blockIdx.y) instead of lanes = proofs
__global__ void __launch_bounds__(64) k_v29_to_fr(const uint4* __restrict__ V29, const uint32_t* __restrict__ slot2node,
                                                  uint32_t nslots, Fr* __restrict__ V, uint32_t B, uint32_t nb,
                                                  uint32_t lg = 0) {
  const uint32_t p = lg ? blockIdx.y : blockIdx.x * 64 + threadIdx.x, sl = lg ? blockIdx.x * 64 + threadIdx.x : blockIdx.y;
  if (p >= nb || sl >= nslots) return;
  const uint4* g = V29 + ((size_t)sl * B + p) * 3;
  const uint4 x = g[0], y = g[1], z = g[2];
  Fr29 v;
  v.v[0] = x.x; v.v[1] = x.y; v.v[2] = x.z; v.v[3] = x.w;
  v.v[4] = y.x; v.v[5] = y.y; v.v[6] = y.z; v.v[7] = y.w;
  v.v[8] = z.x;
  V[(size_t)slot2node[sl] * B + p] = v.to_fq();
}

// =====================================================================================================
// 2. a = A.w, b = B.w, c = a o b on the padded domain (qap.rs:40-67)
// =====================================================================================================
struct CsrView {
  const uint32_t* ptr;
  const uint32_t* col;  // already mapped to graph node ids
  const Fr* coef;
};
// LG (small batches): lanes = rows of ONE proof (blockIdx.y) instead of lanes = proofs -- with lanes = proofs a single
// proof launches 8 192 waves with one useful lane each, which also crowd the walks that run beside them
// Long rows (LG): the circuit's matrices hold 2 entries in most rows and 60 + 60 in ninety of them (Poseidon's mix
// layers), and a lane that walks 120 entries alone -- two dependent loads and a product each -- is the whole kernel
// (0.40 ms for one proof).  Rows with more than MV_LONG entries in A or B are therefore taken out of the lanes = rows
// part and given a wave each (blocks >= nshort): a lane per entry, then a shuffle tree of field additions (exact, so the
// order of the sum does not matter).
constexpr uint32_t MV_LONG = 8;
__device__ __forceinline__ Fr fr_shfl_down(const Fr& x, int off) {
  Fr r;
#pragma unroll
  for (int k = 0; k < 8; k++) r.v[k] = (uint32_t)__shfl_down((int)x.v[k], off, 64);
  return r;
}
template <bool LG>
__global__ void __launch_bounds__(256) k_matvec(CsrView A, CsrView Bm, const Fr* __restrict__ V,
                                                const uint32_t* __restrict__ sig2node, uint32_t nc, uint32_t ni,
                                                uint32_t n, Fr* __restrict__ abc, uint32_t B, uint32_t nb,
                                                const uint32_t* __restrict__ long_rows = nullptr, uint32_t nshort = 0) {
  uint32_t p = LG ? blockIdx.y : blockIdx.x * 64 + threadIdx.x;
  if (LG && blockIdx.x >= nshort) {   // a wave per long row
    const uint32_t row = long_rows[blockIdx.x - nshort], lane = threadIdx.x;
    Fr a = Fr::zero(), b = Fr::zero();
    for (uint32_t k = A.ptr[row] + lane; k < A.ptr[row + 1]; k += 64) a = a + A.coef[k] * V[(size_t)A.col[k] * B + p];
    for (uint32_t k = Bm.ptr[row] + lane; k < Bm.ptr[row + 1]; k += 64) b = b + Bm.coef[k] * V[(size_t)Bm.col[k] * B + p];
#pragma unroll
    for (int off = 32; off; off >>= 1) {
      a = a + fr_shfl_down(a, off);
      b = b + fr_shfl_down(b, off);
    }
    if (lane == 0) {
      const size_t o = (size_t)row * B + p;
      abc[o] = a;
      abc[(size_t)n * B + o] = b;
      abc[2 * (size_t)n * B + o] = a * b;
    }
    return;
  }
  uint32_t row = LG ? blockIdx.x * 64 + threadIdx.x
                    : __builtin_amdgcn_readfirstlane(blockIdx.y * blockDim.y + threadIdx.y);  // wave-uniform
  if (row >= n) return;
  if (p >= nb) return;
  Fr a = Fr::zero(), b = Fr::zero();
  if (row < nc) {
    if (LG && long_rows && (A.ptr[row + 1] - A.ptr[row] > MV_LONG || Bm.ptr[row + 1] - Bm.ptr[row] > MV_LONG)) return;
    for (uint32_t k = A.ptr[row]; k < A.ptr[row + 1]; k++) a = a + A.coef[k] * V[(size_t)A.col[k] * B + p];
    for (uint32_t k = Bm.ptr[row]; k < Bm.ptr[row + 1]; k++) b = b + Bm.coef[k] * V[(size_t)Bm.col[k] * B + p];
  } else if (row < nc + ni) {
    a = V[(size_t)sig2node[row - nc] * B + p];  // a[nc..nc+ni] = w[0..ni] (qap.rs:54-58)
  }
  size_t o = (size_t)row * B + p;
  abc[o] = a;
  abc[(size_t)n * B + o] = b;
  abc[2 * (size_t)n * B + o] = (row < nc) ? a * b : Fr::zero();
}

// =====================================================================================================
// 3. radix-2^K register-blocked NTT passes over [index][proof] data (ark-poly Radix2EvaluationDomain
//    fft/ifft semantics; call sites qap.rs:69-90).  DIF takes natural order to bit-reversed, DIT takes
//    bit-reversed back to natural, so iNTT(DIF) -> coset scale -> NTT(DIT) needs no reordering pass.
// =====================================================================================================
#ifndef RLN_NTT_WAVES
#define RLN_NTT_WAVES 1
#endif
__device__ __forceinline__ Fr29 load_fr29(const uint32_t* __restrict__ p) {
  Fr29 w;
#pragma unroll
  for (int k = 0; k < 9; k++) w.v[k] = p[k];
  return w;
}
// constants for Fr29::mul_mont: the Fr29 image (x 2^261, normalised) of 8 x 32 Montgomery values
__global__ void __launch_bounds__(256) k_consts_to29(const Fr* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fr29 v = Fr29::from_fq(src[t]);
  v.normalize();
#pragma unroll
  for (int k = 0; k < 9; k++) dst[(size_t)t * 9 + k] = v.v[k];
}
// M29: twiddles as Fr29 constants and Fr29::mul_mont products; otherwise 8 x 32 twiddles and products (RLNAMD_NTT29=0)
// LG (small batches): lanes = groups of ONE proof (blockIdx.x = proof) instead of lanes = proofs -- a single proof then
// fills its waves (3 072 eight-point blocks = 48 waves per pass) instead of running 3 072 waves with one useful lane each
// (all ten passes of one proof 0.78 -> see profiles/r3); twiddle indices become per-lane values.
template <int K, bool DIF, bool M29, bool LG = false>
__global__ void __launch_bounds__(256, RLN_NTT_WAVES) k_ntt_pass(Fr* __restrict__ data, const uint32_t* __restrict__ tw, int logn, int s0,
                                                  const uint32_t* __restrict__ scale, uint32_t B, uint32_t nb) {
  auto tmul = [&](const Fr& a, const uint32_t* __restrict__ tab, uint32_t idx) -> Fr {
    if constexpr (M29)
      return Fr29::mul_mont(a, load_fr29(tab + 9 * (size_t)idx));
    else
      return a * reinterpret_cast<const Fr*>(tab)[idx];
  };
  constexpr int R = 1 << K;
  const uint32_t n = 1u << logn;
  auto uni = [](uint32_t v) -> uint32_t { return LG ? v : __builtin_amdgcn_readfirstlane(v); };
  uint32_t p = LG ? blockIdx.x : blockIdx.x * 64 + threadIdx.x;
  uint32_t g = LG ? blockIdx.y * 64 + threadIdx.x
                  : __builtin_amdgcn_readfirstlane(blockIdx.y * blockDim.y + threadIdx.y);  // one group per wave
  if (g >= (n >> K)) return;
  if (p >= nb) return;
  Fr* x = data + (size_t)blockIdx.z * n * B + p;
  uint32_t stride, base;
  if (DIF) {
    stride = n >> (s0 + K);  // h_last
    uint32_t blk = g / stride, lo = g % stride;
    base = blk * (n >> s0) + lo;
  } else {
    stride = 1u << s0;  // h_first
    uint32_t blk = g / stride, lo = g % stride;
    base = blk * (stride << K) + lo;
  }
  const uint32_t lo = g % stride;
  Fr e[R];
#pragma unroll
  for (int m = 0; m < R; m++) e[m] = x[(size_t)(base + m * stride) * B];
#pragma unroll
  for (int t = 0; t < K; t++) {
    const int half = DIF ? (R >> (t + 1)) : (1 << t);
#pragma unroll
    for (int m = 0; m < R; m++) {
      if (m & half) continue;
      uint32_t j = (uint32_t)(m & (half - 1)) * stride + lo;
      uint32_t ti = DIF ? (j << (s0 + t)) : (j << (logn - 1 - (s0 + t)));
      // the twiddle index is the same for all 64 lanes (lanes = proofs): force the scalar path so the
      // twiddle rides in SGPRs instead of VGPRs.  Twiddles are held as Fr29 constants (w 2^261, 9 words): the
      // product with an 8 x 32 value needs no conversion (Fr29::mul_mont, ~290 instead of ~375 instructions)
      const uint32_t tix = uni(ti);
      if (DIF) {
        Fr u = e[m], v = e[m + half];
        e[m] = u + v;
        e[m + half] = tmul(u - v, tw, tix);
      } else {
        Fr u = e[m], v = tmul(e[m + half], tw, tix);
        e[m] = u + v;
        e[m + half] = u - v;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < R; m++) {
    uint32_t pos = base + m * stride;
    Fr o = e[m];
    if (scale) o = tmul(o, scale, uni(pos));
    x[(size_t)pos * B] = o;
  }
}

// Small batches: nine levels in ONE kernel.  The 512 points {base + m stride} that nine consecutive levels close over
// are one wave's work: three radix-8 sub-passes, eight points per lane, exchanged through 16 KB of LDS instead of
// through HBM and two kernel boundaries (a lone proof's pass is all latency: 50 - 65 us for 6 us of arithmetic).  In the
// set's local coordinates m = 0..511 the sub-passes are a 512-point transform's (local strides 64, 8, 1 for DIF and
// 1, 8, 64 for DIT); their twiddles are the big transform's, indexed exactly as k_ntt_pass does for the pass
// (s0 + 3 q, K = 3): global stride = stride * ls, lo = lo_set + lo_local * stride.  Same butterflies, same products, same
// order per point: bit-identical to the passes it replaces.  Grid (proof, set, vector), one wave per workgroup.
template <bool DIF>
__global__ void __launch_bounds__(64) k_ntt_fused9(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, int s0,
                                                   const Fr* __restrict__ scale, uint32_t B, uint32_t nb) {
  __shared__ Fr buf[512];
  const uint32_t p = blockIdx.x, set = blockIdx.y, lane = threadIdx.x;
  const uint32_t n = 1u << logn;
  if (p >= nb) return;
  Fr* x = data + (size_t)blockIdx.z * n * B + p;
  uint32_t stride, base, lo;
  if (DIF) {
    stride = n >> (s0 + 9);
    lo = set % stride;
    base = (set / stride) * (n >> s0) + lo;
  } else {
    stride = 1u << s0;
    lo = set % stride;
    base = (set / stride) * (stride << 9) + lo;
  }
#pragma unroll 1
  for (int q = 0; q < 3; q++) {
    const uint32_t ls = DIF ? (64u >> (3 * q)) : (1u << (3 * q));
    const uint32_t lo_l = lane % ls, base_l = (lane / ls) * (ls * 8) + lo_l;
    const int s0q = s0 + 3 * q;
    const uint32_t stride_q = stride * ls, lo_q = lo + lo_l * stride;
    Fr e[8];
    if (q == 0) {
#pragma unroll
      for (int m = 0; m < 8; m++) e[m] = x[(size_t)(base + (base_l + m * ls) * stride) * B];
    } else {
#pragma unroll
      for (int m = 0; m < 8; m++) e[m] = buf[base_l + m * ls];
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int half = DIF ? (8 >> (t + 1)) : (1 << t);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        if (m & half) continue;
        const uint32_t j = (uint32_t)(m & (half - 1)) * stride_q + lo_q;
        const uint32_t ti = DIF ? (j << (s0q + t)) : (j << (logn - 1 - (s0q + t)));
        if (DIF) {
          const Fr u = e[m], v = e[m + half];
          e[m] = u + v;
          e[m + half] = (u - v) * tw[ti];
        } else {
          const Fr u = e[m], v = e[m + half] * tw[ti];
          e[m] = u + v;
          e[m + half] = u - v;
        }
      }
    }
    if (q == 2) {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const uint32_t pos = base + (base_l + m * ls) * stride;
        Fr o = e[m];
        if (scale) o = o * scale[pos];
        x[(size_t)pos * B] = o;
      }
    } else {
      __syncthreads();   // (one wave: orders the reads above against the writes below)
#pragma unroll
      for (int m = 0; m < 8; m++) buf[base_l + m * ls] = e[m];
      __syncthreads();
    }
  }
}

// h = a o b - c  (qap.rs:84-95), written over the `a` vector
__global__ void __launch_bounds__(256) k_hquot(Fr* __restrict__ abc, uint32_t n, uint32_t B, uint32_t nb, uint32_t lg) {
  uint32_t p = lg ? blockIdx.y : blockIdx.x * 64 + threadIdx.x;   // lg: lanes = coefficients of one proof
  uint32_t i = lg ? blockIdx.x * 64 + threadIdx.x : blockIdx.y * blockDim.y + threadIdx.y;
  if (p >= nb || i >= n) return;
  size_t o = (size_t)i * B + p;
  abc[o] = abc[o] * abc[(size_t)n * B + o] - abc[2 * (size_t)n * B + o];
}

// =====================================================================================================
// 4. scalars -> signed digits (window j: cw[j] bits), layout [scalar][half][window][proof] (int16)
// =====================================================================================================
// Digits of the magnitude `l` (NL limbs, destroyed) under schedule ws; the scalar's sign flips every digit.  A window
// of c bits yields d in [-2^(c-1), 2^(c-1)]; both ends select table entry 2^(c-1) - 1, but only one of them fits an
// int16 at c = 16, so a window value of exactly 2^(c-1) goes to the end the sign leaves representable.
template <int NL>
__device__ __forceinline__ void emit_digits(uint32_t* l, bool neg, const WinSched& ws, int16_t* __restrict__ out, uint32_t B) {
  uint32_t carry = 0;
#pragma unroll 1
  for (int j = 0; j < ws.W; j++) {
    const int c = ws.cw[j];
    const uint32_t mask = (c >= 32) ? 0xFFFFFFFFu : ((1u << c) - 1), E = 1u << (c - 1);
    uint32_t raw = (l[0] & mask) + carry;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) l[i] = (l[i] >> c) | (l[i + 1] << (32 - c));
    l[NL - 1] >>= c;
    int d;
    if (raw > E || (raw == E && !neg)) {
      d = (int)raw - (int)(mask + 1);
      carry = 1;
    } else {
      d = (int)raw;
      carry = 0;
    }
    out[(size_t)j * B] = (int16_t)(neg ? -d : d);
  }
}
// Scalar ids: [0, ns) witness signals, [ns, ns + n) quotient coefficients h, then r, s, -(r s).  dig1 holds the G1
// schedule for all of them; dig2 the G2 schedule for the ones the G2 walk uses (witness, r, s, -(r s): id - n).
// nh = 2: every scalar is split as k1 + lambda k2 (glv.h) and both halves are recoded; nh = 1: the plain 254-bit walk.
__global__ void __launch_bounds__(256) k_recode(const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node,
                                                uint32_t ns, const Fr* __restrict__ H, uint32_t n,
                                                const uint32_t* __restrict__ rs, WinSched ws1, WinSched ws2, uint32_t nh,
                                                int16_t* __restrict__ dig1, int16_t* __restrict__ dig2, uint32_t B,
                                                uint32_t nb, uint32_t part, uint32_t lg) {
  // part 0: every scalar; 1: the witness scalars and r, s, -(r s) (all the G2 walk needs: it can start before the
  // quotient h exists); 2: the coefficients of h only
  // part 3 (small full proofs, fused plan): the products s w_i, r w_i and r s under the ids ns + n + 3 + ..., G1 only
  uint32_t p = lg ? blockIdx.y : blockIdx.x * 64 + threadIdx.x;   // lg (small batches): lanes = scalars of one proof
  uint32_t sid = lg ? blockIdx.x * 64 + threadIdx.x : blockIdx.y * blockDim.y + threadIdx.y;
  if (part == 1) {
    if (sid >= ns + 3) return;
    if (sid >= ns) sid += n;
  } else if (part == 2) {
    if (sid >= n) return;
    sid += ns;
  } else if (part == 3) {
    if (sid >= 2 * ns + 1) return;
    sid += ns + n + 3;
  }
  if (p >= nb || sid >= 3 * ns + n + 4) return;
  Fr x;
  if (sid >= ns + n + 3) {
    const uint32_t q = sid - (ns + n + 3);
    const Fr r = Fr::from_canonical(rs + (size_t)p * 16), s = Fr::from_canonical(rs + (size_t)p * 16 + 8);
    if (q < ns)
      x = s * V[(size_t)sig2node[q] * B + p];
    else if (q < 2 * ns)
      x = r * V[(size_t)sig2node[q - ns] * B + p];
    else
      x = r * s;
  } else if (sid < ns) {
    x = V[(size_t)sig2node[sid] * B + p];
  } else if (sid < ns + n) {
    x = H[(size_t)(sid - ns) * B + p];
  } else {
    Fr r = Fr::from_canonical(rs + (size_t)p * 16);
    Fr s = Fr::from_canonical(rs + (size_t)p * 16 + 8);
    uint32_t which = sid - ns - n;  // 0: r, 1: s, 2: -(r s)
    x = which == 0 ? r : which == 1 ? s : (r * s).neg();
  }
  uint32_t l[8];
  x.to_canonical(l);
  const bool g2 = sid < ns || (sid >= ns + n && sid < ns + n + 3);
  const uint32_t sid2 = sid < ns ? sid : sid - n;
  if (nh == 2) {
    uint32_t k[2][4], neg[2];
    glv_split(l, k[0], &neg[0], k[1], &neg[1]);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t t[4];
      if (g2) {
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = k[h][i];
        emit_digits<4>(t, neg[h] != 0, ws2, dig2 + ((size_t)sid2 * 2 + h) * ws2.W * B + p, B);
      }
      emit_digits<4>(k[h], neg[h] != 0, ws1, dig1 + ((size_t)sid * 2 + h) * ws1.W * B + p, B);
    }
  } else {
    if (g2) {
      uint32_t t[8];
#pragma unroll
      for (int i = 0; i < 8; i++) t[i] = l[i];
      emit_digits<8>(t, false, ws2, dig2 + (size_t)sid2 * ws2.W * B + p, B);
    }
    emit_digits<8>(l, false, ws1, dig1 + (size_t)sid * ws1.W * B + p, B);
  }
}

// =====================================================================================================
// 5. table-driven MSM: acc += +-T[point][window][|digit|-1]
// =====================================================================================================
template <class F>
__global__ void __launch_bounds__(64) k_msm(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ sid,
                                            const uint32_t* __restrict__ rows, const ChunkDesc* __restrict__ chunks,
                                            uint32_t nchunks, const int16_t* __restrict__ digits,
                                            XYZZ<F>* __restrict__ part, WinSched ws, uint32_t B, uint32_t pgroups,
                                            uint32_t nh) {
  // XCD-aware decode: hardware places block L on XCD L % 8; all proof groups of one chunk share the same
  // table rows, so they are given consecutive slots on ONE XCD and meet in that XCD's L2.
  uint32_t L = blockIdx.x;
  uint32_t xcd = L & 7, q = L >> 3;
  uint32_t chunk = (q / pgroups) * 8 + xcd, pg = q % pgroups;
  if (chunk >= nchunks) return;
  uint32_t p = pg * 64 + threadIdx.x;  // padded lanes run on zero digits
  ChunkDesc cd = chunks[chunk];
  XYZZ<F> acc = XYZZ<F>::inf();
  const int W = ws.W;
#pragma unroll 1
  for (uint32_t i = cd.pt_begin; i < cd.pt_end; i++) {
    // table row; the walk (full / partial / finish) is a list of rows.  Bit 31: the row is walked with the digits of
    // the scalar's second GLV half (the sum of those rows is mapped through phi afterwards, k_glv_fold)
    const uint32_t kk = rows[i], k = kk & 0x7FFFFFFFu;
    const int16_t* dg = digits + ((size_t)sid[i] * nh + (kk >> 31)) * W * B + p;
    const Affine<F>* row = table + (size_t)k * ws.stride;
#pragma unroll 1
    for (int j = 0; j < W; j++) {
      int d = dg[(size_t)j * B];
      if (d != 0) {
        uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        Affine<F> pt = row[ws.ro[j] + e];
        if (d < 0) pt.y = pt.y.neg();
        acc.madd(pt);
      }
    }
  }
  part[(size_t)chunk * B + p] = acc;
}

template <class A, class E>
__global__ void __launch_bounds__(256) k_table_to29(const A* __restrict__ src, E* __restrict__ dst, size_t n) {
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  dst[t] = to_table29(src[t]);
}

// dst[r][p] = sum of src[i][p] over ranges[r] -- used twice (chunks -> groups -> segments) so the
// per-proof reduction is a two-level tree instead of one long serial chain
template <class F>
__global__ void __launch_bounds__(64) k_sum_ranges(const XYZZ<F>* __restrict__ src, const ChunkDesc* __restrict__ ranges,
                                                   uint32_t nranges, XYZZ<F>* __restrict__ dst, uint32_t B, uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  uint32_t r = blockIdx.y;
  if (p >= nb || r >= nranges) return;
  ChunkDesc cd = ranges[r];
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t i = cd.pt_begin; i < cd.pt_end; i++) acc.add(src[(size_t)i * B + p]);
  dst[(size_t)r * B + p] = acc;
}

// The same reduction for small batches, lanes = partial sums instead of lanes = proofs: with one proof in the batch
// k_sum_ranges leaves 63 lanes idle and a G2 segment is a serial chain of 16 + 30 general additions (~ 2 ms); here the
// 64 lanes of the wave of (proof, segment) each add their share of the segment's chunks and meet in a six-level tree
// through LDS: 8 + 6 additions.
// which segments / tasks a launch covers (grid.y = n): a small batch finishes s A and r B1 from the h-independent rows
// while the h rows are still being walked, so the back-end kernels run twice on disjoint task lists
struct TaskSel {
  uint8_t id[8];
};
static TaskSel task_sel(std::initializer_list<uint32_t> ids) {
  TaskSel t{};
  uint32_t k = 0;
  for (uint32_t v : ids) t.id[k++] = (uint8_t)v;
  return t;
}
constexpr uint32_t SUM_TREE_LANES = 512;
template <class F>
__global__ void __launch_bounds__(SUM_TREE_LANES) k_sum_tree(const XYZZ<F>* __restrict__ part, const ChunkDesc* __restrict__ segchunks,
                                                  XYZZ<F>* __restrict__ dst, uint32_t B, uint32_t PB, TaskSel sel) {
  // 512 lanes per (proof, segment): the short chunks of the small-batch plans leave ~2 000 partial sums per segment;
  // four per lane and a nine-level tree (part stride PB, result stride B).  The additions are a dependent chain for the
  // lone waves of a single proof (30 us each in Fq2), so the lane count is what sets the kernel's length: 256 lanes were
  // 8 + 8 additions.  Only the upper half of a level passes through LDS.
  __shared__ XYZZ<F> sh[SUM_TREE_LANES / 2];
  __builtin_amdgcn_s_setprio(3);
  const uint32_t p = blockIdx.x, sgi = sel.id[blockIdx.y], l = threadIdx.x;
  const ChunkDesc cd = segchunks[sgi];
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t i = cd.pt_begin + l; i < cd.pt_end; i += SUM_TREE_LANES) acc.add(part[(size_t)i * PB + p]);
#pragma unroll 1
  for (uint32_t stride = SUM_TREE_LANES / 2; stride >= 1; stride >>= 1) {
    if (l >= stride && l < 2 * stride) sh[l - stride] = acc;
    __syncthreads();
    if (l < stride) acc.add(sh[l]);
    __syncthreads();
  }
  if (l == 0) dst[(size_t)sgi * B + p] = acc;
}

// GLV: segment t holds sum k1_i P_i, segment nseg + t holds sum k2_i P_i; the result is the first plus phi of the
// second, phi(X, Y, ZZ, ZZZ) = (beta X, Y, ZZ, ZZZ) (x = X / ZZ).  One Fq product per output point and proof.
__global__ void __launch_bounds__(64) k_glv_fold(G1XYZZ* __restrict__ sums1, G2XYZZ* __restrict__ sums2, uint32_t nseg1,
                                                 uint32_t B, uint32_t nb, TaskSel sel) {
  __builtin_amdgcn_s_setprio(3);
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t t = sel.id[blockIdx.y];
  if (t < nseg1) {
    G1XYZZ a = sums1[(size_t)t * B + p], b = sums1[(size_t)(nseg1 + t) * B + p];
    b.X = b.X * Fq::from_canonical(GlvParams::BETA_G1);
    a.add(b);
    sums1[(size_t)t * B + p] = a;
  } else {
    G2XYZZ a = sums2[p], b = sums2[(size_t)B + p];
    b.X = b.X.mul_fq(Fq::from_canonical(GlvParams::BETA_G2));
    a.add(b);
    sums2[p] = a;
  }
}

// one-time comb table: row (k, j) = { d * 2^(c j) * P_k : d = 1..2^(c-1) } in affine form.
// Built by doubling the known prefix (multiples 1..m -> m+1..2m are "T[i] + T[m]" and one doubling) with
// one shared inversion per level (Montgomery's trick; prefix products parked in `scratch`).
template <class F>
__global__ void __launch_bounds__(64) k_table_build(const Affine<F>* __restrict__ pts, uint32_t npts, WinSched ws,
                                                    Affine<F>* __restrict__ table, F* __restrict__ scratch) {
  const uint32_t W = (uint32_t)ws.W;
  size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= (size_t)npts * W) return;
  uint32_t k = (uint32_t)(t / W), j = (uint32_t)(t % W);
  const uint32_t E = 1u << (ws.cw[j] - 1);
  XYZZ<F> b = XYZZ<F>::from_affine(pts[k]);
  for (uint32_t i = 0; i < (uint32_t)ws.bo[j]; i++) b = b.dbl();
  Affine<F> base = b.to_affine();
  const size_t off = (size_t)k * ws.stride + ws.ro[j];  // even: every row has >= 2 entries (cw >= 2)
  Affine<F>* row = table + off;
  F* pre = scratch + off / 2;
  row[0] = base;
  for (uint32_t m = 1; m < E; m <<= 1) {
    const Affine<F> Pm = row[m - 1];
    F run = F::one();
    for (uint32_t i = 1; i <= m; i++) {
      F den = (i < m) ? (row[i - 1].x - Pm.x) : Pm.y.dbl();
      pre[i - 1] = run;
      run = run * den;
    }
    F inv = run.inv();
    for (uint32_t i = m; i >= 1; i--) {
      F den, lam, x3, y3;
      if (i < m) {
        Affine<F> Pi = row[i - 1];
        den = Pi.x - Pm.x;
        F di = inv * pre[i - 1];
        lam = (Pi.y - Pm.y) * di;
        x3 = lam.sqr() - Pi.x - Pm.x;
        y3 = lam * (Pi.x - x3) - Pi.y;
      } else {
        den = Pm.y.dbl();
        F di = inv * pre[i - 1];
        F x2 = Pm.x.sqr();
        lam = (x2.dbl() + x2) * di;
        x3 = lam.sqr() - Pm.x.dbl();
        y3 = lam * (Pm.x - x3) - Pm.y;
      }
      inv = inv * den;
      row[m + i - 1] = {x3, y3};
    }
  }
}

// =====================================================================================================
// 6. finalize: A, B affine; C = s*A + r*B1 + (L + H - rs*delta) ; compressed encoding
//    (partial_proof.rs:232-273; the alpha/beta/delta/query[0] terms are folded into the MSM segments)
// =====================================================================================================
__device__ __forceinline__ bool fq_is_neg_dev(const Fq& y) {
  uint32_t c[8];
  y.to_canonical(c);
  return limbs_gt(c, FqParams::HALF);
}
__device__ __forceinline__ void store_fq(uint32_t* dst, const Fq& x) { x.to_canonical(dst); }

// Partial proofs (partial_proof.rs:108-179, 182-274).  k_partial_out: the four sums of the "known" walk leave as
// canonical affine coordinates [pi_a | rho | pi_b | pi_c] (320 B).  k_add_partial: the same four points, given
// back with the full witness, are added to the sums of the "unknown + H + blinding" walk before finalize.
__global__ void __launch_bounds__(64) k_partial_out(const G1XYZZ* __restrict__ sums1, const G2XYZZ* __restrict__ sums2,
                                                    uint32_t* __restrict__ out, uint32_t B, uint32_t nb) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  uint32_t* o = out + (size_t)p * 80;
  const uint32_t t = blockIdx.y;  // 0 pi_a, 1 rho, 2 pi_c, 3 pi_b
  if (t < 3) {
    G1Affine a = sums1[(size_t)t * B + p].to_affine();
    uint32_t* d = o + (t == 0 ? 0 : t == 1 ? 16 : 64);
    a.x.to_canonical(d);
    a.y.to_canonical(d + 8);
  } else {
    G2Affine b = sums2[p].to_affine();
    b.x.c0.to_canonical(o + 32);
    b.x.c1.to_canonical(o + 40);
    b.y.c0.to_canonical(o + 48);
    b.y.c1.to_canonical(o + 56);
  }
}
__global__ void __launch_bounds__(64) k_add_partial(G1XYZZ* __restrict__ sums1, G2XYZZ* __restrict__ sums2,
                                                    const uint32_t* __restrict__ pp, uint32_t B, uint32_t nb) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t* o = pp + (size_t)p * 80;
  const uint32_t t = blockIdx.y;
  if (t < 3) {
    const uint32_t* d = o + (t == 0 ? 0 : t == 1 ? 16 : 64);
    G1Affine a{Fq::from_canonical(d), Fq::from_canonical(d + 8)};
    G1XYZZ acc = sums1[(size_t)t * B + p];
    acc.madd(a);
    sums1[(size_t)t * B + p] = acc;
  } else {
    G2Affine b{{Fq::from_canonical(o + 32), Fq::from_canonical(o + 40)}, {Fq::from_canonical(o + 48), Fq::from_canonical(o + 56)}};
    G2XYZZ acc = sums2[p];
    acc.madd(b);
    sums2[p] = acc;
  }
}

// F1: the three MSM sums that become proof elements go to affine form in parallel (one inversion each)
__global__ void __launch_bounds__(64) k_fin_affine(const G1XYZZ* __restrict__ sums1, const G2XYZZ* __restrict__ sums2,
                                                   G1Affine* __restrict__ affA, G1Affine* __restrict__ affB1,
                                                   G2Affine* __restrict__ affB2, uint32_t B, uint32_t nb, TaskSel sel) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t task = sel.id[blockIdx.y];
  if (task == 0)
    affA[p] = sums1[p].to_affine();
  else if (task == 1)
    affB1[p] = sums1[(size_t)B + p].to_affine();
  else
    affB2[p] = sums2[p].to_affine();
}

// F2: the two variable-base products s*A and r*B1 (partial_proof.rs:257-260), one lane each, fixed 4-bit
// windows over a 15-entry table parked in HBM: 252 doublings + 63 additions instead of a bit-serial
// double-and-add whose lanes diverge on every scalar bit.
__global__ void __launch_bounds__(64) k_fin_smul(const G1Affine* __restrict__ affA, const G1Affine* __restrict__ affB1,
                                                 const uint32_t* __restrict__ rs, G1XYZZ* __restrict__ tbl,
                                                 G1XYZZ* __restrict__ prod, uint32_t B, uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t task = blockIdx.y;  // 0: s*A, 1: r*B1
  const G1Affine P = task == 0 ? affA[p] : affB1[p];
  const uint32_t* k = rs + (size_t)p * 16 + (task == 0 ? 8 : 0);
  G1XYZZ* T = tbl + (size_t)task * 16 * B + p;  // T[d] at T[d * B]
  G1XYZZ cur = G1XYZZ::from_affine(P);
  T[(size_t)1 * B] = cur;
#pragma unroll 1
  for (int d = 2; d < 16; d++) {
    cur.madd(P);
    T[(size_t)d * B] = cur;
  }
  // k P = (+-k1) P + (+-k2) phi(P), |k1|, |k2| < 2^126 (glv.h): one ladder of 32 four-bit windows for both halves
  // (126 doublings instead of 252 on this latency path); phi(T[d]) = (beta X, Y, ZZ, ZZZ) is one product per use
  uint32_t kk[8], k1[4], k2[4], n1, n2;
#pragma unroll
  for (int i = 0; i < 8; i++) kk[i] = k[i];
  glv_split(kk, k1, &n1, k2, &n2);
  const Fq beta = Fq::from_canonical(GlvParams::BETA_G1);
  G1XYZZ acc = G1XYZZ::inf();
#pragma unroll 1
  for (int w = 31; w >= 0; w--) {
    if (w != 31) {
      acc = acc.dbl();
      acc = acc.dbl();
      acc = acc.dbl();
      acc = acc.dbl();
    }
    const uint32_t d1 = (k1[w >> 3] >> ((w & 7) * 4)) & 15, d2 = (k2[w >> 3] >> ((w & 7) * 4)) & 15;
    if (d1) {
      G1XYZZ t = T[(size_t)d1 * B];
      if (n1) t.Y = t.Y.neg();
      acc.add(t);
    }
    if (d2) {
      G1XYZZ t = T[(size_t)d2 * B];
      t.X = t.X * beta;
      if (n2) t.Y = t.Y.neg();
      acc.add(t);
    }
  }
  prod[(size_t)task * B + p] = acc;  // r == 0 gives infinity, matching g1_b = 0 (partial_proof.rs:242-248)
}

// F3: C = s*A + r*B1 + (L + H - rs*delta); canonical coordinates and the compressed encoding
__global__ void __launch_bounds__(64) k_fin_out(const G1XYZZ* __restrict__ sums1, const G1XYZZ* __restrict__ prod,
                                                const G1Affine* __restrict__ affA, const G2Affine* __restrict__ affB2,
                                                uint32_t* __restrict__ coords, uint8_t* __restrict__ comp, uint32_t B,
                                                uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  G1XYZZ Cacc = sums1[2 * (size_t)B + p];
  Cacc.add(prod[p]);
  Cacc.add(prod[(size_t)B + p]);
  G1Affine C = Cacc.to_affine();
  G1Affine A = affA[p];
  G2Affine B2 = affB2[p];
  uint32_t* o = coords + (size_t)p * 64;
  store_fq(o, A.x);
  store_fq(o + 8, A.y);
  store_fq(o + 16, B2.x.c0);
  store_fq(o + 24, B2.x.c1);
  store_fq(o + 32, B2.y.c0);
  store_fq(o + 40, B2.y.c1);
  store_fq(o + 48, C.x);
  store_fq(o + 56, C.y);
  // ark-serialize compressed Proof{a,b,c}: x with flags in the top byte (0x80: y > -y, 0x40: infinity)
  uint32_t w[32];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    w[i] = o[i];
    w[8 + i] = o[16 + i];
    w[16 + i] = o[24 + i];
    w[24 + i] = o[48 + i];
  }
  if (A.is_inf()) w[7] |= 0x40000000u; else if (fq_is_neg_dev(A.y)) w[7] |= 0x80000000u;
  if (B2.is_inf()) w[23] |= 0x40000000u;
  else if (B2.y.c1.is_zero() ? fq_is_neg_dev(B2.y.c0) : fq_is_neg_dev(B2.y.c1)) w[23] |= 0x80000000u;
  if (C.is_inf()) w[31] |= 0x40000000u; else if (fq_is_neg_dev(C.y)) w[31] |= 0x80000000u;
  uint32_t* cw = (uint32_t*)(comp + (size_t)p * 128);
#pragma unroll
  for (int i = 0; i < 32; i++) cw[i] = w[i];
}

// =====================================================================================================
// 7. proof values by the Poseidon formulae (witness.rs:759-828): root, a1, y, nullifier
// =====================================================================================================
struct InputSlots {
  uint32_t secret, limit, msg_id, path, path_idx, x, ext, depth;
};
__global__ void __launch_bounds__(64) k_proof_values(const uint32_t* __restrict__ inputs, uint32_t n_inputs,
                                                     InputSlots sl, PoseidonView p2, PoseidonView p3, PoseidonView p4,
                                                     uint32_t* __restrict__ values, uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t* in = inputs + (size_t)p * n_inputs * 8;
  auto ld = [&](uint32_t slot) { return Fr::from_canonical(in + (size_t)slot * 8); };
  Fr secret = ld(sl.secret), limit = ld(sl.limit), msg = ld(sl.msg_id), x = ld(sl.x), ext = ld(sl.ext);
  Fr h1[1] = {secret};
  Fr idc = poseidon_hash_dev<2>(h1, p2);
  Fr h2[2] = {idc, limit};
  Fr root = poseidon_hash_dev<3>(h2, p3);
#pragma unroll 1
  for (uint32_t i = 0; i < sl.depth; i++) {
    Fr e = ld(sl.path + i);
    const uint32_t* bi = in + (size_t)(sl.path_idx + i) * 8;
    uint32_t nz = 0;
    for (int q = 0; q < 8; q++) nz |= bi[q];
    if (nz == 0) {
      h2[0] = root;
      h2[1] = e;
    } else {
      h2[0] = e;
      h2[1] = root;
    }
    root = poseidon_hash_dev<3>(h2, p3);
  }
  Fr h3[3] = {secret, ext, msg};
  Fr a1 = poseidon_hash_dev<4>(h3, p4);
  Fr y = secret + x * a1;
  h1[0] = a1;
  Fr nullifier = poseidon_hash_dev<2>(h1, p2);
  uint32_t* o = values + (size_t)p * 40;
  y.to_canonical(o);
  root.to_canonical(o + 8);
  nullifier.to_canonical(o + 16);
  x.to_canonical(o + 24);
  ext.to_canonical(o + 32);
}

// public signals w[1..npub] of every proof straight from the witness (the circuit's own outputs; for the
// single-message circuit they equal k_proof_values' y, root, nullifier, x, external_nullifier)
__global__ void __launch_bounds__(256) k_public_signals(const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node,
                                                        uint32_t npub, uint32_t B, uint32_t nb, uint32_t* __restrict__ out) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  uint32_t k = blockIdx.y * blockDim.y + threadIdx.y;
  if (p >= nb || k >= npub) return;
  V[(size_t)sig2node[1 + k] * B + p].to_canonical(out + ((size_t)p * npub + k) * 8);
}

// Small batches: k_proof_values is a lone lane's chain of 24 Poseidon hashes (5.3 ms for one proof -- longer than the
// whole rest of the proof once the interpreter runs in 2.9 ms), and the interpreter has just computed the same five
// values as the circuit's outputs: take them from the witness (single-message circuit: w[1..5] = y, root, nullifier,
// x, external_nullifier, the order of k_proof_values; rln.circom's public signals, protocol/proof.rs:37-52)
__global__ void __launch_bounds__(64) k_values_from_witness(const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node,
                                                            uint32_t B, uint32_t nb, uint32_t* __restrict__ values) {
  const uint32_t p = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (p >= nb) return;
  V[(size_t)sig2node[1 + k] * B + p].to_canonical(values + (size_t)p * 40 + k * 8);
}

// gathers for the parity taps
__global__ void k_gather_col(const Fr* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t count, uint32_t B,
                             uint32_t p, uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint32_t row = idx ? idx[i] : i;
  src[(size_t)row * B + p].to_canonical(out + (size_t)i * 8);
}

// generate_zk_proof_with_witness (protocol/proof.rs:705-732): an externally calculated witness replaces the
// graph interpreter's.  given = [proof][signal] canonical LE; each signal is stored at the node it aliases.
__global__ void k_scatter_witness(const uint32_t* __restrict__ given, const uint32_t* __restrict__ sig2node,
                                  uint32_t NS, Fr* __restrict__ V, uint32_t* __restrict__ err, uint32_t B, uint32_t nb) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
  if (p >= nb || j >= NS) return;
  V[(size_t)sig2node[j] * B + p] = Fr::from_canonical(given + ((size_t)p * NS + j) * 8);
  if (j == 0) err[p] = WERR_NONE;
}

// Streamed inputs: the batch's inputs, (r, s) and partial points move from the slot's pinned staging buffer to its device
// buffers by a kernel on the batch's own front-end stream (the pinned pages are device-visible).  A hipMemcpyAsync
// here goes through the runtime's copy path (SDMA / blit + cross-queue signalling), which with the HIP runtime torch
// bundles (7.0) cost 8 ms per 1024-proof batch against 0 with ROCm 7.2's -- the same-box A/B is in profiles/r3_*.
// Single-wave workgroups: a 256-thread workgroup needs four free wave slots on one CU at the same instant, which the
// single-wave MSM workgroups streaming through the chip rarely leave (rocprofv3: 3.0 ms on average, 26.7 ms at worst
// for this 30 us copy when it was launched as 256-thread workgroups).
__global__ void __launch_bounds__(64) k_stage_in(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16) {
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

// Zeroisation of what a finished batch leaves behind.  The reference zeroises the identity secret wherever it holds it
// (IdSecret: Zeroize + ZeroizeOnDrop, rln/src/utils.rs:440-527) and the witness calculator's inputs buffer
// (circuit/iden3calc.rs:45-56).  Here the secret sits in the inputs of the batch and in its witness values: columns
// [0, n) of the stored rows of V (rows == nullptr: every row) and of V29.
__global__ void __launch_bounds__(64) k_wipe_cols(Fr* __restrict__ V, const uint32_t* __restrict__ rows, uint32_t nrows,
                                                  uint32_t B, uint32_t n) {
  const uint32_t p = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y;
  if (p >= n || r >= nrows) return;
  V[(size_t)(rows ? rows[r] : r) * B + p] = Fr::zero();
}
__global__ void __launch_bounds__(64) k_wipe_v29(uint4* __restrict__ V29, uint32_t nrows, uint32_t B, uint32_t n) {
  const uint32_t j = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y;
  if (j >= 3 * n || r >= nrows) return;
  V29[(size_t)r * B * 3 + j] = make_uint4(0, 0, 0, 0);
}

// =====================================================================================================
// host side
// =====================================================================================================
// Everything one in-flight batch owns.  Two slots let batch k+1 run its latency-bound front end (witness
// interpreter, NTT) and batch k-1 its back end (reduction, finalize) on their own streams while batch k
// keeps the chip busy with the MSM.
struct Slot {
  DevBuf<uint32_t> err, coords, values;
  DevBuf<uint8_t> comp;
  DevBuf<Fr> V, abc;
  DevBuf<uint4> V29;              // Fr29 interpreter: stored node values, [slot][proof][12 words]
  DevBuf<int16_t> digits, digits2;  // signed window digits under the G1 / G2 schedule
  DevBuf<G1XYZZ> part1, grp1, sums1, prod, tbl;
  DevBuf<G2XYZZ> part2, grp2, sums2;
  DevBuf<G1Affine> affA, affB1;
  DevBuf<G2Affine> affB2;
  DevBuf<uint32_t> pp_out;       // partial mode output, 320 B per proof
  uint32_t* h_pp = nullptr;
  int mode = 0;
  // streamed batches (Prover::submit): every slot owns its inputs, (r, s) and partial points plus the pinned staging
  // buffer they are copied from, so a caller with a stream of distinct batches never drains the pipeline
  DevBuf<uint32_t> inputs, rs, pp_in;
  uint8_t* h_in = nullptr;      // pinned staging: inputs | rs | partial points
  hipEvent_t evU = nullptr;     // H2D of this slot's inputs done
  hipEvent_t evE = nullptr;     // small batches: the walk of the h-independent G1 rows done
  uint64_t ticket = 0;          // submit() ticket of the batch the slot holds (0: resident-input run)
  uint8_t* h_comp = nullptr;    // pinned: every run ends with the proofs + values copied to the host
  uint32_t* h_values = nullptr;
  uint32_t* h_err = nullptr;
  hipEvent_t evA = nullptr, evB = nullptr, evB2 = nullptr, evR = nullptr, evC = nullptr, evW = nullptr, evV = nullptr;
  hipEvent_t evX = nullptr;     // the witness is in V (before the small-batch recodes that follow it on the same stream)
  hipEvent_t t[15] = {};  // timing marks
  bool used = false;
  bool marked = false;          // the timing marks t[] of the slot's batch were recorded
  bool wiped = false;           // the batch's inputs and witness values have been overwritten (Prover::wipe)
  size_t n = 0;
};

struct Prover::Impl {
  hipStream_t sA = nullptr, sAb = nullptr, sA2 = nullptr, sB = nullptr, sB2 = nullptr, sC = nullptr;
  hipStream_t sV[2] = {nullptr, nullptr};  // proof values (24 chained Poseidon hashes per proof, latency-bound)
  int wstreams = 2;  // graph interpreters in flight (RLNAMD_WSTREAMS): 16 latency-bound waves each
  uint32_t seq = 0;  // batches enqueued: consecutive front ends alternate between sA and sA2
  bool split_msm = true;   // G2 walk on its own stream: its workgroups fill the G1 kernel's tail (RLNAMD_MSM_SPLIT)
  float ms[PROVER_STAGES] = {0};
  DevBuf<unsigned long long> walk_clk;  // clock tap of the two walks: G1 cycles, G1 ticks, G2 cycles, G2 ticks
  bool wit29 = true;             // RLNAMD_WIT29: graph interpreter in the 9 x 29-bit form (k_witness29)
  uint32_t lanechunk_max = 128;  // RLNAMD_LANECHUNK: largest batch that takes the small-batch shapes
  uint32_t lanechunk_walk_max = 48;  // RLNAMD_LANECHUNK_WALK: largest lone batch whose walks run with lanes = chunks
  uint32_t witlanes_max = 256;   // RLNAMD_WITLANES_MAX: largest batch interpreted with lanes = nodes (a wave and 157 KB of LDS per proof)
  DevBuf<GNode29> nodes29;
  DevBuf<unsigned long long> wit_prof;
  DevBuf<uint32_t> consts29, slot2node;
  WitLanes witlanes;             // lanes = independent nodes: the interpreter of batches walked with lanes = chunks
  uint32_t nstore29 = 0, nprog29 = 0;   // stored values, program nodes (after fusion)

  uint32_t N = 0, NS = 0, NI = 0, nc = 0, ni = 0, n = 0;
  int logn = 0;
  DevBuf<GNode> nodes;
  DevBuf<Fr> consts;
  DevBuf<uint32_t> sig2node;
  DevBuf<uint32_t> a_ptr, a_col, b_ptr, b_col;
  DevBuf<uint32_t> mv_long;       // rows with more than MV_LONG entries in A or B
  uint32_t n_mv_long = 0;
  DevBuf<Fr> a_coef, b_coef;
  DevBuf<Fr> tw_f, tw_i, coset;
  DevBuf<uint32_t> tw_f29, tw_i29, coset29;  // the same constants as Fr29 (9 words each) for Fr29::mul_mont
  // MSM
  DevBuf<G1Affine> t1;
  DevBuf<G1Affine29> t1_29;  // G1 table in the 9 x 29-bit form (default; RLNAMD_FQ29=0 keeps the 8 x 32 walk)
  bool use29 = true;
  DevBuf<G2Affine29> t2_29;
  bool use29_g2 = true;
  DevBuf<G2Affine> t2;
  DevBuf<uint32_t> sid1, sid2;
  // a walk = a list of table rows cut into chunks, plus the two-level reduction ranges; one per mode
  struct Plan {
    DevBuf<uint32_t> rows;
    DevBuf<ChunkDesc> chunks, groups, segs, segchunks;   // segchunks: the chunk range of every segment (k_sum_tree)
    DevBuf<uint32_t> rsid;                  // scalar id of every entry of `rows`
    DevBuf<uint32_t> early_ids, late_ids;   // chunk indices without / with rows that depend on the quotient h
    uint32_t nchunks = 0, ngroups = 0, nseg = 0, n_early = 0, n_late = 0;
  };
  Plan plan1[3], plan2[3];  // [PROVE_FULL, PROVE_PARTIAL, PROVE_FINISH]
  // the same walks cut into shorter chunks for batches walked with lanes = chunks: a walk lasts as long as its longest
  // chunk (a lane's serial chain of additions), and a handful of proofs cannot fill the chip anyway
  Plan plan1s[3], plan2s[3];
  Plan plan1f[3];               // [PROVE_FULL] only: the fused small-batch plan (s A and r B1 as rows of the C segment)
  uint32_t max_chunks1s = 0, max_chunks2s = 0, small_stride = 64;   // partial sums of a small batch: [chunk][64]
  uint32_t max_chunks1 = 0, max_chunks2 = 0, max_groups1 = 0, max_groups2 = 0;
  uint32_t npts1 = 0, npts2 = 0;
  std::vector<uint8_t> known;  // per witness signal: computable from the partial witness (evaluate_partial)
  DevBuf<uint32_t> pp_in;      // resident partial-proof points for finish mode, 320 B per proof
  DevBuf<uint32_t> wgiven;     // externally calculated witnesses for the next run (upload_witness), else empty
  size_t wgiven_n = 0;
  InputSlots slots{};
  bool have_values_kernel = false;
  // resident inputs (shared by both slots; upload() drains the pipeline first)
  DevBuf<uint32_t> inputs, rs;
  static constexpr int NSLOT = 6;
  Slot slot[NSLOT];
  int nslot = 5, nstreamA = 2;  // RLNAMD_SLOTS / RLNAMD_ASTREAMS
  WinSched ws{}, ws2{};         // window schedules of the G1 and G2 comb tables
  uint32_t nh = 2;              // halves per scalar: 2 = GLV split (k1 + lambda k2), 1 = plain 254-bit walk
  bool recode_front = true;     // RLNAMD_RECODE_FRONT
  bool ntt29 = false;           // RLNAMD_NTT29: NTT products through Fr29::mul_mont (measured: no gain, see launch site)
  uint32_t msm_lds = 0;         // RLNAMD_MSM_WAVES (waves per SIMD the G1 walk may occupy; 0 = no cap)
  int cur = 0;
  Slot* last = nullptr;
  uint64_t tickets = 0;         // submit() tickets handed out

  // Overwrites what batch `S` knew about its witnesses (see k_wipe_cols): the slot's staged inputs (pinned host + device),
  // its (r, s) and the witness values, on the back-end stream behind the batch's last reader; evC is recorded again, so
  // whoever reuses the slot -- or reads the resident inputs next -- waits for the wipe as well.
  size_t batch_cap = 0;   // = Prover::B_
  void wipe_slot(Slot& S, bool resident) {
    const size_t n = S.n ? S.n : batch_cap;
    const size_t B = batch_cap;
    if (!n) return;
    RLN_HIP(hipStreamWaitEvent(sC, S.evC, 0));
    if (resident) {
      RLN_HIP(hipMemsetAsync(inputs.p, 0, std::min(inputs.bytes(), n * (size_t)NI * 32), sC));
      RLN_HIP(hipMemsetAsync(rs.p, 0, std::min(rs.bytes(), n * 64), sC));
      if (wgiven.p) RLN_HIP(hipMemsetAsync(wgiven.p, 0, wgiven.bytes(), sC));
      wgiven_n = 0;
    } else {
      volatile uint8_t* h = S.h_in;   // volatile: the stores may not be elided (explicit_bzero semantics)
      for (size_t i = 0; i < n * (size_t)NI * 32; i++) h[i] = 0;
      for (size_t i = 0; i < n * 64; i++) h[B * (size_t)NI * 32 + i] = 0;
      RLN_HIP(hipMemsetAsync(S.inputs.p, 0, n * (size_t)NI * 32, sC));
      RLN_HIP(hipMemsetAsync(S.rs.p, 0, n * 64, sC));
    }
    const uint32_t pg = div_up(n, 64);
    if (wit29 && S.V29.p) {
      hipLaunchKernelGGL(k_wipe_cols, dim3(pg, nstore29), dim3(64), 0, sC, S.V.p, slot2node.p, nstore29, (uint32_t)B, (uint32_t)n);
      hipLaunchKernelGGL(k_wipe_v29, dim3(div_up(3 * n, 64), nstore29 + 1), dim3(64), 0, sC, S.V29.p, nstore29 + 1, (uint32_t)B,
                         (uint32_t)n);
      // an externally supplied witness (upload_witness) was stored at the signal rows
      hipLaunchKernelGGL(k_wipe_cols, dim3(pg, NS), dim3(64), 0, sC, S.V.p, sig2node.p, NS, (uint32_t)B, (uint32_t)n);
    } else {
      hipLaunchKernelGGL(k_wipe_cols, dim3(pg, N), dim3(64), 0, sC, S.V.p, (const uint32_t*)nullptr, N, (uint32_t)B, (uint32_t)n);
    }
    RLN_HIP(hipGetLastError());
    RLN_HIP(hipEventRecord(S.evC, sC));
    S.wiped = true;
  }

  void sync_all() {
    RLN_HIP(hipStreamSynchronize(sA));
    RLN_HIP(hipStreamSynchronize(sAb));
    RLN_HIP(hipStreamSynchronize(sV[0]));
    RLN_HIP(hipStreamSynchronize(sV[1]));
    RLN_HIP(hipStreamSynchronize(sA2));
    RLN_HIP(hipStreamSynchronize(sB));
    if (sB2) RLN_HIP(hipStreamSynchronize(sB2));
    RLN_HIP(hipStreamSynchronize(sC));
  }
};

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

static uint32_t bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}

// chunks of one segment -> groups of <= 16 chunks -> the segment
static void make_reduce_ranges(const std::vector<uint32_t>& segfirst, std::vector<ChunkDesc>& groups,
                               std::vector<ChunkDesc>& segs) {
  const uint32_t G = 16;
  for (size_t sgi = 0; sgi + 1 < segfirst.size(); sgi++) {
    uint32_t g0 = (uint32_t)groups.size();
    for (uint32_t c = segfirst[sgi]; c < segfirst[sgi + 1]; c += G)
      groups.push_back({c, std::min(c + G, segfirst[sgi + 1])});
    segs.push_back({g0, (uint32_t)groups.size()});
  }
}

template <class F>
static void build_table(const std::vector<Affine<F>>& pts, const WinSched& ws, DevBuf<Affine<F>>& table, hipStream_t s) {
  size_t npts = pts.size();
  const size_t stride = ws.stride;
  table.alloc(npts * stride);
  DevBuf<Affine<F>> d_pts(npts);
  d_pts.upload(pts.data(), npts, s);
  // scratch is half a table; build in slabs of points so it never exceeds ~4 GiB
  size_t per_pt = stride / 2 * sizeof(F);
  size_t slab = std::max<size_t>(1, ((size_t)4 << 30) / per_pt);
  slab = std::min(slab, npts);
  DevBuf<F> scratch(slab * stride / 2);
  for (size_t k0 = 0; k0 < npts; k0 += slab) {
    size_t cnt = std::min(slab, npts - k0);
    size_t threads = cnt * ws.W;
    hipLaunchKernelGGL(k_table_build<F>, dim3(div_up(threads, 64)), dim3(64), 0, s, d_pts.p + k0, (uint32_t)cnt, ws,
                       table.p + k0 * stride, scratch.p);
    RLN_HIP(hipGetLastError());
  }
  RLN_HIP(hipStreamSynchronize(s));
}

// G1 table in the 9 x 29 form: slabs are built in the 8 x 32 form (k_table_build reads its own rows back) and
// converted into place
template <class F, class Entry>
static void build_table29(const std::vector<Affine<F>>& pts, const WinSched& ws, DevBuf<Entry>& table, hipStream_t s) {
  size_t npts = pts.size();
  const size_t stride = ws.stride;
  table.alloc(npts * stride);
  DevBuf<Affine<F>> d_pts(npts);
  d_pts.upload(pts.data(), npts, s);
  size_t per_pt = stride / 2 * sizeof(F);
  size_t slab = std::max<size_t>(1, ((size_t)4 << 30) / per_pt);
  slab = std::min(slab, npts);
  DevBuf<F> scratch(slab * stride / 2);
  DevBuf<Affine<F>> tmp(slab * stride);
  for (size_t k0 = 0; k0 < npts; k0 += slab) {
    size_t cnt = std::min(slab, npts - k0);
    size_t threads = cnt * ws.W;
    hipLaunchKernelGGL(k_table_build<F>, dim3(div_up(threads, 64)), dim3(64), 0, s, d_pts.p + k0, (uint32_t)cnt, ws,
                       tmp.p, scratch.p);
    hipLaunchKernelGGL((k_table_to29<Affine<F>, Entry>), dim3(div_up(cnt * stride, 256)), dim3(256), 0, s, tmp.p,
                       table.p + k0 * stride, cnt * stride);
    RLN_HIP(hipGetLastError());
  }
  RLN_HIP(hipStreamSynchronize(s));
}

// c-bit windows, the first `wide` of them one bit wider; W = the fewest windows that cover `total` bits: 255 for the
// plain walk (254-bit scalars plus the carry of the signed recoding), 127 for the halves of a GLV split (< 2^126)
static WinSched make_sched(int c, int wide, int total) {
  if (c < 2 || c > 16 || wide < 0 || c + (wide > 0 ? 1 : 0) > 16) throw Error("window bits must be in [2, 16]");
  WinSched ws{};
  int W = (total - wide + c - 1) / c;
  if (wide > W) throw Error("more wide windows than windows");
  if (W > 32) throw Error("window bits too small: more than 32 windows");
  ws.W = W;
  uint32_t bit = 0, off = 0;
  for (int j = 0; j < W; j++) {
    int cw = c + (j < wide ? 1 : 0);
    ws.cw[j] = (uint8_t)cw;
    ws.bo[j] = (uint16_t)bit;
    ws.ro[j] = off;
    bit += cw;
    off += 1u << (cw - 1);
  }
  ws.stride = off;
  return ws;
}

Prover::Prover(const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len, ProverConfig cfg)
    : d_(new Impl) {
  require_gpu();
  zk_ = parse_arkzkey(zkey, zkey_len);
  graph_ = parse_graph(graph, graph_len);
  (void)prepared(zk_);  // verifier precomputation now, so concurrent verify calls only read it
  Impl& D = *d_;
  // window_bits = g1 + 10000 * g2, each spec = c + 100 * wide: c-bit windows, the first `wide` of them (c + 1)-bit
  // (see WinSched); g2 = 0: the G2 table takes the G1 schedule.  With the GLV split (default; RLNAMD_GLV=0 keeps the
  // plain 254-bit walk) the windows cover the 127-bit halves: spec 114 = 15 + 8 x 14 bits, 9 windows, 18 additions
  // per G1 point; spec 715 = 7 x 16 + 15 bits, 8 windows, 16 additions per G2 point.
  const int wb = cfg.window_bits > 0 ? cfg.window_bits : env_int("RLNAMD_WINDOW_BITS", 8);
  const int spec1 = wb % 10000, spec2 = wb / 10000 ? wb / 10000 : spec1;
  D.nh = env_int("RLNAMD_GLV", 1) != 0 ? 2 : 1;
  const int total = D.nh == 2 ? GlvParams::HALF_BITS : 255;
  c_ = spec1 % 100;
  const int wide = spec1 >= 100 ? spec1 / 100 : env_int("RLNAMD_WINDOW_WIDE", 0);
  D.ws = make_sched(c_, wide, total);
  D.ws2 = make_sched(spec2 % 100, spec2 >= 100 ? spec2 / 100 : (wb / 10000 ? 0 : wide), total);
  W_ = D.ws.W * D.nh;
  c2_ = spec2 % 100;
  W2_ = D.ws2.W * D.nh;
  glv_ = D.nh == 2;
  B_ = ((cfg.max_batch ? cfg.max_batch : 1) + 63) / 64 * 64;
  D.batch_cap = B_;

  // ---- consistency between zkey and graph (what arkworks asserts inside the prover)
  D.N = (uint32_t)graph_.nodes.size();
  D.NS = (uint32_t)graph_.signals.size();
  D.NI = graph_.inputs_size;
  D.nc = (uint32_t)zk_.num_constraints;
  D.ni = (uint32_t)zk_.num_instance_variables;
  if (zk_.a_query.size() != D.NS || zk_.b_g1_query.size() != D.NS || zk_.b_g2_query.size() != D.NS)
    throw Error("zkey/graph mismatch: query length != number of witness signals");
  if (zk_.gamma_abc_g1.size() != D.ni || zk_.l_query.size() + D.ni != D.NS)
    throw Error("MalformedVerifyingKey: instance/aux split does not match the witness length");
  uint32_t dom = 1;
  D.logn = 0;
  while (dom < D.nc + D.ni) {
    dom <<= 1;
    D.logn++;
  }
  D.n = dom;
  if (D.logn < 1 || D.logn > 27) throw Error("PolynomialDegreeTooLarge");
  if (zk_.h_query.size() < D.n) throw Error("zkey h_query shorter than the evaluation domain");

  {
    // the short latency-bound stages get the high-priority queues so their few waves are dispatched ahead
    // of the MSM's thousands of workgroups
    int lo = 0, hi = 0;
    RLN_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char* pe = getenv("RLNAMD_PRIO");  // experiment knob: three chars h/l for streams A, B, C
    auto pick = [&](int i, int dflt) { return (pe && strlen(pe) == 3) ? (pe[i] == 'h' ? hi : lo) : dflt; };
    RLN_HIP(hipStreamCreateWithPriority(&D.sA, hipStreamNonBlocking, pick(0, hi)));
    RLN_HIP(hipStreamCreateWithPriority(&D.sA2, hipStreamNonBlocking, pick(0, hi)));
    RLN_HIP(hipStreamCreateWithPriority(&D.sAb, hipStreamNonBlocking, pick(0, hi)));
    for (auto& v : D.sV) RLN_HIP(hipStreamCreateWithPriority(&v, hipStreamNonBlocking, pick(2, hi)));
    D.wstreams = env_int("RLNAMD_WSTREAMS", 2);
    RLN_HIP(hipStreamCreateWithPriority(&D.sB, hipStreamNonBlocking, pick(1, lo)));
    RLN_HIP(hipStreamCreateWithPriority(&D.sC, hipStreamNonBlocking, pick(2, hi)));
    D.nslot = std::min(std::max(env_int("RLNAMD_SLOTS", 5), 2), (int)Impl::NSLOT);
    {
      int mw = env_int("RLNAMD_MSM_WAVES", 0);
      D.msm_lds = mw > 0 ? (uint32_t)(160 * 1024 / (4 * mw)) & ~255u : 0;
    }
    D.nstreamA = env_int("RLNAMD_ASTREAMS", 2);
    D.use29 = env_int("RLNAMD_FQ29", 1) != 0;
    D.use29_g2 = env_int("RLNAMD_FQ29_G2", D.use29 ? 1 : 0) != 0;
    D.recode_front = env_int("RLNAMD_RECODE_FRONT", 1) != 0;
    D.ntt29 = env_int("RLNAMD_NTT29", 0) != 0;
    // the G1 walk is ~12 rounds of 2.8 ms workgroups: on one stream its last round leaves SIMDs idle until the G2
    // walk may start; on two streams the walks of neighbouring batches fill each other's tails (+3.3 - 3.7 % measured)
    D.split_msm = env_int("RLNAMD_MSM_SPLIT", 1) != 0;
    D.walk_clk.alloc(4);
    RLN_HIP(hipMemset(D.walk_clk.p, 0, 4 * sizeof(unsigned long long)));
    if (D.split_msm) RLN_HIP(hipStreamCreateWithPriority(&D.sB2, hipStreamNonBlocking, pick(1, lo)));
  }
  hipStream_t s = D.sB;

  // ---- graph program
  D.nodes.alloc(D.N);
  {
    // device program: operands tagged with where the interpreter finds them (k_witness)
    std::vector<GNode> prog(graph_.nodes);
    auto enc = [&](uint32_t n, uint32_t o) -> uint32_t {
      if (o >= n) throw Error("Graph error: node operand refers forward");
      if (graph_.nodes[o].op == G_CONST) return OPK_CONST | graph_.nodes[o].a;
      if (n - o < WIT_RING) return OPK_RING | o;
      return OPK_FAR | o;
    };
    std::vector<uint8_t> store(D.N, 0);
    for (uint32_t sgn : graph_.signals) store[sgn] = 1;
    for (uint32_t n = 0; n < D.N; n++) {
      GNode& g = prog[n];
      if (g.op == G_INPUT) store[n] = 1;
      if (g.op == G_INPUT || g.op == G_CONST) continue;
      const uint32_t oa = g.a, ob = g.b, oc = g.c;
      g.a = enc(n, oa);
      if ((g.a & OPK_MASK) == OPK_FAR) store[oa] = 1;
      if (g.op != G_NEG && g.op != G_ID) {
        g.b = enc(n, ob);
        if ((g.b & OPK_MASK) == OPK_FAR) store[ob] = 1;
      }
      if (g.op == G_TERN) {
        g.c = enc(n, oc);
        if ((g.c & OPK_MASK) == OPK_FAR) store[oc] = 1;
      }
    }
    for (uint32_t n = 0; n < D.N; n++)
      if (store[n]) prog[n].op |= G_STORE;
    D.nodes.upload(prog.data(), D.N, s);
    RLN_HIP(hipStreamSynchronize(s));
  }
  RLN_HIP(hipFuncSetAttribute((const void*)k_witness, hipFuncAttributeMaxDynamicSharedMemorySize, WIT_RING * 8 * 64 * 4 + WIT_LDS_CONSTS * 32));
  D.consts.alloc(std::max<size_t>(graph_.constants.size(), 1));
  if (!graph_.constants.empty()) D.consts.upload(graph_.constants.data(), graph_.constants.size(), s);
  D.wit29 = env_int("RLNAMD_WIT29", 1) != 0;
  // Largest batch that takes the small-batch shapes (lanes = chunks walks, a wave per proof in the interpreter, early walks
  // and back end).  tools/lanechunk_sweep.py / tools/midstream.py: one batch alone is faster that way up to ~450 proofs
  // (64: 12.4 vs 23.2 ms, 128: 17.8 vs 27.4, 256: 28.4 vs 36.8), a STREAM of such batches up to ~150 (chunks of 64: 9.4 k
  // vs 8.1 k proofs/s, 128: equal, 256: 10.1 k vs 12.1 k) -- 128 wins or ties on both.
  D.lanechunk_max = (uint32_t)std::max(0, env_int("RLNAMD_LANECHUNK", 128));
  D.witlanes_max = (uint32_t)std::max(0, env_int("RLNAMD_WITLANES_MAX", 256));
  D.lanechunk_walk_max = (uint32_t)std::max(0, env_int("RLNAMD_LANECHUNK_WALK", 48));
  // partial sums of a small batch: [chunk][stride]
  D.small_stride = std::max<uint32_t>(64, (std::min<uint32_t>(D.lanechunk_max, (uint32_t)B_) + 63) / 64 * 64);
  std::vector<GNode29> wit29_prog;
  std::vector<uint32_t> wit29_slot2node;
  if (D.wit29) {
    // The program of k_witness29.  (1) Fusion: an Add one of whose operands is a product used nowhere else (and is no
    // witness signal) becomes ONE node, a * b + c (W29_FMA: the addend enters the product's final carry chain,
    // Fr29::mul_add) -- in the shipped circuits every addition of a Poseidon round is of that kind, 23 414 nodes become
    // ~15 000.  (2) Program order = node order without the fused products; the LDS ring is addressed by program
    // index.  (3) Stored values (witness signals, inputs, operands further back than the ring) live in a compact array
    // indexed by `slot`.  (4) W29_RED where the static bound of a value (in units of r) would pass WIT29_BMAX.
    const uint32_t NONE = 0xFFFFFFFFu;
    const std::vector<GNode>& G = graph_.nodes;
    auto is_const = [&](uint32_t o) { return G[o].op == G_CONST; };
    auto nops = [&](const GNode& g) {
      return (g.op == G_INPUT || g.op == G_CONST) ? 0 : (g.op == G_NEG || g.op == G_ID) ? 1 : g.op == G_TERN ? 3 : 2;
    };
    std::vector<uint32_t> uses(D.N, 0);
    for (uint32_t n = 0; n < D.N; n++) {
      const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
      for (int k = 0; k < nops(G[n]); k++) {
        if (o[k] >= n) throw Error("Graph error: node operand refers forward");
        uses[o[k]]++;
      }
    }
    std::vector<uint8_t> is_signal(D.N, 0);
    for (uint32_t sgn : graph_.signals) is_signal[sgn] = 1;
    std::vector<uint32_t> fused_mul(D.N, NONE);   // for an Add: the product folded into it
    std::vector<uint8_t> removed(D.N, 0);
    const bool fuse = env_int("RLNAMD_WIT29_FUSE", 1) != 0;
    for (uint32_t n = 0; fuse && n < D.N; n++) {
      if (G[n].op != G_ADD) continue;
      for (uint32_t m : {G[n].b, G[n].a}) {
        if (G[m].op == G_MUL && uses[m] == 1 && !is_signal[m] && !removed[m] && G[n].a != G[n].b) {
          fused_mul[n] = m;
          removed[m] = 1;
          break;
        }
      }
    }
    // program nodes: operands as ORIGINAL node ids
    struct PNode { uint32_t op, node, src[3]; };
    std::vector<PNode> P;
    std::vector<uint32_t> pidx(D.N, NONE);
    for (uint32_t n = 0; n < D.N; n++) {
      if (removed[n]) continue;
      PNode q{G[n].op, n, {G[n].a, G[n].b, G[n].c}};
      if (fused_mul[n] != NONE) {
        const uint32_t m = fused_mul[n];
        q.op = W29_FMA;
        q.src[0] = G[m].a;
        q.src[1] = G[m].b;
        q.src[2] = G[n].a == m ? G[n].b : G[n].a;
      }
      pidx[n] = (uint32_t)P.size();
      P.push_back(q);
    }
    auto pn_ops = [&](const PNode& q) { return q.op == W29_FMA ? 3 : nops(GNode{q.op, 0, 0, 0}); };
    std::vector<uint8_t> store(D.N, 0);
    for (uint32_t n = 0; n < D.N; n++) store[n] = is_signal[n] || G[n].op == G_INPUT;
    for (uint32_t i = 0; i < P.size(); i++)
      for (int k = 0; k < pn_ops(P[i]); k++) {
        const uint32_t o = P[i].src[k];
        if (!is_const(o) && i - pidx[o] >= WIT29_RING) store[o] = 1;
      }
    std::vector<uint32_t> slot_of(D.N, 0);
    std::vector<uint32_t>& slot2node = wit29_slot2node;
    for (uint32_t n = 0; n < D.N; n++)
      if (store[n] && !removed[n]) {
        slot_of[n] = (uint32_t)slot2node.size();
        slot2node.push_back(n);
      }
    if (slot2node.size() >= 65536) D.wit29 = false;   // the descriptor has 16 bits for the slot: larger graphs keep k_witness
    std::vector<GNode29>& prog = wit29_prog;
    prog.resize(P.size());
    std::vector<double> bnd(D.N, 1.01);
    for (uint32_t i = 0; D.wit29 && i < P.size(); i++) {
      const PNode& q = P[i];
      GNode29 d{};
      uint32_t flags = store[q.node] ? W29_STORE : 0;
      d.a = q.src[0];   // G_INPUT: input index, G_CONST: constant index
      double b = 1.01;  // inputs, constants, slow operations: a fresh product with a constant
      // the fast path of the kernel: Mul / Add / a * b + c with every operand in LDS (ring or constant table)
      bool rare = q.op != G_MUL && q.op != G_ADD && q.op != W29_FMA;
      if (q.op != G_INPUT && q.op != G_CONST) {
        auto enc = [&](uint32_t o) -> uint32_t {
          if (is_const(o)) {
            if (G[o].a >= WIT29_LDS_CONSTS) rare = true;
            return OPK_CONST | G[o].a;
          }
          if (i - pidx[o] < WIT29_RING) return OPK_RING | pidx[o];
          rare = true;
          return OPK_FAR | slot_of[o];
        };
        auto bo = [&](uint32_t o) { return is_const(o) ? 1.01 : bnd[o]; };
        const int k = pn_ops(q);
        double bs[3] = {0, 0, 0};
        uint32_t e[3] = {0, 0, 0};
        for (int j = 0; j < k; j++) {
          e[j] = enc(q.src[j]);
          bs[j] = bo(q.src[j]);
        }
        d.a = e[0];
        d.b = e[1];
        d.c = e[2];
        if (q.op == G_MUL) b = 1.0 + 0.006 * bs[0] * bs[1];
        else if (q.op == W29_FMA) b = 1.0 + 0.006 * bs[0] * bs[1] + bs[2];
        else if (q.op == G_ADD) b = bs[0] + bs[1];
        else if (q.op == G_SUB) b = bs[0] + 8.0;
        else if (q.op == G_NEG) b = 8.0;
        else if (q.op == G_TERN) b = std::max(bs[1], bs[2]);
      }
      if (b > WIT29_BMAX) {
        flags |= W29_RED;
        rare = true;
        b = 1.0 + 0.006 * b;
      }
      if (rare) flags |= W29_RARE;
      bnd[q.node] = b;
      d.w0 = q.op | flags | (slot_of[q.node] << 16);
      prog[i] = d;
    }
  }
  if (D.wit29) {
    std::vector<GNode29>& prog = wit29_prog;
    std::vector<uint32_t>& slot2node = wit29_slot2node;
    D.nprog29 = (uint32_t)prog.size();
    D.nstore29 = (uint32_t)slot2node.size();
    prog.resize(((size_t)D.nprog29 / WIT29_CH + 4) * WIT29_CH, GNode29{});   // the kernel prefetches two chunks past the end
    D.nodes29.alloc(prog.size());
    D.nodes29.upload(prog.data(), prog.size(), s);
    D.slot2node.alloc(std::max<size_t>(slot2node.size(), 1));
    if (!slot2node.empty()) D.slot2node.upload(slot2node.data(), slot2node.size(), s);
    D.consts29.alloc(std::max<size_t>(graph_.constants.size(), 1) * 9);
    if (!graph_.constants.empty())
      hipLaunchKernelGGL(k_consts_to29, dim3(div_up(graph_.constants.size(), 256)), dim3(256), 0, s, D.consts.p,
                         D.consts29.p, (uint32_t)graph_.constants.size());
    RLN_HIP(hipGetLastError());
    RLN_HIP(hipStreamSynchronize(s));
    {   // the same stored values, produced by a wave per proof (witness_lanes.h) when a batch is below a wave of proofs
      std::vector<uint32_t> store_slot(D.N, 0xFFFFFFFFu);
      for (uint32_t i = 0; i < slot2node.size(); i++) store_slot[slot2node[i]] = i;
      D.witlanes.build(graph_, store_slot, (uint32_t)slot2node.size(), s);   // V29 has one row more than stored values
    }
    // 152 KiB of dynamic LDS: a device / partition with a smaller limit keeps the 8 x 32 interpreter (k_witness), the
    // same fallback as for graphs with 65 536 or more stored values -- a resource limit must not fail the constructor
    if (hipFuncSetAttribute((const void*)k_witness29<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            WIT29_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_witness29<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            WIT29_LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      D.wit29 = false;
      D.witlanes.ok = false;
    }
  }
  D.sig2node.alloc(D.NS);
  D.sig2node.upload(graph_.signals.data(), D.NS, s);

  // ---- matrices as CSR over graph node ids
  auto csr = [&](const std::vector<SparseRow>& m, DevBuf<uint32_t>& ptr, DevBuf<uint32_t>& col, DevBuf<Fr>& coef) {
    std::vector<uint32_t> hp(D.nc + 1, 0), hc;
    std::vector<Fr> hv;
    for (uint32_t r = 0; r < D.nc; r++) {
      for (size_t k = 0; k < m[r].col.size(); k++) {
        hc.push_back(graph_.signals[m[r].col[k]]);
        hv.push_back(m[r].coeff[k]);
      }
      hp[r + 1] = (uint32_t)hc.size();
    }
    ptr.alloc(hp.size());
    ptr.upload(hp.data(), hp.size(), s);
    col.alloc(std::max<size_t>(hc.size(), 1));
    coef.alloc(std::max<size_t>(hv.size(), 1));
    if (!hc.empty()) {
      col.upload(hc.data(), hc.size(), s);
      coef.upload(hv.data(), hv.size(), s);
    }
    RLN_HIP(hipStreamSynchronize(s));
  };
  csr(zk_.a, D.a_ptr, D.a_col, D.a_coef);
  csr(zk_.b, D.b_ptr, D.b_col, D.b_coef);
  {   // rows the small-batch mat-vec gives a wave each (k_matvec)
    std::vector<uint32_t> lr;
    for (uint32_t r = 0; r < D.nc; r++)
      if (zk_.a[r].col.size() > MV_LONG || zk_.b[r].col.size() > MV_LONG) lr.push_back(r);
    D.n_mv_long = (uint32_t)lr.size();
    D.mv_long.alloc(std::max<size_t>(lr.size(), 1));
    if (!lr.empty()) D.mv_long.upload(lr.data(), lr.size(), s);
    RLN_HIP(hipStreamSynchronize(s));
  }

  // ---- NTT tables: w = W^(2^(28-logn)), g = root of the doubled domain, coset[pos] = g^bitrev(pos) / n
  {
    Fr root28 = Fr::from_canonical(FR_ROOT_2_28);
    Fr g = root28;
    for (int i = 0; i < 28 - (D.logn + 1); i++) g = g.sqr();  // order 2n
    Fr w = g.sqr();                                           // order n
    Fr wi = w.inv();
    std::vector<Fr> tf(D.n / 2), ti(D.n / 2), cs(D.n);
    Fr a = Fr::one(), b = Fr::one();
    for (uint32_t k = 0; k < D.n / 2; k++) {
      tf[k] = a;
      ti[k] = b;
      a = a * w;
      b = b * wi;
    }
    Fr ninv = Fr::from_u32(D.n).inv();
    std::vector<Fr> gp(D.n);
    Fr acc = ninv;
    for (uint32_t i = 0; i < D.n; i++) {
      gp[i] = acc;
      acc = acc * g;
    }
    for (uint32_t pos = 0; pos < D.n; pos++) cs[pos] = gp[bitrev(pos, D.logn)];
    D.tw_f.alloc(tf.size());
    D.tw_i.alloc(ti.size());
    D.coset.alloc(cs.size());
    D.tw_f.upload(tf.data(), tf.size(), s);
    D.tw_i.upload(ti.data(), ti.size(), s);
    D.coset.upload(cs.data(), cs.size(), s);
    D.tw_f29.alloc(tf.size() * 9);
    D.tw_i29.alloc(ti.size() * 9);
    D.coset29.alloc(cs.size() * 9);
    hipLaunchKernelGGL(k_consts_to29, dim3(div_up(tf.size(), 256)), dim3(256), 0, s, D.tw_f.p, D.tw_f29.p, (uint32_t)tf.size());
    hipLaunchKernelGGL(k_consts_to29, dim3(div_up(ti.size(), 256)), dim3(256), 0, s, D.tw_i.p, D.tw_i29.p, (uint32_t)ti.size());
    hipLaunchKernelGGL(k_consts_to29, dim3(div_up(cs.size(), 256)), dim3(256), 0, s, D.coset.p, D.coset29.p, (uint32_t)cs.size());
    RLN_HIP(hipGetLastError());
    RLN_HIP(hipStreamSynchronize(s));
  }

  // ---- which witness signals are fixed by the partial witness (evaluate_partial, graph.rs:274-312): a node
  //      is known iff all its operands are; the unknown inputs are the per-message ones
  //      (inputs_for_partial_witness_calculation, witness.rs:887-937)
  {
    std::vector<uint8_t> in_known(D.NI, 1), node_known(D.N, 0);
    for (const char* name : {"messageId", "selectorUsed", "x", "externalNullifier"}) {
      auto it = graph_.input_mapping.find(name);
      if (it == graph_.input_mapping.end()) continue;
      for (uint32_t k = 0; k < it->second.second; k++) in_known[it->second.first + k] = 0;
    }
    for (uint32_t i = 0; i < D.N; i++) {
      const GNode& nd = graph_.nodes[i];
      bool k;
      if (nd.op == G_INPUT) k = in_known[nd.a];
      else if (nd.op == G_CONST) k = true;
      else if (nd.op == G_NEG || nd.op == G_ID) k = node_known[nd.a];
      else if (nd.op == G_TERN) k = node_known[nd.a] && node_known[nd.b] && node_known[nd.c];
      else k = node_known[nd.a] && node_known[nd.b];
      node_known[i] = k;
    }
    D.known.resize(D.NS);
    for (uint32_t i = 0; i < D.NS; i++) D.known[i] = node_known[graph_.signals[i]];
  }

  // ---- MSM segments.  Scalar ids: [0, NS) witness, [NS, NS+n) h, then r, s, -(r s).
  //      Every table row belongs to one output segment; the three walks are subsets of the rows:
  //      full = all, partial = rows whose scalar is a known witness signal (incl. w_0 = 1, which carries
  //      alpha / beta / query[0]), finish = the rest (unknown signals, h, blinding terms).
  const uint32_t SID_R = D.NS + D.n, SID_S = SID_R + 1, SID_NRS = SID_R + 2;
  // A walk = a list of (table row, scalar id, output segment) entries cut into chunks.  `dig_sid` = the id the digits
  // of an entry live under (G2: the ids above the h block move down), `is_h` = the scalar is a coefficient of h.
  struct VRow { uint32_t k, sid, dig_sid, seg; bool is_h; };
  auto make_plans = [&](const std::vector<VRow>& vrows, uint32_t nseg, uint32_t chunk_pts, Impl::Plan* plans,
                        uint32_t* max_chunks, uint32_t* max_groups, int only_mode) {
    for (int mode = 0; mode < 3; mode++) {
      if (only_mode >= 0 && mode != only_mode) continue;
      std::vector<uint32_t> rows, rsid, segfirst, early_ids, late_ids;
      std::vector<ChunkDesc> chunks;
      // reduction segment h * nseg + sg: the rows of output sg walked with GLV half h (bit 31 of the row entry)
      for (uint32_t h = 0; h < D.nh; h++)
        for (uint32_t sg = 0; sg < nseg; sg++) {
          segfirst.push_back((uint32_t)chunks.size());
          // rows whose scalar is a coefficient of h come last and start a chunk of their own, so that a small batch can
          // walk everything else while the NTTs still run (early_ids / late_ids)
          for (int late = 0; late < 2; late++) {
            uint32_t first = (uint32_t)rows.size();
            for (const VRow& v : vrows) {
              if (v.seg != sg || (int)v.is_h != late) continue;
              bool is_known = v.sid < D.NS && D.known[v.sid];
              if (mode == PROVE_FULL || (mode == PROVE_PARTIAL) == is_known) {
                rows.push_back(v.k | (h << 31));
                rsid.push_back(v.dig_sid);
              }
            }
            for (uint32_t k = first; k < rows.size(); k += chunk_pts) {
              (late ? late_ids : early_ids).push_back((uint32_t)chunks.size());
              chunks.push_back({k, (uint32_t)std::min<size_t>(k + chunk_pts, rows.size())});
            }
          }
        }
      segfirst.push_back((uint32_t)chunks.size());
      std::vector<ChunkDesc> groups, segs;
      make_reduce_ranges(segfirst, groups, segs);
      Impl::Plan& P = plans[mode];
      P.nchunks = (uint32_t)chunks.size();
      P.ngroups = (uint32_t)groups.size();
      P.nseg = nseg * D.nh;
      P.rows.alloc(std::max<size_t>(rows.size(), 1));
      P.rsid.alloc(std::max<size_t>(rsid.size(), 1));
      if (!rsid.empty()) P.rsid.upload(rsid.data(), rsid.size(), s);
      P.chunks.alloc(std::max<size_t>(chunks.size(), 1));
      P.groups.alloc(std::max<size_t>(groups.size(), 1));
      P.segs.alloc(segs.size());
      if (!rows.empty()) P.rows.upload(rows.data(), rows.size(), s);
      if (!chunks.empty()) P.chunks.upload(chunks.data(), chunks.size(), s);
      if (!groups.empty()) P.groups.upload(groups.data(), groups.size(), s);
      P.segs.upload(segs.data(), segs.size(), s);
      std::vector<ChunkDesc> segchunks;
      for (size_t sgi = 0; sgi + 1 < segfirst.size(); sgi++) segchunks.push_back({segfirst[sgi], segfirst[sgi + 1]});
      P.segchunks.alloc(segchunks.size());
      P.segchunks.upload(segchunks.data(), segchunks.size(), s);
      P.n_early = (uint32_t)early_ids.size();
      P.n_late = (uint32_t)late_ids.size();
      P.early_ids.alloc(std::max<size_t>(early_ids.size(), 1));
      P.late_ids.alloc(std::max<size_t>(late_ids.size(), 1));
      if (!early_ids.empty()) P.early_ids.upload(early_ids.data(), early_ids.size(), s);
      if (!late_ids.empty()) P.late_ids.upload(late_ids.data(), late_ids.size(), s);
      RLN_HIP(hipStreamSynchronize(s));
      *max_chunks = std::max(*max_chunks, P.nchunks);
      *max_groups = std::max(*max_groups, P.ngroups);
    }
  };
  {
    std::vector<G1Affine> pts;
    std::vector<uint32_t> sids, row_seg;
    auto push = [&](const G1Affine& P, uint32_t sid, uint32_t seg) {
      if (P.is_inf()) return;
      pts.push_back(P);
      sids.push_back(sid);
      row_seg.push_back(seg);
    };
    // seg 0: A = alpha + sum_i w_i A_i + r delta      (w_0 = 1 carries a_query[0] and alpha)
    for (uint32_t i = 0; i < D.NS; i++) push(zk_.a_query[i], i, 0);
    push(zk_.alpha_g1, 0, 0);
    push(zk_.delta_g1, SID_R, 0);
    // seg 1: B1 = beta + sum_i w_i B_i + s delta
    for (uint32_t i = 0; i < D.NS; i++) push(zk_.b_g1_query[i], i, 1);
    push(zk_.beta_g1, 0, 1);
    push(zk_.delta_g1, SID_S, 1);
    // seg 2: Cpart = sum_j w_(ni+j) L_j + sum_k h_k H_k - (r s) delta
    for (uint32_t j = 0; j < zk_.l_query.size(); j++) push(zk_.l_query[j], D.ni + j, 2);
    for (uint32_t k = 0; k < D.n; k++) push(zk_.h_query[k], D.NS + k, 2);
    push(zk_.delta_g1, SID_NRS, 2);
    D.npts1 = (uint32_t)pts.size();
    D.sid1.alloc(sids.size());
    D.sid1.upload(sids.data(), sids.size(), s);
    std::vector<VRow> vrows;
    for (uint32_t k = 0; k < sids.size(); k++)
      vrows.push_back({k, sids[k], sids[k], row_seg[k], sids[k] >= D.NS && sids[k] < D.NS + D.n});
    // rows (x halves) per single-wave workgroup: ~150 additions each, as before the split (8 rows x 19 windows)
    make_plans(vrows, 3, (uint32_t)std::max(1, env_int("RLNAMD_MSM_CHUNK", D.nh == 2 ? 16 : 8)), D.plan1, &D.max_chunks1,
               &D.max_groups1, -1);
    {
      uint32_t unused = 0;
      make_plans(vrows, 3, (uint32_t)std::max(1, env_int("RLNAMD_MSM_CHUNK_SMALL", 4)), D.plan1s, &D.max_chunks1s, &unused, -1);
    }
    {
      // Small full proofs, fused plan: s A + r B1 - r s delta = s alpha + r beta + r s delta + sum (s w_i) A_i + sum (r w_i) B1_i,
      // so the two variable-base products of the back end (k_fin_smul: a lone lane's ladder of 127 doublings, the longest
      // kernel behind the interpreter) become extra rows of the C segment -- the A and B1 rows walked a second time under
      // the scalar ids of s w_i and r w_i (k_recode part 3) -- and the B1 segment is not walked at all.  More additions
      // in total (+ 25 % G1 rows), which is why only batches below the small-batch threshold take this plan.
      std::vector<VRow> f;
      const uint32_t NX = D.NS + D.n + 3;   // first extra scalar id: s w_i at NX + i, r w_i at NX + NS + i, r s at NX + 2 NS
      for (uint32_t k = 0; k < sids.size(); k++) {
        const uint32_t sd = sids[k], sg = row_seg[k];
        const bool is_h = sd >= D.NS && sd < D.NS + D.n;
        if (sg == 0) {
          f.push_back({k, sd, sd, 0, false});                                   // A itself is an output
          if (sd < D.NS) f.push_back({k, sd, NX + sd, 2, false});               // (s w_i) A_i   (alpha carries sid 0: s alpha)
          // delta with r (part of A) contributes s r delta to s A: counted once below
        } else if (sg == 1) {
          if (sd < D.NS) f.push_back({k, sd, NX + D.NS + sd, 2, false});        // (r w_i) B1_i  (beta carries sid 0: r beta)
        } else if (sd == SID_NRS) {
          f.push_back({k, sd, NX + 2 * D.NS, 2, false});                        // + r s delta instead of - r s delta
        } else {
          f.push_back({k, sd, sd, 2, is_h});                                    // L and H rows
        }
      }
      uint32_t unused = 0;
      make_plans(f, 3, (uint32_t)std::max(1, env_int("RLNAMD_MSM_CHUNK_SMALL", 4)), D.plan1f, &D.max_chunks1s, &unused,
                 PROVE_FULL);
    }
    if (D.use29) build_table29<Fq, G1Affine29>(pts, D.ws, D.t1_29, s); else build_table<Fq>(pts, D.ws, D.t1, s);
  }
  {
    std::vector<G2Affine> pts;
    std::vector<uint32_t> sids, row_seg;
    auto push = [&](const G2Affine& P, uint32_t sid) {
      if (P.is_inf()) return;
      pts.push_back(P);
      sids.push_back(sid);
      row_seg.push_back(0);
    };
    for (uint32_t i = 0; i < D.NS; i++) push(zk_.b_g2_query[i], i);
    push(zk_.beta_g2, 0);
    push(zk_.delta_g2, SID_S);
    D.npts2 = (uint32_t)pts.size();
    // the G2 digit array holds the witness scalars and r, s, -(r s) only (k_recode): ids above the h block move down
    std::vector<uint32_t> dsid(sids);
    for (uint32_t& v : dsid)
      if (v >= D.NS) v -= D.n;
    D.sid2.alloc(dsid.size());
    D.sid2.upload(dsid.data(), dsid.size(), s);
    std::vector<VRow> vrows;
    for (uint32_t k = 0; k < sids.size(); k++) vrows.push_back({k, sids[k], dsid[k], 0u, false});
    make_plans(vrows, 1, (uint32_t)std::max(1, env_int("RLNAMD_MSM_CHUNK_G2", D.nh == 2 ? 8 : 4)), D.plan2, &D.max_chunks2,
               &D.max_groups2, -1);
    {
      uint32_t unused = 0;
      make_plans(vrows, 1, (uint32_t)std::max(1, env_int("RLNAMD_MSM_CHUNK_G2_SMALL", 2)), D.plan2s, &D.max_chunks2s, &unused, -1);
    }
    if (D.use29_g2) build_table29<Fq2, G2Affine29>(pts, D.ws2, D.t2_29, s); else build_table<Fq2>(pts, D.ws2, D.t2, s);
  }

  // ---- named input slots for the proof-values kernel (single message-id circuit, witness.rs:832-881);
  //      other circuits (multi message-id) take their public values from the witness instead
  D.have_values_kernel = false;
  {
    auto find = [&](const char* name, uint32_t want_len, uint32_t* off) {
      auto it = graph_.input_mapping.find(name);
      if (it == graph_.input_mapping.end() || it->second.second != want_len) return false;
      *off = it->second.first;
      return true;
    };
    D.slots.depth = graph_.tree_depth;
    bool ok = graph_.max_out == 1 && D.ni == 6;
    ok = ok && find("identitySecret", 1, &D.slots.secret) && find("userMessageLimit", 1, &D.slots.limit) &&
         find("messageId", 1, &D.slots.msg_id) && find("pathElements", graph_.tree_depth, &D.slots.path) &&
         find("identityPathIndex", graph_.tree_depth, &D.slots.path_idx) && find("x", 1, &D.slots.x) &&
         find("externalNullifier", 1, &D.slots.ext);
    D.have_values_kernel = ok;
  }
  poseidon_dev();

  // ---- workspace
  const size_t B = B_;
  D.inputs.alloc(B * D.NI * 8);
  D.rs.alloc(B * 16);
  D.pp_in.alloc(B * 80);
  RLN_HIP(hipMemsetAsync(D.pp_in.p, 0, D.pp_in.bytes(), s));
  RLN_HIP(hipMemsetAsync(D.inputs.p, 0, D.inputs.bytes(), s));
  RLN_HIP(hipMemsetAsync(D.rs.p, 0, D.rs.bytes(), s));
  for (int si = 0; si < D.nslot; si++) {
    Slot& S = D.slot[si];
    S.err.alloc(B);
    S.coords.alloc(B * 64);
    S.values.alloc(B * 40);
    S.comp.alloc(B * 128);
    S.V.alloc((size_t)D.N * B);
    if (D.wit29) S.V29.alloc(((size_t)D.nstore29 + 1) * 3 * B);   // + the trash row of k_witness_lanes
    S.abc.alloc(3 * (size_t)D.n * B);
    S.digits.alloc((size_t)(3 * D.NS + D.n + 4) * D.nh * D.ws.W * B);   // + s w_i, r w_i, r s of the fused small-batch plan
    S.digits2.alloc((size_t)(D.NS + 3) * D.nh * D.ws2.W * B);
    S.part1.alloc(std::max((size_t)D.max_chunks1 * B, (size_t)D.max_chunks1s * D.small_stride));
    S.grp1.alloc((size_t)D.max_groups1 * B);
    S.sums1.alloc(3 * D.nh * B);
    S.part2.alloc(std::max((size_t)D.max_chunks2 * B, (size_t)D.max_chunks2s * D.small_stride));
    S.grp2.alloc((size_t)D.max_groups2 * B);
    S.sums2.alloc(D.nh * B);
    S.prod.alloc(2 * B);
    S.tbl.alloc(2 * 16 * B);
    S.affA.alloc(B);
    S.affB1.alloc(B);
    S.affB2.alloc(B);
    S.pp_out.alloc(B * 80);
    S.inputs.alloc(B * D.NI * 8);
    S.rs.alloc(B * 16);
    S.pp_in.alloc(B * 80);
    RLN_HIP(hipMemsetAsync(S.inputs.p, 0, S.inputs.bytes(), s));
    RLN_HIP(hipMemsetAsync(S.rs.p, 0, S.rs.bytes(), s));
    RLN_HIP(hipMemsetAsync(S.pp_in.p, 0, S.pp_in.bytes(), s));
    RLN_HIP(hipHostMalloc((void**)&S.h_in, B * ((size_t)D.NI * 32 + 64 + 320), hipHostMallocDefault));
    RLN_HIP(hipEventCreateWithFlags(&S.evU, hipEventDisableTiming));
    RLN_HIP(hipEventCreateWithFlags(&S.evE, hipEventDisableTiming));
    RLN_HIP(hipHostMalloc((void**)&S.h_pp, B * 320, hipHostMallocDefault));
    RLN_HIP(hipHostMalloc((void**)&S.h_comp, B * 128, hipHostMallocDefault));
    RLN_HIP(hipHostMalloc((void**)&S.h_values, B * 160, hipHostMallocDefault));
    RLN_HIP(hipHostMalloc((void**)&S.h_err, B * 4, hipHostMallocDefault));
    RLN_HIP(hipEventCreateWithFlags(&S.evA, hipEventDisableTiming));
    RLN_HIP(hipEventCreateWithFlags(&S.evB, hipEventDisableTiming));
    RLN_HIP(hipEventCreateWithFlags(&S.evB2, hipEventDisableTiming));
    RLN_HIP(hipEventCreateWithFlags(&S.evR, hipEventDisableTiming));
    RLN_HIP(hipEventCreateWithFlags(&S.evW, hipEventDisableTiming));
    RLN_HIP(hipEventCreateWithFlags(&S.evX, hipEventDisableTiming));
    RLN_HIP(hipEventCreateWithFlags(&S.evV, hipEventDisableTiming));
    RLN_HIP(hipEventCreateWithFlags(&S.evC, hipEventDisableTiming));
    for (auto& e : S.t) RLN_HIP(hipEventCreate(&e));
    RLN_HIP(hipMemsetAsync(S.digits.p, 0, S.digits.bytes(), s));
    RLN_HIP(hipMemsetAsync(S.digits2.p, 0, S.digits2.bytes(), s));
  }
  RLN_HIP(hipStreamSynchronize(s));
}

Prover::~Prover() {
  if (!d_) return;
  Impl& D = *d_;
  for (hipStream_t st : {D.sA, D.sAb, D.sA2, D.sB, D.sB2, D.sC, D.sV[0], D.sV[1]})
    if (st) (void)hipStreamSynchronize(st);
  // freed device memory is not cleared by the runtime: nothing secret-dependent goes back to the allocator
  try {
    if (D.sC) {
      for (int k = 0; k < D.nslot; k++)
        if (D.slot[k].used && !D.slot[k].wiped && D.slot[k].evC) D.wipe_slot(D.slot[k], D.slot[k].ticket == 0);
      (void)hipStreamSynchronize(D.sC);
    }
  } catch (...) {
  }
  for (Slot& S : D.slot) {
    if (S.h_pp) (void)hipHostFree(S.h_pp);
    if (S.h_in) (void)hipHostFree(S.h_in);
    if (S.evU) (void)hipEventDestroy(S.evU);
    if (S.evE) (void)hipEventDestroy(S.evE);
    if (S.h_comp) (void)hipHostFree(S.h_comp);
    if (S.h_values) (void)hipHostFree(S.h_values);
    if (S.h_err) (void)hipHostFree(S.h_err);
    for (hipEvent_t e : {S.evA, S.evB, S.evB2, S.evR, S.evC, S.evW, S.evV, S.evX})
      if (e) (void)hipEventDestroy(e);
    for (auto& e : S.t)
      if (e) (void)hipEventDestroy(e);
  }
  for (hipStream_t st : {D.sA, D.sAb, D.sA2, D.sB, D.sB2, D.sC, D.sV[0], D.sV[1]})
    if (st) (void)hipStreamDestroy(st);
}

size_t Prover::table_bytes() const { return d_->t1.bytes() + d_->t1_29.bytes() + d_->t2.bytes() + d_->t2_29.bytes(); }
size_t Prover::g1_rows() const { return d_->npts1; }
size_t Prover::g2_rows() const { return d_->npts2; }

void Prover::upload(size_t n, const uint8_t* inputs, const uint8_t* rs) {
  if (n > B_) throw Error("batch larger than the prover workspace (max_batch)");
  Impl& D = *d_;
  D.sync_all();  // in-flight batches still read the resident inputs
  RLN_HIP(hipMemcpyAsync(D.inputs.p, inputs, n * D.NI * 32, hipMemcpyHostToDevice, D.sA));
  RLN_HIP(hipMemcpyAsync(D.rs.p, rs, n * 64, hipMemcpyHostToDevice, D.sA));
  RLN_HIP(hipStreamSynchronize(D.sA));
}

void Prover::upload_witness(size_t n, const uint8_t* w_le) {
  if (n > B_) throw Error("batch larger than the prover workspace (max_batch)");
  Impl& D = *d_;
  D.sync_all();
  size_t words = n * (size_t)D.NS * 8;
  if (D.wgiven.n < words) D.wgiven.alloc(words);
  RLN_HIP(hipMemcpyAsync(D.wgiven.p, w_le, words * 4, hipMemcpyHostToDevice, D.sA));
  RLN_HIP(hipStreamSynchronize(D.sA));
  D.wgiven_n = n;
}

template <bool DIF, bool M29>
static void launch_ntt(Fr* data, const uint32_t* tw, int logn, const uint32_t* final_scale, uint32_t B, uint32_t nb,
                       hipStream_t s, bool lanes_are_groups = false) {
  int s0 = 0;
  const bool fuse9 = env_int("RLNAMD_NTT_FUSE9", 1) != 0;
  if (lanes_are_groups && !M29 && fuse9 && logn >= 9) {   // nine levels in one kernel (k_ntt_fused9)
    const Fr* sc = (9 == logn) ? reinterpret_cast<const Fr*>(final_scale) : nullptr;
    hipLaunchKernelGGL((k_ntt_fused9<DIF>), dim3(nb, (1u << logn) >> 9, 3), dim3(64), 0, s, data,
                       reinterpret_cast<const Fr*>(tw), logn, 0, sc, B, nb);
    RLN_HIP(hipGetLastError());
    s0 = 9;
  }
  while (lanes_are_groups && s0 < logn) {   // small batch: blockIdx.x = proof, lanes = groups (radix-8 passes + a radix-2 tail)
    const int rem = logn - s0, K = rem >= 3 ? 3 : 1;
    const uint32_t groups = (1u << logn) >> K;
    dim3 block(64, 1), grid(nb, div_up(groups, 64), 3);
    const uint32_t* sc = (s0 + K == logn) ? final_scale : nullptr;
    if (K == 3)
      hipLaunchKernelGGL((k_ntt_pass<3, DIF, M29, true>), grid, block, 0, s, data, tw, logn, s0, sc, B, nb);
    else
      hipLaunchKernelGGL((k_ntt_pass<1, DIF, M29, true>), grid, block, 0, s, data, tw, logn, s0, sc, B, nb);
    RLN_HIP(hipGetLastError());
    s0 += K;
  }
  while (s0 < logn) {
    int rem = logn - s0;
    static const int maxk = env_int("RLNAMD_NTT_MAXK", 3);
    int K = rem > maxk ? 3 : rem;  // 13 -> 3,3,3,3,1 (a 16-point block spills; measured 6.7 -> 5.0 ms)
    if (K > maxk) K = maxk;
    uint32_t groups = (1u << logn) >> K;
    // one wave per workgroup: a 4-wave workgroup needs four free wave slots on one CU at the same moment, which the
    // single-wave MSM workgroups streaming through the chip never leave (measured: mat-vec 0.6 -> 32 ms, NTT 5 -> 19 ms)
    static const int wpb = env_int("RLNAMD_NTT_WPB", 1);
    dim3 block(64, wpb), grid(div_up(nb, 64), div_up(groups, wpb), 3);
    const uint32_t* sc = (s0 + K == logn) ? final_scale : nullptr;
    switch (K) {
      case 1: hipLaunchKernelGGL((k_ntt_pass<1, DIF, M29>), grid, block, 0, s, data, tw, logn, s0, sc, B, nb); break;
      case 2: hipLaunchKernelGGL((k_ntt_pass<2, DIF, M29>), grid, block, 0, s, data, tw, logn, s0, sc, B, nb); break;
      case 3: hipLaunchKernelGGL((k_ntt_pass<3, DIF, M29>), grid, block, 0, s, data, tw, logn, s0, sc, B, nb); break;
      default: hipLaunchKernelGGL((k_ntt_pass<4, DIF, M29>), grid, block, 0, s, data, tw, logn, s0, sc, B, nb); break;
    }
    RLN_HIP(hipGetLastError());
    s0 += K;
  }
}

// Enqueue one batch; returns as soon as the work is queued.  Stage A (stream sA): proof values, witness,
// matvec, NTTs.  Stage B (sB): digit recoding and the two MSMs.  Stage C (sC): two-level reduction, the
// three finalize kernels, D2H of proofs + values into pinned memory.  Consecutive batches alternate slots.
void Prover::run_async(size_t n, int mode) { enqueue(n, mode, nullptr, nullptr, nullptr); }

// Streamed batch ("H2D of witness inputs -> D2H of proofs"; the reference takes a fresh witness per call,
// protocol/proof.rs:753-777): the inputs go through the slot's pinned staging buffer to the slot's own device buffers on
// the front-end stream of this batch, so consecutive submits of DIFFERENT batches overlap like run_async's do.  Blocks
// only when every workspace slot is in flight (then until the oldest batch has finished).
uint64_t Prover::submit(size_t n, const uint8_t* inputs, const uint8_t* rs, int mode, const uint8_t* partial320) {
  if (n == 0) throw Error("empty batch");
  if (!inputs || !rs) throw Error("submit: inputs and rs are required");
  if (mode == PROVE_FINISH && !partial320) throw Error("submit: finish mode needs the partial points");
  return enqueue(n, mode, inputs, rs, partial320);
}

void Prover::collect(uint64_t ticket, size_t n, uint8_t* proofs, uint8_t* values, uint32_t* errors, uint8_t* coords,
                     uint8_t* partial320, bool wipe_after) {
  Impl& D = *d_;
  Slot* Sp = nullptr;
  for (int k = 0; k < D.nslot; k++)
    if (D.slot[k].used && D.slot[k].ticket == ticket && ticket != 0) Sp = &D.slot[k];
  if (!Sp) throw Error("collect: unknown or expired ticket (its workspace slot has been reused)");
  Slot& S = *Sp;
  if (n > S.n) throw Error("collect: more proofs requested than the batch holds");
  RLN_HIP(hipEventSynchronize(S.evC));
  if (S.mode == PROVE_PARTIAL) {
    if (partial320) memcpy(partial320, S.h_pp, n * 320);
  } else {
    if (proofs) memcpy(proofs, S.h_comp, n * 128);
    if (values) {
      if (D.have_values_kernel)
        memcpy(values, S.h_values, n * 160);
      else if (D.ni == 6) {
        std::vector<uint8_t> pub;
        fetch_public_slot(&S, n, &pub);
        memcpy(values, pub.data(), n * 160);
      } else
        memset(values, 0, n * 160);
    }
    if (coords) RLN_HIP(hipMemcpy(coords, S.coords.p, n * 256, hipMemcpyDeviceToHost));
  }
  if (errors) memcpy(errors, S.h_err, n * 4);
  if (wipe_after && !S.wiped) D.wipe_slot(S, false);
}

// ticket 0: the resident-input run (upload / run / download): the shared input buffers and the last batch's witness.
void Prover::wipe(uint64_t ticket) {
  Impl& D = *d_;
  if (ticket == 0) {
    D.sync_all();
    if (D.last) D.wipe_slot(*D.last, true);
    else {
      RLN_HIP(hipMemsetAsync(D.inputs.p, 0, D.inputs.bytes(), D.sC));
      RLN_HIP(hipMemsetAsync(D.rs.p, 0, D.rs.bytes(), D.sC));
    }
    RLN_HIP(hipStreamSynchronize(D.sC));
    return;
  }
  for (int k = 0; k < D.nslot; k++)
    if (D.slot[k].used && D.slot[k].ticket == ticket) {
      RLN_HIP(hipEventSynchronize(D.slot[k].evC));
      if (!D.slot[k].wiped) D.wipe_slot(D.slot[k], false);
      return;
    }
}

void Prover::collect_public(uint64_t ticket, size_t n, std::vector<uint8_t>* out_le) {
  Impl& D = *d_;
  for (int k = 0; k < D.nslot; k++)
    if (D.slot[k].used && D.slot[k].ticket == ticket && ticket != 0) {
      if (n > D.slot[k].n) throw Error("collect: more proofs requested than the batch holds");
      if (D.slot[k].wiped) throw Error("collect_public: the batch has been wiped (collect it with wipe_after = false first)");
      RLN_HIP(hipEventSynchronize(D.slot[k].evC));
      fetch_public_slot(&D.slot[k], n, out_le);
      return;
    }
  throw Error("collect: unknown or expired ticket (its workspace slot has been reused)");
}

int Prover::slots() const { return d_->nslot; }

// Any number of proofs through the streamed path: chunks of at most capacity() proofs, as many in flight as there are
// workspace slots, results written in index order.  This is what a caller with more proofs than one workspace holds
// gets instead of an upload / run / download loop that drains the pipeline after every chunk.
void Prover::prove_stream(size_t n, const uint8_t* inputs, const uint8_t* rs, uint8_t* proofs, uint8_t* values,
                          uint32_t* errors) {
  Impl& D = *d_;
  struct Pending { uint64_t ticket; size_t off, cnt; };
  std::deque<Pending> q;
  const size_t NIB = (size_t)D.NI * 32;
  auto take = [&]() {
    Pending f = q.front();
    q.pop_front();
    collect(f.ticket, f.cnt, proofs ? proofs + f.off * 128 : nullptr, values ? values + f.off * 160 : nullptr,
            errors ? errors + f.off : nullptr);
  };
  try {
    for (size_t off = 0; off < n; off += B_) {
      size_t cnt = std::min(B_, n - off);
      if ((int)q.size() == D.nslot) take();   // the slot the next submit reuses
      q.push_back({submit(cnt, inputs + off * NIB, rs + off * 64), off, cnt});
    }
    while (!q.empty()) take();
  } catch (...) {
    D.sync_all();
    for (int k = 0; k < D.nslot; k++)   // the error path leaves no witness behind either
      if (D.slot[k].used && !D.slot[k].wiped) D.wipe_slot(D.slot[k], false);
    D.sync_all();
    throw;
  }
}

uint64_t Prover::enqueue(size_t n, int mode, const uint8_t* h_inputs, const uint8_t* h_rs, const uint8_t* h_pp320) {
  if (n == 0) return 0;
  if (n > B_) throw Error("batch larger than the prover workspace (max_batch)");
  if (mode < PROVE_FULL || mode > PROVE_FINISH) throw Error("unknown prover mode");
  Impl& D = *d_;
  // lone: nothing else in flight -- the batch may trade throughput for latency (the fused plan's + 25 % G1 rows, the
  // single-stream chains, the wave-per-proof interpreter above the small-batch threshold)
  const int lone_force = env_int("RLNAMD_LONE", -1);   // -1: detect; 0 / 1: force (measurements, tests)
  const bool lone = lone_force >= 0 ? lone_force != 0 : (!D.last || hipEventQuery(D.last->evC) == hipSuccess);
  (void)hipGetLastError();   // hipErrorNotReady is not an error here
  const bool small = n <= D.lanechunk_max && n <= D.small_stride && D.use29 && D.use29_g2;   // lanes = chunks
  // Small batches (latency, not throughput): the whole front end stays on ONE stream (every cross-stream event hop costs
  // 0.1 - 0.15 ms), the digits of the witness scalars are recoded right behind the interpreter, and both walks start on
  // everything that does not depend on the quotient h while mat-vec / NTTs still run; only the h rows of the G1 walk
  // wait for them.
  const bool early = n <= D.lanechunk_max && D.use29 && D.use29_g2 && D.split_msm && D.recode_front &&
                     mode != PROVE_PARTIAL && env_int("RLNAMD_EARLY_WALK", 1) != 0;
  // small full proofs: s A and r B1 are rows of the C segment (plan1f), no k_fin_smul
  // (up to 96 proofs: above, the walks are issue-bound even for a lone batch and the extra rows cost more than the ladder
  // they replace -- 128 proofs 16.6 -> 15.3 ms without them, 64 proofs 10.1 -> 10.3 ms)
  const bool fused = lone && n <= 96 && early && small && mode == PROVE_FULL && D.nh == 2 && env_int("RLNAMD_FUSED_SMUL", 1) != 0 &&
                     env_int("RLNAMD_EARLY_FIN", 1) != 0;   // (its back end is the split one below)
  const Impl::Plan& P1 = fused ? D.plan1f[PROVE_FULL] : small ? D.plan1s[mode] : D.plan1[mode];
  const Impl::Plan& P2 = small ? D.plan2s[mode] : D.plan2[mode];
  const uint32_t PB = small ? D.small_stride : (uint32_t)B_;   // stride of the partial-sum arrays
  // mid-size small batches: the short-chunk plans walked with lanes = proofs (walk29.h).  A lone batch: above 48 proofs
  // (64: 11.3 -> 9.9 ms, 128: 18.1 -> 16.3 ms; 32: 6.9 ms against 8.4).  In a stream of batches the lanes = chunks form
  // pays its scattered gathers in throughput much earlier (streams of 64 / 128-proof batches: 9.5 -> 10.8 k, 10.7 -> 11.9 k
  // proofs/s), so there it stops at 16 proofs.
  const bool walk_lp = small && early && (n > D.lanechunk_walk_max || (!lone && n >= 16));
  Slot& S = D.slot[D.cur];
  const bool streamed = h_inputs != nullptr;
  if (streamed) {
    if (D.wgiven_n) throw Error("upload_witness applies to the resident-input run that follows it, not to submit");
    // the slot's previous batch must be finished before its staging buffer (and its result buffers) are reused
    if (S.used) RLN_HIP(hipEventSynchronize(S.evC));
    memcpy(S.h_in, h_inputs, n * (size_t)D.NI * 32);
    memcpy(S.h_in + B_ * (size_t)D.NI * 32, h_rs, n * 64);
    if (h_pp320) memcpy(S.h_in + B_ * ((size_t)D.NI * 32 + 64), h_pp320, n * 320);
  }
  const uint32_t* in_p = streamed ? S.inputs.p : D.inputs.p;
  const uint32_t* rs_p = streamed ? S.rs.p : D.rs.p;
  const uint32_t* pp_p = (streamed && h_pp320) ? S.pp_in.p : D.pp_in.p;
  S.mode = mode;
  S.ticket = streamed ? ++D.tickets : 0;
  D.cur = (D.cur + 1) % D.nslot;
  // Front end in two pipeline stages on their own streams: A1 = graph interpreter (16 latency-bound waves per 1024
  // proofs, ~28 ms), A2 = mat-vec + NTTs + quotient (throughput kernels squeezed in beside the MSM, ~25 ms contended).
  // Chained on one stream they were the critical path (54 ms against 49 ms of MSM).
  const uint32_t sq = D.seq++;
  hipStream_t sA = (D.nstreamA > 1 && D.wstreams > 1 && (sq & 1)) ? D.sAb : D.sA;
  const uint32_t B = (uint32_t)B_, nb = (uint32_t)n;
  hipStream_t sA2 = (D.nstreamA > 1 && !early) ? D.sA2 : sA;
  const uint32_t pg = div_up(nb, 64);
  const uint32_t nbp = pg * 64;  // padded lanes compute on stale / zero inputs; results ignored
  // ---------------- stage A
  if (S.used) RLN_HIP(hipStreamWaitEvent(sA, S.evC, 0));  // slot free again
  if (streamed) {
    static const bool h2d_kernel = env_int("RLNAMD_H2D_KERNEL", 1) != 0;
    auto h2d = [&](void* dst, const uint8_t* src, size_t bytes) {   // sizes are multiples of 32
      if (h2d_kernel)
        hipLaunchKernelGGL(k_stage_in, dim3(div_up(bytes / 16, 64)), dim3(64), 0, sA, (const uint4*)src, (uint4*)dst,
                           (uint32_t)(bytes / 16));
      else
        RLN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, sA));
    };
    h2d(S.inputs.p, S.h_in, n * (size_t)D.NI * 32);
    h2d(S.rs.p, S.h_in + B_ * (size_t)D.NI * 32, n * 64);
    if (h_pp320) h2d(S.pp_in.p, S.h_in + B_ * ((size_t)D.NI * 32 + 64), n * 320);
    RLN_HIP(hipGetLastError());
    RLN_HIP(hipEventRecord(S.evU, sA));
  }
  // Timing marks (stage_ms): a timed event record is a barrier packet and a timestamp write on its stream -- three of
  // them sit between the interpreter and the mat-vec of a single proof (~0.1 ms of its 5 ms).  Small batches record
  // them only when asked to (RLNAMD_MARKS_SMALL=1; tools/single_latency.py); their stage_ms reads 0 otherwise.
  const bool marks = nb > D.lanechunk_max || env_int("RLNAMD_MARKS_SMALL", 0) != 0;
  S.marked = marks;
#define MARK(i, stream)                                   \
  do {                                                    \
    if (marks) RLN_HIP(hipEventRecord(S.t[i], stream));   \
  } while (0)
  // The lanes = nodes interpreter (a wave and 157 KB of LDS per proof, ~25 x the instructions per proof of k_witness29,
  // 2.0 ms against 11 ms): always below the small-batch threshold; up to witlanes_max only for a LONE batch -- in a stream
  // of such batches it costs throughput (profiles/r3_rocprof_summary.md, section 10), and there the previous batch is still in flight.
  const bool wl_used = D.wit29 && D.witlanes.ok && (nb <= D.lanechunk_max || (nb <= D.witlanes_max && lone));
  MARK(1, sA);
  if (D.wit29) {
    static const bool prof = env_int("RLNAMD_WIT_PROF", 0) != 0;   // diagnostic: cycles per node class, on stderr
    if (prof) {
      DevBuf<unsigned long long>& pb = D.wit_prof;
      if (!pb.p) pb.alloc(16);
      hipLaunchKernelGGL(k_witness29<true>, dim3(pg), dim3(64), WIT29_LDS_BYTES, sA, D.nodes29.p, D.nprog29,
                         D.consts29.p, (uint32_t)graph_.constants.size(), in_p, D.NI, S.V29.p, S.err.p, B, nbp, pb.p);
      unsigned long long h[16];
      RLN_HIP(hipStreamSynchronize(sA));
      RLN_HIP(hipMemcpy(h, pb.p, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "wit29 prof: mul %llu cyc / %llu, add %llu / %llu, const+input %llu / %llu, other %llu / %llu; total %llu cyc, %.3f ms, clock %.0f MHz\n",
              h[0], h[4], h[1], h[5], h[2], h[6], h[3], h[7], h[8], h[9] / 1e5, h[9] ? 100.0 * h[8] / h[9] : 0.0);
    } else if (wl_used) {
      D.witlanes.launch(sA, D.consts29.p, in_p, D.NI, S.V29.p, S.err.p, B, nb);
    } else
    hipLaunchKernelGGL(k_witness29<false>, dim3(pg), dim3(64), WIT29_LDS_BYTES, sA, D.nodes29.p, D.nprog29,
                       D.consts29.p, (uint32_t)graph_.constants.size(), in_p, D.NI, S.V29.p, S.err.p, B, nbp, nullptr);
    if (nb <= D.lanechunk_max)
      hipLaunchKernelGGL(k_v29_to_fr, dim3(div_up(D.nstore29, 64), nb), dim3(64, 1), 0, sA, S.V29.p, D.slot2node.p,
                         D.nstore29, S.V.p, B, nb, 1u);
    else
      hipLaunchKernelGGL(k_v29_to_fr, dim3(pg, D.nstore29), dim3(64, 1), 0, sA, S.V29.p, D.slot2node.p, D.nstore29, S.V.p,
                         B, nbp);
  } else {
    hipLaunchKernelGGL(k_witness, dim3(pg), dim3(64), WIT_RING * 8 * 64 * 4 + WIT_LDS_CONSTS * 32, sA, D.nodes.p, D.N, D.consts.p,
                       (uint32_t)graph_.constants.size(), in_p, D.NI, S.V.p,
                       S.err.p, B, nbp);
  }
  if (D.wgiven_n) {
    if (D.wgiven_n != n || mode != PROVE_FULL) throw Error("upload_witness: the next run must be a full proof of the same batch");
    hipLaunchKernelGGL(k_scatter_witness, dim3(pg, div_up(D.NS, 4)), dim3(64, 4), 0, sA, D.wgiven.p, D.sig2node.p,
                       D.NS, S.V.p, S.err.p, B, nb);
    D.wgiven_n = 0;
  }
  // Small batches (latency): the digits of the witness scalars and of r, s are recoded right behind the interpreter,
  // so the G2 walk -- the longer of the two, and independent of the quotient h -- starts beside mat-vec / NTT instead
  // of behind them; only h's digits wait for the NTTs.
  const bool early_g2 = early;
  RLN_HIP(hipEventRecord(S.evX, sA));   // mat-vec / NTT (sA2) need the witness, not the recodes below
  if (early) {
    // on the walks' own stream: mat-vec and the NTTs (sA) start at once, beside the recodes instead of behind them
    // The G2 chain (recode, walk, sum, inversion: 1.3 ms for one proof) is the longest thing behind the interpreter and
    // every cross-stream hop costs it 50 - 100 us, so it runs on ONE stream (sB2); the G1 walk's stream takes the hop.
    // (A lone batch; in a stream of batches the recodes stay on the front-end stream, where they do not queue behind the
    // previous batch's walks.)
    hipStream_t sR1 = lone ? D.sB2 : sA, sR3 = lone ? D.sB : sA;
    if (lone) RLN_HIP(hipStreamWaitEvent(D.sB2, S.evX, 0));
    hipLaunchKernelGGL(k_recode, dim3(div_up(D.NS + 3, 64), nb), dim3(64, 1), 0, sR1, S.V.p, D.sig2node.p, D.NS,
                       S.abc.p, D.n, rs_p, D.ws, D.ws2, D.nh, S.digits.p, S.digits2.p, B, nb, 1u, 1u);
    if (lone) {
      RLN_HIP(hipEventRecord(S.evW, D.sB2));
      RLN_HIP(hipStreamWaitEvent(D.sB, S.evW, 0));
    }
    if (fused)
      hipLaunchKernelGGL(k_recode, dim3(div_up(2 * D.NS + 1, 64), nb), dim3(64, 1), 0, sR3, S.V.p, D.sig2node.p, D.NS,
                         S.abc.p, D.n, rs_p, D.ws, D.ws2, D.nh, S.digits.p, S.digits2.p, B, nb, 3u, 1u);
    if (!lone) {
      RLN_HIP(hipEventRecord(S.evW, sA));
      RLN_HIP(hipStreamWaitEvent(D.sB, S.evW, 0));
      RLN_HIP(hipStreamWaitEvent(D.sB2, S.evW, 0));
    }
    MARK(14, D.sB);
    if (P1.n_early && walk_lp)
      hipLaunchKernelGGL((k_msm29<G1Acc29, G1Affine29, G1XYZZ, 4>), dim3(div_up(P1.n_early, 8) * 8 * pg), dim3(64), D.msm_lds,
                         D.sB, D.t1_29.p, P1.rsid.p, P1.rows.p, P1.chunks.p, P1.n_early, S.digits.p, S.part1.p, D.ws, B, pg,
                         D.nh, nullptr, P1.early_ids.p, PB);
    else if (P1.n_early)
      hipLaunchKernelGGL((k_msm29<G1Acc29, G1Affine29, G1XYZZ, 2, true>), dim3(div_up(P1.n_early, 64), nb), dim3(64), 0,
                         D.sB, D.t1_29.p, P1.rsid.p, P1.rows.p, P1.chunks.p, P1.n_early, S.digits.p, S.part1.p, D.ws, B, PB,
                         D.nh, nullptr, P1.early_ids.p);
    RLN_HIP(hipEventRecord(S.evE, D.sB));
  }
  MARK(2, sA);
  if (sA2 != sA) RLN_HIP(hipStreamWaitEvent(sA2, S.evX, 0));
  MARK(12, sA2);
  if (mode != PROVE_PARTIAL) {  // the quotient h depends on the whole witness: not part of a partial proof
    CsrView A{D.a_ptr.p, D.a_col.p, D.a_coef.p}, Bm{D.b_ptr.p, D.b_col.p, D.b_coef.p};
    if (nb <= D.lanechunk_max)
      hipLaunchKernelGGL(k_matvec<true>, dim3(div_up(D.n, 64) + D.n_mv_long, nb), dim3(64, 1), 0, sA2, A, Bm, S.V.p,
                         D.sig2node.p, D.nc, D.ni, D.n, S.abc.p, B, nb, D.mv_long.p, div_up(D.n, 64));
    else
      hipLaunchKernelGGL(k_matvec<false>, dim3(pg, D.n), dim3(64, 1), 0, sA2, A, Bm, S.V.p, D.sig2node.p, D.nc,
                         D.ni, D.n, S.abc.p, B, nbp);
  }
  MARK(3, sA2);
  if (mode != PROVE_PARTIAL) {
    // Twiddle products through Fr29::mul_mont need ~290 instead of ~375 instructions, but same-box A/B runs gave
    // 20.21 / 20.26 k against 20.30 / 20.24 k proofs/s: beside the table walks the passes are bound by HBM and by
    // waiting for SIMD slots, not by their instruction count.  Kept selectable (RLNAMD_NTT29=1), off by default.
    if (D.ntt29) {
      launch_ntt<true, true>(S.abc.p, D.tw_i29.p, D.logn, D.coset29.p, B, nbp, sA2);  // iNTT (DIF) + g^i / n
      launch_ntt<false, true>(S.abc.p, D.tw_f29.p, D.logn, nullptr, B, nbp, sA2);     // NTT (DIT)
    } else {
      const bool lg = nb <= D.lanechunk_max;   // below a wave of proofs: lanes = groups
      launch_ntt<true, false>(S.abc.p, (const uint32_t*)D.tw_i.p, D.logn, (const uint32_t*)D.coset.p, B, lg ? nb : nbp, sA2, lg);
      launch_ntt<false, false>(S.abc.p, (const uint32_t*)D.tw_f.p, D.logn, nullptr, B, lg ? nb : nbp, sA2, lg);
    }
    if (nb <= D.lanechunk_max)
      hipLaunchKernelGGL(k_hquot, dim3(div_up(D.n, 64), nb), dim3(64, 1), 0, sA2, S.abc.p, D.n, B, nb, 1u);
    else
      hipLaunchKernelGGL(k_hquot, dim3(pg, D.n), dim3(64, 1), 0, sA2, S.abc.p, D.n, B, nbp, 0u);
  }
  MARK(4, sA2);
  // digit recoding either closes the front end (the MSM stream then carries nothing but the two table walks) or
  // opens the MSM stage (RLNAMD_RECODE_FRONT=0)
  hipStream_t sR = D.recode_front ? sA2 : D.sB;
  if (!D.recode_front) {
    RLN_HIP(hipEventRecord(S.evA, sA2));
    RLN_HIP(hipStreamWaitEvent(D.sB, S.evA, 0));
  }
  MARK(5, sR);
  if (early_g2)
    hipLaunchKernelGGL(k_recode, dim3(div_up(D.n, 64), nb), dim3(64, 1), 0, sR, S.V.p, D.sig2node.p, D.NS, S.abc.p, D.n,
                       rs_p, D.ws, D.ws2, D.nh, S.digits.p, S.digits2.p, B, nb, 2u, 1u);
  else
    hipLaunchKernelGGL(k_recode, dim3(pg, D.NS + D.n + 3), dim3(64, 1), 0, sR, S.V.p, D.sig2node.p, D.NS,
                       S.abc.p, D.n, rs_p, D.ws, D.ws2, D.nh, S.digits.p, S.digits2.p, B, nbp, 0u, 0u);
  MARK(6, sR);
  // ---------------- stage B
  if (D.recode_front && !early) {
    RLN_HIP(hipEventRecord(S.evA, sA2));
    RLN_HIP(hipStreamWaitEvent(D.sB, S.evA, 0));
  }
  if (!early) MARK(14, D.sB);
  // below half a wave of proofs the walks run with lanes = chunks (walk29.h); RLNAMD_LANECHUNK overrides the threshold
  const bool lanechunk = nb <= D.lanechunk_max;
  hipStream_t s2 = D.split_msm ? D.sB2 : D.sB;
  if (D.split_msm && !early) {   // (early: sB2 already waits for the witness + part-1 digits, all the G2 walk reads)
    RLN_HIP(hipEventRecord(S.evR, D.sB));
    RLN_HIP(hipStreamWaitEvent(D.sB2, S.evR, 0));
  }
  if (early) {   // the h rows, on the front-end stream itself (no event hop); everything else is already walking
    if (P1.n_late && walk_lp)
      hipLaunchKernelGGL((k_msm29<G1Acc29, G1Affine29, G1XYZZ, 4>), dim3(div_up(P1.n_late, 8) * 8 * pg), dim3(64), D.msm_lds,
                         sA, D.t1_29.p, P1.rsid.p, P1.rows.p, P1.chunks.p, P1.n_late, S.digits.p, S.part1.p, D.ws, B, pg,
                         D.nh, nullptr, P1.late_ids.p, PB);
    else if (P1.n_late)
      hipLaunchKernelGGL((k_msm29<G1Acc29, G1Affine29, G1XYZZ, 2, true>), dim3(div_up(P1.n_late, 64), nb), dim3(64), 0, sA,
                         D.t1_29.p, P1.rsid.p, P1.rows.p, P1.chunks.p, P1.n_late, S.digits.p, S.part1.p, D.ws, B, PB, D.nh,
                         nullptr, P1.late_ids.p);
    RLN_HIP(hipEventRecord(S.evR, sA));
    RLN_HIP(hipStreamWaitEvent(D.sB, S.evR, 0));   // evB below then covers both launches
  } else if (P1.nchunks) {
    uint32_t blocks = div_up(P1.nchunks, 8) * 8 * pg;
    if (D.use29 && lanechunk)
      hipLaunchKernelGGL((k_msm29<G1Acc29, G1Affine29, G1XYZZ, 2, true>), dim3(div_up(P1.nchunks, 64), nb), dim3(64), 0, D.sB,
                         D.t1_29.p, P1.rsid.p, P1.rows.p, P1.chunks.p, P1.nchunks, S.digits.p, S.part1.p, D.ws, B, PB, D.nh,
                         nullptr);
    else if (D.use29)
      // dynamic LDS that the kernel never touches caps it at D.msm_waves waves per SIMD: at 4 x 128 VGPRs the
      // register file is full and the front end's NTT / mat-vec workgroups wait for an MSM workgroup (~1 ms) to retire
      hipLaunchKernelGGL((k_msm29<G1Acc29, G1Affine29, G1XYZZ, 4>), dim3(blocks), dim3(64), D.msm_lds, D.sB, D.t1_29.p, P1.rsid.p, P1.rows.p, P1.chunks.p,
                         P1.nchunks, S.digits.p, S.part1.p, D.ws, B, pg, D.nh, D.walk_clk.p);
    else
      hipLaunchKernelGGL(k_msm<Fq>, dim3(blocks), dim3(64), 0, D.sB, D.t1.p, P1.rsid.p, P1.rows.p, P1.chunks.p,
                         P1.nchunks, S.digits.p, S.part1.p, D.ws, B, pg, D.nh);
  }
  MARK(7, D.sB);
  MARK(11, s2);
  if (P2.nchunks) {
    uint32_t blocks = div_up(P2.nchunks, 8) * 8 * pg;
    if (D.use29_g2 && walk_lp)
      hipLaunchKernelGGL((k_msm29<G2Acc29, G2Affine29, G2XYZZ, 2>), dim3(blocks), dim3(64), 0, s2, D.t2_29.p, P2.rsid.p,
                         P2.rows.p, P2.chunks.p, P2.nchunks, S.digits2.p, S.part2.p, D.ws2, B, pg, D.nh, nullptr, nullptr, PB);
    else if (D.use29_g2 && lanechunk)
      hipLaunchKernelGGL((k_msm29<G2Acc29, G2Affine29, G2XYZZ, 1, true>), dim3(div_up(P2.nchunks, 64), nb), dim3(64), 0, s2,
                         D.t2_29.p, P2.rsid.p, P2.rows.p, P2.chunks.p, P2.nchunks, S.digits2.p, S.part2.p, D.ws2, B, PB, D.nh,
                         nullptr);
    else if (D.use29_g2)
      hipLaunchKernelGGL((k_msm29<G2Acc29, G2Affine29, G2XYZZ, 2>), dim3(blocks), dim3(64), 0, s2, D.t2_29.p, P2.rsid.p,
                         P2.rows.p, P2.chunks.p, P2.nchunks, S.digits2.p, S.part2.p, D.ws2, B, pg, D.nh,
                         D.walk_clk.p ? D.walk_clk.p + 2 : nullptr);
    else
      hipLaunchKernelGGL(k_msm<Fq2>, dim3(blocks), dim3(64), 0, s2, D.t2.p, P2.rsid.p, P2.rows.p, P2.chunks.p,
                         P2.nchunks, S.digits2.p, S.part2.p, D.ws2, B, pg, D.nh);
  }
  MARK(8, s2);
  RLN_HIP(hipEventRecord(S.evB, D.sB));
  if (D.split_msm) RLN_HIP(hipEventRecord(S.evB2, D.sB2));
  // ---------------- stage C
  // proof values (Poseidon chain, latency-bound, depends on the inputs only): the back-end stream has slack
  hipStream_t sV = D.sV[sq & 1];
  if (S.used) {
    RLN_HIP(hipStreamWaitEvent(D.sC, S.evC, 0));
    RLN_HIP(hipStreamWaitEvent(sV, S.evC, 0));
  }
  if (streamed) RLN_HIP(hipStreamWaitEvent(sV, S.evU, 0));
  MARK(0, sV);
  // (whenever the batch is small enough for the lanes = nodes interpreter: the Poseidon chain alone is 5.3 ms)
  const bool values_w = (early || wl_used) && D.have_values_kernel && D.ni == 6 && env_int("RLNAMD_VALUES_WITNESS", 1) != 0;
  if (values_w) {   // small batches: the circuit's own outputs (see k_values_from_witness)
    RLN_HIP(hipStreamWaitEvent(sV, S.evX, 0));   // sA: witness stored
    hipLaunchKernelGGL(k_values_from_witness, dim3(pg, 5), dim3(64), 0, sV, S.V.p, D.sig2node.p, B, nbp, S.values.p);
  } else if (D.have_values_kernel)
    hipLaunchKernelGGL(k_proof_values, dim3(pg), dim3(64), 0, sV, in_p, D.NI, D.slots, poseidon_view(2),
                       poseidon_view(3), poseidon_view(4), S.values.p, nbp);
  MARK(13, sV);
  // Small full proofs: A and B1 are sums over h-independent rows only, so their reduction, the two inversions and the two
  // variable-base products s A, r B1 (the longest kernel of the back end) run on the idle sA2 as soon as the early G1
  // walk is done -- beside the NTTs and the walk of the h rows, not behind them.  sums1 segments: h * 3 + {A, B1, C}.
  const bool early_fin = early && mode == PROVE_FULL && D.nh == 2 && env_int("RLNAMD_EARLY_FIN", 1) != 0;
  const TaskSel all6 = task_sel({0, 1, 2, 3, 4, 5}), all4 = task_sel({0, 1, 2, 3}), all3 = task_sel({0, 1, 2});
  // below a wave of proofs s A / r B1 are a lone lane's chain: NAF ladder in the 9 x 29 form (fin29.hip)
  const bool fin29 = nb <= D.lanechunk_max && D.use29 && env_int("RLNAMD_FIN29", 1) != 0;
  hipStream_t sF = D.sC;   // the stream of k_fin_out and of the copies to the host
  if (early_fin) {
    RLN_HIP(hipStreamWaitEvent(D.sA2, S.evE, 0));   // sB: the early G1 walk
    if (fused) {
      // fused plan: only A has to be reduced and inverted early; s A and r B1 are inside the C segment, B1 is never formed
      hipLaunchKernelGGL(k_sum_tree<Fq>, dim3(nb, 2), dim3(SUM_TREE_LANES), 0, D.sA2, S.part1.p, P1.segchunks.p, S.sums1.p, B, PB,
                         task_sel({0, 3}));
      hipLaunchKernelGGL(k_glv_fold, dim3(pg, 1), dim3(64), 0, D.sA2, S.sums1.p, S.sums2.p, 3u, B, nbp, task_sel({0}));
      hipLaunchKernelGGL(k_fin_affine, dim3(pg, 1), dim3(64), 0, D.sA2, S.sums1.p, S.sums2.p, S.affA.p, S.affB1.p,
                         S.affB2.p, B, nbp, task_sel({0}));
      RLN_HIP(hipMemsetAsync(S.prod.p, 0, S.prod.bytes(), D.sA2));   // ZZ = 0: two points at infinity for k_fin_out
    } else {
      hipLaunchKernelGGL(k_sum_tree<Fq>, dim3(nb, 4), dim3(SUM_TREE_LANES), 0, D.sA2, S.part1.p, P1.segchunks.p, S.sums1.p, B,
                         PB, task_sel({0, 1, 3, 4}));
      hipLaunchKernelGGL(k_glv_fold, dim3(pg, 2), dim3(64), 0, D.sA2, S.sums1.p, S.sums2.p, 3u, B, nbp, task_sel({0, 1}));
      hipLaunchKernelGGL(k_fin_affine, dim3(pg, 2), dim3(64), 0, D.sA2, S.sums1.p, S.sums2.p, S.affA.p, S.affB1.p,
                         S.affB2.p, B, nbp, task_sel({0, 1}));
      if (fin29)
        launch_fin_smul29(D.sA2, S.affA.p, S.affB1.p, rs_p, S.prod.p, B, nb);
      else
        hipLaunchKernelGGL(k_fin_smul, dim3(pg, 2), dim3(64), 0, D.sA2, S.affA.p, S.affB1.p, rs_p, S.tbl.p, S.prod.p, B, nbp);
    }
    RLN_HIP(hipGetLastError());
    RLN_HIP(hipEventRecord(S.evA, D.sA2));
    // The G2 sum and inversion (0.65 ms for one proof, behind the G2 walk only) beside the C segment (sC, behind the h
    // rows) instead of in front of it; k_fin_out waits for both, for s A, r B1 and the values.
    // (s2: the G2 walk's stream -- its back end follows it without a hop; in a stream of batches it goes to sC, where
    // it does not hold up the next batch's G2 walk)
    hipStream_t sG = lone ? s2 : D.sC;
    if (!lone) RLN_HIP(hipStreamWaitEvent(D.sC, S.evB2, 0));
    hipLaunchKernelGGL(k_sum_tree<Fq2>, dim3(nb, P2.nseg), dim3(SUM_TREE_LANES), 0, sG, S.part2.p, P2.segchunks.p, S.sums2.p, B, PB, all6);
    hipLaunchKernelGGL(k_glv_fold, dim3(pg, 1), dim3(64), 0, sG, S.sums1.p, S.sums2.p, 3u, B, nbp, task_sel({3}));
    hipLaunchKernelGGL(k_fin_affine, dim3(pg, 1), dim3(64), 0, sG, S.sums1.p, S.sums2.p, S.affA.p, S.affB1.p,
                       S.affB2.p, B, nbp, task_sel({2}));
    if (lone) RLN_HIP(hipEventRecord(S.evB2, s2));
    RLN_HIP(hipEventRecord(S.evV, sV));
    // The C segment, k_fin_out and the copies home on the front-end stream itself, right behind the walk of the h rows:
    // the chain interpreter -> NTT -> h -> walk -> sum -> output crosses no stream (each hop is 50 - 100 us).
    // (Only for a lone batch: in a stream the front-end stream must be free for the batch after next -- there the C
    // segment stays on sC, behind evB, which covers both G1 walks.)
    if (lone) {
      sF = sA;
      if (S.used) RLN_HIP(hipStreamWaitEvent(sF, S.evC, 0));
      MARK(9, sF);
      RLN_HIP(hipStreamWaitEvent(sF, S.evE, 0));   // the early G1 walk: the C segment's h-independent rows
    } else {
      MARK(9, sF);
      RLN_HIP(hipStreamWaitEvent(sF, S.evB, 0));
    }
    hipLaunchKernelGGL(k_sum_tree<Fq>, dim3(nb, 2), dim3(SUM_TREE_LANES), 0, sF, S.part1.p, P1.segchunks.p, S.sums1.p, B,
                       PB, task_sel({2, 5}));
    hipLaunchKernelGGL(k_glv_fold, dim3(pg, 1), dim3(64), 0, sF, S.sums1.p, S.sums2.p, 3u, B, nbp, task_sel({2}));
    RLN_HIP(hipStreamWaitEvent(sF, S.evB2, 0));
    RLN_HIP(hipStreamWaitEvent(sF, S.evA, 0));
    RLN_HIP(hipStreamWaitEvent(sF, S.evV, 0));
  } else {
    RLN_HIP(hipEventRecord(S.evV, sV));
    RLN_HIP(hipStreamWaitEvent(D.sC, S.evV, 0));
    RLN_HIP(hipStreamWaitEvent(D.sC, S.evB, 0));
    if (D.split_msm) RLN_HIP(hipStreamWaitEvent(D.sC, S.evB2, 0));
    MARK(9, D.sC);
  }
  if (early_fin) {
  } else if (lanechunk) {   // small batch: lanes = partial sums (k_sum_tree)
    hipLaunchKernelGGL(k_sum_tree<Fq>, dim3(nb, P1.nseg), dim3(SUM_TREE_LANES), 0, D.sC, S.part1.p, P1.segchunks.p, S.sums1.p, B, PB, all6);
    hipLaunchKernelGGL(k_sum_tree<Fq2>, dim3(nb, P2.nseg), dim3(SUM_TREE_LANES), 0, D.sC, S.part2.p, P2.segchunks.p, S.sums2.p, B, PB, all6);
  } else {
    if (P1.ngroups)
      hipLaunchKernelGGL(k_sum_ranges<Fq>, dim3(pg, P1.ngroups), dim3(64), 0, D.sC, S.part1.p, P1.groups.p, P1.ngroups,
                         S.grp1.p, B, nbp);
    if (P2.ngroups)
      hipLaunchKernelGGL(k_sum_ranges<Fq2>, dim3(pg, P2.ngroups), dim3(64), 0, D.sC, S.part2.p, P2.groups.p, P2.ngroups,
                         S.grp2.p, B, nbp);
    hipLaunchKernelGGL(k_sum_ranges<Fq>, dim3(pg, P1.nseg), dim3(64), 0, D.sC, S.grp1.p, P1.segs.p, P1.nseg, S.sums1.p, B, nbp);
    hipLaunchKernelGGL(k_sum_ranges<Fq2>, dim3(pg, P2.nseg), dim3(64), 0, D.sC, S.grp2.p, P2.segs.p, P2.nseg, S.sums2.p, B, nbp);
  }
  if (early_fin) {
  } else if (D.nh == 2)  // sums of the second halves through phi, onto the first: afterwards sums1[0..3) / sums2[0] as without GLV
    hipLaunchKernelGGL(k_glv_fold, dim3(pg, 4), dim3(64), 0, D.sC, S.sums1.p, S.sums2.p, 3u, B, nbp, all4);
  if (mode == PROVE_PARTIAL) {
    hipLaunchKernelGGL(k_partial_out, dim3(pg, 4), dim3(64), 0, D.sC, S.sums1.p, S.sums2.p, S.pp_out.p, B, nbp);
    RLN_HIP(hipGetLastError());
    RLN_HIP(hipMemcpyAsync(S.h_pp, S.pp_out.p, n * 320, hipMemcpyDeviceToHost, D.sC));
  } else {
    if (mode == PROVE_FINISH)
      hipLaunchKernelGGL(k_add_partial, dim3(pg, 4), dim3(64), 0, D.sC, S.sums1.p, S.sums2.p, pp_p, B, nbp);
    if (!early_fin) {
      hipLaunchKernelGGL(k_fin_affine, dim3(pg, 3), dim3(64), 0, D.sC, S.sums1.p, S.sums2.p, S.affA.p, S.affB1.p,
                         S.affB2.p, B, nbp, all3);
      if (fin29)
        launch_fin_smul29(D.sC, S.affA.p, S.affB1.p, rs_p, S.prod.p, B, nb);
      else
        hipLaunchKernelGGL(k_fin_smul, dim3(pg, 2), dim3(64), 0, D.sC, S.affA.p, S.affB1.p, rs_p, S.tbl.p, S.prod.p, B,
                           nbp);
    }
    hipLaunchKernelGGL(k_fin_out, dim3(pg), dim3(64), 0, sF, S.sums1.p, S.prod.p, S.affA.p, S.affB2.p, S.coords.p,
                       S.comp.p, B, nbp);
    RLN_HIP(hipGetLastError());
    RLN_HIP(hipMemcpyAsync(S.h_comp, S.comp.p, n * 128, hipMemcpyDeviceToHost, sF));
    RLN_HIP(hipMemcpyAsync(S.h_values, S.values.p, n * 160, hipMemcpyDeviceToHost, sF));
  }
  RLN_HIP(hipMemcpyAsync(S.h_err, S.err.p, n * 4, hipMemcpyDeviceToHost, sF));
  MARK(10, sF);
  RLN_HIP(hipEventRecord(S.evC, sF));
  S.used = true;
  S.wiped = false;
  S.n = n;
  D.last = &S;
  return S.ticket;
}

void Prover::sync() { sync_measure(false); }

void Prover::sync_measure(bool last_only) {
  Impl& D = *d_;
  D.sync_all();
  if (D.last) {
    // stage spans, averaged over the batches still held in the workspace slots (the last <= nslot launches of the
    // same kind): in the pipeline a span includes whatever shared the chip with it
    const int pairs[PROVER_STAGES][2] = {{1, 2}, {12, 3}, {3, 4}, {5, 6}, {14, 7}, {11, 8}, {9, 10}, {0, 13}};
    float acc[PROVER_STAGES] = {0};
    int cnt = 0;
    for (int k = 0; k < D.nslot; k++) {
      Slot& S = D.slot[k];
      if (!S.used || !S.marked || S.mode != D.last->mode || S.n != D.last->n || (last_only && &S != D.last)) continue;
      for (int i = 0; i < PROVER_STAGES; i++) {
        float ms = 0;
        RLN_HIP(hipEventElapsedTime(&ms, S.t[pairs[i][0]], S.t[pairs[i][1]]));
        acc[i] += ms;
      }
      cnt++;
    }
    for (int i = 0; i < PROVER_STAGES; i++) D.ms[i] = cnt ? acc[i] / cnt : 0.f;
  }
}

void Prover::run(size_t n, int mode) {
  sync();  // nothing else in flight: the stage spans of this batch are those of the stages by themselves
  run_async(n, mode);
  sync_measure(true);
}

void Prover::upload_partial(size_t n, const uint8_t* coords320) {
  if (n > B_) throw Error("batch larger than the prover workspace (max_batch)");
  Impl& D = *d_;
  D.sync_all();
  RLN_HIP(hipMemcpyAsync(D.pp_in.p, coords320, n * 320, hipMemcpyHostToDevice, D.sA));
  RLN_HIP(hipStreamSynchronize(D.sA));
}

void Prover::download_partial(size_t n, uint8_t* coords320) {
  Impl& D = *d_;
  sync();
  if (!D.last || D.last->mode != PROVE_PARTIAL || n > D.last->n) throw Error("the last run was not a partial-proof run");
  memcpy(coords320, D.last->h_pp, n * 320);
}

const std::vector<uint8_t>& Prover::known_mask() const { return d_->known; }

void Prover::stage_ms(float out[PROVER_STAGES]) const {
  for (int i = 0; i < PROVER_STAGES; i++) out[i] = d_->ms[i];
}

void Prover::walk_clock_mhz(double out[2]) {
  Impl& D = *d_;
  sync();
  unsigned long long h[4] = {0, 0, 0, 0};
  if (D.walk_clk.p) {
    RLN_HIP(hipMemcpy(h, D.walk_clk.p, sizeof(h), hipMemcpyDeviceToHost));
    RLN_HIP(hipMemset(D.walk_clk.p, 0, sizeof(h)));
  }
  for (int g = 0; g < 2; g++) out[g] = h[2 * g + 1] ? 100.0 * (double)h[2 * g] / (double)h[2 * g + 1] : 0.0;
}

void Prover::download(size_t n, ProofOut* out) {
  if (n > B_) throw Error("batch larger than the prover workspace (max_batch)");
  Impl& D = *d_;
  sync();
  if (!D.last) throw Error("no resident run to read from");
  Slot& S = *D.last;
  std::vector<uint32_t> coords(n * 64);
  RLN_HIP(hipMemcpyAsync(coords.data(), S.coords.p, n * 256, hipMemcpyDeviceToHost, D.sC));
  RLN_HIP(hipStreamSynchronize(D.sC));
  // Without the proof-values kernel (a single message-id circuit whose graph does not carry the shipped input
  // names) the five values are the circuit's own public outputs w[1..6] = y, root, nullifier, x, ext -- the same
  // numbers for every satisfying witness (witness.rs:759-804).  Other shapes (multi message-id) are read by the
  // caller through fetch_public; their `values` are zero here, never stale.
  std::vector<uint8_t> pub;
  const bool from_public = !D.have_values_kernel && D.ni == 6;
  if (from_public) fetch_public(n, &pub);
  for (size_t i = 0; i < n; i++) {
    memcpy(out[i].compressed, S.h_comp + i * 128, 128);
    memcpy(out[i].coords, coords.data() + i * 64, 256);
    if (D.have_values_kernel)
      memcpy(out[i].values, S.h_values + i * 40, 160);
    else if (from_public)
      memcpy(out[i].values, pub.data() + i * 160, 160);
    else
      memset(out[i].values, 0, 160);
    out[i].error = S.h_err[i];
  }
}

void Prover::fetch_public(size_t n, std::vector<uint8_t>* out_le) {
  Impl& D = *d_;
  sync();
  if (!D.last || n > B_) throw Error("no resident run to read from");
  fetch_public_slot(D.last, n, out_le);
}

// the slot's batch must have finished (its evC passed); sC is in stream order behind it
void Prover::fetch_public_slot(void* slot, size_t n, std::vector<uint8_t>* out_le) {
  Impl& D = *d_;
  Slot& S = *(Slot*)slot;
  const uint32_t npub = D.ni - 1;
  DevBuf<uint32_t> tmp(n * npub * 8);
  hipLaunchKernelGGL(k_public_signals, dim3(div_up(n, 64), div_up(npub, 4)), dim3(64, 4), 0, D.sC, S.V.p,
                     D.sig2node.p, npub, (uint32_t)B_, (uint32_t)n, tmp.p);
  out_le->resize(n * npub * 32);
  RLN_HIP(hipMemcpyAsync(out_le->data(), tmp.p, out_le->size(), hipMemcpyDeviceToHost, D.sC));
  RLN_HIP(hipStreamSynchronize(D.sC));
}

void Prover::fetch_witness(size_t p, std::vector<uint8_t>* w_le) {
  Impl& D = *d_;
  sync();
  if (!D.last || p >= B_) throw Error("no resident run to read from");
  DevBuf<uint32_t> tmp((size_t)D.NS * 8);
  hipLaunchKernelGGL(k_gather_col, dim3(div_up(D.NS, 256)), dim3(256), 0, D.sC, D.last->V.p, D.sig2node.p, D.NS,
                     (uint32_t)B_, (uint32_t)p, tmp.p);
  w_le->resize((size_t)D.NS * 32);
  RLN_HIP(hipMemcpyAsync(w_le->data(), tmp.p, w_le->size(), hipMemcpyDeviceToHost, D.sC));
  RLN_HIP(hipStreamSynchronize(D.sC));
}

void Prover::fetch_h(size_t p, std::vector<uint8_t>* h_le) {
  Impl& D = *d_;
  sync();
  if (!D.last || p >= B_) throw Error("no resident run to read from");
  DevBuf<uint32_t> tmp((size_t)D.n * 8);
  hipLaunchKernelGGL(k_gather_col, dim3(div_up(D.n, 256)), dim3(256), 0, D.sC, D.last->abc.p, (const uint32_t*)nullptr,
                     D.n, (uint32_t)B_, (uint32_t)p, tmp.p);
  h_le->resize((size_t)D.n * 32);
  RLN_HIP(hipMemcpyAsync(h_le->data(), tmp.p, h_le->size(), hipMemcpyDeviceToHost, D.sC));
  RLN_HIP(hipStreamSynchronize(D.sC));
}

}  // namespace rlnamd
