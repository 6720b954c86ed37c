// prover_front.hip -- front end of the prover pipeline: the witness-graph interpreters, the QAP mat-vec, the NTT passes
// and the quotient, the digit recoding.  Declarations and shared descriptors: prover_kernels.h; host side: prover.hip.
//
// The 256-bit multiply is inlined (measured on MI355X: G1 MSM 44.5 -> 41.7 ms, G2 MSM 30.7 -> 22.5 ms per 1024
// proofs against the out-of-line form, which costs call overhead and a VGPR-hungry calling convention);
// -DRLN_NOINLINE_MUL restores the shared 2.5 KB body.
#include "prover_kernels.h"

#include "glv.h"

namespace rlnamd {

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
                                                  uint32_t lg) {
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
// LG (small batches): lanes = rows of ONE proof (blockIdx.y) instead of lanes = proofs -- with lanes = proofs a single
// proof launches 8 192 waves with one useful lane each, which also crowd the walks that run beside them
// Long rows (LG): the circuit's matrices hold 2 entries in most rows and 60 + 60 in ninety of them (Poseidon's mix
// layers), and a lane that walks 120 entries alone -- two dependent loads and a product each -- is the whole kernel
// (0.40 ms for one proof).  Rows with more than MV_LONG entries in A or B are therefore taken out of the lanes = rows
// part and given a wave each (blocks >= nshort): a lane per entry, then a shuffle tree of field additions (exact, so the
// order of the sum does not matter).
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
                                                const uint32_t* __restrict__ long_rows, uint32_t nshort) {
  if (LG) __builtin_amdgcn_s_setprio(3);   // small batches: a link of the latency chain (see k_ntt_mid)
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
// (Small batches do not come here: k_ntt_edge / k_ntt_mid below; a circuit with fewer than 512 constraints would, with its
// proofs in the lanes like any batch.)
template <int K, bool DIF>
__global__ void __launch_bounds__(256, RLN_NTT_WAVES) k_ntt_pass(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, int s0,
                                                  const Fr* __restrict__ scale, uint32_t B, uint32_t nb) {
  // (twiddle products through Fr29::mul_mont -- ~290 instead of ~375 instructions -- were measured neutral twice: beside the
  // table walks the passes are bound by HBM and by waiting for SIMD slots; that variant is gone)
  auto tmul = [&](const Fr& a, const Fr* __restrict__ tab, uint32_t idx) -> Fr { return a * tab[idx]; };
  constexpr int R = 1 << K;
  const uint32_t n = 1u << logn;
  auto uni = [](uint32_t v) -> uint32_t { return __builtin_amdgcn_readfirstlane(v); };
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  uint32_t g = __builtin_amdgcn_readfirstlane(blockIdx.y * blockDim.y + threadIdx.y);  // one group per wave
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
      // twiddle rides in SGPRs instead of VGPRs
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

// Small batches (lanes = groups), logn >= 9: the three transforms of the quotient in THREE kernels, one butterfly per lane
// per level.  A lone proof's transform is all latency: with eight points per lane (k_ntt_pass, and the nine-level kernel
// that stood here until round 4) a lane's chain is 12 twiddle products per pass -- 38 + 15 + 10 us inverse,
// 57 + 15 + 42 us forward for 6 us of arithmetic.  Here a workgroup of 256 lanes (a wave per SIMD) owns 512 points in 16 KB
// of LDS and every level costs a lane ONE product:
//   k_ntt_edge<true>   inverse (DIF) levels 0 .. E-1, E = logn - 9: the sets {lo + 512 m}, 512 >> E of them per workgroup
//   k_ntt_mid          inverse levels E .. logn-1 over 512 CONTIGUOUS points, the coset / 1/n scaling, and forward (DIT)
//                      levels 0 .. 8 over the same points (DIF leaves bit-reversed order, DIT starts from it: the low
//                      nine levels of both close over the same 512 positions) -- eighteen levels, one kernel
//   k_ntt_edge<false>  forward levels 9 .. logn-1
// Same butterflies, same twiddles, same products per point as the passes: bit-identical (Fr products are canonical).
__global__ void __launch_bounds__(256) k_ntt_mid(Fr* __restrict__ data, const Fr* __restrict__ tw_i,
                                                 const Fr* __restrict__ tw_f, int logn, const Fr* __restrict__ scale,
                                                 uint32_t B, uint32_t nb) {
  __shared__ Fr buf[512];
  __builtin_amdgcn_s_setprio(3);   // the quotient chain is short and the h rows wait for it: issue ahead of the walks' waves
  const uint32_t p = blockIdx.x, l = threadIdx.x;
  const uint32_t n = 1u << logn, base = blockIdx.y * 512;
  if (p >= nb) return;
  Fr* x = data + (size_t)blockIdx.z * n * B + p;
  buf[l] = x[(size_t)(base + l) * B];
  buf[l + 256] = x[(size_t)(base + l + 256) * B];
  __syncthreads();
  const int E = logn - 9;
#pragma unroll 1
  for (int t = 0; t < 8; t++) {   // inverse level E + t: half = 256 >> t
    const uint32_t hl = 256u >> t, jl = l & (hl - 1), i0 = ((l >> (8 - t)) << (9 - t)) | jl, i1 = i0 + hl;
    const Fr u = buf[i0], v = buf[i1];
    buf[i0] = u + v;
    buf[i1] = (u - v) * tw_i[(size_t)jl << (E + t)];
    __syncthreads();
  }
  {   // the last inverse level (half = 1), the scaling and the first forward level (half = 1): the same two points
    const uint32_t i0 = 2 * l, i1 = i0 + 1;
    const Fr u = buf[i0], v = buf[i1];
    Fr a = u + v, b = u - v;   // (both levels' twiddle is tw[0] = 1: the product would return its operand)
    if (scale) {
      a = a * scale[base + i0];
      b = b * scale[base + i1];
    }
    buf[i0] = a + b;
    buf[i1] = a - b;
    __syncthreads();
  }
#pragma unroll 1
  for (int s = 1; s < 9; s++) {   // forward level s: half = 1 << s
    const uint32_t hl = 1u << s, jl = l & (hl - 1), i0 = ((l >> s) << (s + 1)) | jl, i1 = i0 + hl;
    const Fr u = buf[i0], v = buf[i1] * tw_f[(size_t)jl << (logn - 1 - s)];
    buf[i0] = u + v;
    buf[i1] = u - v;
    __syncthreads();
  }
  x[(size_t)(base + l) * B] = buf[l];
  x[(size_t)(base + l + 256) * B] = buf[l + 256];
}

template <bool DIF>
__global__ void __launch_bounds__(256) k_ntt_edge(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, uint32_t B,
                                                  uint32_t nb) {
  __shared__ Fr buf[512];
  __builtin_amdgcn_s_setprio(3);
  const uint32_t p = blockIdx.x, l = threadIdx.x;
  const uint32_t n = 1u << logn;
  if (p >= nb) return;
  const int E = logn - 9;                       // 1 .. 9
  const uint32_t cl = 512u >> E, lo0 = blockIdx.y * cl;   // local point i = m cl + c  <->  position lo0 + c + 512 m
  Fr* x = data + (size_t)blockIdx.z * n * B + p;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t i = l + 256 * k;
    buf[i] = x[(size_t)(lo0 + (i & (cl - 1)) + 512 * (i >> (9 - E))) * B];
  }
  __syncthreads();
  const uint32_t c = l & (cl - 1), q = l >> (9 - E);
#pragma unroll 1
  for (int t = 0; t < E; t++) {
    const int lh = DIF ? E - 1 - t : t;         // log2 of the level's half in units of m
    const uint32_t hm = 1u << lh, jm = q & (hm - 1), m0 = ((q >> lh) << (lh + 1)) | jm;
    const uint32_t i0 = m0 * cl + c, i1 = i0 + hm * cl;
    const size_t j = (size_t)jm * 512 + lo0 + c;
    if (DIF) {
      const Fr u = buf[i0], v = buf[i1];
      buf[i0] = u + v;
      buf[i1] = (u - v) * tw[j << t];
    } else {
      const Fr u = buf[i0], v = buf[i1] * tw[j << (logn - 1 - (9 + t))];
      buf[i0] = u + v;
      buf[i1] = u - v;
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t i = l + 256 * k;
    x[(size_t)(lo0 + (i & (cl - 1)) + 512 * (i >> (9 - E))) * B] = buf[i];
  }
}

// h = a o b - c  (qap.rs:84-95), written over the `a` vector
__global__ void __launch_bounds__(256) k_hquot(Fr* __restrict__ abc, uint32_t n, uint32_t B, uint32_t nb, uint32_t lg) {
  if (lg) __builtin_amdgcn_s_setprio(3);   // small batches: a link of the latency chain (see k_ntt_mid)
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
                                                uint32_t ns, const Fr* H, uint32_t n,
                                                const uint32_t* __restrict__ rs, WinSched ws1, WinSched ws2, uint32_t nh,
                                                int16_t* __restrict__ dig1, int16_t* __restrict__ dig2, uint32_t B,
                                                uint32_t nb, uint32_t part, uint32_t lg, uint32_t dB) {
  // dB: the proof stride of the digit arrays ([scalar][half][window][dB]).  The batch capacity B in the throughput
  // shapes; the batch SIZE for small batches -- with B = 64 a lone proof's digits sat one per 128-byte line (64 scattered
  // two-byte stores per lane here, a miss per step in the walks); compact, a scalar's windows share one line.
  // part 0: every scalar; 1: the witness scalars and r, s, -(r s) (all the G2 walk needs: it can start before the
  // quotient h exists); 2: the coefficients of h only
  // part 3 (small full proofs, fused plan): the products s w_i, r w_i and r s under the ids ns + n + 3 + ..., G1 only
  if (lg) __builtin_amdgcn_s_setprio(3);   // small batches: a link of the latency chain (see k_ntt_mid)
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
    const size_t o = (size_t)(sid - ns) * B + p;
    // lg == 2 (small batches): the quotient on the fly, h = a o b - c (k_hquot's line: same products, same bytes) -- one
    // kernel and one boundary less on the chain the h rows wait for
    x = lg == 2 ? H[o] * H[(size_t)n * B + o] - H[2 * (size_t)n * B + o] : H[o];
    if (lg == 2) const_cast<Fr*>(H)[o] = x;   // where k_hquot leaves it (Prover::fetch_h reads it there); one lane per element
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
        emit_digits<4>(t, neg[h] != 0, ws2, dig2 + ((size_t)sid2 * 2 + h) * ws2.W * dB + p, dB);
      }
      emit_digits<4>(k[h], neg[h] != 0, ws1, dig1 + ((size_t)sid * 2 + h) * ws1.W * dB + p, dB);
    }
  } else {
    if (g2) {
      uint32_t t[8];
#pragma unroll
      for (int i = 0; i < 8; i++) t[i] = l[i];
      emit_digits<8>(t, false, ws2, dig2 + (size_t)sid2 * ws2.W * dB + p, dB);
    }
    emit_digits<8>(l, false, ws1, dig1 + (size_t)sid * ws1.W * dB + p, dB);
  }
}


// ---- explicit instantiations: every form the host launches
template __global__ void k_witness29<false>(const GNode29* __restrict__ nodes, uint32_t n_nodes, const uint32_t* __restrict__ consts29, uint32_t n_consts, const uint32_t* __restrict__ inputs, uint32_t n_inputs, uint4* __restrict__ V29, uint32_t* __restrict__ err, uint32_t B, uint32_t nb, unsigned long long* __restrict__ prof);
template __global__ void k_matvec<true>(CsrView A, CsrView Bm, const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node, uint32_t nc, uint32_t ni, uint32_t n, Fr* __restrict__ abc, uint32_t B, uint32_t nb, const uint32_t* __restrict__ long_rows, uint32_t nshort);
template __global__ void k_matvec<false>(CsrView A, CsrView Bm, const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node, uint32_t nc, uint32_t ni, uint32_t n, Fr* __restrict__ abc, uint32_t B, uint32_t nb, const uint32_t* __restrict__ long_rows, uint32_t nshort);
template __global__ void k_ntt_pass<1, true>(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, int s0, const Fr* __restrict__ scale, uint32_t B, uint32_t nb);
template __global__ void k_ntt_pass<1, false>(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, int s0, const Fr* __restrict__ scale, uint32_t B, uint32_t nb);
template __global__ void k_ntt_pass<2, true>(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, int s0, const Fr* __restrict__ scale, uint32_t B, uint32_t nb);
template __global__ void k_ntt_pass<2, false>(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, int s0, const Fr* __restrict__ scale, uint32_t B, uint32_t nb);
template __global__ void k_ntt_pass<3, true>(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, int s0, const Fr* __restrict__ scale, uint32_t B, uint32_t nb);
template __global__ void k_ntt_pass<3, false>(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, int s0, const Fr* __restrict__ scale, uint32_t B, uint32_t nb);
template __global__ void k_ntt_edge<true>(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, uint32_t B, uint32_t nb);
template __global__ void k_ntt_edge<false>(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, uint32_t B, uint32_t nb);

}  // namespace rlnamd
