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

#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fq29.h"
#include "witness_ops.h"

namespace rlnamd {

constexpr uint32_t WL_W = 32;                 // micro-ops per step (the widest ready set of the shipped circuits is 21)
constexpr uint32_t WL_SLOTS = 3200;           // LDS value slots of 48 bytes: 150 KiB
constexpr uint32_t WL_LDS_BYTES = WL_SLOTS * 48 + 64;
constexpr uint32_t WL_PF = 8;                 // descriptors prefetched per lane (steps ahead)
constexpr double WL_BMAX = 7.5;
enum : uint32_t { WK_FMA = 0, WK_ADD = 1, WK_MISC = 2, WK_SQR = 3 };   // WK_SQR: every lane computes a * a + c
enum : uint32_t { WO_NOP = 0, WO_COMPUTE = 1, WO_INPUT = 2, WO_RARE = 3 };   // MISC steps: what the lane does
constexpr uint32_t WL_STORE = 1u << 8;
// descriptor: x = lane op | WL_STORE | kind << 12 | graph op << 16;  y = dst | a << 16;  z = b | c << 16;  w = V29 slot
//             (dst, a, b, c: LDS slots; WO_INPUT: a = index into the inputs buffer)

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
  if (lane >= WL_W) return;   // the upper half of the wave has no micro-ops
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
      const uint32_t kind = (__builtin_amdgcn_readfirstlane(q.x) >> 12) & 3;
      const uint32_t dst = q.y & 0xFFFF, sa = q.y >> 16, sb = q.z & 0xFFFF, sc = q.z >> 16;
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
namespace {
struct MicroOp {
  uint32_t lop, gop, node;     // node: the graph node this micro-op defines (NONE for a raw value awaiting its reduction)
  uint32_t src[3];             // value ids (NONE: unused; >= FIX: a fixed LDS slot)
  uint32_t dst;                // value id
  uint32_t imm;                // WO_INPUT: index into the inputs buffer
};
constexpr uint32_t NONE = 0xFFFFFFFFu;
int env_int_wl(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
}  // namespace

void WitLanes::build(const Graph& graph, const std::vector<uint32_t>& store_slot, uint32_t trash_slot, hipStream_t s) {
  ok = false;
  if (env_int_wl("RLNAMD_WITLANES", 1) == 0) return;
  const std::vector<GNode>& G = graph.nodes;
  const uint32_t N = (uint32_t)G.size();
  n_consts = (uint32_t)graph.constants.size();
  const uint32_t Z = n_consts, ONE = n_consts + 1, MONE = n_consts + 2, first_free = n_consts + 3;
  const uint32_t DUMMY = WL_SLOTS - 1;
  const uint32_t FIX = (1u << 30) + N;    // value ids >= FIX address a fixed LDS slot (ZERO / ONE / MINUS_ONE)
  if (first_free + 64 >= DUMMY) return;   // the constants alone (nearly) fill the LDS
  auto nops = [&](const GNode& g) {
    return (g.op == G_INPUT || g.op == G_CONST) ? 0 : (g.op == G_NEG || g.op == G_ID) ? 1 : g.op == G_TERN ? 3 : 2;
  };
  std::vector<uint8_t> is_signal(N, 0);
  for (uint32_t sg : graph.signals) is_signal[sg] = 1;
  std::vector<std::vector<uint32_t>> users(N);
  std::vector<uint32_t> uses(N, 0);
  for (uint32_t n = 0; n < N; n++) {
    const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
    for (int k = 0; k < nops(G[n]); k++) {
      if (o[k] >= n) throw Error("Graph error: node operand refers forward");
      uses[o[k]]++;
      if (users[o[k]].empty() || users[o[k]].back() != n) users[o[k]].push_back(n);
    }
  }
  // Values: one per graph node (id = node) plus temporaries (raw results awaiting a reduction), ids >= N.
  // avail[v]: produced (constants: always).  bound[v] in units of r.
  std::vector<uint8_t> avail(N, 0), done(N, 0);
  std::vector<double> bound(N, 1.01);
  for (uint32_t n = 0; n < N; n++)
    if (G[n].op == G_CONST) avail[n] = done[n] = 1;
  std::vector<std::vector<MicroOp>> steps;
  std::vector<uint32_t> step_kind;
  std::vector<uint32_t> ready;
  std::vector<uint8_t> in_ready(N, 0);
  auto operands_avail = [&](uint32_t n) {
    const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
    for (int k = 0; k < nops(G[n]); k++)
      if (!avail[o[k]]) return false;
    return true;
  };
  for (uint32_t n = 0; n < N; n++)
    if (G[n].op != G_CONST && operands_avail(n)) {
      ready.push_back(n);
      in_ready[n] = 1;
    }
  auto is_rare = [&](uint32_t n) {
    const uint32_t op = G[n].op;
    return !(op == G_MUL || op == G_ADD || op == G_SUB || op == G_NEG);
  };
  struct Pending { uint32_t node, raw; };   // a raw value that still needs x * ONE + ZERO to become `node`
  std::vector<Pending> pending;
  uint32_t next_tmp = N;
  std::vector<double> tmp_bound;
  auto bnd = [&](uint32_t v) { return v < N ? bound[v] : tmp_bound[v - N]; };
  nfma = nadd = nmisc = nsqr = 0;
  while (!ready.empty() || !pending.empty()) {
    std::sort(ready.begin(), ready.end());
    std::vector<MicroOp> ops;
    std::vector<uint32_t> produced, left;
    uint32_t kind;
    bool any_rare = false, any_mul = false;
    for (uint32_t n : ready) {
      any_rare |= is_rare(n);
      any_mul |= G[n].op == G_MUL || G[n].op == G_SUB || G[n].op == G_NEG;
    }
    if (any_rare) {
      kind = WK_MISC;
      for (uint32_t n : ready) {
        if (!is_rare(n) || ops.size() >= WL_W) { left.push_back(n); continue; }
        MicroOp m{G[n].op == G_INPUT ? WO_INPUT : WO_RARE, G[n].op, n, {G[n].a, G[n].b, G[n].c}, n, G[n].a};
        for (int k = nops(G[n]); k < 3; k++) m.src[k] = NONE;
        ops.push_back(m);
        produced.push_back(n);
        bound[n] = G[n].op == G_TERN ? std::max(bound[G[n].b], bound[G[n].c]) : 1.01;
      }
    } else {
      // an Add whose plain sum would pass the bound forces the product form for the whole step
      bool force_fma = !pending.empty();
      for (uint32_t n : ready)
        if (G[n].op == G_ADD && bound[G[n].a] + bound[G[n].b] > WL_BMAX) force_fma = true;
      kind = (any_mul || force_fma) ? WK_FMA : WK_ADD;
      if (kind == WK_FMA) {
        if (pending.size() > WL_W / 2) return;   // not a graph this form is meant for
        for (const Pending& pd : pending) {   // reductions first: their consumers are waiting
          ops.push_back(MicroOp{WO_COMPUTE, G_MUL, pd.node, {pd.raw, FIX + ONE, FIX + Z}, pd.node, 0});
          produced.push_back(pd.node);
          bound[pd.node] = 1.0 + 0.006 * bnd(pd.raw);
        }
        pending.clear();
      }
      for (uint32_t n : ready) {
        if (ops.size() >= WL_W) { left.push_back(n); continue; }
        const GNode& g = G[n];
        if (kind == WK_ADD) {
          ops.push_back(MicroOp{WO_COMPUTE, G_ADD, n, {g.a, g.b, NONE}, n, 0});
          bound[n] = bound[g.a] + bound[g.b];
          produced.push_back(n);
          continue;
        }
        MicroOp m{WO_COMPUTE, g.op, n, {NONE, NONE, NONE}, n, 0};
        double b;
        if (g.op == G_MUL) {
          // fuse with its only user when that is an Add of a value that is already there (a * b + c in one step)
          uint32_t add = NONE, c = NONE;
          if (uses[n] == 1 && !is_signal[n] && store_slot[n] == NONE) {
            const uint32_t u = users[n][0];
            if (G[u].op == G_ADD && G[u].a != G[u].b) {
              const uint32_t other = G[u].a == n ? G[u].b : G[u].a;
              if (avail[other] && bound[other] + 1.0 + 0.006 * bound[g.a] * bound[g.b] <= WL_BMAX) { add = u; c = other; }
            }
          }
          if (add != NONE) {
            m.node = m.dst = add;
            m.src[0] = g.a; m.src[1] = g.b; m.src[2] = c;
            b = 1.0 + 0.006 * bound[g.a] * bound[g.b] + bound[c];
            done[n] = 1;   // never materialised
            bound[add] = b;
            ops.push_back(m);
            produced.push_back(add);
            continue;
          }
          m.src[0] = g.a; m.src[1] = g.b; m.src[2] = FIX + Z;
          b = 1.0 + 0.006 * bound[g.a] * bound[g.b];
        } else if (g.op == G_ADD) {   // x * 1 + y, the larger bound as multiplicand
          const uint32_t x = bound[g.a] >= bound[g.b] ? g.a : g.b, y = x == g.a ? g.b : g.a;
          m.src[0] = x; m.src[1] = FIX + ONE; m.src[2] = y;
          b = 1.0 + 0.006 * bound[x] + bound[y];
        } else if (g.op == G_SUB) {   // a - b = b * (-1) + a
          m.src[0] = g.b; m.src[1] = FIX + MONE; m.src[2] = g.a;
          b = 1.0 + 0.006 * bound[g.b] * 1.05 + bound[g.a];
        } else {                      // G_NEG
          m.src[0] = g.a; m.src[1] = FIX + MONE; m.src[2] = FIX + Z;
          b = 1.0 + 0.006 * bound[g.a] * 1.05;
        }
        if (b > WL_BMAX) {            // leave the raw value in a temporary and reduce it in the next FMA step
          const uint32_t raw = next_tmp++;
          tmp_bound.push_back(b);
          m.dst = raw;
          m.node = NONE;
          pending.push_back({n, raw});
          ops.push_back(m);
          continue;
        }
        bound[n] = b;
        ops.push_back(m);
        produced.push_back(n);
      }
    }
    if (kind == WK_FMA) {   // all products squarings (idle lanes compute ZERO * ZERO + ZERO: a square as well)?
      bool all_sq = !ops.empty();
      for (const MicroOp& m : ops) all_sq = all_sq && m.src[0] == m.src[1];
      if (all_sq) { kind = WK_SQR; nsqr++; }
    }
    if (kind == WK_FMA) nfma++; else if (kind == WK_ADD) nadd++; else if (kind == WK_MISC) nmisc++;
    steps.push_back(ops);
    step_kind.push_back(kind);
    for (uint32_t n : ready) in_ready[n] = 0;
    ready.swap(left);
    for (uint32_t n : ready) in_ready[n] = 1;
    for (uint32_t n : produced) { avail[n] = 1; done[n] = 1; }
    for (uint32_t n : produced)
      for (uint32_t u : users[n])
        if (!done[u] && !in_ready[u] && operands_avail(u)) {
          // a product already folded into its Add is done; an Add whose product was folded is produced by that step
          ready.push_back(u);
          in_ready[u] = 1;
        }
    // a node folded into an FMA (done, not avail) must not be scheduled again: drop it from `ready`
    ready.erase(std::remove_if(ready.begin(), ready.end(), [&](uint32_t n) { return done[n]; }), ready.end());
  }
  for (uint32_t n = 0; n < N; n++)
    if (!done[n]) throw Error("witness lanes: graph node left unscheduled");
  // ---- LDS slots from the liveness of the schedule
  const uint32_t nvals = next_tmp;
  const uint32_t FIXB = FIX;
  std::vector<uint32_t> last_use(nvals, 0), slot(nvals, NONE);
  for (uint32_t t = 0; t < steps.size(); t++)
    for (const MicroOp& m : steps[t])
      for (int k = 0; k < 3; k++)
        if (m.src[k] != NONE && m.src[k] < FIXB) last_use[m.src[k]] = t;
  for (uint32_t n = 0; n < N; n++)
    if (G[n].op == G_CONST) slot[n] = G[n].a;   // constants sit in their own slots
  std::vector<uint32_t> free_slots;
  for (uint32_t sl = DUMMY; sl-- > first_free;) free_slots.push_back(sl);
  std::vector<std::vector<uint32_t>> dies(steps.size() + 1);
  peak_slots = 0;
  uint32_t live = 0;
  // idle lanes and the padding steps (the kernel runs whole groups of WL_PF steps and prefetches one group further)
  // compute ZERO * ZERO + ZERO into the dummy slot
  std::vector<uint4> img((steps.size() + 2 * WL_PF) * (size_t)WL_W,
                         make_uint4(0, DUMMY | (Z << 16), Z | (Z << 16), trash_slot));
  auto slot_of_val = [&](uint32_t v) -> uint32_t {
    if (v == NONE) return Z;
    if (v >= FIXB) return v - FIXB;
    if (slot[v] == NONE) throw Error("witness lanes: operand read before it was produced");
    return slot[v];
  };
  for (uint32_t t = 0; t < steps.size(); t++) {
    // operands first (their slots may be those of values that die here), then the results
    std::vector<uint32_t> sa(steps[t].size() * 3);
    for (size_t i = 0; i < steps[t].size(); i++)
      for (int k = 0; k < 3; k++) sa[3 * i + k] = slot_of_val(steps[t][i].src[k]);
    for (size_t i = 0; i < steps[t].size(); i++) {
      const MicroOp& m = steps[t][i];
      if (free_slots.empty()) return;   // more live values than LDS slots: keep k_witness29
      const uint32_t sl = free_slots.back();
      free_slots.pop_back();
      slot[m.dst] = sl;
      live++;
      peak_slots = std::max(peak_slots, live);
      // a value nobody reads (a signal that is only stored) dies at once
      dies[std::max(last_use[m.dst], t)].push_back(m.dst);
      uint32_t x = m.lop | (step_kind[t] << 12) | (m.gop << 16), w = trash_slot;
      if (m.node != NONE && store_slot[m.node] != NONE) {
        x |= WL_STORE;
        w = store_slot[m.node];
      }
      const uint32_t fa = m.lop == WO_INPUT ? m.imm : sa[3 * i];
      if (fa >= 65536) return;          // an input index that does not fit the descriptor: keep k_witness29
      img[(size_t)t * WL_W + i] = make_uint4(x, sl | (fa << 16), sa[3 * i + 1] | (sa[3 * i + 2] << 16), w);
    }
    // every descriptor of the step carries the kind (lane 0's is the one the kernel reads)
    for (uint32_t i = (uint32_t)steps[t].size(); i < WL_W; i++) img[(size_t)t * WL_W + i].x = step_kind[t] << 12;
    if (steps[t].empty()) img[(size_t)t * WL_W].x = step_kind[t] << 12;
    for (uint32_t v : dies[t]) {
      free_slots.push_back(slot[v]);
      live--;
    }
  }
  nsteps = (uint32_t)steps.size();
  prog.alloc(img.size());
  prog.upload(img.data(), img.size(), s);
  RLN_HIP(hipStreamSynchronize(s));
  if (hipFuncSetAttribute((const void*)k_witness_lanes, hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES) !=
      hipSuccess) {   // a device with less LDS per workgroup: keep k_witness29
    (void)hipGetLastError();
    return;
  }
  ok = true;
  if (env_int_wl("RLNAMD_WITLANES_INFO", 0))
    fprintf(stderr, "witness lanes: %u steps (%u fma, %u sqr, %u add, %u misc), peak %u live values, %u constants\n", nsteps,
            nfma, nsqr, nadd, nmisc, peak_slots, n_consts);
}

void WitLanes::launch(hipStream_t s, const uint32_t* d_consts29, const uint32_t* d_inputs, uint32_t n_inputs, uint4* V29,
                      uint32_t* err, uint32_t B, uint32_t nb) const {
  hipLaunchKernelGGL(k_witness_lanes, dim3(nb), dim3(64), WL_LDS_BYTES, s, prog.p, nsteps, d_consts29, n_consts, d_inputs,
                     n_inputs, V29, err, B);
}

}  // namespace rlnamd
