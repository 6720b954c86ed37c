// Host emulator of the lanes = nodes witness interpreter: runs the micro-op program that zerokit_amd/csrc/witness_sched.cpp
// emits (steps, LDS slot assignment, fusion, reductions) with the product's own host field arithmetic and graph
// operations, so the CPU suite can check the SCHEDULE against the golden witness digests without a GPU.  The device
// kernel differs only in how a value is represented (9 x 29-bit limbs, lazily reduced); residues mod r are the same.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "field.h"
#include "witness_ops.h"
#include "witness_sched.h"
#include "zkey.h"

using namespace rlnamd;

static std::string g_err;

extern "C" {
const char* witsched_error() { return g_err.c_str(); }

// runs program P over `inputs_le`, writing its stores into `stored`; returns the error flags
static uint32_t emulate(const WlProgram& P, const uint8_t* inputs_le, std::vector<Fr>& stored) {
  std::vector<Fr> lds(WL_SLOTS, Fr::zero());
  const uint32_t nc = P.n_consts;
  if (P.consts.size() != nc) throw std::runtime_error("program constants");
  for (uint32_t i = 0; i < nc; i++) lds[i] = P.consts[i];
  lds[nc] = Fr::zero();
  lds[nc + 1] = Fr::one();
  lds[nc + 2] = Fr::one().neg();
  uint32_t err = 0;
  for (uint32_t t = 0; t < P.nsteps + WL_PF; t++) {   // the kernel runs whole groups of WL_PF steps: padding included
    const WlDesc* d = &P.img[(size_t)t * WL_W];
    const uint32_t kind = (d[0].x >> 12) & 7;
    // every lane reads its operands before any lane writes: two phases
    std::vector<std::pair<uint32_t, Fr>> writes;
    std::vector<std::pair<uint32_t, Fr>> stores;
    const uint32_t lanes = WL_W, stride = kind == WK_ROW ? 16 : 1;
    for (uint32_t l = 0; l < lanes; l += stride) {
      const WlDesc& q = d[l];
      if (((q.x >> 12) & 7) != kind) throw std::runtime_error("step kind differs between lanes");
      const uint32_t dst = q.y & 0xFFFF, sa = q.y >> 16, sb = q.z & 0xFFFF, sc = q.z >> 16, lop = q.x & 0xFF,
                     gop = (q.x >> 16) & 0xFF;
      Fr v = Fr::zero();
      if (kind == WK_FMA || kind == WK_ROW) {
        v = lds[sa] * lds[sb] + lds[sc];
      } else if (kind == WK_SQR) {
        if (sa != sb) throw std::runtime_error("SQR step with a != b");
        v = lds[sa] * lds[sa] + lds[sc];
      } else if (kind == WK_ADD) {
        v = lds[sa] + lds[sb];
      } else if (lop == WO_INPUT) {
        uint32_t c[8];
        memcpy(c, inputs_le + (size_t)sa * 32, 32);
        if (limbs_geq(c, FrParams::MOD)) err = WERR_INPUT_RANGE;
        v = Fr::from_canonical(c);
      } else if (lop == WO_RARE) {
        if (gop == G_TERN) v = lds[sa].is_zero() ? lds[sc] : lds[sb];
        else v = witness_slow_op(gop, lds[sa], lds[sb], &err);
      }
      if (kind == WK_ROW) {
        for (uint32_t k = 1; k < 16; k++)
          if (memcmp(&d[l + k], &q, sizeof(q)) != 0) throw std::runtime_error("row descriptor not replicated");
      }
      writes.push_back({dst, v});
      if (q.x & WL_STORE) stores.push_back({q.w, v});
    }
    for (auto& w : writes) lds[w.first] = w.second;
    for (auto& st : stores) {
      if (st.first >= stored.size()) throw std::runtime_error("store slot out of range");
      stored[st.first] = st.second;
    }
  }
  return err;
}
// store every witness signal (and every input), as the prover does
static void store_slots(const Graph& g, std::vector<uint32_t>* store_slot, std::vector<uint32_t>* slot2node) {
  const uint32_t N = (uint32_t)g.nodes.size(), NONE = 0xFFFFFFFFu;
  std::vector<uint8_t> is_signal(N, 0);
  for (uint32_t sg : g.signals) is_signal[sg] = 1;
  store_slot->assign(N, NONE);
  for (uint32_t n = 0; n < N; n++)
    if (g.nodes[n].op == G_INPUT || is_signal[n]) {
      (*store_slot)[n] = (uint32_t)slot2node->size();
      slot2node->push_back(n);
    }
}
static void witness_out(const Graph& g, const std::vector<uint32_t>& store_slot, const std::vector<Fr>& stored, uint8_t* out_le) {
  for (size_t i = 0; i < g.signals.size(); i++) {
    uint32_t c[8];
    const uint32_t node = g.signals[i];
    if (g.nodes[node].op == G_CONST) g.constants[g.nodes[node].a].to_canonical(c);
    else stored[store_slot[node]].to_canonical(c);
    memcpy(out_le + 32 * i, c, 32);
  }
}

// graph: graph.bin bytes; inputs_le: inputs_size x 32 canonical LE (slot 0 = 1); rows: 0 lane form, 1 row form.
// witness_out_le: num_signals x 32 canonical LE.  stats[0..7] = steps, row, fma, sqr, add, misc, peak live values, error flag
int witsched_run(const uint8_t* graph, size_t len, const uint8_t* inputs_le, size_t inputs_size, int rows,
                 uint8_t* witness_out_le, uint32_t* stats) {
  try {
    Graph g = parse_graph(graph, len);
    if (inputs_size != g.inputs_size) throw std::runtime_error("inputs size mismatch");
    std::vector<uint32_t> store_slot, slot2node;
    store_slots(g, &store_slot, &slot2node);
    const uint32_t trash = (uint32_t)slot2node.size();
    WlProgram P = wl_schedule(g, store_slot, trash, rows != 0);
    if (!P.ok) throw std::runtime_error("graph does not fit the lanes form");
    std::vector<Fr> stored(slot2node.size() + 1, Fr::zero());
    const uint32_t err = emulate(P, inputs_le, stored);
    witness_out(g, store_slot, stored, witness_out_le);
    stats[0] = P.nsteps; stats[1] = P.nrow; stats[2] = P.nfma; stats[3] = P.nsqr; stats[4] = P.nadd; stats[5] = P.nmisc;
    stats[6] = P.peak_slots; stats[7] = err;
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

// Segments behind hints (witness_sched.h: wl_segments) on the host.  hints_le: n_hints x 32 canonical LE -- the values
// the caller claims for the chain between the hashes (the test computes them with the Python oracle's Poseidon).  The cut
// nodes are found as the prover finds them: every computed node of a plain host evaluation whose value equals a hint.
// Every segment program runs by itself on rows prefilled with junk; the witness read off the rows must be the full one,
// and every cut node's computed value must equal its hint.  stats[0..7] = of the LONGEST segment; stats[8] = segments,
// [9] = cut nodes, [10] = sum of all segments' steps, [11] = hint mismatches.
int witsched_run_segments(const uint8_t* graph, size_t len, const uint8_t* inputs_le, size_t inputs_size,
                          const uint8_t* hints_le, uint32_t n_hints, int rows, uint8_t* witness_out_le, uint32_t* stats) {
  try {
    Graph g = parse_graph(graph, len);
    if (inputs_size != g.inputs_size) throw std::runtime_error("inputs size mismatch");
    std::vector<uint32_t> store_slot, slot2node;
    store_slots(g, &store_slot, &slot2node);
    uint32_t e0 = 0;
    std::vector<Fr> plain = wl_eval_host(g, inputs_le, &e0);
    std::vector<Fr> hints(n_hints);
    std::vector<std::vector<uint32_t>> cuts(n_hints);
    for (uint32_t j = 0; j < n_hints; j++) {
      uint32_t c[8];
      memcpy(c, hints_le + 32 * (size_t)j, 32);
      hints[j] = Fr::from_canonical(c);
      for (uint32_t n = 0; n < g.nodes.size(); n++)
        if (g.nodes[n].op != G_INPUT && g.nodes[n].op != G_CONST && plain[n] == hints[j]) cuts[j].push_back(n);
      if (cuts[j].empty()) throw std::runtime_error("a hint matches no node of the graph");
    }
    // cut nodes are stored values too (the check reads them)
    for (auto& cj : cuts)
      for (uint32_t n : cj)
        if (store_slot[n] == 0xFFFFFFFFu) { store_slot[n] = (uint32_t)slot2node.size(); slot2node.push_back(n); }
    const uint32_t trash = (uint32_t)slot2node.size();
    WlSegments S = wl_segments(g, cuts);
    std::vector<uint8_t> ext(inputs_le, inputs_le + inputs_size * 32);
    ext.insert(ext.end(), hints_le, hints_le + 32 * (size_t)n_hints);
    Fr junk = Fr::one() + Fr::one() + Fr::one();
    std::vector<Fr> stored(slot2node.size() + 1, junk);
    uint32_t err = 0, longest = 0, total = 0;
    WlProgram L;
    for (size_t k = 0; k < S.graphs.size(); k++) {
      WlProgram Q = wl_schedule(S.graphs[k], wl_segment_store_slots(S, k, store_slot), trash, rows != 0);
      if (!Q.ok) throw std::runtime_error("a segment does not fit the lanes form");
      err |= emulate(Q, ext.data(), stored);
      total += Q.nsteps;
      if (getenv("WITSCHED_PRINT_SEGMENTS"))
        fprintf(stderr, "segment %zu: %zu nodes, %u steps (row %u, misc %u), %u constants\n", k, S.graphs[k].nodes.size(), Q.nsteps, Q.nrow, Q.nmisc, Q.n_consts);
      if (Q.nsteps >= longest) { longest = Q.nsteps; L = Q; }
    }
    uint32_t mism = 0;
    for (size_t i = 0; i < S.cut_nodes.size(); i++)
      if (!(stored[store_slot[S.cut_nodes[i]]] == hints[S.cut_hint[i]])) mism++;
    witness_out(g, store_slot, stored, witness_out_le);
    stats[0] = L.nsteps; stats[1] = L.nrow; stats[2] = L.nfma; stats[3] = L.nsqr; stats[4] = L.nadd; stats[5] = L.nmisc;
    stats[6] = L.peak_slots; stats[7] = err;
    stats[8] = (uint32_t)S.graphs.size(); stats[9] = (uint32_t)S.cut_nodes.size(); stats[10] = total; stats[11] = mism;
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

// The finish path of round 6 on the host: (1) the FULL program over the PARTIAL witness (the per-message inputs zeroed:
// what RLNAMD_MODE_PARTIAL runs) leaves the stored rows; (2) every row of an unknown node is overwritten with junk (the
// cone must produce all of them); (3) the CONE program (wl_cone: the unknown nodes + the few known ones they read) runs
// over the full inputs on top of those rows.  witness_out_le = the witness read off the rows afterwards: it must be the
// full witness.  stats as above for the cone program; stats[8..11] = cone nodes, unknown nodes, recomputed known nodes,
// known stored rows.
int witsched_run_cone(const uint8_t* graph, size_t len, const uint8_t* inputs_le, size_t inputs_size, int rows,
                      uint8_t* witness_out_le, uint32_t* stats) {
  try {
    Graph g = parse_graph(graph, len);
    if (inputs_size != g.inputs_size) throw std::runtime_error("inputs size mismatch");
    std::vector<uint32_t> store_slot, slot2node;
    store_slots(g, &store_slot, &slot2node);
    const uint32_t trash = (uint32_t)slot2node.size();
    WlProgram P = wl_schedule(g, store_slot, trash, rows != 0);
    if (!P.ok) throw std::runtime_error("graph does not fit the lanes form");
    WlCone C = wl_cone(g);
    std::vector<uint8_t> partial_in(inputs_le, inputs_le + inputs_size * 32);
    for (uint32_t n = 0; n < g.nodes.size(); n++)
      if (g.nodes[n].op == G_INPUT && !C.node_known[n]) memset(partial_in.data() + 32 * (size_t)g.nodes[n].a, 0, 32);
    std::vector<Fr> stored(slot2node.size() + 1, Fr::zero());
    (void)emulate(P, partial_in.data(), stored);
    uint32_t known_rows = 0;
    Fr junk = Fr::one() + Fr::one() + Fr::one();
    for (uint32_t sl = 0; sl < slot2node.size(); sl++) {
      if (C.node_known[slot2node[sl]]) known_rows++;
      else stored[sl] = junk;
    }
    std::vector<uint32_t> cone_store = wl_cone_store_slots(C, store_slot);
    WlProgram Q = wl_schedule(C.graph, cone_store, trash, rows != 0);
    if (!Q.ok) throw std::runtime_error("the cone does not fit the lanes form");
    const uint32_t err = emulate(Q, inputs_le, stored);
    witness_out(g, store_slot, stored, witness_out_le);
    stats[0] = Q.nsteps; stats[1] = Q.nrow; stats[2] = Q.nfma; stats[3] = Q.nsqr; stats[4] = Q.nadd; stats[5] = Q.nmisc;
    stats[6] = Q.peak_slots; stats[7] = err;
    uint32_t unk = 0, rec = 0;
    for (uint32_t c = 0; c < C.node_of.size(); c++) { unk += !C.node_known[C.node_of[c]]; rec += C.recomputed[c]; }
    stats[8] = (uint32_t)C.node_of.size(); stats[9] = unk; stats[10] = rec; stats[11] = known_rows;
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}
}
