// See witness_sched.h.  Step kinds:
//   FMA   every lane computes a * b + c in the 9 x 29 form: a Mul is a * b + ZERO, an Add that shares the step with a
//         product is x * ONE + y, a Sub a + b * MINUS_ONE -- no divergence inside a step;   SQR: a * a + c;
//   ROW   the same products with ONE product per 16-lane row (at most WL_ROWS per step);
//   ADD   every lane computes a + b (only when no product is ready);
//   MISC  inputs (canonical -> Montgomery) and the rare operations (comparisons, shifts, bit operations, division,
//         TernCond ...: graph.rs:72-143, 314-466).
// Bounds: values are kept below WL_BMAX r (fq29.h: products take operands up to 10 r, K8 - x up to 7.9 r).  A product
// reduces its multiplicands (result < PB0 r + a b / 2^261 + c; PB0 = 1 in lane form, 1.06 in row form where m's limbs
// below the top one are only near-normalised), so an Add riding in a product step takes the operand with the larger bound as
// multiplicand; when a result would still pass WL_BMAX it is followed by a reduction x * ONE + ZERO.
#include "witness_sched.h"
#include "witness_ops.h"

#include <stdio.h>
#include <stdlib.h>

#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <stdexcept>

namespace rlnamd {

namespace {
struct MicroOp {
  uint32_t lop, gop, node;     // node: the graph node this micro-op defines (NONE for a raw value awaiting its reduction)
  uint32_t src[3];             // value ids (NONE: unused; >= FIX: a fixed LDS slot)
  uint32_t dst;                // value id
  uint32_t imm;                // WO_INPUT: index into the inputs buffer
};
constexpr uint32_t NONE = 0xFFFFFFFFu;
// a * b + c as ONE node: emitted by hoist_linear_forms for its chains (the graph format has no such operation), passed
// through by reassociate_sums, scheduled as a product step with its addend
constexpr uint32_t G_FMA = 25;
int env_int_wl(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
}  // namespace

// Sums re-associated by arrival time.  The circuit compiler leaves a sum of k terms as a left-deep chain of binary Adds
// in source order (Poseidon's mix: ((M0 x0 + M1 x1) + M2 x2) + c), so a term that arrives last -- the S-box output --
// is followed by the chain's remaining Adds, one step each.  The chain's intermediate Adds that nobody else reads (one
// user, not a witness signal, not stored) are free to be re-associated: the terms are combined earliest-first, the
// late product last, where it fuses with the finished partial sum into ONE a * b + c step.  Field addition is
// associative and the intermediate sums are not outputs, so every stored value is unchanged.  Arrival times are
// estimated on the unbounded-width schedule (a step per product / fused product, an Add a step of its own unless it
// fuses).  Measured on the tree_height = 20 graph: 7 276 -> 6 099 steps on the critical path.
static void reassociate_sums(const Graph& in, const std::vector<uint32_t>& store_in, Graph* out,
                             std::vector<uint32_t>* store_out) {
  const std::vector<GNode>& G = in.nodes;
  const uint32_t N = (uint32_t)G.size();
  auto nops = [&](const GNode& g) {
    return (g.op == G_INPUT || g.op == G_CONST) ? 0 : (g.op == G_NEG || g.op == G_ID) ? 1 : (g.op == G_TERN || g.op == G_FMA) ? 3 : 2;
  };
  std::vector<uint32_t> uses(N, 0);
  std::vector<uint8_t> is_signal(N, 0);
  for (uint32_t sg : in.signals) is_signal[sg] = 1;
  for (uint32_t n = 0; n < N; n++) {
    const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
    for (int k = 0; k < nops(G[n]); k++) {
      if (o[k] >= n) throw std::runtime_error("Graph error: node operand refers forward");
      uses[o[k]]++;
    }
  }
  auto private_node = [&](uint32_t n) { return uses[n] == 1 && !is_signal[n] && store_in[n] == NONE; };
  std::vector<uint8_t> absorbed(N, 0);
  for (uint32_t n = 0; n < N; n++)
    if (G[n].op == G_ADD) {
      if (G[G[n].a].op == G_ADD && private_node(G[n].a)) absorbed[G[n].a] = 1;
      if (G[n].b != G[n].a && G[G[n].b].op == G_ADD && private_node(G[n].b)) absorbed[G[n].b] = 1;
    }
  *out = Graph();
  out->constants = in.constants;
  out->input_mapping = in.input_mapping;
  out->tree_depth = in.tree_depth;
  out->max_out = in.max_out;
  out->inputs_size = in.inputs_size;
  std::vector<GNode>& H = out->nodes;
  std::vector<uint32_t> remap(N, NONE), T;   // T: estimated step at which a node of the new graph is there
  std::vector<uint32_t>& st = *store_out;
  st.clear();
  auto emit = [&](const GNode& g, uint32_t t, uint32_t store) {
    H.push_back(g);
    T.push_back(t);
    st.push_back(store);
    return (uint32_t)H.size() - 1;
  };
  std::vector<uint32_t> terms, stack;
  struct Item {
    uint32_t t, node, tm;   // tm: NONE, or the step its operands are there for a product that may still fuse
  };
  for (uint32_t n = 0; n < N; n++) {
    if (absorbed[n]) continue;
    GNode g = G[n];
    const int no = nops(g);
    if (g.op == G_ADD && ((G[g.a].op == G_ADD && absorbed[g.a]) || (G[g.b].op == G_ADD && absorbed[g.b]))) {
      terms.clear();
      stack.assign({g.b, g.a});
      while (!stack.empty()) {
        const uint32_t o = stack.back();
        stack.pop_back();
        if (absorbed[o]) {
          stack.push_back(G[o].b);
          stack.push_back(G[o].a);
        } else {
          terms.push_back(remap[o]);
        }
      }
      std::vector<Item> items;
      for (size_t i = 0; i < terms.size(); i++) {
        const uint32_t v = terms[i];
        uint32_t tm = NONE;
        if (H[v].op == G_MUL) tm = std::max(T[H[v].a], T[H[v].b]);   // may fuse (if private: checked below)
        items.push_back({T[v], v, tm});
      }
      // products that other nodes read as well cannot fuse: checked on the old graph
      {
        size_t i = 0;
        stack.assign({g.b, g.a});
        while (!stack.empty()) {
          const uint32_t o = stack.back();
          stack.pop_back();
          if (absorbed[o]) {
            stack.push_back(G[o].b);
            stack.push_back(G[o].a);
          } else {
            if (!(G[o].op == G_MUL && private_node(o))) items[i].tm = NONE;
            i++;
          }
        }
      }
      auto later = [](const Item& x, const Item& y) { return x.t != y.t ? x.t > y.t : x.node > y.node; };
      std::make_heap(items.begin(), items.end(), later);
      while (items.size() > 1) {
        std::pop_heap(items.begin(), items.end(), later);
        const Item x = items.back();
        items.pop_back();
        std::pop_heap(items.begin(), items.end(), later);
        const Item y = items.back();
        items.pop_back();
        const bool fuse = y.tm != NONE && x.t <= y.tm;
        const uint32_t t = fuse ? y.tm + 1 : std::max(x.t, y.t) + 1;
        const bool last = items.empty();
        const uint32_t z = emit(GNode{G_ADD, x.node, y.node, 0}, t, last ? store_in[n] : NONE);
        items.push_back({t, z, NONE});
        std::push_heap(items.begin(), items.end(), later);
      }
      remap[n] = items[0].node;
      continue;
    }
    uint32_t* o[3] = {&g.a, &g.b, &g.c};
    uint32_t t = 0;
    for (int k = 0; k < no; k++) {
      *o[k] = remap[*o[k]];
      t = std::max(t, T[*o[k]]);
    }
    if (no) t++;
    if (g.op == G_ADD) {   // a plain Add fuses with a private product whose other operand is there in time
      for (int k = 0; k < 2; k++) {
        const uint32_t m = k ? G[n].b : G[n].a, c = k ? g.a : g.b;
        if (G[n].a != G[n].b && G[m].op == G_MUL && private_node(m)) {
          const uint32_t tm = std::max(T[H[remap[m]].a], T[H[remap[m]].b]);
          if (T[c] <= tm) t = std::min(t, tm + 1);
        }
      }
    }
    remap[n] = emit(g, t, store_in[n]);
  }
  out->signals.reserve(in.signals.size());
  for (uint32_t sg : in.signals) out->signals.push_back(remap[sg]);
}

// Linear forms on the critical path.  Poseidon's partial round, as the circuit compiler leaves it, is FOUR dependent
// products: x^2, x^4, y = x^4 x + c, then x' = K00 y + K01 s1' + K02 s2' with s_i' = s_i + K_i y.  Everything behind y is
// linear, and a linear form can be re-expressed over EARLIER values:
//     x' = (K00 + K01 K1 + K02 K2) y + (K01 K1' + K02 K2') y_prev + K01 s1_prev + K02 s2_prev
// (two rounds of the s_i recurrence unrolled, like terms merged into one folded constant each), and the one late term
// takes its constant on the early factor: kappa (x^4 x + c) = x^4 (kappa x) + kappa c -- kappa x is there one step after x,
// beside x^2.  Three dependent products per round for two more micro-ops per round (the s_i updates keep their
// definition; x' trades three products and two additions for five fused a * b + c).  Generic over the graph: a
// materialised linear node (an Add / a product with one constant factor that is read more than once, stored, or a witness
// signal) near the critical path is expanded through the linear nodes below it -- through private ones freely, through
// at most two shared ones -- into sum coef_k atom_k + const over nonlinear atoms; the form is taken when the arrival
// estimate of its fused chain beats the node's own.  Every value the graph outputs keeps its definition or an
// algebraically equal one (field arithmetic is exact: bit-identical witness), what loses its last reader is dropped.
// Shipped depth-20 circuit: multiplication depth 5 736 -> ~4 300; 6 122 -> 4 813 row-form steps (the row form has four
// product slots per step: 18 600 micro-ops are 4 650 steps at the least).  RLNAMD_WL_HOIST=0 keeps the graph as it came.
static void hoist_linear_forms(const Graph& in, const std::vector<uint32_t>& store_in, Graph* out,
                               std::vector<uint32_t>* store_out) {
  const std::vector<GNode>& G = in.nodes;
  const uint32_t N = (uint32_t)G.size();
  auto nops = [&](const GNode& g) {
    return (g.op == G_INPUT || g.op == G_CONST) ? 0 : (g.op == G_NEG || g.op == G_ID) ? 1 : (g.op == G_TERN || g.op == G_FMA) ? 3 : 2;
  };
  auto cost = [&](const GNode& g) { return (g.op == G_INPUT || g.op == G_CONST || g.op == G_ADD) ? 0u : 1u; };
  auto is_const = [&](uint32_t n) { return G[n].op == G_CONST; };
  // criticality on the multiplication-depth estimate (additions free): only nodes within `near` of the longest path
  std::vector<uint32_t> D(N, 0), TL(N, 0), uses(N, 0);
  uint32_t CP = 0;
  for (uint32_t n = 0; n < N; n++) {
    const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
    uint32_t d = 0;
    for (int k = 0; k < nops(G[n]); k++) {
      if (o[k] >= n) throw std::runtime_error("Graph error: node operand refers forward");
      d = std::max(d, D[o[k]]);
      uses[o[k]]++;
    }
    D[n] = d + cost(G[n]);
  }
  for (uint32_t n = N; n-- > 0;) {
    const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
    for (int k = 0; k < nops(G[n]); k++) TL[o[k]] = std::max(TL[o[k]], TL[n] + cost(G[n]));
    CP = std::max(CP, D[n] + TL[n]);
  }
  const uint32_t near = (uint32_t)env_int_wl("RLNAMD_WL_NEAR", 0);
  const size_t max_terms = (size_t)env_int_wl("RLNAMD_WL_TERMS", 6);
  const int max_cross = env_int_wl("RLNAMD_WL_CROSS", 2);
  // a chain's partial sum grows by ~1.06 r per fused product (row form): reduced (x * ONE) after this many pieces, so that
  // what the last products add to stays inside the interpreter's value range (WL_BMAX) without a reduction BEHIND them
  const size_t reduce_every = (size_t)env_int_wl("RLNAMD_WL_REDUCE_EVERY", 5);
  // (a full round's outputs have three late-factor pieces each: one per form was tried -- 4 881 steps against 4 813)
  const size_t max_late = (size_t)env_int_wl("RLNAMD_WL_MAX_LATE", 3);
  std::vector<uint8_t> is_signal(N, 0);
  for (uint32_t sg : in.signals) is_signal[sg] = 1;
  auto private_node = [&](uint32_t n) { return uses[n] == 1 && !is_signal[n] && store_in[n] == NONE; };
  auto linear = [&](uint32_t n) {
    return G[n].op == G_ADD || (G[n].op == G_MUL && is_const(G[n].a) != is_const(G[n].b));
  };
  Graph H;
  H.constants = in.constants;
  std::vector<uint32_t> remap(N, NONE), st, T;   // T: estimated step at which a node of the new graph is there
  auto emit = [&](const GNode& g, uint32_t t, uint32_t store) {
    H.nodes.push_back(g);
    st.push_back(store);
    T.push_back(t);
    return (uint32_t)H.nodes.size() - 1;
  };
  auto To = [&](uint32_t old) { return T[remap[old]]; };
  std::map<std::array<uint32_t, 8>, uint32_t> const_node;   // value -> G_CONST node of the new graph
  for (uint32_t n = 0; n < N; n++)
    if (is_const(n)) {
      std::array<uint32_t, 8> key;
      memcpy(key.data(), in.constants[G[n].a].v, 32);
      const_node.emplace(key, NONE);   // filled when the node is emitted below
    }
  auto const_of = [&](const Fr& v) {
    std::array<uint32_t, 8> key;
    memcpy(key.data(), v.v, 32);
    auto it = const_node.find(key);
    if (it != const_node.end() && it->second != NONE) return it->second;
    H.constants.push_back(v);
    const uint32_t nd = emit(GNode{G_CONST, (uint32_t)H.constants.size() - 1, 0, 0}, 0, NONE);
    const_node[key] = nd;
    return nd;
  };
  struct Term {
    uint32_t atom;   // node of the OLD graph
    Fr coef;
  };
  struct Form {
    std::vector<Term> t;
    Fr c = Fr::zero();
    bool ok = true;
  };
  // expand `n` (times coef) into f; `cross`: shared linear nodes that may still be looked through
  struct Expander {
    const std::vector<GNode>& G;
    const Graph& in;
    decltype(linear)& lin;
    decltype(private_node)& priv;
    size_t max_terms;
    void run(uint32_t n, const Fr& coef, int cross, bool top, Form& f, int depth = 0) {
      if (!f.ok) return;
      if (depth > 48) {   // (a chain of thousands of private Adds over one atom merges into one term: bound the recursion)
        f.ok = false;
        return;
      }
      const GNode& g = G[n];
      if (g.op == G_CONST) {
        f.c = f.c + coef * in.constants[g.a];
        return;
      }
      if (lin(n) && (top || priv(n) || cross > 0)) {
        const int cx = (top || priv(n)) ? cross : cross - 1;
        if (g.op == G_ADD) {
          run(g.a, coef, cx, false, f, depth + 1);
          run(g.b, coef, cx, false, f, depth + 1);
        } else {
          const bool ka = G[g.a].op == G_CONST;
          run(ka ? g.b : g.a, coef * in.constants[G[ka ? g.a : g.b].a], cx, false, f, depth + 1);
        }
        return;
      }
      for (Term& t : f.t)
        if (t.atom == n) {
          t.coef = t.coef + coef;
          return;
        }
      f.t.push_back(Term{n, coef});
      if (f.t.size() > max_terms) f.ok = false;
    }
  };
  Expander ex{G, in, linear, private_node, max_terms};
  // a term of a form, ready to be emitted
  struct Piece {
    uint32_t ready, tm;   // tm: NONE for a plain addend, else the step its product's operands are there
    uint32_t atom;
    Fr coef;
    uint32_t late, early;   // late-factor form (NONE: plain K * atom)
  };
  size_t rewritten = 0, by_cross[8] = {0};
  for (uint32_t n = 0; n < N; n++) {
    const GNode& g = G[n];
    // ---- the node as it stands
    GNode h = g;
    uint32_t* o[3] = {&h.a, &h.b, &h.c};
    uint32_t t_def = 0;
    for (int k = 0; k < nops(g); k++) {
      *o[k] = remap[*o[k]];
      t_def = std::max(t_def, T[*o[k]]);
    }
    if (nops(g)) t_def++;
    if (g.op == G_ADD && g.a != g.b)   // a plain Add fuses with a private product whose other operand is there in time
      for (int k = 0; k < 2; k++) {
        const uint32_t m = k ? g.b : g.a, c = k ? g.a : g.b;
        if (G[m].op == G_MUL && private_node(m)) {
          const uint32_t tm = std::max(To(G[m].a), To(G[m].b));
          if (To(c) <= tm) t_def = std::min(t_def, tm + 1);
        }
      }
    if (is_const(n)) {
      std::array<uint32_t, 8> key;
      memcpy(key.data(), in.constants[g.a].v, 32);
      uint32_t& slot = const_node[key];
      if (slot == NONE) slot = emit(h, 0, store_in[n]);
      remap[n] = slot;
      if (store_in[n] != NONE && st[slot] == NONE) st[slot] = store_in[n];
      continue;
    }
    bool done = false;
    if (linear(n) && !private_node(n) && CP - (D[n] + TL[n]) <= near) {
      // ---- candidate forms: looked through 0, 1, .. max_cross shared linear nodes
      uint32_t best_t = t_def;
      std::vector<Piece> best;
      Fr best_c = Fr::zero();
      int best_cross = -1;
      for (int cross = 0; cross <= max_cross; cross++) {
        Form f;
        ex.run(n, Fr::one(), cross, true, f);
        if (!f.ok || f.t.empty()) continue;
        // variant 0: the late-factor form only for the piece that arrives last; variant 1: wherever it is earlier
        for (int variant = 0; variant < 2; variant++) {
          std::vector<Piece> ps;
          Fr c = f.c;
          uint32_t last_ready = 0;
          if (variant == 0)
            for (const Term& tr : f.t)
              if (!tr.coef.is_zero()) last_ready = std::max(last_ready, To(tr.atom) + (tr.coef == Fr::one() ? 0u : 1u));
          for (const Term& tr : f.t) {
            if (tr.coef.is_zero()) continue;
            Piece p{0, NONE, tr.atom, tr.coef, NONE, NONE};
            if (tr.coef == Fr::one()) {
              p.ready = To(tr.atom);
            } else {
              // the late-factor form: atom = A * B, or (A * B) + const with the product private, A later than B
              uint32_t P = NONE;
              Fr pc = Fr::zero();
              if (variant == 1 || To(tr.atom) + 1 == last_ready) {
                if (G[tr.atom].op == G_MUL && !is_const(G[tr.atom].a) && !is_const(G[tr.atom].b)) {
                  P = tr.atom;
                } else if (G[tr.atom].op == G_ADD && is_const(G[tr.atom].a) != is_const(G[tr.atom].b)) {
                  const uint32_t cn = is_const(G[tr.atom].a) ? G[tr.atom].a : G[tr.atom].b;
                  const uint32_t pn = cn == G[tr.atom].a ? G[tr.atom].b : G[tr.atom].a;
                  if (G[pn].op == G_MUL && !is_const(G[pn].a) && !is_const(G[pn].b) && private_node(pn)) {
                    P = pn;
                    pc = in.constants[G[cn].a];
                  }
                }
              }
              if (P != NONE && To(G[P].a) != To(G[P].b)) {
                p.late = To(G[P].a) > To(G[P].b) ? G[P].a : G[P].b;
                p.early = p.late == G[P].a ? G[P].b : G[P].a;
                p.tm = std::max(To(p.late), To(p.early) + 1);
                if (p.tm + 1 < To(tr.atom) + 2) {   // better than K * atom
                  c = c + tr.coef * pc;
                } else {
                  p.late = p.early = NONE;
                }
              }
              if (p.late == NONE) p.tm = To(tr.atom);
              p.ready = p.tm + 1;
            }
            ps.push_back(p);
          }
          if (ps.empty()) continue;
          std::stable_sort(ps.begin(), ps.end(), [](const Piece& x, const Piece& y) { return x.ready < y.ready; });
          // the fused chain: constant first, then the pieces as they arrive
          uint32_t acc = 0;
          bool have = !c.is_zero();
          for (size_t k = 0; k < ps.size(); k++) {
            const Piece& p = ps[k];
            if (have && reduce_every && k && k % reduce_every == 0 && k + 1 < ps.size()) acc++;
            if (!have) {
              acc = p.ready;
              have = true;
            } else if (p.tm != NONE) {
              acc = std::max(acc, p.tm) + 1;
            } else {
              acc = std::max(acc, p.ready) + 1;
            }
          }
          size_t n_late = 0;
          for (const Piece& p : ps) n_late += p.late != NONE;
          if (n_late > max_late) continue;
          if (acc < best_t) {
            best_t = acc;
            best = ps;
            best_c = c;
            best_cross = cross * 2 + variant;
          }
        }
      }
      if (!best.empty()) {
        // the chain as explicit a * b + c nodes, in arrival order
        uint32_t acc = NONE, acc_t = 0;
        if (!best_c.is_zero()) acc = const_of(best_c);
        const uint32_t first_new = (uint32_t)H.nodes.size();   // (constants are shared: the chain's own nodes start here)
        for (size_t k = 0; k < best.size(); k++) {
          const Piece& p = best[k];
          if (acc != NONE && reduce_every && k && k % reduce_every == 0 && k + 1 < best.size()) {
            acc_t++;
            acc = emit(GNode{G_MUL, acc, const_of(Fr::one()), 0}, acc_t, NONE);
          }
          if (p.tm == NONE) {
            const uint32_t node = remap[p.atom];
            if (acc == NONE) {
              acc = node;
              acc_t = p.ready;
            } else {
              acc_t = std::max(acc_t, p.ready) + 1;
              acc = emit(GNode{G_ADD, acc, node, 0}, acc_t, NONE);
            }
            continue;
          }
          uint32_t fa, fb;
          if (p.late != NONE) {
            fa = remap[p.late];
            fb = emit(GNode{G_MUL, const_of(p.coef), remap[p.early], 0}, To(p.early) + 1, NONE);
          } else {
            fa = const_of(p.coef);
            fb = remap[p.atom];
          }
          if (acc == NONE) {
            acc_t = p.ready;
            acc = emit(GNode{G_MUL, fa, fb, 0}, acc_t, NONE);
          } else {
            acc_t = std::max(acc_t, p.tm) + 1;
            acc = emit(GNode{G_FMA, fa, fb, acc}, acc_t, NONE);
          }
        }
        if (acc < first_new)   // a single plain term: a node that exists already (it has its own store slot) -- a copy carries this one's
          acc = emit(GNode{G_ADD, acc, const_of(Fr::zero()), 0}, acc_t + 1, NONE);
        st[acc] = store_in[n];
        remap[n] = acc;
        done = true;
        rewritten++;
        if (best_cross >= 0 && best_cross < 8) by_cross[best_cross]++;
      }
    }
    if (!done) remap[n] = emit(h, t_def, store_in[n]);
  }
  if (getenv("RLNAMD_WL_DEBUG"))
    fprintf(stderr, "wl debug: %zu linear forms rewritten (shared nodes looked through x 2 + variant: %zu %zu %zu %zu %zu %zu)\n", rewritten,
            by_cross[0], by_cross[1], by_cross[2], by_cross[3], by_cross[4], by_cross[5]);
  // drop what lost its last reader, keep signals, stored values and inputs
  const uint32_t M = (uint32_t)H.nodes.size();
  std::vector<uint8_t> keep(M, 0);
  for (uint32_t sg : in.signals) keep[remap[sg]] = 1;
  for (uint32_t m = 0; m < M; m++)
    if (st[m] != NONE || H.nodes[m].op == G_INPUT) keep[m] = 1;
  for (uint32_t m = M; m-- > 0;) {
    if (!keep[m]) continue;
    const uint32_t o[3] = {H.nodes[m].a, H.nodes[m].b, H.nodes[m].c};
    for (int k = 0; k < nops(H.nodes[m]); k++) keep[o[k]] = 1;
  }
  *out = Graph();
  out->constants = H.constants;
  out->input_mapping = in.input_mapping;
  out->tree_depth = in.tree_depth;
  out->max_out = in.max_out;
  out->inputs_size = in.inputs_size;
  // (the surviving constants first, then the rest in emission order: every node still follows what it reads)
  std::vector<uint32_t> pos(M, NONE);
  store_out->clear();
  for (int pass = 0; pass < 2; pass++)
    for (uint32_t m = 0; m < M; m++) {
      if (!keep[m] || (H.nodes[m].op == G_CONST) != (pass == 0)) continue;
      GNode h = H.nodes[m];
      uint32_t* o[3] = {&h.a, &h.b, &h.c};
      for (int k = 0; k < nops(H.nodes[m]); k++) *o[k] = pos[*o[k]];
      pos[m] = (uint32_t)out->nodes.size();
      out->nodes.push_back(h);
      store_out->push_back(st[m]);
    }
  out->signals.reserve(in.signals.size());
  for (uint32_t sg : in.signals) out->signals.push_back(pos[remap[sg]]);
}

static WlProgram wl_schedule_graph(const Graph& graph, const std::vector<uint32_t>& store_slot, uint32_t trash_slot,
                                   bool rows);

WlProgram wl_schedule(const Graph& graph, const std::vector<uint32_t>& store_slot, uint32_t trash_slot, bool rows) {
  if (!env_int_wl("RLNAMD_WL_REASSOC", 1)) {
    WlProgram P = wl_schedule_graph(graph, store_slot, trash_slot, rows);
    P.consts = graph.constants;
    return P;
  }
  Graph g1, g2;
  std::vector<uint32_t> st1, st2;
  const Graph* src = &graph;
  const std::vector<uint32_t>* st = &store_slot;
  if (env_int_wl("RLNAMD_WL_HOIST", 1)) {
    hoist_linear_forms(graph, store_slot, &g1, &st1);
    src = &g1;
    st = &st1;
  }
  reassociate_sums(*src, *st, &g2, &st2);
  WlProgram P = wl_schedule_graph(g2, st2, trash_slot, rows);
  P.consts = g2.constants;
  return P;
}

static WlProgram wl_schedule_graph(const Graph& graph, const std::vector<uint32_t>& store_slot, uint32_t trash_slot,
                                   bool rows) {
  WlProgram R;
  uint32_t &nsteps = R.nsteps, &nrow = R.nrow, &nfma = R.nfma, &nsqr = R.nsqr, &nadd = R.nadd, &nmisc = R.nmisc;
  uint32_t &peak_slots = R.peak_slots, &n_consts = R.n_consts;
  const std::vector<GNode>& G = graph.nodes;
  const uint32_t N = (uint32_t)G.size();
  n_consts = (uint32_t)graph.constants.size();
  const uint32_t Z = n_consts, ONE = n_consts + 1, MONE = n_consts + 2, first_free = n_consts + 3;
  const uint32_t DUMMY = WL_SLOTS - 1;
  const uint32_t FIX = (1u << 30) + N;    // value ids >= FIX address a fixed LDS slot (ZERO / ONE / MINUS_ONE)
  if (first_free + 64 >= DUMMY) return R;   // the constants alone (nearly) fill the LDS
  // a constant that is itself a witness signal has a store slot and no micro-op that would fill it: such a graph keeps
  // the other interpreter (none of the shipped circuits has one: their constant 1 is input 0)
  for (uint32_t n = 0; n < N; n++)
    if (graph.nodes[n].op == G_CONST && store_slot[n] != NONE) return R;
  auto nops = [&](const GNode& g) {
    return (g.op == G_INPUT || g.op == G_CONST) ? 0 : (g.op == G_NEG || g.op == G_ID) ? 1 : (g.op == G_TERN || g.op == G_FMA) ? 3 : 2;
  };
  std::vector<uint8_t> is_signal(N, 0);
  for (uint32_t sg : graph.signals) is_signal[sg] = 1;
  std::vector<std::vector<uint32_t>> users(N);
  std::vector<uint32_t> uses(N, 0);
  for (uint32_t n = 0; n < N; n++) {
    const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
    for (int k = 0; k < nops(G[n]); k++) {
      if (o[k] >= n) throw std::runtime_error("Graph error: node operand refers forward");
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
    return !(op == G_MUL || op == G_ADD || op == G_SUB || op == G_NEG || op == G_FMA);
  };
  struct Pending { uint32_t node, raw; };   // a raw value that still needs x * ONE + ZERO to become `node`
  std::vector<Pending> pending;
  uint32_t next_tmp = N;
  std::vector<double> tmp_bound;
  auto bnd = [&](uint32_t v) { return v < N ? bound[v] : tmp_bound[v - N]; };
  nfma = nadd = nmisc = nsqr = nrow = 0;
  // row form (one product per 16-lane row, wl_row_mul_add): at most WL_ROWS products per step, result < 2.05 r + ...
  const double PB0 = rows ? 1.06 : 1.0;
  const size_t fma_cap = rows ? WL_ROWS : WL_W;
  // height = longest chain of nodes from a node to a sink: when a step cannot take every ready node, the ones the
  // longest chains hang on go first
  std::vector<uint32_t> height(N, 0);
  for (uint32_t n = N; n-- > 0;) {
    const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
    for (int k = 0; k < nops(G[n]); k++) height[o[k]] = std::max(height[o[k]], height[n] + 1);
  }
  while (!ready.empty() || !pending.empty()) {
    std::sort(ready.begin(), ready.end(), [&](uint32_t x, uint32_t y) {
      return height[x] != height[y] ? height[x] > height[y] : x < y;
    });
    std::vector<MicroOp> ops;
    std::vector<uint32_t> produced, left;
    uint32_t kind;
    bool any_rare = false, any_mul = false;
    for (uint32_t n : ready) {
      any_rare |= is_rare(n);
      any_mul |= G[n].op == G_MUL || G[n].op == G_SUB || G[n].op == G_NEG || G[n].op == G_FMA;
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
      const size_t cap = kind == WK_FMA ? fma_cap : (size_t)WL_W;
      if (kind == WK_FMA) {
        std::vector<Pending> later;
        for (const Pending& pd : pending) {   // reductions first: their consumers are waiting
          if (ops.size() >= cap) { later.push_back(pd); continue; }
          ops.push_back(MicroOp{WO_COMPUTE, G_MUL, pd.node, {pd.raw, FIX + ONE, FIX + Z}, pd.node, 0});
          produced.push_back(pd.node);
          bound[pd.node] = PB0 + 0.006 * bnd(pd.raw);
        }
        pending.swap(later);
      }
      for (uint32_t n : ready) {
        if (ops.size() >= cap) { left.push_back(n); continue; }
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
              if (avail[other] && bound[other] + PB0 + 0.006 * bound[g.a] * bound[g.b] <= WL_BMAX) { add = u; c = other; }
            }
          }
          if (add != NONE) {
            m.node = m.dst = add;
            m.src[0] = g.a; m.src[1] = g.b; m.src[2] = c;
            b = PB0 + 0.006 * bound[g.a] * bound[g.b] + bound[c];
            done[n] = 1;   // never materialised
            bound[add] = b;
            ops.push_back(m);
            produced.push_back(add);
            continue;
          }
          m.src[0] = g.a; m.src[1] = g.b; m.src[2] = FIX + Z;
          b = PB0 + 0.006 * bound[g.a] * bound[g.b];
        } else if (g.op == G_ADD) {   // x * 1 + y, the larger bound as multiplicand
          const uint32_t x = bound[g.a] >= bound[g.b] ? g.a : g.b, y = x == g.a ? g.b : g.a;
          m.src[0] = x; m.src[1] = FIX + ONE; m.src[2] = y;
          b = PB0 + 0.006 * bound[x] + bound[y];
        } else if (g.op == G_FMA) {   // a * b + c as the graph pass left it
          m.gop = G_MUL;
          m.src[0] = g.a; m.src[1] = g.b; m.src[2] = g.c;
          b = PB0 + 0.006 * bound[g.a] * bound[g.b] + bound[g.c];
        } else if (g.op == G_SUB) {   // a - b = b * (-1) + a
          m.src[0] = g.b; m.src[1] = FIX + MONE; m.src[2] = g.a;
          b = PB0 + 0.006 * bound[g.b] * 1.05 + bound[g.a];
        } else {                      // G_NEG
          m.src[0] = g.a; m.src[1] = FIX + MONE; m.src[2] = FIX + Z;
          b = PB0 + 0.006 * bound[g.a] * 1.05;
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
    if (kind == WK_FMA && rows) {
      kind = WK_ROW;
      nrow++;
    } else if (kind == WK_FMA) {   // all products squarings (idle lanes compute ZERO * ZERO + ZERO: a square as well)?
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
    if (!done[n]) throw std::runtime_error("witness lanes: graph node left unscheduled");
  if (getenv("RLNAMD_WL_DEBUG")) {
    size_t nops_total = 0, nadd1 = 0, nred = 0, nprod = 0, nfused = 0;
    for (const auto& st_ : steps)
      for (const MicroOp& m : st_) {
        nops_total++;
        if (m.lop != WO_COMPUTE) continue;
        if (m.src[1] == FIX + ONE && m.src[2] == FIX + Z) nred++;
        else if (m.src[1] == FIX + ONE || m.src[1] == FIX + MONE) nadd1++;
        else if (m.src[2] != FIX + Z && m.src[2] != NONE) nfused++;
        else nprod++;
      }
    fprintf(stderr, "wl debug: steps %zu micro-ops %zu: products %zu fused a*b+c %zu adds-as-products %zu reductions %zu; constants %u\n",
            steps.size(), nops_total, nprod, nfused, nadd1, nred, n_consts);
  }
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
  std::vector<WlDesc>& img = R.img;
  // (row form: the padding steps are row steps too, so that the last group can be a group of row steps)
  img.assign((steps.size() + 2 * WL_PF) * (size_t)WL_W,
             WlDesc{rows ? (WK_ROW << 12) : 0u, DUMMY | (Z << 16), Z | (Z << 16), trash_slot});
  auto slot_of_val = [&](uint32_t v) -> uint32_t {
    if (v == NONE) return Z;
    if (v >= FIXB) return v - FIXB;
    if (slot[v] == NONE) throw std::runtime_error("witness lanes: operand read before it was produced");
    return slot[v];
  };
  for (uint32_t t = 0; t < steps.size(); t++) {
    // operands first (their slots may be those of values that die here), then the results
    std::vector<uint32_t> sa(steps[t].size() * 3);
    for (size_t i = 0; i < steps[t].size(); i++)
      for (int k = 0; k < 3; k++) sa[3 * i + k] = slot_of_val(steps[t][i].src[k]);
    for (size_t i = 0; i < steps[t].size(); i++) {
      const MicroOp& m = steps[t][i];
      if (free_slots.empty()) return R;   // more live values than LDS slots: keep k_witness29
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
      if (fa >= 65536) return R;          // an input index that does not fit the descriptor: keep k_witness29
      const WlDesc desc{x, sl | (fa << 16), sa[3 * i + 1] | (sa[3 * i + 2] << 16), w};
      if (step_kind[t] == WK_ROW) {     // the row's sixteen lanes all read the row's descriptor
        for (uint32_t l = 0; l < 16; l++) img[(size_t)t * WL_W + 16 * i + l] = desc;
      } else {
        img[(size_t)t * WL_W + i] = desc;
      }
    }
    // every descriptor of the step carries the kind (lane 0's is the one the kernel reads)
    {
      const uint32_t used = step_kind[t] == WK_ROW ? 16 * (uint32_t)steps[t].size() : (uint32_t)steps[t].size();
      for (uint32_t i = used; i < WL_W; i++) img[(size_t)t * WL_W + i].x = step_kind[t] << 12;
    }
    for (uint32_t v : dies[t]) {
      free_slots.push_back(slot[v]);
      live--;
    }
  }
  nsteps = (uint32_t)steps.size();
  // groups of WL_PF steps that hold row steps only: marked on their first step (every lane's copy)
  for (size_t t0 = 0; t0 < steps.size(); t0 += WL_PF) {
    bool all = true;
    for (size_t t = t0; t < t0 + WL_PF; t++) all = all && (t >= steps.size() ? rows : step_kind[t] == WK_ROW);
    if (all)
      for (uint32_t l = 0; l < WL_W; l++) img[t0 * WL_W + l].x |= WL_GROUP_ROWS;
  }
  R.ok = true;
  return R;
}


// ------------------------------------------------------------------------------------------------- the unknown cone
WlCone wl_cone(const Graph& g) {
  const std::vector<GNode>& G = g.nodes;
  const uint32_t N = (uint32_t)G.size();
  auto nops = [&](const GNode& q) {
    return (q.op == G_INPUT || q.op == G_CONST) ? 0 : (q.op == G_NEG || q.op == G_ID) ? 1 : q.op == G_TERN ? 3 : 2;
  };
  WlCone C;
  std::vector<uint8_t> in_known(g.inputs_size, 1);
  for (const char* name : {"messageId", "selectorUsed", "x", "externalNullifier"}) {
    auto it = g.input_mapping.find(name);
    if (it == g.input_mapping.end()) continue;
    for (uint32_t k = 0; k < it->second.second; k++)
      if (it->second.first + k < in_known.size()) in_known[it->second.first + k] = 0;
  }
  C.node_known.assign(N, 0);
  for (uint32_t i = 0; i < N; i++) {
    const GNode& q = G[i];
    const uint32_t o[3] = {q.a, q.b, q.c};
    bool k = true;
    if (q.op == G_INPUT) k = q.a < in_known.size() && in_known[q.a];
    else
      for (int j = 0; j < nops(q); j++) {
        if (o[j] >= i) throw std::runtime_error("Graph error: node operand refers forward");
        k = k && C.node_known[o[j]];
      }
    C.node_known[i] = k;
  }
  // members: every unknown node; from there down, every operand (known ones included) until inputs and constants
  std::vector<uint8_t> member(N, 0);
  for (uint32_t i = N; i-- > 0;) {
    if (!C.node_known[i]) member[i] = 1;
    if (!member[i]) continue;
    const uint32_t o[3] = {G[i].a, G[i].b, G[i].c};
    for (int j = 0; j < nops(G[i]); j++) member[o[j]] = 1;
  }
  std::vector<uint32_t> remap(N, NONE);
  for (uint32_t i = 0; i < N; i++) {
    if (!member[i]) continue;
    remap[i] = (uint32_t)C.node_of.size();
    C.node_of.push_back(i);
    GNode q = G[i];
    const int k = nops(q);
    if (k > 0) q.a = remap[q.a];
    if (k > 1) q.b = remap[q.b];
    if (k > 2) q.c = remap[q.c];
    C.graph.nodes.push_back(q);
    C.recomputed.push_back(C.node_known[i] && q.op != G_INPUT && q.op != G_CONST);
  }
  // only the constants the cone reads (the program copies its constants into LDS before its first step: 1 133 for the
  // whole depth-20 graph, a third of a lone finish's interpreter time when the cone carried them all)
  {
    std::vector<uint32_t> cmap(g.constants.size(), NONE);
    for (GNode& q : C.graph.nodes) {
      if (q.op != G_CONST) continue;
      if (q.a >= g.constants.size()) throw std::runtime_error("Graph error: constant index out of range");
      if (cmap[q.a] == NONE) {
        cmap[q.a] = (uint32_t)C.graph.constants.size();
        C.graph.constants.push_back(g.constants[q.a]);
      }
      q.a = cmap[q.a];
    }
  }
  C.graph.input_mapping = g.input_mapping;
  C.graph.tree_depth = g.tree_depth;
  C.graph.max_out = g.max_out;
  C.graph.inputs_size = g.inputs_size;
  for (uint32_t sg : g.signals)
    if (!C.node_known[sg]) C.graph.signals.push_back(remap[sg]);
  return C;
}

std::vector<uint32_t> wl_cone_store_slots(const WlCone& cone, const std::vector<uint32_t>& store_slot_full) {
  std::vector<uint32_t> st(cone.node_of.size(), NONE);
  for (uint32_t c = 0; c < cone.node_of.size(); c++)
    if (!cone.node_known[cone.node_of[c]]) st[c] = store_slot_full[cone.node_of[c]];
  return st;
}


// ------------------------------------------------------------------------------------------------- segments behind hints
std::vector<Fr> wl_eval_host(const Graph& g, const uint8_t* inputs_le, uint32_t* err) {
  const std::vector<GNode>& G = g.nodes;
  std::vector<Fr> v(G.size(), Fr::zero());
  *err = WERR_NONE;
  for (uint32_t n = 0; n < G.size(); n++) {
    const GNode& q = G[n];
    switch (q.op) {
      case G_INPUT: {
        uint32_t c[8];
        memcpy(c, inputs_le + (size_t)q.a * 32, 32);
        if (limbs_geq(c, FrParams::MOD)) *err = WERR_INPUT_RANGE;
        v[n] = Fr::from_canonical(c);
        break;
      }
      case G_CONST: v[n] = g.constants[q.a]; break;
      case G_MUL: v[n] = v[q.a] * v[q.b]; break;
      case G_ADD: v[n] = v[q.a] + v[q.b]; break;
      case G_SUB: v[n] = v[q.a] - v[q.b]; break;
      case G_NEG: v[n] = v[q.a].neg(); break;
      case G_TERN: v[n] = v[q.a].is_zero() ? v[q.c] : v[q.b]; break;
      default: v[n] = witness_slow_op(q.op, v[q.a], q.op == G_ID ? Fr::zero() : v[q.b], err); break;
    }
  }
  return v;
}

WlSegments wl_segments(const Graph& g, const std::vector<std::vector<uint32_t>>& cuts) {
  const std::vector<GNode>& G = g.nodes;
  const uint32_t N = (uint32_t)G.size();
  auto nops = [&](const GNode& q) {
    return (q.op == G_INPUT || q.op == G_CONST) ? 0 : (q.op == G_NEG || q.op == G_ID) ? 1 : q.op == G_TERN ? 3 : 2;
  };
  WlSegments S;
  S.n_hints = (uint32_t)cuts.size();
  std::vector<uint32_t> hint_of(N, NONE);
  for (uint32_t j = 0; j < cuts.size(); j++)
    for (uint32_t n : cuts[j]) {
      if (n >= N || G[n].op == G_INPUT || G[n].op == G_CONST) throw std::runtime_error("wl_segments: a cut must be a computed node");
      hint_of[n] = j;
      S.cut_nodes.push_back(n);
      S.cut_hint.push_back(j);
    }
  // key of a node: the hints it reaches backwards without crossing a cut
  std::vector<std::vector<uint32_t>> key(N);
  std::map<std::vector<uint32_t>, uint32_t> seg_of_key;
  std::vector<uint32_t> seg(N, NONE);
  for (uint32_t n = 0; n < N; n++) {
    const GNode& q = G[n];
    if (q.op == G_INPUT || q.op == G_CONST) continue;
    const uint32_t o[3] = {q.a, q.b, q.c};
    std::vector<uint32_t> k;
    for (int j = 0; j < nops(q); j++) {
      if (o[j] >= n) throw std::runtime_error("Graph error: node operand refers forward");
      if (hint_of[o[j]] != NONE) k.push_back(hint_of[o[j]]);
      else k.insert(k.end(), key[o[j]].begin(), key[o[j]].end());
    }
    std::sort(k.begin(), k.end());
    k.erase(std::unique(k.begin(), k.end()), k.end());
    auto it = seg_of_key.find(k);
    if (it == seg_of_key.end()) it = seg_of_key.emplace(k, (uint32_t)seg_of_key.size()).first;
    seg[n] = it->second;
    key[n] = std::move(k);
  }
  // nodes of one key that share nothing but inputs, constants and cuts are independent too (the identity commitment and
  // a1 both start from the inputs: together they fill the four product rows of a step twice over): connected components,
  // the small ones (a stray addition is not worth a wave and a CU's LDS) kept with the largest of their key
  {
    std::vector<uint32_t> parent(N);
    for (uint32_t n = 0; n < N; n++) parent[n] = n;
    auto find = [&](uint32_t x) {
      while (parent[x] != x) x = parent[x] = parent[parent[x]];
      return x;
    };
    for (uint32_t n = 0; n < N; n++) {
      if (seg[n] == NONE) continue;
      const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
      for (int j = 0; j < nops(G[n]); j++)
        if (seg[o[j]] == seg[n] && hint_of[o[j]] == NONE) parent[find(o[j])] = find(n);
    }
    std::map<uint32_t, uint32_t> comp_size;
    for (uint32_t n = 0; n < N; n++)
      if (seg[n] != NONE) comp_size[find(n)]++;
    std::vector<uint32_t> biggest(seg_of_key.size(), NONE);   // per key: the root of its largest component
    for (auto& cs : comp_size) {
      const uint32_t k = seg[cs.first];
      if (biggest[k] == NONE || comp_size[biggest[k]] < cs.second) biggest[k] = cs.first;
    }
    std::map<uint32_t, uint32_t> seg_of_comp;
    uint32_t next = 0;
    std::vector<uint32_t> seg2(N, NONE);
    for (uint32_t n = 0; n < N; n++) {
      if (seg[n] == NONE) continue;
      uint32_t r = find(n);
      if (comp_size[r] < 256) r = biggest[seg[n]];
      auto it = seg_of_comp.find(r);
      if (it == seg_of_comp.end()) it = seg_of_comp.emplace(r, next++).first;
      seg2[n] = it->second;
    }
    seg.swap(seg2);
    seg_of_key.clear();
    for (uint32_t k = 0; k < next; k++) seg_of_key.emplace(std::vector<uint32_t>{k, 0xFFFFFFFFu}, k);   // (only its size is used below)
  }
  const uint32_t K = (uint32_t)seg_of_key.size();
  if (K == 0) return S;
  // an input no computed node reads is still a stored value (it may be a witness signal): segment 0 takes those
  std::vector<uint8_t> input_read(N, 0);
  for (uint32_t n = 0; n < N; n++) {
    const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
    for (int j = 0; j < nops(G[n]); j++)
      if (G[o[j]].op == G_INPUT) input_read[o[j]] = 1;
  }
  S.graphs.resize(K);
  S.node_of.resize(K);
  S.owned.resize(K);
  // a segment: its own nodes and, downwards from them, everything they read until cuts, inputs and constants
  for (uint32_t k = 0; k < K; k++) {
    std::vector<uint8_t> member(N, 0);
    for (uint32_t n = N; n-- > 0;) {
      if (seg[n] == k || (k == 0 && G[n].op == G_INPUT && !input_read[n])) member[n] = 1;
      if (!member[n]) continue;
      const uint32_t o[3] = {G[n].a, G[n].b, G[n].c};
      for (int j = 0; j < nops(G[n]); j++)
        if (hint_of[o[j]] == NONE) member[o[j]] = 1;      // (a cut is read as a hint, not computed here)
    }
    Graph& H = S.graphs[k];
    std::vector<uint32_t> remap(N, NONE), hint_node(S.n_hints, NONE);
    auto operand = [&](uint32_t o) -> uint32_t {
      if (hint_of[o] == NONE) return remap[o];
      uint32_t& hn = hint_node[hint_of[o]];
      if (hn == NONE) {   // the hint as an input of the segment, the first time it is read
        hn = (uint32_t)H.nodes.size();
        H.nodes.push_back(GNode{G_INPUT, g.inputs_size + hint_of[o], 0, 0});
        S.node_of[k].push_back(NONE);
        S.owned[k].push_back(0);
      }
      return hn;
    };
    for (uint32_t n = 0; n < N; n++) {
      if (!member[n]) continue;
      GNode q = G[n];
      const int c = nops(q);
      if (c > 0) q.a = operand(G[n].a);
      if (c > 1) q.b = operand(G[n].b);
      if (c > 2) q.c = operand(G[n].c);
      remap[n] = (uint32_t)H.nodes.size();
      H.nodes.push_back(q);
      S.node_of[k].push_back(n);
      S.owned[k].push_back(seg[n] == k || G[n].op == G_INPUT);
    }
    {   // only the constants the segment reads
      std::vector<uint32_t> cmap(g.constants.size(), NONE);
      for (GNode& q : H.nodes) {
        if (q.op != G_CONST) continue;
        if (q.a >= g.constants.size()) throw std::runtime_error("Graph error: constant index out of range");
        if (cmap[q.a] == NONE) {
          cmap[q.a] = (uint32_t)H.constants.size();
          H.constants.push_back(g.constants[q.a]);
        }
        q.a = cmap[q.a];
      }
    }
    H.input_mapping = g.input_mapping;
    H.tree_depth = g.tree_depth;
    H.max_out = g.max_out;
    H.inputs_size = g.inputs_size + S.n_hints;
    for (uint32_t sg : g.signals)
      if (seg[sg] == k) H.signals.push_back(remap[sg]);
  }
  return S;
}

std::vector<uint32_t> wl_segment_store_slots(const WlSegments& S, size_t k, const std::vector<uint32_t>& store_slot_full) {
  std::vector<uint32_t> st(S.node_of[k].size(), NONE);
  for (uint32_t c = 0; c < st.size(); c++)
    if (S.owned[k][c] && S.node_of[k][c] != NONE) st[c] = store_slot_full[S.node_of[k][c]];
  return st;
}

}  // namespace rlnamd
