// Sanitizer harness ("ASan/UBSan in CPU tests").  The product's HOST code that reads bytes it does not
// control -- the arkzkey and graph parsers (zkey.cpp), the config_path JSON parser (tree_config.h), proof decompression
// and the pairing verifier (pairing.h) -- plus the interpreter's scheduler (witness_sched.cpp), compiled by g++ with
// -fsanitize=address,undefined and run on the shipped resources, on a golden proof and on a few thousand truncated /
// bit-flipped inputs.  Malformed input must end in rlnamd::Error, never in a sanitizer report.  CPU only: the GPU pool
// refuses sanitizer runs, and no HIP call is reached here.
// Round 6: the WIRE parsers of the zerokit C ABI (ffi_wire.h: what a relay node feeds with bytes off
// the network -- rln proofs, proof values, witnesses, partial witnesses, partial proofs; V1 LE / BE and the V3 forms)
// on golden records and on > 10 000 truncations, bit flips, hostile length prefixes and trailing bytes: every input is
// parsed or refused with one of the reference's error texts (rln/src/error.rs), and what parses re-serialises.
//   usage: sanitize_main <zkey> <graph> <proof128 + public inputs file> [<V1 LE rln proof record> <partial320 file>]
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "ffi_wire.h"
#include "pairing.h"
#include "tree_config.h"
#include "witness_sched.h"
#include "zkey.h"
using namespace rlnamd;

static std::vector<uint8_t> slurp(const char* path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) return v;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static bool verify(const Zkey& zk, const uint8_t* proof, const uint8_t* pub, size_t n) {
  G1Affine A, C;
  G2Affine B;
  if (!g1_decompress(proof, &A) || !g2_decompress(proof + 32, &B) || !g1_decompress(proof + 96, &C) || !g2_in_subgroup(B))
    return false;
  std::vector<Fr> x(n);
  for (size_t i = 0; i < n; i++) {
    uint32_t c[8];
    memcpy(c, pub + 32 * i, 32);
    if (limbs_geq(c, FrParams::MOD)) return false;
    x[i] = Fr::from_canonical(c);
  }
  return groth16_verify(zk, A, B, C, x);
}

// ---------------------------------------------------------------------------------------------- wire parsers (ffi_wire.h)
// the texts a deserialiser or the validation behind it may end in (rln/src/error.rs:19-33,59-93,121-165)
static bool reference_text(const std::string& m) {
  static const char* ok[] = {
      "Input data too short: expected at least ", "Non-canonical field element: value is not in [0, r-1]",
      "Non-canonical bool byte: expected 0x00 or 0x01, got 0x", "Expected to read ", "Unknown message mode version byte: 0x",
      "Proof serialization error: the input buffer contained invalid data", "The field ", "User message limit cannot be zero",
      "Merkle proof length mismatch: expected ", "Message id (", "Duplicate message ID found in message_ids",
      "At least one selector_used value must be true",
      // V3 (SerializationErrorV3 / arkworks SerializationError)
      "I/O error: ", "the input buffer contained invalid data", "Arkworks canonical serialization error: "};
  for (const char* p : ok)
    if (m.compare(0, strlen(p), p) == 0) return true;
  return false;
}
struct WireStats { size_t parsed = 0, refused = 0; int failures = 0; };
template <class F>
static void feed(WireStats& st, const char* who, const std::vector<uint8_t>& q, F&& parse) {
  // the bytes sit in an allocation of exactly their length: one byte read past the end is an ASan report
  uint8_t* heap = (uint8_t*)malloc(q.size() ? q.size() : 1);
  if (!q.empty()) memcpy(heap, q.data(), q.size());
  Vec_uint8_t v{heap, q.size(), q.size()};
  try {
    parse(&v);
    st.parsed++;
  } catch (const Error& e) {
    st.refused++;
    if (!reference_text(e.what())) { fprintf(stderr, "%s: not a reference error text: %s\n", who, e.what()); st.failures++; }
  } catch (const std::exception& e) {
    fprintf(stderr, "%s: foreign exception: %s\n", who, e.what());
    st.failures++;
  }
  free(heap);
}
template <class F>
static void fuzz(WireStats& st, const char* who, const std::vector<uint8_t>& rec, F&& parse, int flips) {
  for (size_t k = 0; k <= rec.size(); k++) feed(st, who, std::vector<uint8_t>(rec.begin(), rec.begin() + k), parse);   // every prefix
  for (int k = 0; k < flips; k++) {                                                                                    // bit flips
    std::vector<uint8_t> q = rec;
    for (int m = 0; m < 1 + (int)(rnd() % 3); m++) q[rnd() % q.size()] ^= (uint8_t)(1u << (rnd() % 8));
    feed(st, who, q, parse);
  }
  static const uint64_t hostile[] = {~0ull, 1ull << 63, 1ull << 61, (1ull << 32) + 1, 0x0100000000000000ull, 1ull << 40, 65537, 0};
  for (size_t off = 0; off + 8 <= rec.size(); off++)                                                                   // a length at every offset
    for (int h = 0; h < 2; h++) {
      std::vector<uint8_t> q = rec;
      uint64_t v = hostile[rnd() % 8];
      if (h) v = rec.size() - off + (rnd() % 3);
      for (int i = 0; i < 8; i++) q[off + i] = (uint8_t)(v >> (8 * ((rnd() & 1) ? i : 7 - i)));
      feed(st, who, q, parse);
    }
  for (int k = 1; k <= 3; k++) {                                                                                       // trailing bytes
    std::vector<uint8_t> q = rec;
    q.insert(q.end(), (size_t)k * 7, (uint8_t)rnd());
    feed(st, who, q, parse);
  }
}
static CFr fr_small(uint64_t v) { return cfr_from_u64(v); }
static int wire_section(const std::vector<uint8_t>& golden_le, const std::vector<uint8_t>& partial320) {
  WireStats st;
  // ---- golden V1 LE record (tests/golden/rln_h20_vectors.json: rln_proof_le, generated by oracle/pyref): parse,
  //      re-serialise (identical), BE round trip, V3 LE / mixed round trips
  std::unique_ptr<FFI_RLNProof> pr;
  {
    Vec_uint8_t v{const_cast<uint8_t*>(golden_le.data()), golden_le.size(), golden_le.size()};
    pr.reset(proof_from_bytes(&v, false));
    if (proof_bytes(*pr, false) != golden_le) { fprintf(stderr, "V1 LE proof does not re-serialise\n"); st.failures++; }
  }
  const std::vector<uint8_t> rec_be = proof_bytes(*pr, true), rec_v3 = v3_proof_bytes(*pr, false), rec_v3m = v3_proof_bytes(*pr, true);
  auto same_proof = [&](FFI_RLNProof* q) {
    std::unique_ptr<FFI_RLNProof> g(q);
    return proof_bytes(*g, false) == golden_le;
  };
  { Vec_uint8_t v{const_cast<uint8_t*>(rec_be.data()), rec_be.size(), rec_be.size()}; if (!same_proof(proof_from_bytes(&v, true))) st.failures++; }
  { Vec_uint8_t v{const_cast<uint8_t*>(rec_v3.data()), rec_v3.size(), rec_v3.size()}; if (!same_proof(v3_proof_from(&v, false))) st.failures++; }
  { Vec_uint8_t v{const_cast<uint8_t*>(rec_v3m.data()), rec_v3m.size(), rec_v3m.size()}; if (!same_proof(v3_proof_from(&v, true))) st.failures++; }
  if (rec_v3.size() != 289 || golden_le.size() != 290) { fprintf(stderr, "record sizes (SURVEY Appendix B)\n"); st.failures++; }
  // ---- a multi-message-id proof record (values: ys / nullifiers / selector_used vectors) around the same proof bytes
  FFI_RLNProof multi = *pr;
  multi.values.multi = true;
  multi.values.ys = {fr_small(5), fr_small(0), fr_small(7), fr_small(9)};
  multi.values.nullifiers = {fr_small(11), fr_small(0), fr_small(13), fr_small(15)};
  multi.values.selector_used = {1, 0, 1, 1};
  // ---- witnesses: single and multi, the partial witness, the partial proof (points of the pyref fixture)
  FFI_RLNWitnessInput w;
  w.identity_secret = fr_small(12345); w.user_message_limit = fr_small(100); w.message_id = fr_small(1);
  for (int i = 0; i < 20; i++) { w.path_elements.push_back(fr_small(1000 + i)); w.identity_path_index.push_back((uint8_t)(i & 1)); }
  w.x = fr_small(42); w.external_nullifier = fr_small(100);
  FFI_RLNWitnessInput wm = w;
  wm.multi = true; wm.message_ids = {fr_small(0), fr_small(4), fr_small(5), fr_small(9)}; wm.selector_used = {0, 1, 1, 1};
  FFI_RLNPartialWitnessInput pw;
  pw.identity_secret = w.identity_secret; pw.user_message_limit = w.user_message_limit;
  pw.path_elements = w.path_elements; pw.identity_path_index = w.identity_path_index;
  FFI_RLNPartialProof pp;
  pp.mask.assign(5843, 1);
  for (size_t i = 0; i < pp.mask.size(); i += 11) pp.mask[i] = 0;
  memcpy(pp.coords, partial320.data(), 320);
  struct Rec { const char* who; std::vector<uint8_t> bytes; void (*parse)(const Vec_uint8_t*); int flips; };
  const std::vector<Rec> recs = {
      {"V1 proof LE", golden_le, [](const Vec_uint8_t* v) { delete proof_from_bytes(v, false); }, 500},
      {"V1 proof BE", rec_be, [](const Vec_uint8_t* v) { delete proof_from_bytes(v, true); }, 500},
      {"V1 multi proof LE", proof_bytes(multi, false), [](const Vec_uint8_t* v) { delete proof_from_bytes(v, false); }, 500},
      {"V1 multi proof BE", proof_bytes(multi, true), [](const Vec_uint8_t* v) { delete proof_from_bytes(v, true); }, 300},
      {"V1 values LE", values_bytes(multi.values, false), [](const Vec_uint8_t* v) { Cursor c{v->ptr, v->len, 0, false}; (void)values_from(c); }, 300},
      {"V1 values BE", values_bytes(pr->values, true), [](const Vec_uint8_t* v) { Cursor c{v->ptr, v->len, 0, true}; (void)values_from(c); }, 300},
      {"V1 witness LE", witness_bytes(w, false), [](const Vec_uint8_t* v) { delete witness_from_bytes(v, false); }, 400},
      {"V1 witness BE", witness_bytes(w, true), [](const Vec_uint8_t* v) { delete witness_from_bytes(v, true); }, 400},
      {"V1 multi witness LE", witness_bytes(wm, false), [](const Vec_uint8_t* v) { delete witness_from_bytes(v, false); }, 400},
      {"V1 partial witness LE", partial_witness_bytes(pw, false), [](const Vec_uint8_t* v) { delete partial_witness_from_bytes(v, false); }, 300},
      {"V1 partial witness BE", partial_witness_bytes(pw, true), [](const Vec_uint8_t* v) { delete partial_witness_from_bytes(v, true); }, 300},
      {"V3 proof LE", rec_v3, [](const Vec_uint8_t* v) { delete v3_proof_from(v, false); }, 500},
      {"V3 proof mixed", rec_v3m, [](const Vec_uint8_t* v) { delete v3_proof_from(v, true); }, 500},
      {"V3 multi proof LE", v3_proof_bytes(multi, false), [](const Vec_uint8_t* v) { delete v3_proof_from(v, false); }, 300},
      {"V3 multi proof mixed", v3_proof_bytes(multi, true), [](const Vec_uint8_t* v) { delete v3_proof_from(v, true); }, 300},
      {"V3 witness LE", v3_witness_bytes(w, false), [](const Vec_uint8_t* v) { delete v3_witness_from(v, false); }, 400},
      {"V3 witness BE", v3_witness_bytes(w, true), [](const Vec_uint8_t* v) { delete v3_witness_from(v, true); }, 400},
      {"V3 multi witness LE", v3_witness_bytes(wm, false), [](const Vec_uint8_t* v) { delete v3_witness_from(v, false); }, 300},
      {"V3 multi witness BE", v3_witness_bytes(wm, true), [](const Vec_uint8_t* v) { delete v3_witness_from(v, true); }, 300},
      {"V3 partial witness LE", v3_partial_witness_bytes(pw, false), [](const Vec_uint8_t* v) { delete v3_partial_witness_from(v, false); }, 200},
      {"V3 partial witness BE", v3_partial_witness_bytes(pw, true), [](const Vec_uint8_t* v) { delete v3_partial_witness_from(v, true); }, 200},
  };
  // every golden record parses and re-serialises to itself
  for (const Rec& r : recs) {
    WireStats one;
    feed(one, r.who, r.bytes, r.parse);
    if (one.parsed != 1) { fprintf(stderr, "%s: the golden record does not parse\n", r.who); st.failures++; }
  }
  {
    auto rt = [&](const std::vector<uint8_t>& b, bool be, bool v3) {
      Vec_uint8_t v{const_cast<uint8_t*>(b.data()), b.size(), b.size()};
      std::unique_ptr<FFI_RLNWitnessInput> q(v3 ? v3_witness_from(&v, be) : witness_from_bytes(&v, be));
      return (v3 ? v3_witness_bytes(*q, be) : witness_bytes(*q, be)) == b;
    };
    for (int be = 0; be < 2; be++)
      for (int v3 = 0; v3 < 2; v3++)
        if (!rt(v3 ? v3_witness_bytes(w, be) : witness_bytes(w, be), be, v3) || !rt(v3 ? v3_witness_bytes(wm, be) : witness_bytes(wm, be), be, v3)) {
          fprintf(stderr, "witness round trip (be %d, v3 %d)\n", be, v3);
          st.failures++;
        }
  }
  for (const Rec& r : recs) fuzz(st, r.who, r.bytes, r.parse, r.flips);
  // the partial proof is 6 011 bytes of mask: prefixes at a stride, the point bytes bit by bit (decompression, subgroup check)
  {
    const std::vector<uint8_t> b1 = partial_proof_bytes(pp), b3 = v3_partial_proof_bytes(pp);
    auto p1 = [](const Vec_uint8_t* v) { delete partial_proof_from_bytes(v); };
    auto p3 = [](const Vec_uint8_t* v) { delete v3_partial_proof_from(v); };
    WireStats one;
    feed(one, "V1 partial proof", b1, p1);
    feed(one, "V3 partial proof", b3, p3);
    if (one.parsed != 2 || b1.size() != 6012 || b3.size() != 6011) { fprintf(stderr, "partial proof golden record\n"); st.failures++; }
    {
      Vec_uint8_t v{const_cast<uint8_t*>(b1.data()), b1.size(), b1.size()};
      std::unique_ptr<FFI_RLNPartialProof> q(partial_proof_from_bytes(&v));
      if (q->mask != pp.mask || memcmp(q->coords, pp.coords, 320) != 0) { fprintf(stderr, "partial proof round trip\n"); st.failures++; }
    }
    for (size_t k = 0; k <= b1.size(); k += (k < 16 || k + 200 > b1.size()) ? 1 : 97) {
      feed(st, "V1 partial proof", std::vector<uint8_t>(b1.begin(), b1.begin() + k), p1);
      if (k < b3.size()) feed(st, "V3 partial proof", std::vector<uint8_t>(b3.begin(), b3.begin() + k), p3);
    }
    for (int k = 0; k < 600; k++) {
      std::vector<uint8_t> q = (k & 1) ? b3 : b1;
      const size_t tail = 160 + 9, off = (k % 3 == 0) ? rnd() % 9 : q.size() - 1 - rnd() % (k % 3 == 1 ? 160 : tail);
      q[off] ^= (uint8_t)(1u << (rnd() % 8));
      if (k & 1) feed(st, "V3 partial proof", q, p3); else feed(st, "V1 partial proof", q, p1);
    }
    static const uint64_t hostile[] = {~0ull, 1ull << 63, 6012, 5844, 1ull << 32};
    for (uint64_t h : hostile) {
      std::vector<uint8_t> q = b1;
      for (int i = 0; i < 8; i++) q[1 + i] = (uint8_t)(h >> (8 * i));
      feed(st, "V1 partial proof", q, p1);
      q = b3;
      for (int i = 0; i < 8; i++) q[i] = (uint8_t)(h >> (8 * i));
      feed(st, "V3 partial proof", q, p3);
    }
  }
  // semantic refusals carry the reference's texts
  {
    FFI_RLNWitnessInput bad = w;
    bad.message_id = fr_small(100);   // == limit
    feed(st, "witness: message id at the limit", witness_bytes(bad, false), [](const Vec_uint8_t* v) { delete witness_from_bytes(v, false); });
    bad = w; bad.user_message_limit = fr_small(0);
    feed(st, "witness: zero limit", witness_bytes(bad, true), [](const Vec_uint8_t* v) { delete witness_from_bytes(v, true); });
    bad = w; bad.identity_path_index.pop_back();
    feed(st, "witness: ragged path", witness_bytes(bad, false), [](const Vec_uint8_t* v) { delete witness_from_bytes(v, false); });
    bad = wm; bad.message_ids[2] = bad.message_ids[1];
    feed(st, "witness: duplicate ids", witness_bytes(bad, false), [](const Vec_uint8_t* v) { delete witness_from_bytes(v, false); });
    if (st.refused < 4) { fprintf(stderr, "semantic refusals missing\n"); st.failures++; }
  }
  printf("sanitize_main: wire parsers: %zu inputs, %zu parsed, %zu refused with a reference error text, %d failures\n",
         st.parsed + st.refused, st.parsed, st.refused, st.failures);
  if (st.parsed + st.refused < 10000) { fprintf(stderr, "fewer than 10 000 wire inputs\n"); st.failures++; }
  return st.failures;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::vector<uint8_t> zb = slurp(argv[1]), gb = slurp(argv[2]), pb = slurp(argv[3]);
  if (zb.empty() || gb.empty() || pb.size() < 128 + 32) return 2;
  int failures = 0;
  // 1. the shipped files parse; the golden proof verifies; a flipped bit does not
  Zkey zk = parse_arkzkey(zb.data(), zb.size());
  Graph g = parse_graph(gb.data(), gb.size());
  const size_t npub = (pb.size() - 128) / 32;
  if (!verify(zk, pb.data(), pb.data() + 128, npub)) { fprintf(stderr, "golden proof rejected\n"); failures++; }
  for (int k = 0; k < 24; k++) {
    std::vector<uint8_t> q = pb;
    q[rnd() % q.size()] ^= (uint8_t)(1u << (rnd() % 8));
    if (q != pb && verify(zk, q.data(), q.data() + 128, npub)) { fprintf(stderr, "mutated proof accepted\n"); failures++; }
  }
  // 2. truncated and mutated zkey / graph files: an Error or a successful parse, nothing else
  size_t threw = 0, parsed = 0;
  for (int k = 0; k < 150; k++) {
    std::vector<uint8_t> q(zb.begin(), zb.begin() + (k < 60 ? rnd() % 4096 : zb.size()));
    if (k >= 60) for (int m = 0; m < 4; m++) q[rnd() % std::min<size_t>(q.size(), 2048)] = (uint8_t)rnd();
    try { (void)parse_arkzkey(q.data(), q.size()); parsed++; } catch (const std::exception&) { threw++; }
  }
  for (int k = 0; k < 400; k++) {
    std::vector<uint8_t> q(gb.begin(), gb.begin() + (k < 150 ? rnd() % gb.size() : gb.size()));
    if (k >= 150) for (int m = 0; m < 1 + (int)(rnd() % 6); m++) q[rnd() % q.size()] = (uint8_t)rnd();
    try {
      Graph h = parse_graph(q.data(), q.size());
      parsed++;
    } catch (const std::exception&) { threw++; }
  }
  // 3. the interpreter's scheduler on the shipped graph, both forms, with and without re-associated sums
  {
    std::vector<uint32_t> store_slot(g.nodes.size(), 0xFFFFFFFFu);
    uint32_t ns = 0;
    for (uint32_t n = 0; n < g.nodes.size(); n++)
      if (g.nodes[n].op == G_INPUT) store_slot[n] = ns++;
    for (uint32_t sgn : g.signals) if (store_slot[sgn] == 0xFFFFFFFFu) store_slot[sgn] = ns++;
    for (int rows = 0; rows < 2; rows++) {
      WlProgram p = wl_schedule(g, store_slot, ns, rows != 0);
      if (!p.ok || p.nsteps == 0) { fprintf(stderr, "no schedule\n"); failures++; }
    }
    // round 6: the unknown cone of evaluate_partial cut out of the graph and scheduled like it (wl_cone)
    {
      WlCone c = wl_cone(g);
      WlProgram p = wl_schedule(c.graph, wl_cone_store_slots(c, store_slot), ns, true);
      if (!p.ok || p.nsteps == 0 || c.node_of.size() >= g.nodes.size() / 4) { fprintf(stderr, "no cone schedule\n"); failures++; }
    }
  }
  // 3b. graphs the scheduler's rewriting passes were not written for: a chain of 40 000 Adds over ONE value (a linear form
  //     of one term however deep it is looked into), a chain of constant products, a sum of a node with itself -- a
  //     schedule, a refusal (ok = false) or an Error, nothing else
  for (int shape = 0; shape < 3; shape++) {
    Graph h;
    h.constants.push_back(Fr::one() + Fr::one());
    h.inputs_size = 2;
    h.nodes.push_back(GNode{G_INPUT, 1, 0, 0});
    h.nodes.push_back(GNode{G_CONST, 0, 0, 0});
    h.nodes.push_back(GNode{G_MUL, 0, 0, 0});                        // x * x: a product below the chains
    uint32_t last = 2;
    const uint32_t len = shape == 0 ? 40000u : 3000u;
    for (uint32_t k = 0; k < len; k++) {
      if (shape == 0) h.nodes.push_back(GNode{G_ADD, last, 2, 0});
      else if (shape == 1) h.nodes.push_back(GNode{G_MUL, 1, last, 0});
      else h.nodes.push_back(GNode{G_ADD, last, last, 0});
      last = (uint32_t)h.nodes.size() - 1;
    }
    h.nodes.push_back(GNode{G_MUL, last, last, 0});
    h.signals = {0, (uint32_t)h.nodes.size() - 1};
    std::vector<uint32_t> st(h.nodes.size(), 0xFFFFFFFFu);
    st[0] = 0;
    st[h.nodes.size() - 1] = 1;
    try { (void)wl_schedule(h, st, 2, true); parsed++; } catch (const std::exception&) { threw++; }
  }
  // 4. config_path JSON: well-formed, malformed, hostile
  const char* cfgs[] = {"{}", "{\"temporary\": true}", "{\"devices\": [0, 1, 2]}", "{\"devices\": [0, ]}", "{\"devices\": [",
                        "{\"path\": \"/tmp/x\\\"y\", \"temporary\": false}", "{\"a\": [[[[{\"b\": \"]\"}]]]], \"max_batch\": 64}",
                        "{\"a\": [", "{\"a\": \"", "{", "", "[1]", "{\"tree_depth\": 99999999999999999999999}",
                        "{\"window_bits\": -5, \"devices\": [-1]}", "{\"devices\": [1e9]}", "{\"x\": {\"y\": {\"z\": [1, 2, {\"w\": null}]}}}"};
  for (const char* c : cfgs) {
    try { (void)parse_tree_config(c); parsed++; } catch (const std::exception&) { threw++; }
  }
  for (int k = 0; k < 2000; k++) {   // random bytes from the JSON alphabet
    static const char al[] = "{}[]\",:0123456789-truefalsn \\abde";
    std::string sx;
    for (int m = 0; m < 1 + (int)(rnd() % 40); m++) sx += al[rnd() % (sizeof al - 1)];
    try { (void)parse_tree_config(sx); parsed++; } catch (const std::exception&) { threw++; }
  }
  // 5. the wire parsers of the C ABI
  if (argc >= 6) {
    std::vector<uint8_t> le = slurp(argv[4]), p320 = slurp(argv[5]);
    if (le.size() < 129 || p320.size() != 320) return 2;
    failures += wire_section(le, p320);
  }
  printf("sanitize_main: %zu inputs parsed, %zu rejected with an error, %d failures\n", parsed, threw, failures);
  return failures ? 1 : 0;
}
