// Sanitizer harness ("ASan/UBSan in CPU tests").  The product's HOST code that reads bytes it does not
// control -- the arkzkey and graph parsers (zkey.cpp), the config_path JSON parser (tree_config.h), proof decompression
// and the pairing verifier (pairing.h) -- plus the interpreter's scheduler (witness_sched.cpp), compiled by g++ with
// -fsanitize=address,undefined and run on the shipped resources, on a golden proof and on a few thousand truncated /
// bit-flipped inputs.  Malformed input must end in rlnamd::Error, never in a sanitizer report.  CPU only: the GPU pool
// refuses sanitizer runs, and no HIP call is reached here.
//   usage: sanitize_main <zkey> <graph> <proof128 + public inputs file>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
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
  printf("sanitize_main: %zu inputs parsed, %zu rejected with an error, %d failures\n", parsed, threw, failures);
  return failures ? 1 : 0;
}
