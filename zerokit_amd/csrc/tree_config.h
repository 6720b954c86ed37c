// tree_config.h -- the `config_path` JSON of ffi_rln_new* (PmTreeConfig, rln/src/pm_tree_adapter.rs:71-176) plus this
// backend's own keys in the same object.  Pure host code with no HIP call, in a header of its own so that the CPU suite
// can compile it with AddressSanitizer / UBSan (tests/host/sanitize_main.cpp): the text comes from a file the caller names.
#pragma once
#include <ctype.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"
#include "prover.h"

namespace rlnamd {

// ---- tree persistence --------------------------------------------------------------------------------------------
// The reference keeps the default (pmtree-ft) tree in a sled database under `path` (pm_tree_adapter.rs:71-176,
// 191-239).  sled's on-disk format belongs to a third-party crate (sled 0.34.7) that is not in the tree, so the state is
// kept in ONE snapshot file of our own, `<path>/rlnamd_tree.bin`: the same config keys and lifecycle (load when
// present, otherwise start empty; written by ffi_flush and when the object is freed), not readable by sled.  The
// tree itself stays in HBM; a snapshot holds depth, next_index, the metadata bytes and the leaves below next_index.
struct TreeConfig {
  std::string path;
  bool has_path = false;
  bool temporary = true;   // DEFAULT_TEMPORARY (pm_tree_adapter.rs:67)
  long tree_depth = -1;
  // prover sizing, keys of THIS backend in the same JSON object (the reference's PmTreeConfig::from_str picks its keys
  // out of a serde_json::Value and ignores the rest, so one config file serves both): "window_bits" = the comb schedule
  // of rlnamd_prover_new (7150114 = the 228 GiB bench schedule), "max_batch" = workspace capacity in proofs.
  // 0 / absent: RLNAMD_WINDOW_BITS / RLNAMD_MAX_BATCH, else the defaults (20 GiB tables: G1 c = 10, G2 c = 12; 256 proofs --
  // a single proof is as fast as with 64, a batch call streams at 15 k instead of 11 k proofs/s).
  long window_bits = 0, max_batch = 0;
  // "profile": a name for the two numbers above, so that a caller need not know the schedule's encoding --
  //   "latency"     the defaults (20 GiB of tables, 256 proofs of workspace): one proof 2.2 ms, batch calls 15 k proofs/s
  //   "throughput"  the bench's operating point (window_bits 7150114: 228 GiB; max_batch 1024): batch calls 21 k proofs/s,
  //                 5 - 7 s to build, nothing else of that size fits the device
  //   "small"       window_bits 8, max_batch 64: 7.7 GiB, one proof 2.7 ms, batch calls 9 k proofs/s
  // explicit "window_bits" / "max_batch" keys override what the profile implies; any other name is a configuration error.
  std::string profile;
  // "devices": [0, 1, ...] -- two or more entries put an rlnamd_pool (a prover replica + a host thread per listed device)
  // behind the object: ffi_generate_rln_proofs_batch then shards n > max_batch proofs over the devices by index
  // (BASELINE config 4: 65 536 = 8 x 8 192).  Everything else -- single proofs, the tree, verification -- runs on the
  // first listed device, which must be the calling thread's current device (device 0 unless the host chose otherwise).
  std::vector<int> devices;
  bool has_devices = false;
  // with "devices": "dynamic_shards": true = rlnamd_pool_set_dynamic (the replicas draw chunks from a shared cursor instead
  // of taking contiguous shards); "failover": k = rlnamd_pool_set_failover (a failing device's chunks are proved again
  // by the others, up to k times per call, and the device is left out of the next "revive_after" batch calls -- 8 unless
  // the key says otherwise, 0 = for the object's lifetime)
  bool dynamic_shards = false;
  long failover = 0;
  long revive_after = 8;
  // "partial_cache": N -- entries of the prover's partial-proof cache (rln_amd.h: rlnamd_prover_collect_partial_cached; ~0.26 MB
  // each on the depth-20 circuit): how many partial proofs made by ffi_generate_partial_zk_proof can be finished through the
  // short path at a time.  -1: the prover's default (RLNAMD_PARTIAL_CACHE or 64); 0: off
  long partial_cache = -1;
  // "auto_partial": N -- ffi_generate_rln_proof remembers the partial proofs of up to N members and finishes instead of
  // proving from scratch when a member proves again at the same root (ffi.cpp: FFI_RLN::auto_partial).  0 (default): off
  long auto_partial = 0;
  // "gather_calls": N -- single-proof calls that arrive from other threads while a proof is on the device go out together,
  // as one batch of up to N, when it returns (ffi.cpp: prove_one).  -1 (default): on, up to the workspace's capacity;
  // 0 or 1: every call is its own batch, one after the other
  long gather_calls = -1;
  // "gather_window_us": how long the leader of a gathered batch waits for callers it saw within the last 20 ms
  // (500 unless given; 0: it takes what is queued)
  long gather_window_us = 500;
  bool persistent() const { return !temporary && has_path; }
  ProverConfig prover_config() const {
    ProverConfig cfg;
    const char* mb = getenv("RLNAMD_MAX_BATCH");
    long pw = 0, pb = 0;   // what the profile implies
    if (profile == "throughput") { pw = 7150114; pb = 1024; }
    else if (profile == "small") { pw = 8; pb = 64; }
    cfg.max_batch = max_batch > 0 ? (size_t)max_batch : pb > 0 ? (size_t)pb : (mb && *mb ? (size_t)atoll(mb) : 256);
    cfg.window_bits = window_bits > 0 ? (int)window_bits : (int)pw;   // 0: Prover takes RLNAMD_WINDOW_BITS or its default schedule
    cfg.partial_cache = partial_cache >= 0 ? partial_cache : (auto_partial > 0 ? std::max(64l, auto_partial + 8) : -1);
    return cfg;
  }
};

// flat JSON object with string / number / bool / null values (PmTreeConfig::from_str, pm_tree_adapter.rs:139-176)
inline TreeConfig parse_tree_config(const std::string& js) {
  TreeConfig c;
  size_t i = 0;
  auto bad = [&](const char* what) -> Error {
    return Error(std::string("Configuration error: Error while reading pmtree config: ") + what + " at column " +
                 std::to_string(i));
  };
  auto ws = [&]() { while (i < js.size() && isspace((unsigned char)js[i])) i++; };
  auto str = [&]() {
    std::string o;
    if (js[i] != '"') throw bad("expected a string");
    for (i++; i < js.size() && js[i] != '"'; i++) {
      if (js[i] == '\\' && i + 1 < js.size()) {
        char e = js[++i];
        o += e == 'n' ? '\n' : e == 't' ? '\t' : e;
      } else {
        o += js[i];
      }
    }
    if (i >= js.size()) throw bad("unterminated string");
    i++;
    return o;
  };
  ws();
  if (i >= js.size() || js[i] != '{') throw bad("expected value");
  i++;
  ws();
  while (i < js.size() && js[i] != '}') {
    std::string key = str();
    ws();
    if (i >= js.size() || js[i] != ':') throw bad("expected `:`");
    i++;
    ws();
    if (i >= js.size()) throw bad("EOF while parsing a value");
    if (js[i] == '"') {
      std::string v = str();
      if (key == "path") { c.path = v; c.has_path = true; }
      if (key == "profile") {
        if (v != "latency" && v != "throughput" && v != "small")
          throw Error("Configuration error: profile: expected \"latency\", \"throughput\" or \"small\", got \"" + v + "\"");
        c.profile = v;
      }
    } else if (!js.compare(i, 4, "true") || !js.compare(i, 5, "false")) {
      bool v = js[i] == 't';
      i += v ? 4 : 5;
      if (key == "temporary") c.temporary = v;
      if (key == "dynamic_shards") c.dynamic_shards = v;
    } else if (!js.compare(i, 4, "null")) {
      i += 4;
    } else if (js[i] == '[' && key == "devices") {
      // strict, as serde_json would read a Vec<i32>: `[` (int (`,` int)*)? `]`, no trailing comma, no bare `-`
      std::vector<int> vals;
      i++;
      ws();
      if (i < js.size() && js[i] == ']') {
        i++;
      } else {
        for (;;) {
          ws();
          size_t j = i;
          if (j < js.size() && js[j] == '-') j++;
          size_t d0 = j;
          while (j < js.size() && isdigit((unsigned char)js[j])) j++;
          if (j == d0 || j - d0 > 9) throw bad("expected value");
          vals.push_back(atoi(js.substr(i, j - i).c_str()));
          i = j;
          ws();
          if (i >= js.size()) throw bad("EOF while parsing a list");
          if (js[i] == ',') { i++; continue; }
          if (js[i] == ']') { i++; break; }
          throw bad("expected `,` or `]`");
        }
      }
      for (int d : vals)
        if (d < 0) throw Error("Configuration error: devices: negative device ordinal");
      c.devices = vals;
      c.has_devices = true;
    } else if (js[i] == '[' || js[i] == '{') {
      // any other array / object (a key this library does not know): skipped as a whole, like serde ignores unknown fields
      int depth = 0;
      bool in_str = false;
      for (; i < js.size(); i++) {
        const char ch = js[i];
        if (in_str) {
          if (ch == '\\') i++;
          else if (ch == '"') in_str = false;
          continue;
        }
        if (ch == '"') in_str = true;
        else if (ch == '[' || ch == '{') depth++;
        else if (ch == ']' || ch == '}') {
          if (--depth == 0) { i++; break; }
        }
      }
      if (depth != 0) throw bad("EOF while parsing a value");
    } else if (isdigit((unsigned char)js[i]) || js[i] == '-') {
      size_t j = i;
      while (j < js.size() && (isdigit((unsigned char)js[j]) || strchr("+-.eE", js[j]))) j++;
      long num = atol(js.substr(i, j - i).c_str());
      if (key == "tree_depth") c.tree_depth = num;
      if (key == "window_bits") c.window_bits = num;
      if (key == "max_batch") c.max_batch = num;
      if (key == "failover") {
        if (num < 0 || num > 64) throw Error("Configuration error: failover: expected 0 .. 64 rounds");
        c.failover = num;
      }
      if (key == "auto_partial") {
        if (num < 0 || num > 65536) throw Error("Configuration error: auto_partial: expected 0 .. 65536 members");
        c.auto_partial = num;
      }
      if (key == "gather_calls") {
        if (num < 0 || num > 65536) throw Error("Configuration error: gather_calls: expected 0 .. 65536 calls");
        c.gather_calls = num;
      }
      if (key == "gather_window_us") {
        if (num < 0 || num > 100000) throw Error("Configuration error: gather_window_us: expected 0 .. 100000 microseconds");
        c.gather_window_us = num;
      }
      if (key == "partial_cache") {
        if (num < 0 || num > 1000000) throw Error("Configuration error: partial_cache: expected 0 .. 1000000 entries");
        c.partial_cache = num;
      }
      if (key == "revive_after") {
        if (num < 0 || num > 1000000) throw Error("Configuration error: revive_after: expected 0 .. 1000000 calls");
        c.revive_after = num;
      }
      i = j;
    } else {
      throw bad("expected value");
    }
    ws();
    if (i < js.size() && js[i] == ',') { i++; ws(); }
    else if (i < js.size() && js[i] != '}') throw bad("expected `,` or `}`");
  }
  if (i >= js.size()) throw bad("EOF while parsing an object");
  // resolve_path (pm_tree_adapter.rs:93-100)
  if (!c.temporary && !c.has_path) throw Error("Configuration error: Error while creating pmtree config: missing path");
  struct stat st;
  if (c.temporary && c.has_path && stat(c.path.c_str(), &st) == 0)
    throw Error("Configuration error: Error while creating pmtree config: path already exists");
  return c;
}

// the config_path argument of ffi_rln_new*: a JSON file; unreadable / missing / oversized file == "" == defaults
// (ffi_rln.rs:28-45: `.unwrap_or_default()`)
inline TreeConfig tree_config_from_file(const char* config_path) {
  std::string js;
  if (config_path && *config_path) {
    FILE* f = fopen(config_path, "rb");
    if (f) {
      char buf[4096];
      size_t n;
      while ((n = fread(buf, 1, sizeof buf, f)) > 0 && js.size() <= (1u << 20)) js.append(buf, n);
      fclose(f);
      if (js.size() > (1u << 20)) js.clear();  // MAX_CONFIG_SIZE
    }
  }
  if (js.empty()) return TreeConfig();
  return parse_tree_config(js);
}


}  // namespace rlnamd
