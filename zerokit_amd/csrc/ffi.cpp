// zerokit-compatible C ABI (include/rln.h) on top of the HIP modules.
//
// Mirrors the delegation structure of the reference: ffi_* (rln/src/ffi/*.rs) -> RLN (rln/src/public.rs)
// -> protocol / tree.  Every compute call lands in a HIP kernel (prover.hip, merkle.hip, poseidon.hip);
// only byte shuffling, validation and proof verification (public.rs:725-745, CPU in the reference too)
// run on the host.
#include "../../include/rln.h"

#include <ctype.h>
#include <dlfcn.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>
#include <sys/stat.h>

#include <algorithm>
#include <fstream>
#include <memory>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "../../include/rln_amd.h"
#include "capi_util.h"
#include "common.h"
#include "keccak.h"
#include "merkle.h"
#include "tree_any.h"
#include "tree_config.h"   // TreeConfig, parse_tree_config, tree_config_from_file
#include "pairing.h"
#include "poseidon.h"
#include "prover.h"
#include "gather.h"
#include "ffi_wire.h"   // CFr, the FFI object structs, Cursor / V3Reader, every (de)serialiser and validation

using namespace rlnamd;


// ---------------------------------------------------------------------------------- small host helpers
namespace {

Vec_uint8_t no_err() { return {nullptr, 0, 0}; }
Vec_uint8_t make_str(const std::string& s) {  // NUL-terminated, len excludes the terminator
  uint8_t* p = (uint8_t*)malloc(s.size() + 1);
  memcpy(p, s.c_str(), s.size() + 1);
  return {p, s.size(), s.size() + 1};
}
Vec_uint8_t make_bytes(const std::vector<uint8_t>& v) {
  uint8_t* p = (uint8_t*)malloc(v.size() ? v.size() : 1);
  if (!v.empty()) memcpy(p, v.data(), v.size());
  return {p, v.size(), v.size() ? v.size() : 1};
}
Vec_CFr_t make_vec_cfr(const std::vector<CFr>& v) {
  CFr* p = (CFr*)malloc((v.size() ? v.size() : 1) * sizeof(CFr));
  if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(CFr));
  return {(CFr_t*)p, v.size(), v.size() ? v.size() : 1};
}
CFr_t* box_cfr(const CFr& v) {
  CFr* p = (CFr*)malloc(sizeof(CFr));
  *p = v;
  return (CFr_t*)p;
}
const CFr& R(const CFr_t* p) { return *(const CFr*)p; }


std::string lib_dir() {
  Dl_info info;
  if (dladdr((void*)&lib_dir, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    size_t k = p.find_last_of('/');
    return k == std::string::npos ? "." : p.substr(0, k);
  }
  return ".";
}
std::vector<uint8_t> read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw Error("I/O error: cannot open " + path);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
// The reference embeds the depth-20 resources with include_bytes! (circuit/mod.rs:29-42); here they are
// the same files shipped beside the library (override with RLNAMD_RESOURCES).
std::string resource_dir(size_t depth) {
  const char* env = getenv("RLNAMD_RESOURCES");
  std::string base = env && *env ? env : lib_dir() + "/../resources";
  return base + "/tree_depth_" + std::to_string(depth);
}

// Uniform in [0, r) by rejection: 32 fresh bytes from the kernel's CSPRNG, masked to 254 bits, redrawn as a whole
// while >= r (the loop of <Fr as UniformRand>::rand, ark-ff 0.5.0; the reference calls it with thread_rng,
// protocol/proof.rs:743-745, protocol/keygen.rs).  r / 2^254 = 0.756, so 1.32 draws on average.
CFr random_fr() {
  CFr r;
  for (;;) {
    uint8_t b[32];
    size_t got = 0;
    while (got < sizeof b) {
      ssize_t k = getrandom(b + got, sizeof b - got, 0);
      if (k < 0) {
        if (errno == EINTR) continue;
        throw Error(std::string("getrandom failed: ") + strerror(errno));
      }
      got += (size_t)k;
    }
    b[31] &= 0x3F;
    if (!is_canonical(b)) continue;
    memcpy(r.le, b, 32);
    return r;
  }
}

Fr to_fr(const CFr& v) {
  uint32_t c[8];
  memcpy(c, v.le, 32);
  return Fr::from_canonical(c);
}
CFr from_fr(const Fr& f) {
  CFr r;
  uint32_t c[8];
  f.to_canonical(c);
  memcpy(r.le, c, 32);
  return r;
}

// ChaCha20Rng::from_seed (rand_chacha 0.3.1, pinned in Cargo.lock): 20 rounds, 64-bit block counter from 0,
// stream id 0; the output is the keystream read as little-endian words.
struct ChaCha20Rng {
  uint32_t key[8];
  uint64_t counter = 0;
  uint8_t buf[64];
  size_t pos = 64;
  explicit ChaCha20Rng(const uint8_t seed[32]) { memcpy(key, seed, 32); }
  static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  void block() {
    uint32_t st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    memcpy(st + 4, key, 32);
    st[12] = (uint32_t)counter;
    st[13] = (uint32_t)(counter >> 32);
    st[14] = st[15] = 0;
    uint32_t x[16];
    memcpy(x, st, 64);
    auto qr = [&](int a, int b, int c, int d) {
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
      x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
      x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    };
    for (int i = 0; i < 10; i++) {
      qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
      qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) {
      uint32_t w = x[i] + st[i];
      memcpy(buf + 4 * i, &w, 4);
    }
    counter++;
    pos = 0;
  }
  void fill(uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
      if (pos == 64) block();
      out[i] = buf[pos++];
    }
  }
  // <Fr as UniformRand>::rand (ark-ff 0.5.0 fields/models/fp/mod.rs): four u64 limbs, top two bits shaved,
  // rejected while >= r; the accepted limbs ARE the Montgomery representation, so the value is raw / 2^256.
  CFr next_fr() {
    for (;;) {
      uint8_t raw[32];
      fill(raw, 32);
      raw[31] &= 0x3F;
      if (!is_canonical(raw)) continue;
      CFr c;
      memcpy(c.le, raw, 32);
      uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
      Fr m;  // the Montgomery form of 1/R is the plain integer 1
      memcpy(m.v, one, 32);
      return from_fr(to_fr(c) * m);
    }
  }
};
ChaCha20Rng seeded_rng(const Vec_uint8_t* seed) {  // keygen.rs:50-58: the seed is keccak256 of the signal
  uint8_t h[32];
  keccak256(seed ? seed->ptr : nullptr, seed ? seed->len : 0, h);
  return ChaCha20Rng(h);
}
CFr poseidon_host_call(const std::vector<CFr>& in) {  // one-lane launch of the batch kernel
  CFr out;
  memset(out.le, 0, 32);
  if (rlnamd_poseidon_hash((const uint8_t*)in.data(), 1, in.size(), out.le) != 0)
    throw Error(std::string("poseidon: ") + rlnamd_last_error());
  return out;
}

// entry points without an error channel: see the keygen block of the extern "C" section
template <class T, class F>
T no_channel(const char* who, T failed, F&& f) noexcept {
  try {
    return f();
  } catch (const std::exception& e) {
    fprintf(stderr, "librln: %s: %s\n", who, e.what());
  } catch (...) {
    fprintf(stderr, "librln: %s: unknown error\n", who);
  }
  return failed;
}
const Vec_CFr_t kNoVecCFr = {nullptr, 0, 0};

// compute_id_secret (protocol/slashing.rs:12-36)
CFr compute_id_secret(const CFr& x1, const CFr& y1, const CFr& x2, const CFr& y2) {
  Fr dx = to_fr(x1) - to_fr(x2);
  if (dx.is_zero()) throw Error("Cannot recover secret: division by zero (shares have the same x value)");
  Fr a1 = (to_fr(y1) - to_fr(y2)) * dx.inv();
  return from_fr(to_fr(y1) - to_fr(x1) * a1);
}

// BigInt::from_str + calculated_witness_to_field_elements (protocol/proof.rs:593-614): decimal, optional sign,
// reduced mod r; negative values map to r - |w|
CFr parse_bigint_fr(const uint8_t* p, size_t n) {
  size_t i = 0;
  bool neg = false;
  if (i < n && (p[i] == '+' || p[i] == '-')) neg = p[i++] == '-';
  if (i >= n) throw Error("Failed to parse witness: cannot parse integer from empty string");
  if (p[i] == '_') throw Error("Failed to parse witness: invalid digit found in string");
  Fr acc = Fr::zero(), ten = to_fr(cfr_from_u64(10));
  bool over = false;  // |w| > r is only an error for negative values
  std::vector<uint8_t> digits;
  for (; i < n; i++) {
    if (p[i] == '_') continue;
    if (p[i] < '0' || p[i] > '9') throw Error("Failed to parse witness: invalid digit found in string");
    acc = acc * ten + to_fr(cfr_from_u64(p[i] - '0'));
    digits.push_back(p[i] - '0');
  }
  if (neg) {
    // compare |w| with r through its decimal length / value (r has 77 digits)
    static const char* RDEC = "21888242871839275222246405745257275088548364400416034343698204186575808495617";
    size_t lead = 0;
    while (lead + 1 < digits.size() && digits[lead] == 0) lead++;
    size_t len = digits.size() - lead;
    if (len > 77) over = true;
    else if (len == 77)
      for (size_t k = 0; k < 77; k++) {
        int d = digits[lead + k], r = RDEC[k] - '0';
        if (d != r) {
          over = d > r;
          break;
        }
      }
    if (over) throw Error("Cannot convert bigint to biguint: negative value below -r");
    acc = Fr::zero() - acc;
  }
  return from_fr(acc);
}

std::string json_str_array(const std::vector<std::string>& v) {
  std::string s = "[";
  for (size_t i = 0; i < v.size(); i++) s += (i ? ",\"" : "\"") + v[i] + "\"";
  return s + "]";
}
}  // namespace

// -------------------------------------------------------------------------------------- object model

struct FFI_RLN {
  // generate / verify take &self in the reference and may be called from several threads (SURVEY section 8b,
  // "Threading"); the prover owns one set of device workspaces, so proving calls on one object take turns
  std::shared_ptr<std::mutex> prove_mu = std::make_shared<std::mutex>();   // (shared: a partial proof's cache entry is released under it)
  std::shared_ptr<Prover> prover;   // owned, or replica 0 of `pool` (then the pool owns it)
  rlnamd_pool* pool = nullptr;      // config "devices" with two or more entries: batch calls shard over the devices
  // The object lives on ONE device -- its tree, its prover (replica 0 of a pool) and every later call from the caller's
  // thread use the CURRENT device -- so a "devices" list must start with the device that is current when the object is
  // made; anything else would put the tree and the single proofs on another GPU than the one the list names.
  int home_device = 0;
  // "auto_partial": N in the config_path JSON (0 = off, the default).  The reference's own headline optimisation --
  // cache the part of the proof that depends on the member, finish per message (rln/README.md:360-375) -- applied behind
  // the UNCHANGED ffi_generate_rln_proof: the object remembers, for up to N members (identity secret, limit, Merkle
  // path), the partial proof and the prover's cache handle.  The first proof of a member at a root is a full proof, and
  // its partial proof is made on the device behind the call's return; every later proof of that member at that root is
  // a finish through the cone (0.8 instead of 2.1 ms).  Same proof for the same (r, s): nothing a verifier can see.
  // What it costs is a policy decision, hence opt-in: the member's witness values stay on the device, and the key
  // (secret included) in host memory, until the entry is evicted (LRU), the tree moves on, or the object is freed.
  size_t auto_partial = 0;
  // single calls from several threads gathered into batches: gather.h
  struct GatherReq {
    FFI_RLNWitnessInput* w = nullptr;
    const FFI_RLNPartialProof* pp = nullptr;   // the finish queue: the partial proof to finish
    CFr rs[2];
    bool has_rs = false, done = false;
    FFI_RLNProof* out = nullptr;
    std::string err;
    // the caller's own thread packs its inputs and hashes its hints before it queues (prove_one): the chains of a
    // gathered batch are then hashed by as many threads as there are callers, not by the leader
    std::vector<uint8_t> inputs;
    std::vector<uint32_t> hints;
    ~GatherReq() {
      secure_zero(inputs.data(), inputs.size());
      secure_zero(hints.data(), hints.size() * 4);
    }
    void gather_failed() {
      if (!out && err.empty()) err = "Error producing proof: out of memory";
    }
  };
  typedef GatherQueue<GatherReq> Gather;
  Gather gather,   // ffi_generate_rln_proof and its twins
      gather_fin;   // ffi_finish_rln_proof and its twins: finishes of partial proofs gathered the same way
  struct Memo {
    std::vector<uint8_t> key;   // identity secret | limit | path elements | path index
    uint8_t coords[320];
    uint64_t handle = 0, stamp = 0;
  };
  std::vector<Memo> memo;
  uint64_t memo_clock = 0, memo_hits = 0, memo_misses = 0;
  uint64_t pending_ticket = 0;          // a partial batch of one proof enqueued behind the last miss ...
  std::vector<uint8_t> pending_key;     // ... for this member
  void memo_drop(Memo& m) {             // (caller holds prove_mu)
    if (m.handle) prover->release_partial(&m.handle, 1);
    secure_zero(m.key.data(), m.key.size());
    m.key.clear();
    m.handle = 0;
  }
  void memo_adopt_pending() {           // the partial proof enqueued behind the last miss, if any: collect it into the memo
    if (!pending_ticket) return;
    const uint64_t t = pending_ticket;
    pending_ticket = 0;
    Memo m;
    m.key.swap(pending_key);
    uint32_t err = 0;
    try {
      prover->collect_partial_cached(t, 1, m.coords, &m.handle, &err);
    } catch (...) {
      secure_zero(m.key.data(), m.key.size());
      return;
    }
    if (err || !m.handle) {   // not a witness, or no room in the prover's cache: nothing to remember
      memo_drop(m);
      return;
    }
    m.stamp = ++memo_clock;
    if (memo.size() >= auto_partial) {   // evict the least recently used member
      size_t lru = 0;
      for (size_t i = 1; i < memo.size(); i++)
        if (memo[i].stamp < memo[lru].stamp) lru = i;
      memo_drop(memo[lru]);
      memo[lru] = std::move(m);
    } else {
      memo.push_back(std::move(m));
    }
  }
  void memo_clear() {
    if (pending_ticket && prover) {
      try {
        prover->wipe(pending_ticket);
      } catch (...) {
      }
    }
    pending_ticket = 0;
    secure_zero(pending_key.data(), pending_key.size());
    pending_key.clear();
    for (Memo& m : memo) memo_drop(m);
    memo.clear();
  }
  void make_prover(const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len, const TreeConfig& tcfg) {
    const ProverConfig cfg = tcfg.prover_config();
    auto_partial = tcfg.auto_partial > 0 ? (size_t)tcfg.auto_partial : 0;
    gather_wanted = tcfg.gather_calls;
    if (const char* e = getenv("RLNAMD_GATHER_CALLS"))
      if (*e) gather_wanted = atol(e);
    gather.window_us = tcfg.gather_window_us;
    if (const char* e = getenv("RLNAMD_GATHER_WINDOW_US"))
      if (*e) gather.window_us = std::min(std::max(0l, atol(e)), 100000l);
    if (hipGetDevice(&home_device) != hipSuccess) home_device = 0;
    if (tcfg.has_devices) {
      if (tcfg.devices.empty()) throw Error("Configuration error: devices: empty list");
      if (tcfg.devices[0] != home_device)
        throw Error("Configuration error: devices: the list starts with device " + std::to_string(tcfg.devices[0]) +
                    " but the calling thread's current device is " + std::to_string(home_device) +
                    " (the object's tree and single proofs live on the current device: make it current first)");
    }
    if (tcfg.devices.size() >= 2) {
      if (rlnamd_pool_new(zkey, zkey_len, graph, graph_len, cfg.max_batch, cfg.window_bits, tcfg.devices.data(),
                          tcfg.devices.size(), &pool) != RLNAMD_OK)
        throw Error(std::string("Configuration error: devices: ") + rlnamd_last_error());
      rlnamd_pool_set_dynamic(pool, tcfg.dynamic_shards ? 1 : 0);
      rlnamd_pool_set_failover(pool, (int)tcfg.failover);
      rlnamd_pool_set_probation(pool, tcfg.failover > 0 ? (size_t)tcfg.revive_after : 0);
      prover = std::shared_ptr<Prover>(rlnamd_pool_replica_prover(pool, 0), [](Prover*) {});
    } else {
      // A default object (no "profile" / "window_bits" / "max_batch" key, no RLNAMD_* sizing) takes the latency point:
      // ~20 GiB of comb tables + ~3 GiB of workspaces.  Several objects per process, or a device that other work has
      // filled, may not have that: then the object is built at the "small" point (7.7 GiB + 0.8 GiB: the same bytes
      // out, single proofs within 0.2 ms of the default, batch calls slower) instead of failing where the reference's
      // constructor cannot fail for lack of memory; below that the error names the key to set.
      ProverConfig use = cfg;
      const bool sized_by_caller = tcfg.window_bits > 0 || tcfg.max_batch > 0 || !tcfg.profile.empty() ||
                                   (getenv("RLNAMD_WINDOW_BITS") && *getenv("RLNAMD_WINDOW_BITS")) ||
                                   (getenv("RLNAMD_MAX_BATCH") && *getenv("RLNAMD_MAX_BATCH"));
      size_t free_b = 0, total_b = 0;
      if (!sized_by_caller && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t GiB = (size_t)1 << 30;
        if (const char* t = getenv("RLNAMD_ASSUME_FREE_GIB"))   // test hook: the decision below without filling a device
          if (*t) free_b = (size_t)atoll(t) * GiB;
        if (free_b < 26 * GiB) {
          if (free_b < 10 * GiB)
            throw Error("Configuration error: " + std::to_string(free_b / GiB) + " GiB of device memory free; an RLN object needs ~23 GiB "
                        "at the default operating point and ~9 GiB at {\"profile\": \"small\"} (config_path JSON)");
          use.window_bits = 8;
          use.max_batch = 64;
        }
      }
      prover.reset(new Prover(zkey, zkey_len, graph, graph_len, use));
    }
    const size_t cap = prover->capacity();
    gather.most = gather_wanted < 0 ? cap : gather_wanted <= 1 ? 0 : std::min((size_t)gather_wanted, cap);
    gather_fin.most = gather.most;
    gather_fin.window_us = gather.window_us;
  }
  long gather_wanted = -1;
  TreeAny tree;   // dense in HBM up to depth 30, sparse (host-indexed, device-hashed) for 31 .. 63
  bool stateless = false;  // V3 only (RLNV3<Stateless, _>): no tree, tree calls return an error
  size_t next_index = 0;
  std::vector<uint8_t> leaf_set;  // cached_leaves_indices
  std::vector<uint8_t> metadata;
  std::string store;              // snapshot file of a persistent tree ("" = temporary tree)

  ~FFI_RLN() {
    try {
      flush();  // sled flushes when the database is dropped
    } catch (...) {
    }
    try {
      memo_clear();
    } catch (...) {
    }
    prover.reset();
    if (pool) rlnamd_pool_free(pool);
  }

  // PoseidonTree::default(depth) (public.rs:298-303).  `self.tree = PoseidonTree::default(d)?` leaves the old tree
  // in place when the constructor fails, so the new tree is built beside the old one and swapped in -- together with
  // next_index, leaf_set and metadata -- only once init has succeeded (bad depth, hipMalloc failure: nothing changes).
  void new_tree(size_t depth) {
    if (depth >= 64) throw Error("Merkle tree error: Tree depth exceeds maximum allowed (must be < 64)");  // InvalidDepth
    uint8_t zero[32] = {0};
    bool dropped_old = false;
    // The replacement is built beside the old tree so that a failure changes nothing -- unless the device cannot hold
    // both dense trees (depth 28: 16 GiB each, next to up to 228 GiB of comb tables): then the old tree goes first and
    // the reset is no longer atomic (a failure after this point leaves an empty depth-0 tree behind, reported by the
    // error).  Depths above 30 take the sparse tree (tree_any.h): nothing proportional to 2^depth is allocated.
    if (depth <= (size_t)TreeAny::MAX_DENSE_DEPTH) {
      size_t free_b = 0, total_b = 0;
      const size_t need = ((size_t)64 << depth) + ((size_t)1 << 26);   // 2^(depth+1) nodes of 32 B + slack
      if (tree.depth > 0 && !tree.sparse && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < need) {
        TreeAny empty;
        tree = std::move(empty);
        dropped_old = true;
      }
    }
    TreeAny fresh;
    try {
      fresh.init((int)depth, zero);
    } catch (const std::exception& e) {
      if (!dropped_old) throw;
      // the non-atomic case: say what state the object is in now
      leaf_set.clear();
      next_index = 0;
      metadata.clear();
      throw Error(std::string(e.what()) + " (the previous tree had to be released first to make room and is lost: the "
                  "object now holds an empty depth-0 tree; call ffi_set_tree again with a depth that fits)");
    }
    // cached_leaves_indices: one byte per leaf for the dense tree; the sparse tree keeps none (write-only bookkeeping)
    std::vector<uint8_t> fresh_set(fresh.sparse ? 0 : (size_t)1 << depth, 0);
    tree = std::move(fresh);
    leaf_set.swap(fresh_set);
    next_index = 0;
    metadata.clear();
  }
  // ffi_set_tree / ffi_init_tree_with_leaves: the stored tree is flushed and replaced by a default (temporary) one
  void replace_with_default_tree(size_t depth) {
    flush();
    new_tree(depth);  // throws before anything is dropped
    store.clear();
  }
  // PmTree::new (pm_tree_adapter.rs:191-239): depth check against the config, load the stored tree when there is
  // one (its depth must match), else start empty; cached_leaves_indices rebuilt from the leaves below next_index
  void open_tree(size_t depth, const TreeConfig& cfg) {
    if (cfg.tree_depth >= 0 && (size_t)cfg.tree_depth != depth)
      throw Error("Merkle tree error: Tree depth exceeds maximum allowed (must be < 64)");  // InvalidDepth
    store.clear();
    // a snapshot holds the dense prefix of leaves below next_index; the sparse tree of depths 31 .. 63 has no such prefix
    // (one leaf at index 2^35 would make it terabytes), so persistence stops at the dense tree
    if (cfg.persistent() && depth > (size_t)TreeAny::MAX_DENSE_DEPTH)
      throw Error("Configuration error: a persistent tree (temporary = false) is supported up to depth " +
                  std::to_string(TreeAny::MAX_DENSE_DEPTH) + "; depth " + std::to_string(depth) +
                  " takes the sparse in-memory tree, which has no snapshot form");
    new_tree(depth);
    if (!cfg.persistent()) return;
    if (mkdir(cfg.path.c_str(), 0777) != 0 && errno != EEXIST)
      throw Error("Merkle tree error: cannot create " + cfg.path + ": " + strerror(errno));
    std::string file = cfg.path + "/rlnamd_tree.bin";
    FILE* f = fopen(file.c_str(), "rb");
    if (f) {
      struct { char magic[8]; uint64_t depth, next, meta_len; } h;
      std::vector<uint8_t> meta, leaves;
      bool ok = fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, "RLNAMDT1", 8) && h.depth < 64 &&
                h.next <= ((uint64_t)1 << h.depth) && h.meta_len <= (1u << 30);
      if (ok) {
        meta.resize(h.meta_len);
        leaves.resize(h.next * 32);
        ok = (meta.empty() || fread(meta.data(), meta.size(), 1, f) == 1) &&
             (leaves.empty() || fread(leaves.data(), leaves.size(), 1, f) == 1);
      }
      fclose(f);
      if (!ok) throw Error("Merkle tree error: " + file + " is not a tree snapshot of this library");
      if (h.depth != depth) throw Error("Merkle tree error: Tree depth exceeds maximum allowed (must be < 64)");
      if (h.next) {
        tree.set_range_host(0, leaves.data(), h.next);
        static const uint8_t zero[32] = {0};
        for (size_t i = 0; i < h.next && !leaf_set.empty(); i++) leaf_set[i] = memcmp(leaves.data() + 32 * i, zero, 32) != 0;
      }
      next_index = h.next;
      metadata = meta;
    }
    store = file;
  }
  void flush() {  // written to a sibling file, then renamed over the snapshot
    if (store.empty()) return;
    std::vector<uint8_t> leaves(next_index * 32);
    if (next_index) tree.get_leaves_host(0, next_index, leaves.data());
    struct { char magic[8]; uint64_t depth, next, meta_len; } h;
    memcpy(h.magic, "RLNAMDT1", 8);
    h.depth = (uint64_t)tree.depth;
    h.next = next_index;
    h.meta_len = metadata.size();
    std::string tmp = store + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) throw Error("Merkle tree error: cannot write " + tmp + ": " + strerror(errno));
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && (metadata.empty() || fwrite(metadata.data(), metadata.size(), 1, f) == 1) &&
              (leaves.empty() || fwrite(leaves.data(), leaves.size(), 1, f) == 1);
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp.c_str(), store.c_str()) != 0)
      throw Error("Merkle tree error: cannot write " + store + ": " + strerror(errno));
  }
  void set_range(size_t start, const std::vector<CFr>& leaves) {  // full_merkle_tree.rs:197-223
    if (start + leaves.size() > tree.capacity() || start + leaves.size() < start)
      throw Error("set_range got too many leaves");
    if (leaves.empty()) return;
    tree.set_range_host(start, (const uint8_t*)leaves.data(), leaves.size());
    for (size_t i = 0; i < leaves.size() && !leaf_set.empty(); i++) leaf_set[start + i] = 1;
    next_index = std::max(next_index, start + leaves.size());
  }
  // one leaf (:141-147).  The write is recorded and hashed by the first reader, together with every other write made
  // until then (TreeAny::set_leaf): ONE pass over the union of the dirty paths instead of `depth` dependent hashes per call
  void set(size_t index, const CFr& leaf) {
    if (index >= tree.capacity()) throw Error("set_range got too many leaves");
    tree.set_leaf(index, leaf.le);
    if (!leaf_set.empty()) leaf_set[index] = 1;
    next_index = std::max(next_index, index + 1);
  }
  CFr get(size_t index) {
    if (index >= tree.capacity()) throw Error("Leaf index out of bounds");
    CFr r;
    tree.get_node_host(tree.capacity() - 1 + index, r.le);
    return r;
  }
  void del(size_t index) {  // :271-285; out-of-capacity indices are an error (rln/tests/ffi.rs:1087-1089)
    if (index >= tree.capacity()) throw Error("Leaf index out of bounds");
    if (index < next_index) {
      set(index, cfr_from_u64(0));
      if (!leaf_set.empty()) leaf_set[index] = 0;
    }
  }
  // override_range with the default (pmtree-ft) build's behaviour: empty `indices` allowed
  // (pm_tree_adapter.rs:320-356, validation override_range_validation.rs:20-65)
  void override_range(size_t start, const std::vector<CFr>& leaves, std::vector<size_t> indices) {
    const size_t cap = tree.capacity();
    for (size_t i : indices)
      if (i >= cap) throw Error("Invalid indices");
    std::sort(indices.begin(), indices.end());
    indices.erase(std::unique(indices.begin(), indices.end()), indices.end());
    size_t end = 0;
    bool have_end = !leaves.empty();
    if (have_end) {
      end = start + leaves.size();
      if (end < start || end > cap) throw Error("set_range got too many leaves");
    }
    if (!indices.empty() && have_end && (indices[0] > start || indices[0] >= end)) throw Error("Invalid indices");
    if (leaves.empty() && indices.empty()) throw Error("Leaf index out of bounds");
    if (indices.empty()) {
      if (leaves.size() == 1) set(start, leaves[0]); else set_range(start, leaves);
      return;
    }
    if (leaves.empty()) {
      if (indices.size() == 1) {
        del(indices[0]);
        return;
      }
      // remove_indices (pm_tree_adapter.rs:417-435): the whole span [first, last] is reset
      size_t s = indices.front(), e = indices.back() + 1;
      set_range(s, std::vector<CFr>(e - s, cfr_from_u64(0)));
      for (size_t i = s; i < e && !leaf_set.empty(); i++) leaf_set[i] = 0;
      return;
    }
    // remove_indices_and_set_leaves (pm_tree_adapter.rs:437-480); the merged buffer is written at `start`
    // (SURVEY Appendix C.1), exactly as the reference does
    size_t min_index = indices[0];
    std::vector<CFr> vals(end - min_index, cfr_from_u64(0));
    for (size_t i = min_index; i < start; i++)
      if (!std::binary_search(indices.begin(), indices.end(), i)) vals[i - min_index] = get(i);
    for (size_t i = 0; i < leaves.size(); i++) vals[start - min_index + i] = leaves[i];
    if (start + vals.size() > cap) throw Error("set_range got too many leaves");
    set_range(start, vals);
    for (size_t i : indices)
      if (!leaf_set.empty()) leaf_set[i] = 0;
    for (size_t i = start; i < end - min_index && i < cap && !leaf_set.empty(); i++) leaf_set[i] = 1;
  }
};

namespace {


// inputs_for_witness_calculation + populate_inputs (witness.rs:832-881, iden3calc.rs:122-181)
void fill_inputs(const Prover& P, const FFI_RLNWitnessInput& w, uint8_t* buf) {
  const Graph& g = P.graph();
  memset(buf, 0, (size_t)g.inputs_size * 32);
  buf[0] = 1;
  auto put = [&](const char* name, const CFr* vals, size_t count) {
    auto it = g.input_mapping.find(name);
    if (it == g.input_mapping.end()) throw Error(std::string("Error calculating witness: missing input ") + name);
    if (it->second.second != count)
      throw Error(std::string("Error calculating witness: invalid input length for ") + name + ": expected " +
                  std::to_string(it->second.second) + ", got " + std::to_string(count));
    memcpy(buf + (size_t)it->second.first * 32, vals, count * 32);
  };
  put("identitySecret", &w.identity_secret, 1);
  put("userMessageLimit", &w.user_message_limit, 1);
  if (w.multi) {
    put("messageId", w.message_ids.data(), w.message_ids.size());
    std::vector<CFr> sel;
    for (uint8_t b : w.selector_used) sel.push_back(cfr_from_u64(b ? 1 : 0));
    put("selectorUsed", sel.data(), sel.size());
  } else {
    put("messageId", &w.message_id, 1);
  }
  put("pathElements", w.path_elements.data(), w.path_elements.size());
  std::vector<CFr> idx;
  for (uint8_t b : w.identity_path_index) idx.push_back(cfr_from_u64(b));
  put("identityPathIndex", idx.data(), idx.size());
  put("x", &w.x, 1);
  put("externalNullifier", &w.external_nullifier, 1);
}

void check_against_graph(const Prover& P, const FFI_RLNWitnessInput& w) {  // proof.rs:644-700
  size_t d = P.graph().tree_depth;
  if (w.path_elements.size() != d)
    throw Error("The field path_elements has length " + std::to_string(w.path_elements.size()) +
                ", but the field tree_depth has length " + std::to_string(d));
  if (w.identity_path_index.size() != d)
    throw Error("The field identity_path_index has length " + std::to_string(w.identity_path_index.size()) +
                ", but the field tree_depth has length " + std::to_string(d));
  const bool graph_multi = P.graph().input_mapping.count("selectorUsed") != 0;  // mode.rs:145-157
  if (w.multi != graph_multi)
    throw Error(std::string("Witness message mode ") + (w.multi ? "MultiV1" : "SingleV1") +
                " does not match graph mode " + (graph_multi ? "MultiV1" : "SingleV1"));
  if (w.multi) {
    size_t mo = P.graph().max_out;
    if (w.message_ids.size() != mo)
      throw Error("The field message_ids has length " + std::to_string(w.message_ids.size()) +
                  ", but the field max_out has length " + std::to_string(mo));
    if (w.selector_used.size() != mo)
      throw Error("The field selector_used has length " + std::to_string(w.selector_used.size()) +
                  ", but the field max_out has length " + std::to_string(mo));
  }
}

void fill_outputs(const ProofOut& po, FFI_RLNProof* pr) {
  memcpy(pr->proof, po.compressed, 128);
  memcpy(pr->values.y.le, po.values[0], 32);
  memcpy(pr->values.root.le, po.values[1], 32);
  memcpy(pr->values.nullifier.le, po.values[2], 32);
  memcpy(pr->values.x.le, po.values[3], 32);
  memcpy(pr->values.external_nullifier.le, po.values[4], 32);
}
// multi message-id public signals [ys | root | nullifiers | x | ext | selector_used] -> proof values
void values_from_public(const uint8_t* pub, size_t mo, FFI_RLNProofValues* v) {
  auto at = [&](size_t k) {
    CFr c;
    memcpy(c.le, pub + 32 * k, 32);
    return c;
  };
  v->multi = true;
  v->ys.clear();
  v->nullifiers.clear();
  v->selector_used.clear();
  for (size_t k = 0; k < mo; k++) v->ys.push_back(at(k));
  v->root = at(mo);
  for (size_t k = 0; k < mo; k++) v->nullifiers.push_back(at(mo + 1 + k));
  v->x = at(2 * mo + 1);
  v->external_nullifier = at(2 * mo + 2);
  for (size_t k = 0; k < mo; k++) v->selector_used.push_back(cfr_is_zero(at(2 * mo + 3 + k)) ? 0 : 1);
}

// Every resident-input run below ends with the prover's copies of the inputs and of the witness overwritten (the
// reference zeroises the witness calculator's inputs buffer, circuit/iden3calc.rs:45-56), on the error paths as well.
struct WipeResident {
  Prover& P;
  ~WipeResident() {
    try {
      P.wipe(0);
    } catch (...) {
    }
  }
};

// the member memo's key (FFI_RLN::auto_partial): identity secret | limit | path elements | path index
void memo_key(const FFI_RLNWitnessInput& w, std::vector<uint8_t>& key) {
  key.clear();
  key.reserve(64 + 33 * w.path_elements.size());
  key.insert(key.end(), w.identity_secret.le, w.identity_secret.le + 32);
  key.insert(key.end(), w.user_message_limit.le, w.user_message_limit.le + 32);
  for (const CFr& e : w.path_elements) key.insert(key.end(), e.le, e.le + 32);
  key.insert(key.end(), w.identity_path_index.begin(), w.identity_path_index.end());
}

// generate_rln_proof for a slice of witnesses (public.rs:624-631)
void prove_many(FFI_RLN& rln, FFI_RLNWitnessInput* const* ws, size_t n, const CFr* rs, FFI_RLNProof** out) {
  std::lock_guard<std::mutex> guard(*rln.prove_mu);
  Prover& P = *rln.prover;
  const size_t ni = P.inputs_per_proof();
  for (size_t i = 0; i < n; i++) check_against_graph(P, *ws[i]);
  const bool multi = n > 0 && ws[0]->multi;
  std::vector<FFI_RLNProof*> made;
  const size_t npub = P.num_public(), mo = P.graph().max_out;
  auto pack = [&](size_t off, size_t m, std::vector<uint8_t>& inputs, std::vector<uint8_t>& rsb) {
    // one allocation for the whole call: a vector that grew would hand its earlier buffer -- identity secrets in it --
    // back to the heap uncleared (ZeroOnExit sees the final buffer only)
    const size_t most = std::max(m, std::min(n, P.capacity()));
    if (inputs.capacity() < most * ni * 32) { secure_zero(inputs.data(), inputs.size()); inputs.clear(); inputs.reserve(most * ni * 32); }
    if (rsb.capacity() < most * 64) { secure_zero(rsb.data(), rsb.size()); rsb.clear(); rsb.reserve(most * 64); }
    inputs.assign(m * ni * 32, 0);
    rsb.resize(m * 64);
    for (size_t i = 0; i < m; i++) {
      fill_inputs(P, *ws[off + i], inputs.data() + i * ni * 32);
      CFr r = rs ? rs[2 * (off + i)] : random_fr();      // proof.rs:743-745
      CFr s = rs ? rs[2 * (off + i) + 1] : random_fr();
      memcpy(rsb.data() + i * 64, r.le, 32);
      memcpy(rsb.data() + i * 64 + 32, s.le, 32);
    }
  };
  try {
    if (rln.auto_partial) rln.memo_adopt_pending();
    if (rln.auto_partial && n == 1 && !multi) {
      // the member memo (FFI_RLN::auto_partial): a finish through the cone when this member's partial proof is at hand,
      // else a full proof with the member's partial proof enqueued behind it
      const FFI_RLNWitnessInput& w = *ws[0];
      std::vector<uint8_t> key;
      memo_key(w, key);
      ZeroOnExit zk{key};
      std::vector<uint8_t> inputs, rsb;
      ZeroOnExit z1{inputs}, z2{rsb};
      pack(0, 1, inputs, rsb);
      FFI_RLN::Memo* hit = nullptr;
      for (FFI_RLN::Memo& m : rln.memo)
        if (m.key == key) hit = &m;
      uint8_t proof[128], values[160];
      uint32_t err = 0;
      uint64_t ticket;
      if (hit) {
        hit->stamp = ++rln.memo_clock;
        rln.memo_hits++;
        ticket = P.submit_finish(1, inputs.data(), rsb.data(), hit->coords, &hit->handle);
      } else {
        rln.memo_misses++;
        ticket = P.submit(1, inputs.data(), rsb.data());
      }
      P.collect(ticket, 1, proof, values, &err);
      if (err) throw Error("Error calculating witness: graph evaluation failed (code " + std::to_string(err) + ")");
      if (!hit) {   // this member's partial proof, behind the caller's back: adopted by the next proving call
        FFI_RLNWitnessInput pw = w;
        pw.message_id = cfr_from_u64(0);
        pw.x = cfr_from_u64(0);
        pw.external_nullifier = cfr_from_u64(0);
        std::vector<uint8_t> pin(ni * 32), zero_rs(64, 0);
        ZeroOnExit z3{pin};
        fill_inputs(P, pw, pin.data());
        rln.pending_ticket = P.submit(1, pin.data(), zero_rs.data(), PROVE_PARTIAL);
        rln.pending_key = key;
      }
      std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
      memcpy(pr->proof, proof, 128);
      memcpy(pr->values.y.le, values, 32);
      memcpy(pr->values.root.le, values + 32, 32);
      memcpy(pr->values.nullifier.le, values + 64, 32);
      memcpy(pr->values.x.le, values + 96, 32);
      memcpy(pr->values.external_nullifier.le, values + 128, 32);
      made.push_back(pr.release());
    } else if (n <= P.capacity()) {
      // one batch: the latency path (a single proof walks with lanes = chunks, see Prover::run_async)
      // (streamed since round 6: submit + collect is the resident upload / run / download without its two pipeline
      // drains -- 0.1 ms of a single proof; the batch's inputs and witness are wiped behind the copy-out)
      std::vector<uint8_t> inputs, rsb, proofs(n * 128), values(n * 160), pub;
      ZeroOnExit z1{inputs}, z2{rsb};
      std::vector<uint32_t> errs(n);
      pack(0, n, inputs, rsb);
      const uint64_t ticket = P.submit(n, inputs.data(), rsb.data());
      try {
        P.collect(ticket, n, proofs.data(), values.data(), errs.data(), nullptr, nullptr, !multi);
        if (multi) P.collect_public(ticket, n, &pub);  // ys, root, nullifiers, x, ext, selectors (witness.rs:777-802)
      } catch (...) {
        if (multi) P.wipe(ticket);
        throw;
      }
      if (multi) P.wipe(ticket);
      for (size_t i = 0; i < n; i++) {
        if (errs[i]) throw Error("Error calculating witness: graph evaluation failed (code " + std::to_string(errs[i]) + ")");
        std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
        memcpy(pr->proof, proofs.data() + 128 * i, 128);
        const uint8_t* v = values.data() + 160 * i;
        memcpy(pr->values.y.le, v, 32);
        memcpy(pr->values.root.le, v + 32, 32);
        memcpy(pr->values.nullifier.le, v + 64, 32);
        memcpy(pr->values.x.le, v + 96, 32);
        memcpy(pr->values.external_nullifier.le, v + 128, 32);
        if (multi) values_from_public(pub.data() + i * npub * 32, mo, &pr->values);
        made.push_back(pr.release());
      }
    } else if (rln.pool && !multi) {
      // several devices behind the object: contiguous index shards, one per replica, each streamed (rlnamd_pool_prove)
      std::vector<uint8_t> inputs, rsb, proofs(n * 128), values(n * 160);
      ZeroOnExit z1{inputs}, z2{rsb};
      std::vector<uint32_t> errs(n);
      pack(0, n, inputs, rsb);
      if (rlnamd_pool_prove(rln.pool, n, inputs.data(), rsb.data(), proofs.data(), values.data(), errs.data()) != RLNAMD_OK)
        throw Error(rlnamd_last_error());
      for (size_t i = 0; i < n; i++) {
        if (errs[i]) throw Error("Error calculating witness: graph evaluation failed (code " + std::to_string(errs[i]) + ")");
        std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
        memcpy(pr->proof, proofs.data() + 128 * i, 128);
        const uint8_t* v = values.data() + 160 * i;
        memcpy(pr->values.y.le, v, 32);
        memcpy(pr->values.root.le, v + 32, 32);
        memcpy(pr->values.nullifier.le, v + 64, 32);
        memcpy(pr->values.x.le, v + 96, 32);
        memcpy(pr->values.external_nullifier.le, v + 128, 32);
        made.push_back(pr.release());
      }
    } else {
      // more proofs than one workspace holds: chunks of capacity() streamed through submit / collect, every workspace
      // slot in flight -- the host packs chunk k + 1 while the device proves chunk k, nothing drains in between
      struct Pending { uint64_t ticket; size_t m; };
      std::deque<Pending> q;
      std::vector<uint8_t> inputs, rsb, proofs, values, pub;
      ZeroOnExit z1{inputs}, z2{rsb};
      std::vector<uint32_t> errs;
      auto take = [&]() {
        Pending f = q.front();
        q.pop_front();
        proofs.resize(f.m * 128);
        values.resize(f.m * 160);
        errs.resize(f.m);
        P.collect(f.ticket, f.m, proofs.data(), values.data(), errs.data(), nullptr, nullptr, !multi);
        if (multi) {
          P.collect_public(f.ticket, f.m, &pub);
          P.wipe(f.ticket);
        }
        for (size_t i = 0; i < f.m; i++) {
          if (errs[i]) throw Error("Error calculating witness: graph evaluation failed (code " + std::to_string(errs[i]) + ")");
          std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
          memcpy(pr->proof, proofs.data() + 128 * i, 128);
          const uint8_t* v = values.data() + 160 * i;
          memcpy(pr->values.y.le, v, 32);
          memcpy(pr->values.root.le, v + 32, 32);
          memcpy(pr->values.nullifier.le, v + 64, 32);
          memcpy(pr->values.x.le, v + 96, 32);
          memcpy(pr->values.external_nullifier.le, v + 128, 32);
          if (multi) values_from_public(pub.data() + i * npub * 32, mo, &pr->values);
          made.push_back(pr.release());
        }
      };
      try {
        for (size_t off = 0; off < n; off += P.capacity()) {
          size_t m = std::min(n - off, P.capacity());
          pack(off, m, inputs, rsb);
          if ((int)q.size() == P.slots()) take();
          q.push_back({P.submit(m, inputs.data(), rsb.data()), m});
        }
        while (!q.empty()) take();
      } catch (...) {
        // nothing of this call stays in flight behind the error, and the batches still queued (never collected, so never
        // wiped) do not keep their staged inputs and witnesses either
        try {
          P.sync();
          for (const Pending& f : q) P.wipe(f.ticket);
          P.sync();
        } catch (...) {
        }
        throw;
      }
    }
  } catch (...) {
    for (auto* p : made) delete p;
    throw;
  }
  for (size_t i = 0; i < n; i++) out[i] = made[i];
}

// the gathered calls of FFI_RLN::Gather as one batch; never throws: every request leaves with a proof or its error text
void run_gathered(FFI_RLN& rln, const std::vector<FFI_RLN::GatherReq*>& batch) {
  const size_t n = batch.size();
  std::vector<FFI_RLNWitnessInput*> ws(n);
  std::vector<CFr> rs(2 * n);
  struct WipeRs {
    std::vector<CFr>& v;
    ~WipeRs() { secure_zero(v.data(), v.size() * sizeof(CFr)); }
  } wipe_rs{rs};
  for (size_t i = 0; i < n; i++) {
    ws[i] = batch[i]->w;
    rs[2 * i] = batch[i]->has_rs ? batch[i]->rs[0] : random_fr();      // proof.rs:743-745
    rs[2 * i + 1] = batch[i]->has_rs ? batch[i]->rs[1] : random_fr();
  }
  // every request brought its packed inputs and its hints (single message-id): one submit_hinted, nothing hashed here
  bool packed = !rln.auto_partial && n <= 64;
  for (size_t i = 0; i < n && packed; i++) packed = !batch[i]->hints.empty() && !batch[i]->inputs.empty() && !ws[i]->multi;
  if (packed) {
    try {
      std::lock_guard<std::mutex> guard(*rln.prove_mu);
      Prover& P = *rln.prover;
      const size_t ni = P.inputs_per_proof(), hw = P.hint_words();
      std::vector<uint8_t> inputs(n * ni * 32), rsb(n * 64), proofs(n * 128), values(n * 160);
      ZeroOnExit z1{inputs}, z2{rsb};
      std::vector<uint32_t> hints(n * hw), errs(n);
      struct WipeHints {
        std::vector<uint32_t>& v;
        ~WipeHints() { secure_zero(v.data(), v.size() * 4); }
      } wipe_hints{hints};
      for (size_t i = 0; i < n; i++) {
        if (batch[i]->inputs.size() != ni * 32 || batch[i]->hints.size() != hw) throw Error("gathered request of another shape");
        memcpy(inputs.data() + i * ni * 32, batch[i]->inputs.data(), ni * 32);
        memcpy(hints.data() + i * hw, batch[i]->hints.data(), hw * 4);
        memcpy(rsb.data() + i * 64, rs[2 * i].le, 32);
        memcpy(rsb.data() + i * 64 + 32, rs[2 * i + 1].le, 32);
      }
      const uint64_t ticket = P.submit_hinted(n, inputs.data(), rsb.data(), hints.data());
      P.collect(ticket, n, proofs.data(), values.data(), errs.data());
      for (size_t i = 0; i < n; i++) {
        if (errs[i]) {
          batch[i]->err = "Error calculating witness: graph evaluation failed (code " + std::to_string(errs[i]) + ")";
          continue;
        }
        std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
        memcpy(pr->proof, proofs.data() + 128 * i, 128);
        const uint8_t* v = values.data() + 160 * i;
        memcpy(pr->values.y.le, v, 32);
        memcpy(pr->values.root.le, v + 32, 32);
        memcpy(pr->values.nullifier.le, v + 64, 32);
        memcpy(pr->values.x.le, v + 96, 32);
        memcpy(pr->values.external_nullifier.le, v + 128, 32);
        batch[i]->out = pr.release();
      }
      return;
    } catch (const std::exception&) {
      // the batch as a whole did not go through: below, the way every batch goes that brings no hints
      for (size_t i = 0; i < n; i++)
        if (batch[i]->out) {
          delete batch[i]->out;
          batch[i]->out = nullptr;
        }
      for (size_t i = 0; i < n; i++) batch[i]->err.clear();
    }
  }
  auto alone = [&](size_t i) {
    try {
      FFI_RLNProof* o = nullptr;
      prove_many(rln, &ws[i], 1, &rs[2 * i], &o);
      batch[i]->out = o;
    } catch (const std::exception& e) {
      batch[i]->err = e.what();
      if (batch[i]->err.empty()) batch[i]->err = "Error producing proof";
    }
  };
  if (n == 1) return alone(0);
  if (rln.auto_partial) {
    // An object with the member memo: the requests of remembered members go out as ONE batch of finishes through the
    // cone; the others one by one, the way a lone call goes (a full proof, the member's partial proof behind it).
    std::vector<size_t> hit, rest;
    try {
      std::lock_guard<std::mutex> guard(*rln.prove_mu);
      rln.memo_adopt_pending();
      Prover& P = *rln.prover;
      const size_t ni = P.inputs_per_proof();
      std::vector<FFI_RLN::Memo*> hm;
      std::vector<uint8_t> key;
      ZeroOnExit zk{key};
      for (size_t i = 0; i < n; i++) {
        FFI_RLN::Memo* found = nullptr;
        if (!ws[i]->multi) {
          memo_key(*ws[i], key);
          for (FFI_RLN::Memo& m : rln.memo)
            if (m.key == key) found = &m;
        }
        if (found) {
          hit.push_back(i);
          hm.push_back(found);
        } else {
          rest.push_back(i);
        }
      }
      if (hit.size() >= 2) {
        const size_t nh = hit.size();
        std::vector<uint8_t> inputs(nh * ni * 32), rsb(nh * 64), coords(nh * 320), proofs(nh * 128), values(nh * 160);
        ZeroOnExit z1{inputs}, z2{rsb};
        std::vector<uint64_t> handles(nh);
        std::vector<uint32_t> errs(nh);
        for (size_t k = 0; k < nh; k++) {
          check_against_graph(P, *ws[hit[k]]);
          fill_inputs(P, *ws[hit[k]], inputs.data() + k * ni * 32);
          memcpy(rsb.data() + k * 64, rs[2 * hit[k]].le, 32);
          memcpy(rsb.data() + k * 64 + 32, rs[2 * hit[k] + 1].le, 32);
          memcpy(coords.data() + k * 320, hm[k]->coords, 320);
          handles[k] = hm[k]->handle;
        }
        const uint64_t ticket = P.submit_finish(nh, inputs.data(), rsb.data(), coords.data(), handles.data());
        P.collect(ticket, nh, proofs.data(), values.data(), errs.data());
        for (size_t k = 0; k < nh; k++) {
          hm[k]->stamp = ++rln.memo_clock;
          rln.memo_hits++;
          if (errs[k]) {
            batch[hit[k]]->err = "Error calculating witness: graph evaluation failed (code " + std::to_string(errs[k]) + ")";
            continue;
          }
          std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
          memcpy(pr->proof, proofs.data() + 128 * k, 128);
          const uint8_t* v = values.data() + 160 * k;
          memcpy(pr->values.y.le, v, 32);
          memcpy(pr->values.root.le, v + 32, 32);
          memcpy(pr->values.nullifier.le, v + 64, 32);
          memcpy(pr->values.x.le, v + 96, 32);
          memcpy(pr->values.external_nullifier.le, v + 128, 32);
          batch[hit[k]]->out = pr.release();
        }
      } else {
        rest.insert(rest.end(), hit.begin(), hit.end());
        hit.clear();
      }
    } catch (const std::exception&) {   // the finish batch as a whole did not go through: whoever has no result yet goes alone
      for (size_t i : hit)
        if (!batch[i]->out && batch[i]->err.empty()) rest.push_back(i);
    }
    for (size_t i : rest) alone(i);
    return;
  }
  try {
    std::vector<FFI_RLNProof*> outs(n, nullptr);
    prove_many(rln, ws.data(), n, rs.data(), outs.data());
    for (size_t i = 0; i < n; i++) batch[i]->out = outs[i];
  } catch (const std::exception&) {
    // somebody's witness does not evaluate (or does not fit the circuit): the reference would fail that call only, so
    // every request is proved on its own and keeps its own error
    for (size_t i = 0; i < n; i++) alone(i);
  }
}

// generate_rln_proof for ONE witness from any thread (FFI_RLN::Gather); rs: (r, s) or null (sampled)
FFI_RLNProof* prove_one(FFI_RLN& rln, FFI_RLNWitnessInput* w, const CFr* rs) {
  FFI_RLN::Gather& G = rln.gather;
  if (G.most == 0) {
    FFI_RLNProof* out = nullptr;
    prove_many(rln, &w, 1, rs, &out);
    return out;
  }
  FFI_RLN::GatherReq me;
  me.w = w;
  if (rs) {
    me.rs[0] = rs[0];
    me.rs[1] = rs[1];
    me.has_rs = true;
  }
  struct WipeReq {
    FFI_RLN::GatherReq& r;
    ~WipeReq() { secure_zero(r.rs, sizeof r.rs); }
  } wipe_me{me};
  // this thread's share of the batch's host work, before it queues: the packed inputs and the hints of its proof
  // (Prover::hints_for; graph and slots are read-only: no lock).  A witness that does not fit leaves both empty and gets
  // its error from the proving path.
  if (!rln.auto_partial && !w->multi) {
    try {
      const Prover& P = *rln.prover;
      const size_t hw = P.hint_words();
      if (hw) {
        check_against_graph(P, *w);
        me.inputs.resize(P.inputs_per_proof() * 32);
        fill_inputs(P, *w, me.inputs.data());
        me.hints.resize(hw);
        P.hints_for(me.inputs.data(), me.hints.data());
      }
    } catch (const std::exception&) {
      secure_zero(me.inputs.data(), me.inputs.size());
      me.inputs.clear();
      me.hints.clear();
    }
  }
  G.pass(me, [&](const std::vector<FFI_RLN::GatherReq*>& batch) { run_gathered(rln, batch); });
  if (!me.err.empty()) throw Error(me.err);
  return me.out;
}

// generate_rln_proof_with_witness (public.rs:643-658): the Groth16 proof is made from the supplied witness;
// the proof values come from the witness input (single: the values kernel over the inputs; multi: the public
// signals, identical for any witness that satisfies the circuit).
FFI_RLNProof* prove_with_witness(FFI_RLN& rln, const Vec_String_t* calc, const FFI_RLNWitnessInput& w) {
  std::lock_guard<std::mutex> guard(*rln.prove_mu);
  if (rln.auto_partial) rln.memo_adopt_pending();
  Prover& P = *rln.prover;
  std::vector<uint8_t> given;
  given.reserve((calc ? calc->len : 0) * 32);
  for (size_t i = 0; calc && i < calc->len; i++) {
    CFr v = parse_bigint_fr(calc->ptr[i].ptr, calc->ptr[i].len);
    given.insert(given.end(), v.le, v.le + 32);
  }
  check_against_graph(P, w);
  if (given.size() / 32 != P.num_signals())  // SynthesisError::MalformedVerifyingKey in ark-groth16
    throw Error("Error producing proof: malformed verifying key (witness has " + std::to_string(given.size() / 32) +
                " entries, the circuit " + std::to_string(P.num_signals()) + ")");
  std::vector<uint8_t> inputs(P.inputs_per_proof() * 32), rs(64);
  ZeroOnExit z1{inputs}, z2{rs}, z3{given};
  WipeResident wr{P};
  fill_inputs(P, w, inputs.data());
  CFr r = random_fr(), sb = random_fr();
  memcpy(rs.data(), r.le, 32);
  memcpy(rs.data() + 32, sb.le, 32);
  P.upload(1, inputs.data(), rs.data());
  P.upload_witness(1, given.data());
  P.run(1, PROVE_FULL);
  ProofOut po;
  P.download(1, &po);
  std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
  fill_outputs(po, pr.get());
  if (w.multi) {
    std::vector<uint8_t> pub;
    P.fetch_public(1, &pub);
    values_from_public(pub.data(), P.graph().max_out, &pr->values);
  }
  return pr.release();
}

// generate_partial_zk_proof (proof.rs:783-803)
FFI_RLNPartialProof* prove_partial(FFI_RLN& rln, const FFI_RLNPartialWitnessInput& pw) {
  std::lock_guard<std::mutex> guard(*rln.prove_mu);
  if (rln.auto_partial) rln.memo_adopt_pending();
  Prover& P = *rln.prover;
  size_t d = P.graph().tree_depth;
  if (pw.path_elements.size() != d)
    throw Error("The field path_elements has length " + std::to_string(pw.path_elements.size()) +
                ", but the field tree_depth has length " + std::to_string(d));
  FFI_RLNWitnessInput w;
  w.identity_secret = pw.identity_secret;
  w.user_message_limit = pw.user_message_limit;
  w.message_id = cfr_from_u64(0);
  w.path_elements = pw.path_elements;
  w.identity_path_index = pw.identity_path_index;
  w.x = cfr_from_u64(0);
  w.external_nullifier = cfr_from_u64(0);
  if (P.graph().input_mapping.count("selectorUsed")) {   // the multi-message-id circuit: max_out open message slots (witness.rs:887-937)
    w.multi = true;
    w.message_ids.assign(P.graph().max_out, cfr_from_u64(0));
    w.selector_used.assign(P.graph().max_out, 0);
  }
  std::vector<uint8_t> inputs(P.inputs_per_proof() * 32), rs(64, 0);
  ZeroOnExit z1{inputs};
  fill_inputs(P, w, inputs.data());
  std::unique_ptr<FFI_RLNPartialProof> pp(new FFI_RLNPartialProof);
  // streamed, and the values the partial witness fixes stay on the device under a handle (round 6): finishing THIS object on
  // this prover interprets only the cone of the graph that depends on the message (rln_amd.h)
  uint32_t err = 0;
  const uint64_t ticket = P.submit(1, inputs.data(), rs.data(), PROVE_PARTIAL);
  P.collect_partial_cached(ticket, 1, pp->coords, &pp->handle, &err);
  if (err) throw Error("Error calculating witness: graph evaluation failed (code " + std::to_string(err) + ")");
  if (pp->handle)
    pp->release = [wp = std::weak_ptr<Prover>(rln.prover), mu = rln.prove_mu](uint64_t h) {
      if (auto p = wp.lock()) {
        std::lock_guard<std::mutex> g(*mu);
        p->release_partial(&h, 1);
      }
    };
  const std::vector<uint8_t>& known = P.known_mask();
  pp->mask.assign(known.begin() + 1, known.end());
  return pp.release();
}

// finish_zk_proof_with_rs (proof.rs:821-849) + proof values
FFI_RLNProof* finish_proof(FFI_RLN& rln, const FFI_RLNPartialProof& pp, const FFI_RLNWitnessInput& w, const CFr& r,
                           const CFr& s) {
  std::lock_guard<std::mutex> guard(*rln.prove_mu);
  if (rln.auto_partial) rln.memo_adopt_pending();
  Prover& P = *rln.prover;
  check_against_graph(P, w);
  const std::vector<uint8_t>& known = P.known_mask();
  if (pp.mask.size() + 1 != known.size() || !std::equal(pp.mask.begin(), pp.mask.end(), known.begin() + 1))
    throw Error("Error producing proof: the partial proof's mask does not match this circuit (malformed verifying key)");
  std::vector<uint8_t> inputs(P.inputs_per_proof() * 32), rs(64);
  ZeroOnExit z1{inputs}, z2{rs};
  fill_inputs(P, w, inputs.data());
  memcpy(rs.data(), r.le, 32);
  memcpy(rs.data() + 32, s.le, 32);
  {   // streamed; with a live handle the partial run's values are on this device and only the cone is interpreted
    ProofOut po;
    const uint64_t ticket = P.submit_finish(1, inputs.data(), rs.data(), pp.coords, &pp.handle);
    std::vector<uint8_t> pub;
    try {
      P.collect(ticket, 1, po.compressed, &po.values[0][0], &po.error, nullptr, nullptr, !w.multi);
      if (w.multi && !po.error) P.collect_public(ticket, 1, &pub);   // ys, root, nullifiers, x, ext, selectors (witness.rs:777-802)
    } catch (...) {
      if (w.multi) P.wipe(ticket);
      throw;
    }
    if (w.multi) P.wipe(ticket);
    if (po.error) throw Error("Error calculating witness: graph evaluation failed (code " + std::to_string(po.error) + ")");
    std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
    fill_outputs(po, pr.get());
    if (w.multi) values_from_public(pub.data(), P.graph().max_out, &pr->values);
    return pr.release();
  }
}

// finish_proof for a batch of single-message-id requests (FFI_RLN::gather_fin): ONE submit_finish / collect; throws on
// the first request that cannot be finished (the caller then finishes each on its own, so that every request keeps its error)
void finish_many(FFI_RLN& rln, const std::vector<FFI_RLN::GatherReq*>& batch, const std::vector<CFr>& rs,
                 std::vector<FFI_RLNProof*>& outs) {
  std::lock_guard<std::mutex> guard(*rln.prove_mu);
  if (rln.auto_partial) rln.memo_adopt_pending();
  Prover& P = *rln.prover;
  const size_t n = batch.size(), ni = P.inputs_per_proof();
  const std::vector<uint8_t>& known = P.known_mask();
  std::vector<uint8_t> inputs(n * ni * 32), rsb(n * 64), coords(n * 320), proofs(n * 128), values(n * 160);
  ZeroOnExit z1{inputs}, z2{rsb};
  std::vector<uint64_t> handles(n);
  std::vector<uint32_t> errs(n);
  for (size_t i = 0; i < n; i++) {
    const FFI_RLNWitnessInput& w = *batch[i]->w;
    const FFI_RLNPartialProof& pp = *batch[i]->pp;
    check_against_graph(P, w);
    if (w.multi) throw Error("finish_many: single message-id requests only");
    if (pp.mask.size() + 1 != known.size() || !std::equal(pp.mask.begin(), pp.mask.end(), known.begin() + 1))
      throw Error("Error producing proof: the partial proof's mask does not match this circuit (malformed verifying key)");
    fill_inputs(P, w, inputs.data() + i * ni * 32);
    memcpy(rsb.data() + i * 64, rs[2 * i].le, 32);
    memcpy(rsb.data() + i * 64 + 32, rs[2 * i + 1].le, 32);
    memcpy(coords.data() + i * 320, pp.coords, 320);
    handles[i] = pp.handle;
  }
  const uint64_t ticket = P.submit_finish(n, inputs.data(), rsb.data(), coords.data(), handles.data());
  P.collect(ticket, n, proofs.data(), values.data(), errs.data());
  for (size_t i = 0; i < n; i++)
    if (errs[i]) throw Error("Error calculating witness: graph evaluation failed (code " + std::to_string(errs[i]) + ")");
  for (size_t i = 0; i < n; i++) {
    std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
    memcpy(pr->proof, proofs.data() + 128 * i, 128);
    const uint8_t* v = values.data() + 160 * i;
    memcpy(pr->values.y.le, v, 32);
    memcpy(pr->values.root.le, v + 32, 32);
    memcpy(pr->values.nullifier.le, v + 64, 32);
    memcpy(pr->values.x.le, v + 96, 32);
    memcpy(pr->values.external_nullifier.le, v + 128, 32);
    outs[i] = pr.release();
  }
}

// ffi_finish_rln_proof from any thread: finishes that arrive while a batch is on the device go out together (as
// prove_one does it for full proofs); r, s: the blinding factors (sampled by the caller of this function when the ABI
// does not give them)
FFI_RLNProof* finish_one(FFI_RLN& rln, const FFI_RLNPartialProof& pp, FFI_RLNWitnessInput& w, const CFr& r, const CFr& s) {
  FFI_RLN::Gather& G = rln.gather_fin;
  if (G.most == 0 || w.multi) return finish_proof(rln, pp, w, r, s);
  FFI_RLN::GatherReq me;
  me.w = &w;
  me.pp = &pp;
  me.rs[0] = r;
  me.rs[1] = s;
  me.has_rs = true;
  struct WipeReq {
    FFI_RLN::GatherReq& q;
    ~WipeReq() { secure_zero(q.rs, sizeof q.rs); }
  } wipe_me{me};
  G.pass(me, [&](const std::vector<FFI_RLN::GatherReq*>& batch) {
    const size_t n = batch.size();
    auto alone = [&](size_t i) {
      try {
        batch[i]->out = finish_proof(rln, *batch[i]->pp, *batch[i]->w, batch[i]->rs[0], batch[i]->rs[1]);
      } catch (const std::exception& e) {
        batch[i]->err = e.what();
        if (batch[i]->err.empty()) batch[i]->err = "Error producing proof";
      }
    };
    if (n == 1) return alone(0);
    std::vector<FFI_RLNProof*> outs(n, nullptr);
    try {
      std::vector<CFr> rs(2 * n);
      struct WipeRs {
        std::vector<CFr>& v;
        ~WipeRs() { secure_zero(v.data(), v.size() * sizeof(CFr)); }
      } wipe_rs{rs};
      for (size_t i = 0; i < n; i++) {
        rs[2 * i] = batch[i]->rs[0];
        rs[2 * i + 1] = batch[i]->rs[1];
      }
      finish_many(rln, batch, rs, outs);
      for (size_t i = 0; i < n; i++) batch[i]->out = outs[i];
    } catch (const std::exception&) {
      for (FFI_RLNProof* o : outs) delete o;
      for (size_t i = 0; i < n; i++) alone(i);
    }
  });
  if (!me.err.empty()) throw Error(me.err);
  return me.out;
}


bool verify_zk(FFI_RLN& rln, const FFI_RLNProof& pr) {  // verify_zk_proof (proof.rs:856-894)
  G1Affine A, C;
  G2Affine B;
  if (!g1_decompress(pr.proof, &A) || !g2_decompress(pr.proof + 32, &B) || !g1_decompress(pr.proof + 96, &C))
    return false;
  auto F = [](const CFr& v) {
    uint32_t c[8];
    memcpy(c, v.le, 32);
    return Fr::from_canonical(c);
  };
  std::vector<Fr> in;  // proof.rs:863-885
  const FFI_RLNProofValues& v = pr.values;
  if (!v.multi) {
    in = {F(v.y), F(v.root), F(v.nullifier), F(v.x), F(v.external_nullifier)};
  } else {
    for (auto& y : v.ys) in.push_back(F(y));
    in.push_back(F(v.root));
    for (auto& nl : v.nullifiers) in.push_back(F(nl));
    in.push_back(F(v.x));
    in.push_back(F(v.external_nullifier));
    for (uint8_t b : v.selector_used) in.push_back(F(cfr_from_u64(b ? 1 : 0)));
  }
  if (in.size() + 1 != rln.prover->zkey().gamma_abc_g1.size())
    throw Error("Error producing proof: malformed verifying key (public input count does not match the circuit)");
  return groth16_verify(rln.prover->zkey(), A, B, C, in);
}


template <class ResT, class F>
ResT guard_ptr(F&& f) {
  try {
    return ResT{f(), no_err()};
  } catch (const std::exception& e) {
    return ResT{nullptr, make_str(e.what())};
  }
}
template <class F>
CBoolResult_t guard_bool(F&& f) {
  try {
    return CBoolResult_t{f(), no_err()};
  } catch (const std::exception& e) {
    return CBoolResult_t{false, make_str(e.what())};
  }
}
template <class F>
CResult_Vec_uint8_Vec_uint8_t guard_bytes(F&& f) {
  try {
    return {make_bytes(f()), no_err()};
  } catch (const std::exception& e) {
    return {Vec_uint8_t{nullptr, 0, 0}, make_str(e.what())};
  }
}

FFI_RLN* rln_create(size_t depth, const std::vector<uint8_t>& zkey, const std::vector<uint8_t>& graph,
                    const TreeConfig& tcfg = TreeConfig()) {
  require_gpu();
  std::unique_ptr<FFI_RLN> r(new FFI_RLN);
  r->make_prover(zkey.data(), zkey.size(), graph.data(), graph.size(), tcfg);
  if (r->prover->graph().tree_depth != depth)  // graph_from_raw expected depth (circuit/mod.rs:163-179)
    throw Error("Graph error: tree depth mismatch: expected " + std::to_string(depth) + ", got " +
                std::to_string(r->prover->graph().tree_depth));
  r->open_tree(depth, tcfg);
  return r.release();
}

}  // namespace

extern "C" {

// ================================================================================ RLN object
CResult_FFI_RLN_ptr_Vec_uint8_t ffi_rln_new(size_t tree_depth, const char* config_path) {
  return guard_ptr<CResult_FFI_RLN_ptr_Vec_uint8_t>([&]() -> FFI_RLN_t* {
    TreeConfig tcfg = tree_config_from_file(config_path);  // RLN::new: the config is parsed first (public.rs:113)
    // RLN::new always loads the embedded depth-20 circuit (public.rs:110-128, circuit/mod.rs:29-42);
    // the tree is built with the requested depth.
    std::string dir = resource_dir(20);
    auto zkey = read_file(dir + "/rln_final.arkzkey");
    auto graph = read_file(dir + "/graph.bin");
    require_gpu();
    std::unique_ptr<FFI_RLN> r(new FFI_RLN);
    r->make_prover(zkey.data(), zkey.size(), graph.data(), graph.size(), tcfg);
    r->open_tree(tree_depth, tcfg);
    return (FFI_RLN_t*)r.release();
  });
}
CResult_FFI_RLN_ptr_Vec_uint8_t ffi_rln_new_with_params(size_t tree_depth, const Vec_uint8_t* zkey_data,
                                                        const Vec_uint8_t* graph_data, const char* config_path) {
  return guard_ptr<CResult_FFI_RLN_ptr_Vec_uint8_t>([&]() -> FFI_RLN_t* {
    if (!zkey_data || !graph_data) throw Error("ZKey error: Empty zkey bytes");
    std::vector<uint8_t> z(zkey_data->ptr, zkey_data->ptr + zkey_data->len);
    std::vector<uint8_t> g(graph_data->ptr, graph_data->ptr + graph_data->len);
    return (FFI_RLN_t*)rln_create(tree_depth, z, g, tree_config_from_file(config_path));
  });
}
void ffi_rln_free(FFI_RLN_t* rln) { delete (FFI_RLN*)rln; }
// EXT (include/rln_amd.h): how the prover behind an FFI_RLN / FFI_RLNV3 object was sized (config_path keys, environment)
int rlnamd_ffi_prover_info(const void* ffi_rln, rlnamd_prover_info* info) {
  if (!ffi_rln || !info) return RLNAMD_ERR;
  try {
    rlnamd::fill_prover_info(*((const FFI_RLN*)ffi_rln)->prover, info);
  } catch (...) {
    return RLNAMD_ERR;
  }
  return RLNAMD_OK;
}
// EXT (include/rln_amd.h): the member memo of an object built with "auto_partial": [0] members remembered, [1] proofs
// that were finishes of a remembered partial proof, [2] proofs made from scratch, [3] 1 while a partial proof is pending
int rlnamd_ffi_memo_stats(const void* ffi_rln, uint64_t out[4]) {
  if (!ffi_rln || !out) return RLNAMD_ERR;
  FFI_RLN& r = *(FFI_RLN*)ffi_rln;
  std::lock_guard<std::mutex> guard(*r.prove_mu);
  out[0] = r.memo.size();
  out[1] = r.memo_hits;
  out[2] = r.memo_misses;
  out[3] = r.pending_ticket ? 1 : 0;
  return RLNAMD_OK;
}
// EXT (include/rln_amd.h): the gathering of concurrent single-proof calls: [0] batches led, [1] calls that went out in
// them, [2] the largest batch, [3] the most calls one batch may take (0: off), [4] batches whose leader waited for a
// recent caller, [5] nanoseconds the leaders spent proving their batches, [6] / [7] batches and calls of the finish queue
int rlnamd_ffi_gather_stats(const void* ffi_rln, uint64_t out[8]) {
  if (!ffi_rln || !out) return RLNAMD_ERR;
  FFI_RLN& r = *(FFI_RLN*)ffi_rln;
  std::lock_guard<std::mutex> guard(r.gather.mu);
  out[0] = r.gather.batches;
  out[1] = r.gather.calls;
  out[2] = r.gather.largest;
  out[3] = r.gather.most;
  out[4] = r.gather.waited;
  out[5] = r.gather.busy_ns;
  {
    std::lock_guard<std::mutex> g2(r.gather_fin.mu);
    out[6] = r.gather_fin.batches;
    out[7] = r.gather_fin.calls;
  }
  return RLNAMD_OK;
}
size_t ffi_rln_get_tree_depth(FFI_RLN_t* const* rln) { return ((FFI_RLN*)*rln)->tree.depth; }
size_t ffi_rln_get_max_out(FFI_RLN_t* const* rln) { return ((FFI_RLN*)*rln)->prover->graph().max_out; }

// ================================================================================ proofs
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_generate_rln_proof(FFI_RLN_t* const* rln,
                                                            FFI_RLNWitnessInput_t* const* witness) {
  return guard_ptr<CResult_FFI_RLNProof_ptr_Vec_uint8_t>([&]() -> FFI_RLNProof_t* {
    return (FFI_RLNProof_t*)prove_one(*(FFI_RLN*)*rln, (FFI_RLNWitnessInput*)*witness, nullptr);
  });
}
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_generate_rln_proof_with_rs(FFI_RLN_t* const* rln,
                                                                    FFI_RLNWitnessInput_t* const* witness,
                                                                    const CFr_t* r, const CFr_t* s) {
  return guard_ptr<CResult_FFI_RLNProof_ptr_Vec_uint8_t>([&]() -> FFI_RLNProof_t* {
    CFr rs[2] = {R(r), R(s)};
    return (FFI_RLNProof_t*)prove_one(*(FFI_RLN*)*rln, (FFI_RLNWitnessInput*)*witness, rs);
  });
}
CBoolResult_t ffi_generate_rln_proofs_batch(FFI_RLN_t* const* rln, FFI_RLNWitnessInput_t* const* witnesses, size_t n,
                                            const CFr_t* rs, FFI_RLNProof_t** out) {
  return guard_bool([&]() {
    prove_many(*(FFI_RLN*)*rln, (FFI_RLNWitnessInput* const*)witnesses, n, (const CFr*)rs, (FFI_RLNProof**)out);
    return true;
  });
}
CBoolResult_t ffi_verify_rln_proof(FFI_RLN_t* const* rln, FFI_RLNProof_t* const* proof, const CFr_t* x) {
  return guard_bool([&]() {  // public.rs:725-745: proof -> root -> signal, first failure is an error
    FFI_RLN& r = *(FFI_RLN*)*rln;
    const FFI_RLNProof& pr = *(FFI_RLNProof*)*proof;
    if (!verify_zk(r, pr)) throw Error("Verification error: Invalid proof provided");
    CFr root;
    r.tree.get_node_host(0, root.le);
    if (cfr_cmp(root, pr.values.root) != 0) throw Error("Verification error: Expected one of the provided roots");
    if (cfr_cmp(R(x), pr.values.x) != 0) throw Error("Verification error: Signal value does not match");
    return true;
  });
}
CBoolResult_t ffi_verify_with_roots(FFI_RLN_t* const* rln, FFI_RLNProof_t* const* proof, const Vec_CFr_t* roots,
                                    const CFr_t* x) {
  return guard_bool([&]() {  // public.rs:750-771
    FFI_RLN& r = *(FFI_RLN*)*rln;
    const FFI_RLNProof& pr = *(FFI_RLNProof*)*proof;
    if (!verify_zk(r, pr)) throw Error("Verification error: Invalid proof provided");
    if (roots && roots->len) {
      bool found = false;
      for (size_t i = 0; i < roots->len; i++) found |= cfr_cmp(((const CFr*)roots->ptr)[i], pr.values.root) == 0;
      if (!found) throw Error("Verification error: Expected one of the provided roots");
    }
    if (cfr_cmp(R(x), pr.values.x) != 0) throw Error("Verification error: Signal value does not match");
    return true;
  });
}
FFI_RLNProofValues_t* ffi_rln_proof_get_values(FFI_RLNProof_t* const* proof) {
  return (FFI_RLNProofValues_t*)new FFI_RLNProofValues(((FFI_RLNProof*)*proof)->values);
}
uint8_t ffi_rln_proof_get_version_byte(FFI_RLNProof_t* const* p) { return ((FFI_RLNProof*)*p)->values.multi ? 0x01 : 0x00; }
CResult_Vec_uint8_Vec_uint8_t ffi_rln_proof_to_bytes_le(FFI_RLNProof_t* const* proof) {
  return guard_bytes([&]() { return proof_bytes(*(FFI_RLNProof*)*proof, false); });
}
CResult_Vec_uint8_Vec_uint8_t ffi_rln_proof_to_bytes_be(FFI_RLNProof_t* const* proof) {
  return guard_bytes([&]() { return proof_bytes(*(FFI_RLNProof*)*proof, true); });
}
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_bytes_le_to_rln_proof(const Vec_uint8_t* bytes) {
  return guard_ptr<CResult_FFI_RLNProof_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNProof_t* { return (FFI_RLNProof_t*)proof_from_bytes(bytes, false); });
}
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_bytes_be_to_rln_proof(const Vec_uint8_t* bytes) {
  return guard_ptr<CResult_FFI_RLNProof_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNProof_t* { return (FFI_RLNProof_t*)proof_from_bytes(bytes, true); });
}
void ffi_rln_proof_free(FFI_RLNProof_t* proof) { delete (FFI_RLNProof*)proof; }

// ================================================================================ partial proofs
CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t ffi_rln_partial_witness_input_new(
    const CFr_t* identity_secret, const CFr_t* user_message_limit, const Vec_CFr_t* path_elements,
    const Vec_uint8_t* identity_path_index) {
  return guard_ptr<CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t>([&]() -> FFI_RLNPartialWitnessInput_t* {
    std::unique_ptr<FFI_RLNPartialWitnessInput> w(new FFI_RLNPartialWitnessInput);
    w->identity_secret = R(identity_secret);
    w->user_message_limit = R(user_message_limit);
    w->path_elements.assign((const CFr*)path_elements->ptr, (const CFr*)path_elements->ptr + path_elements->len);
    w->identity_path_index.assign(identity_path_index->ptr, identity_path_index->ptr + identity_path_index->len);
    validate_partial_witness(*w);
    return (FFI_RLNPartialWitnessInput_t*)w.release();
  });
}
FFI_RLNPartialWitnessInput_t* ffi_rln_witness_to_partial_witness(FFI_RLNWitnessInput_t* const* w) {
  const FFI_RLNWitnessInput& f = *(FFI_RLNWitnessInput*)*w;
  FFI_RLNPartialWitnessInput* p = new FFI_RLNPartialWitnessInput;
  p->identity_secret = f.identity_secret;
  p->user_message_limit = f.user_message_limit;
  p->path_elements = f.path_elements;
  p->identity_path_index = f.identity_path_index;
  return (FFI_RLNPartialWitnessInput_t*)p;
}
void ffi_rln_partial_witness_input_free(FFI_RLNPartialWitnessInput_t* w) { delete (FFI_RLNPartialWitnessInput*)w; }
#define PW(w) (*(FFI_RLNPartialWitnessInput*)*(w))
uint8_t ffi_rln_partial_witness_input_get_version_byte(FFI_RLNPartialWitnessInput_t* const*) { return 0x00; }
CFr_t* ffi_rln_partial_witness_input_get_identity_secret(FFI_RLNPartialWitnessInput_t* const* w) {
  return box_cfr(PW(w).identity_secret);
}
CFr_t* ffi_rln_partial_witness_input_get_user_message_limit(FFI_RLNPartialWitnessInput_t* const* w) {
  return box_cfr(PW(w).user_message_limit);
}
Vec_CFr_t ffi_rln_partial_witness_input_get_path_elements(FFI_RLNPartialWitnessInput_t* const* w) {
  return make_vec_cfr(PW(w).path_elements);
}
Vec_uint8_t ffi_rln_partial_witness_input_get_identity_path_index(FFI_RLNPartialWitnessInput_t* const* w) {
  return make_bytes(PW(w).identity_path_index);
}
CResult_Vec_uint8_Vec_uint8_t ffi_rln_partial_witness_to_bytes_le(FFI_RLNPartialWitnessInput_t* const* w) {
  return guard_bytes([&]() { return partial_witness_bytes(PW(w), false); });
}
CResult_Vec_uint8_Vec_uint8_t ffi_rln_partial_witness_to_bytes_be(FFI_RLNPartialWitnessInput_t* const* w) {
  return guard_bytes([&]() { return partial_witness_bytes(PW(w), true); });
}
CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t ffi_bytes_le_to_rln_partial_witness(const Vec_uint8_t* b) {
  return guard_ptr<CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNPartialWitnessInput_t* { return (FFI_RLNPartialWitnessInput_t*)partial_witness_from_bytes(b, false); });
}
CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t ffi_bytes_be_to_rln_partial_witness(const Vec_uint8_t* b) {
  return guard_ptr<CResult_FFI_RLNPartialWitnessInput_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNPartialWitnessInput_t* { return (FFI_RLNPartialWitnessInput_t*)partial_witness_from_bytes(b, true); });
}
CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t ffi_generate_partial_zk_proof(
    FFI_RLN_t* const* rln, FFI_RLNPartialWitnessInput_t* const* partial_witness) {
  return guard_ptr<CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t>([&]() -> FFI_RLNPartialProof_t* {
    return (FFI_RLNPartialProof_t*)prove_partial(*(FFI_RLN*)*rln, *(FFI_RLNPartialWitnessInput*)*partial_witness);
  });
}
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_finish_rln_proof(FFI_RLN_t* const* rln, FFI_RLNPartialProof_t* const* partial,
                                                          FFI_RLNWitnessInput_t* const* witness) {
  return guard_ptr<CResult_FFI_RLNProof_ptr_Vec_uint8_t>([&]() -> FFI_RLNProof_t* {
    return (FFI_RLNProof_t*)finish_one(*(FFI_RLN*)*rln, *(FFI_RLNPartialProof*)*partial,
                                       *(FFI_RLNWitnessInput*)*witness, random_fr(), random_fr());
  });
}
CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_finish_rln_proof_with_rs(FFI_RLN_t* const* rln,
                                                                  FFI_RLNPartialProof_t* const* partial,
                                                                  FFI_RLNWitnessInput_t* const* witness,
                                                                  const CFr_t* r, const CFr_t* s) {
  return guard_ptr<CResult_FFI_RLNProof_ptr_Vec_uint8_t>([&]() -> FFI_RLNProof_t* {
    return (FFI_RLNProof_t*)finish_one(*(FFI_RLN*)*rln, *(FFI_RLNPartialProof*)*partial,
                                       *(FFI_RLNWitnessInput*)*witness, R(r), R(s));
  });
}
CBoolResult_t ffi_finish_rln_proofs_batch(FFI_RLN_t* const* rln, FFI_RLNPartialProof_t* const* partials,
                                          FFI_RLNWitnessInput_t* const* witnesses, size_t n, const CFr_t* rs,
                                          FFI_RLNProof_t** out) {
  return guard_bool([&]() {
    FFI_RLN& r = *(FFI_RLN*)*rln;
    std::lock_guard<std::mutex> guard(*r.prove_mu);
    if (r.auto_partial) r.memo_adopt_pending();
    Prover& P = *r.prover;
    const size_t cap = P.capacity(), ni = P.inputs_per_proof();
    const std::vector<uint8_t>& known = P.known_mask();
    std::vector<FFI_RLNProof*> made;
    // chunks of the workspace's capacity, every slot in flight: the host packs chunk k + 1 while the device finishes chunk k
    struct Pending { uint64_t ticket; size_t m; };
    std::deque<Pending> q;
    std::vector<uint8_t> inputs, rsb, coords, proofs, values;
    ZeroOnExit z1{inputs}, z2{rsb};
    std::vector<uint64_t> handles;
    std::vector<uint32_t> errs;
    inputs.reserve(std::min(n, cap) * ni * 32);   // one allocation: a vector that grew would leave secrets behind uncleared
    rsb.reserve(std::min(n, cap) * 64);
    auto take = [&]() {
      const Pending f = q.front();
      q.pop_front();
      proofs.resize(f.m * 128);
      values.resize(f.m * 160);
      errs.resize(f.m);
      P.collect(f.ticket, f.m, proofs.data(), values.data(), errs.data());
      for (size_t i = 0; i < f.m; i++) {
        if (errs[i]) throw Error("Error calculating witness: graph evaluation failed (code " + std::to_string(errs[i]) + ")");
        std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
        memcpy(pr->proof, proofs.data() + 128 * i, 128);
        const uint8_t* v = values.data() + 160 * i;
        memcpy(pr->values.y.le, v, 32);
        memcpy(pr->values.root.le, v + 32, 32);
        memcpy(pr->values.nullifier.le, v + 64, 32);
        memcpy(pr->values.x.le, v + 96, 32);
        memcpy(pr->values.external_nullifier.le, v + 128, 32);
        made.push_back(pr.release());
      }
    };
    try {
      for (size_t off = 0; off < n; off += cap) {
        const size_t m = std::min(cap, n - off);
        inputs.assign(m * ni * 32, 0);
        rsb.resize(m * 64);
        coords.resize(m * 320);
        handles.resize(m);
        for (size_t i = 0; i < m; i++) {
          const FFI_RLNWitnessInput& w = *(const FFI_RLNWitnessInput*)witnesses[off + i];
          const FFI_RLNPartialProof& pp = *(const FFI_RLNPartialProof*)partials[off + i];
          check_against_graph(P, w);
          if (w.multi) throw Error("ffi_finish_rln_proofs_batch: single message-id witnesses only");
          if (pp.mask.size() + 1 != known.size() || !std::equal(pp.mask.begin(), pp.mask.end(), known.begin() + 1))
            throw Error("Error producing proof: the partial proof's mask does not match this circuit (malformed verifying key)");
          fill_inputs(P, w, inputs.data() + i * ni * 32);
          const CFr rr = rs ? R(&rs[2 * (off + i)]) : random_fr(), ss = rs ? R(&rs[2 * (off + i) + 1]) : random_fr();
          memcpy(rsb.data() + i * 64, rr.le, 32);
          memcpy(rsb.data() + i * 64 + 32, ss.le, 32);
          memcpy(coords.data() + i * 320, pp.coords, 320);
          handles[i] = pp.handle;
        }
        if ((int)q.size() == P.slots()) take();
        q.push_back({P.submit_finish(m, inputs.data(), rsb.data(), coords.data(), handles.data()), m});
      }
      while (!q.empty()) take();
    } catch (...) {
      try {   // nothing of this call stays in flight behind the error; the batches never collected are wiped
        P.sync();
        for (const Pending& f : q) P.wipe(f.ticket);
        P.sync();
      } catch (...) {
      }
      for (FFI_RLNProof* o : made) delete o;
      throw;
    }
    for (size_t i = 0; i < n; i++) out[i] = (FFI_RLNProof_t*)made[i];
    return true;
  });
}
uint8_t ffi_rln_partial_proof_get_version_byte(FFI_RLNPartialProof_t* const*) { return 0x00; }
CResult_Vec_uint8_Vec_uint8_t ffi_rln_partial_proof_to_bytes_le(FFI_RLNPartialProof_t* const* partial) {
  return guard_bytes([&]() { return partial_proof_bytes(*(FFI_RLNPartialProof*)*partial); });
}
CResult_Vec_uint8_Vec_uint8_t ffi_rln_partial_proof_to_bytes_be(FFI_RLNPartialProof_t* const* partial) {
  return guard_bytes([&]() { return partial_proof_bytes(*(FFI_RLNPartialProof*)*partial); });  // always LE
}
CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t ffi_bytes_le_to_rln_partial_proof(const Vec_uint8_t* bytes) {
  return guard_ptr<CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNPartialProof_t* { return (FFI_RLNPartialProof_t*)partial_proof_from_bytes(bytes); });
}
CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t ffi_bytes_be_to_rln_partial_proof(const Vec_uint8_t* bytes) {
  return guard_ptr<CResult_FFI_RLNPartialProof_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNPartialProof_t* { return (FFI_RLNPartialProof_t*)partial_proof_from_bytes(bytes); });
}
void ffi_rln_partial_proof_free(FFI_RLNPartialProof_t* partial) { delete (FFI_RLNPartialProof*)partial; }

// ================================================================================ witness input
CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t ffi_rln_witness_input_new_single(
    const CFr_t* identity_secret, const CFr_t* user_message_limit, const CFr_t* message_id,
    const Vec_CFr_t* path_elements, const Vec_uint8_t* identity_path_index, const CFr_t* x,
    const CFr_t* external_nullifier) {
  return guard_ptr<CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t>([&]() -> FFI_RLNWitnessInput_t* {
    std::unique_ptr<FFI_RLNWitnessInput> w(new FFI_RLNWitnessInput);
    w->identity_secret = R(identity_secret);
    w->user_message_limit = R(user_message_limit);
    w->message_id = R(message_id);
    w->path_elements.assign((const CFr*)path_elements->ptr, (const CFr*)path_elements->ptr + path_elements->len);
    w->identity_path_index.assign(identity_path_index->ptr, identity_path_index->ptr + identity_path_index->len);
    w->x = R(x);
    w->external_nullifier = R(external_nullifier);
    validate_witness(*w);
    return (FFI_RLNWitnessInput_t*)w.release();
  });
}
CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t ffi_rln_witness_input_new_multi(
    const CFr_t* identity_secret, const CFr_t* user_message_limit, const Vec_CFr_t* message_ids,
    const Vec_CFr_t* path_elements, const Vec_uint8_t* identity_path_index, const CFr_t* x,
    const CFr_t* external_nullifier, const Vec_bool_t* selector_used) {
  return guard_ptr<CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t>([&]() -> FFI_RLNWitnessInput_t* {
    std::unique_ptr<FFI_RLNWitnessInput> w(new FFI_RLNWitnessInput);
    w->multi = true;
    w->identity_secret = R(identity_secret);
    w->user_message_limit = R(user_message_limit);
    w->message_id = cfr_from_u64(0);
    w->message_ids.assign((const CFr*)message_ids->ptr, (const CFr*)message_ids->ptr + message_ids->len);
    w->path_elements.assign((const CFr*)path_elements->ptr, (const CFr*)path_elements->ptr + path_elements->len);
    w->identity_path_index.assign(identity_path_index->ptr, identity_path_index->ptr + identity_path_index->len);
    w->x = R(x);
    w->external_nullifier = R(external_nullifier);
    for (size_t i = 0; i < selector_used->len; i++) w->selector_used.push_back(selector_used->ptr[i] ? 1 : 0);
    validate_witness(*w);
    return (FFI_RLNWitnessInput_t*)w.release();
  });
}
#define W(w) (*(FFI_RLNWitnessInput*)*(w))
static Vec_bool_t make_vec_bool(const std::vector<uint8_t>& v) {
  bool* p = (bool*)malloc(v.size() ? v.size() : 1);
  for (size_t i = 0; i < v.size(); i++) p[i] = v[i] != 0;
  return {p, v.size(), v.size() ? v.size() : 1};
}
Vec_CFr_t ffi_rln_witness_input_get_message_ids(FFI_RLNWitnessInput_t* const* w) { return make_vec_cfr(W(w).message_ids); }
Vec_bool_t ffi_rln_witness_input_get_selector_used(FFI_RLNWitnessInput_t* const* w) { return make_vec_bool(W(w).selector_used); }
void ffi_vec_bool_free(Vec_bool_t v) { free(v.ptr); }
uint8_t ffi_rln_witness_input_get_version_byte(FFI_RLNWitnessInput_t* const* w) { return W(w).multi ? 0x01 : 0x00; }
CFr_t* ffi_rln_witness_input_get_identity_secret(FFI_RLNWitnessInput_t* const* w) { return box_cfr(W(w).identity_secret); }
CFr_t* ffi_rln_witness_input_get_user_message_limit(FFI_RLNWitnessInput_t* const* w) { return box_cfr(W(w).user_message_limit); }
CFr_t* ffi_rln_witness_input_get_message_id(FFI_RLNWitnessInput_t* const* w) { return box_cfr(W(w).message_id); }
Vec_CFr_t ffi_rln_witness_input_get_path_elements(FFI_RLNWitnessInput_t* const* w) { return make_vec_cfr(W(w).path_elements); }
Vec_uint8_t ffi_rln_witness_input_get_identity_path_index(FFI_RLNWitnessInput_t* const* w) { return make_bytes(W(w).identity_path_index); }
CFr_t* ffi_rln_witness_input_get_x(FFI_RLNWitnessInput_t* const* w) { return box_cfr(W(w).x); }
CFr_t* ffi_rln_witness_input_get_external_nullifier(FFI_RLNWitnessInput_t* const* w) { return box_cfr(W(w).external_nullifier); }
CResult_Vec_uint8_Vec_uint8_t ffi_rln_witness_to_bytes_le(FFI_RLNWitnessInput_t* const* w) {
  return guard_bytes([&]() { return witness_bytes(W(w), false); });
}
CResult_Vec_uint8_Vec_uint8_t ffi_rln_witness_to_bytes_be(FFI_RLNWitnessInput_t* const* w) {
  return guard_bytes([&]() { return witness_bytes(W(w), true); });
}
CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t ffi_bytes_le_to_rln_witness(const Vec_uint8_t* b) {
  return guard_ptr<CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNWitnessInput_t* { return (FFI_RLNWitnessInput_t*)witness_from_bytes(b, false); });
}
CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t ffi_bytes_be_to_rln_witness(const Vec_uint8_t* b) {
  return guard_ptr<CResult_FFI_RLNWitnessInput_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNWitnessInput_t* { return (FFI_RLNWitnessInput_t*)witness_from_bytes(b, true); });
}
void ffi_rln_witness_input_free(FFI_RLNWitnessInput_t* w) { delete (FFI_RLNWitnessInput*)w; }

// ================================================================================ proof values
#define PV(p) (*(FFI_RLNProofValues*)*(p))
CFr_t* ffi_rln_proof_values_get_root(FFI_RLNProofValues_t* const* pv) { return box_cfr(PV(pv).root); }
CFr_t* ffi_rln_proof_values_get_x(FFI_RLNProofValues_t* const* pv) { return box_cfr(PV(pv).x); }
CFr_t* ffi_rln_proof_values_get_external_nullifier(FFI_RLNProofValues_t* const* pv) { return box_cfr(PV(pv).external_nullifier); }
// wrong-variant getters return an error string (the V3 behaviour, ffi_rln_v3.rs:705-718; V1 panics)
CResult_CFr_ptr_Vec_uint8_t ffi_rln_proof_values_get_y(FFI_RLNProofValues_t* const* pv) {
  if (PV(pv).multi) return {nullptr, make_str("Field `y` does not exist on the `MultiV1` variant")};
  return {box_cfr(PV(pv).y), no_err()};
}
CResult_CFr_ptr_Vec_uint8_t ffi_rln_proof_values_get_nullifier(FFI_RLNProofValues_t* const* pv) {
  if (PV(pv).multi) return {nullptr, make_str("Field `nullifier` does not exist on the `MultiV1` variant")};
  return {box_cfr(PV(pv).nullifier), no_err()};
}
CResult_Vec_CFr_Vec_uint8_t ffi_rln_proof_values_get_ys(FFI_RLNProofValues_t* const* pv) {
  if (!PV(pv).multi) return {Vec_CFr_t{nullptr, 0, 0}, make_str("Field `ys` does not exist on the `SingleV1` variant")};
  return {make_vec_cfr(PV(pv).ys), no_err()};
}
CResult_Vec_CFr_Vec_uint8_t ffi_rln_proof_values_get_nullifiers(FFI_RLNProofValues_t* const* pv) {
  if (!PV(pv).multi)
    return {Vec_CFr_t{nullptr, 0, 0}, make_str("Field `nullifiers` does not exist on the `SingleV1` variant")};
  return {make_vec_cfr(PV(pv).nullifiers), no_err()};
}
CResult_Vec_bool_Vec_uint8_t ffi_rln_proof_values_get_selector_used(FFI_RLNProofValues_t* const* pv) {
  if (!PV(pv).multi)
    return {Vec_bool_t{nullptr, 0, 0}, make_str("Field `selector_used` does not exist on the `SingleV1` variant")};
  return {make_vec_bool(PV(pv).selector_used), no_err()};
}
uint8_t ffi_rln_proof_values_get_version_byte(FFI_RLNProofValues_t* const* pv) { return PV(pv).multi ? 0x01 : 0x00; }
Vec_uint8_t ffi_rln_proof_values_to_bytes_le(FFI_RLNProofValues_t* const* pv) { return make_bytes(values_bytes(PV(pv), false)); }
Vec_uint8_t ffi_rln_proof_values_to_bytes_be(FFI_RLNProofValues_t* const* pv) { return make_bytes(values_bytes(PV(pv), true)); }
static FFI_RLNProofValues* pv_from(const Vec_uint8_t* b, bool be) {
  if (!b || b->len == 0) throw Error("Expected to read 1 bytes but read 0 bytes");
  Cursor c{b->ptr, b->len, 0, be};
  std::unique_ptr<FFI_RLNProofValues> v(new FFI_RLNProofValues(values_from(c)));
  return v.release();
}
CResult_FFI_RLNProofValues_ptr_Vec_uint8_t ffi_bytes_le_to_rln_proof_values(const Vec_uint8_t* b) {
  return guard_ptr<CResult_FFI_RLNProofValues_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNProofValues_t* { return (FFI_RLNProofValues_t*)pv_from(b, false); });
}
CResult_FFI_RLNProofValues_ptr_Vec_uint8_t ffi_bytes_be_to_rln_proof_values(const Vec_uint8_t* b) {
  return guard_ptr<CResult_FFI_RLNProofValues_ptr_Vec_uint8_t>(
      [&]() -> FFI_RLNProofValues_t* { return (FFI_RLNProofValues_t*)pv_from(b, true); });
}
void ffi_rln_proof_values_free(FFI_RLNProofValues_t* pv) { delete (FFI_RLNProofValues*)pv; }

// ================================================================================ tree
#define RLNM(r) (*(FFI_RLN*)*(r))
static std::vector<CFr> vec_of(const Vec_CFr_t* v) {
  return std::vector<CFr>((const CFr*)v->ptr, (const CFr*)v->ptr + v->len);
}
CBoolResult_t ffi_set_tree(FFI_RLN_t** rln, size_t tree_depth) {
  return guard_bool([&]() {  // PoseidonTree::default(depth): the stored tree is dropped (flushed), the new one is temporary
    RLNM(rln).replace_with_default_tree(tree_depth);
    return true;
  });
}
CBoolResult_t ffi_delete_leaf(FFI_RLN_t** rln, size_t index) {
  return guard_bool([&]() { RLNM(rln).del(index); return true; });
}
CBoolResult_t ffi_set_leaf(FFI_RLN_t** rln, size_t index, const CFr_t* leaf) {
  return guard_bool([&]() { RLNM(rln).set(index, R(leaf)); return true; });
}
CResult_CFr_ptr_Vec_uint8_t ffi_get_leaf(FFI_RLN_t* const* rln, size_t index) {
  return guard_ptr<CResult_CFr_ptr_Vec_uint8_t>([&]() { return box_cfr(RLNM(rln).get(index)); });
}
size_t ffi_leaves_set(FFI_RLN_t* const* rln) { return RLNM(rln).next_index; }
CBoolResult_t ffi_set_next_leaf(FFI_RLN_t** rln, const CFr_t* leaf) {
  return guard_bool([&]() { RLNM(rln).set(RLNM(rln).next_index, R(leaf)); return true; });  // update_next :271-274
}
CBoolResult_t ffi_set_leaves_from(FFI_RLN_t** rln, size_t index, const Vec_CFr_t* leaves) {
  return guard_bool([&]() { RLNM(rln).override_range(index, vec_of(leaves), {}); return true; });  // public.rs:364-368
}
CBoolResult_t ffi_init_tree_with_leaves(FFI_RLN_t** rln, const Vec_CFr_t* leaves) {
  return guard_bool([&]() {  // public.rs:376-379
    FFI_RLN& r = RLNM(rln);
    r.replace_with_default_tree(r.tree.depth);  // set_tree(depth) then set_leaves_from (public.rs:376-379)
    r.override_range(0, vec_of(leaves), {});
    return true;
  });
}
CBoolResult_t ffi_atomic_operation(FFI_RLN_t** rln, size_t index, const Vec_CFr_t* leaves, const Vec_size_t* indices) {
  return guard_bool([&]() {
    RLNM(rln).override_range(index, vec_of(leaves), std::vector<size_t>(indices->ptr, indices->ptr + indices->len));
    return true;
  });
}
CBoolResult_t ffi_seq_atomic_operation(FFI_RLN_t** rln, const Vec_CFr_t* leaves, const Vec_uint8_t* indices) {
  return guard_bool([&]() {  // ffi_tree.rs:170-187
    std::vector<size_t> idx(indices->ptr, indices->ptr + indices->len);
    RLNM(rln).override_range(RLNM(rln).next_index, vec_of(leaves), idx);
    return true;
  });
}
CFr_t* ffi_get_root(FFI_RLN_t* const* rln) {
  CFr r;
  memset(r.le, 0, 32);
  try {
    RLNM(rln).tree.get_node_host(0, r.le);
  } catch (...) {
  }
  return box_cfr(r);
}
CResult_FFI_MerkleProof_ptr_Vec_uint8_t ffi_get_merkle_proof(FFI_RLN_t* const* rln, size_t index) {
  return guard_ptr<CResult_FFI_MerkleProof_ptr_Vec_uint8_t>([&]() -> FFI_MerkleProof_t* {
    FFI_RLN& r = RLNM(rln);
    if (index >= r.tree.capacity()) throw Error("Leaf index out of bounds");
    size_t d = r.tree.depth;
    std::vector<CFr> elems(d);
    std::vector<uint8_t> bits(d);
    r.tree.proof_host(index, (uint8_t*)elems.data(), bits.data());
    FFI_MerkleProof_t* mp = (FFI_MerkleProof_t*)malloc(sizeof(FFI_MerkleProof_t));
    mp->path_elements = make_vec_cfr(elems);
    mp->path_index = make_bytes(bits);
    return mp;
  });
}
void ffi_merkle_proof_free(FFI_MerkleProof_t* proof) {
  if (!proof) return;
  free(proof->path_elements.ptr);
  free(proof->path_index.ptr);
  free(proof);
}
CBoolResult_t ffi_set_metadata(FFI_RLN_t** rln, const Vec_uint8_t* metadata) {
  return guard_bool([&]() { RLNM(rln).metadata.assign(metadata->ptr, metadata->ptr + metadata->len); return true; });
}
CResult_Vec_uint8_Vec_uint8_t ffi_get_metadata(FFI_RLN_t* const* rln) {
  return guard_bytes([&]() { return RLNM(rln).metadata; });
}
CBoolResult_t ffi_flush(FFI_RLN_t** rln) {  // deferred leaf writes are hashed now; persistent trees write their snapshot
  return guard_bool([&]() { RLNM(rln).tree.flush_pending(); RLNM(rln).flush(); return true; });
}

// ================================================================================ CFr / Vec helpers
CFr_t* ffi_cfr_zero(void) { return box_cfr(cfr_from_u64(0)); }
CFr_t* ffi_cfr_one(void) { return box_cfr(cfr_from_u64(1)); }
CResult_Vec_uint8_Vec_uint8_t ffi_cfr_to_bytes_le(const CFr_t* cfr) {
  return guard_bytes([&]() { return std::vector<uint8_t>(R(cfr).le, R(cfr).le + 32); });
}
CResult_Vec_uint8_Vec_uint8_t ffi_cfr_to_bytes_be(const CFr_t* cfr) {
  return guard_bytes([&]() {
    std::vector<uint8_t> b(32);
    be32(R(cfr).le, b.data());
    return b;
  });
}
static CFr cfr_from_bytes(const Vec_uint8_t* bytes, bool be) {
  if (!bytes) throw Error("Input data too short: expected at least 32 bytes, got 0 bytes");
  Cursor c{bytes->ptr, bytes->len, 0, be};
  return c.fr();
}
CResult_CFr_ptr_Vec_uint8_t ffi_bytes_le_to_cfr(const Vec_uint8_t* bytes) {
  return guard_ptr<CResult_CFr_ptr_Vec_uint8_t>([&]() { return box_cfr(cfr_from_bytes(bytes, false)); });
}
CResult_CFr_ptr_Vec_uint8_t ffi_bytes_be_to_cfr(const Vec_uint8_t* bytes) {
  return guard_ptr<CResult_CFr_ptr_Vec_uint8_t>([&]() { return box_cfr(cfr_from_bytes(bytes, true)); });
}
CFr_t* ffi_uint_to_cfr(uint32_t value) { return box_cfr(cfr_from_u64(value)); }
Vec_uint8_t ffi_cfr_debug(const CFr_t* cfr) { return make_str(cfr ? cfr_dec(R(cfr)) : "None"); }
void ffi_cfr_free(CFr_t* cfr) {   // a CFr may be an identity secret (keygen): cleared before it goes back to the heap
  if (cfr) secure_zero(cfr, sizeof(CFr));
  free(cfr);
}

Vec_CFr_t ffi_vec_cfr_new(size_t capacity) {
  size_t cap = capacity ? capacity : 1;
  return {(CFr_t*)malloc(cap * sizeof(CFr)), 0, cap};
}
Vec_CFr_t ffi_vec_cfr_from_cfr(const CFr_t* cfr) { return make_vec_cfr({R(cfr)}); }
void ffi_vec_cfr_push(Vec_CFr_t* v, const CFr_t* cfr) {
  if (v->len == v->cap || !v->ptr) {
    size_t cap = v->cap ? v->cap * 2 : 4;
    v->ptr = (CFr_t*)realloc(v->ptr, cap * sizeof(CFr));
    v->cap = cap;
  }
  ((CFr*)v->ptr)[v->len++] = R(cfr);
}
size_t ffi_vec_cfr_len(const Vec_CFr_t* v) { return v->len; }
const CFr_t* ffi_vec_cfr_get(const Vec_CFr_t* v, size_t i) { return i < v->len ? (const CFr_t*)((const CFr*)v->ptr + i) : nullptr; }
static std::vector<uint8_t> vec_cfr_bytes(const Vec_CFr_t* v, bool be) {  // utils.rs:123-156
  std::vector<uint8_t> b;
  put_u64(b, v->len, be);
  for (size_t i = 0; i < v->len; i++) put_fr(b, ((const CFr*)v->ptr)[i], be);
  return b;
}
CResult_Vec_uint8_Vec_uint8_t ffi_vec_cfr_to_bytes_le(const Vec_CFr_t* v) { return guard_bytes([&]() { return vec_cfr_bytes(v, false); }); }
CResult_Vec_uint8_Vec_uint8_t ffi_vec_cfr_to_bytes_be(const Vec_CFr_t* v) { return guard_bytes([&]() { return vec_cfr_bytes(v, true); }); }
static CResult_Vec_CFr_Vec_uint8_t vec_cfr_from(const Vec_uint8_t* bytes, bool be) {
  try {
    Cursor c{bytes->ptr, bytes->len, 0, be};
    return {make_vec_cfr(c.vec_fr()), no_err()};
  } catch (const std::exception& e) {
    return {Vec_CFr_t{nullptr, 0, 0}, make_str(e.what())};
  }
}
CResult_Vec_CFr_Vec_uint8_t ffi_bytes_le_to_vec_cfr(const Vec_uint8_t* bytes) { return vec_cfr_from(bytes, false); }
CResult_Vec_CFr_Vec_uint8_t ffi_bytes_be_to_vec_cfr(const Vec_uint8_t* bytes) { return vec_cfr_from(bytes, true); }
Vec_uint8_t ffi_vec_cfr_debug(const Vec_CFr_t* v) {
  if (!v) return make_str("None");
  std::string s = "[";
  for (size_t i = 0; i < v->len; i++) s += (i ? ", " : "") + cfr_dec(((const CFr*)v->ptr)[i]);
  return make_str(s + "]");
}
void ffi_vec_cfr_free(Vec_CFr_t v) {   // keygen hands the secrets out as a Vec<CFr>
  if (v.ptr) secure_zero(v.ptr, std::max(v.len, v.cap) * sizeof(CFr));   // the whole allocation, not only the live elements
  free(v.ptr);
}

static std::vector<uint8_t> vec_u8_bytes(const Vec_uint8_t* v, bool be) {  // utils.rs:158-190
  std::vector<uint8_t> b;
  put_u64(b, v->len, be);
  b.insert(b.end(), v->ptr, v->ptr + v->len);
  return b;
}
CResult_Vec_uint8_Vec_uint8_t ffi_vec_u8_to_bytes_le(const Vec_uint8_t* v) { return guard_bytes([&]() { return vec_u8_bytes(v, false); }); }
CResult_Vec_uint8_Vec_uint8_t ffi_vec_u8_to_bytes_be(const Vec_uint8_t* v) { return guard_bytes([&]() { return vec_u8_bytes(v, true); }); }
CResult_Vec_uint8_Vec_uint8_t ffi_bytes_le_to_vec_u8(const Vec_uint8_t* bytes) {
  return guard_bytes([&]() { Cursor c{bytes->ptr, bytes->len, 0, false}; return c.vec_u8(); });
}
CResult_Vec_uint8_Vec_uint8_t ffi_bytes_be_to_vec_u8(const Vec_uint8_t* bytes) {
  return guard_bytes([&]() { Cursor c{bytes->ptr, bytes->len, 0, true}; return c.vec_u8(); });
}
Vec_uint8_t ffi_vec_u8_debug(const Vec_uint8_t* v) {
  if (!v) return make_str("None");
  std::string s = "[";
  char buf[8];
  for (size_t i = 0; i < v->len; i++) {
    snprintf(buf, sizeof buf, "%x", v->ptr[i]);
    s += (i ? ", " : "") + std::string(buf);
  }
  return make_str(s + "]");
}
void ffi_vec_u8_free(Vec_uint8_t v) { free(v.ptr); }

CFr_t* ffi_hash_to_field_le(const Vec_uint8_t* input) {
  CFr r;
  hash_to_field_le(input->ptr, input->len, r.le);
  return box_cfr(r);
}
CFr_t* ffi_hash_to_field_be(const Vec_uint8_t* input) {
  CFr r;
  hash_to_field_be(input->ptr, input->len, r.le);
  return box_cfr(r);
}
// The hash / keygen entry points below return bare values in the reference (ffi_utils.rs:359-392: no CResult), so a
// device failure (no GPU, out of memory, lost device) has no error channel.  It must not look like a result: the
// functions print the error and return NULL / an empty Vec (ptr NULL, len 0), as the *_default constructors do, and
// no exception leaves the extern "C" boundary.
CFr_t* ffi_poseidon_hash_pair(const CFr_t* a, const CFr_t* b) {
  return no_channel("ffi_poseidon_hash_pair", (CFr_t*)nullptr,
                    [&]() { return box_cfr(poseidon_host_call({R(a), R(b)})); });
}
Vec_CFr_t ffi_key_gen(void) {  // keygen (protocol/keygen.rs:18-30): random secret, commitment = H(secret)
  return no_channel("ffi_key_gen", kNoVecCFr, [&]() {
    CFr secret = random_fr();
    return make_vec_cfr({secret, poseidon_host_call({secret})});
  });
}
Vec_CFr_t ffi_seeded_key_gen(const Vec_uint8_t* seed) {  // seeded_keygen (protocol/keygen.rs:50-65)
  return no_channel("ffi_seeded_key_gen", kNoVecCFr, [&]() {
    ChaCha20Rng rng = seeded_rng(seed);
    CFr secret = rng.next_fr();
    return make_vec_cfr({secret, poseidon_host_call({secret})});
  });
}
static Vec_CFr_t extended_identity(const CFr& trapdoor, const CFr& nullifier) {  // keygen.rs:31-45
  CFr secret = poseidon_host_call({trapdoor, nullifier});
  return make_vec_cfr({trapdoor, nullifier, secret, poseidon_host_call({secret})});
}
Vec_CFr_t ffi_extended_key_gen(void) {
  return no_channel("ffi_extended_key_gen", kNoVecCFr, [&]() { return extended_identity(random_fr(), random_fr()); });
}
Vec_CFr_t ffi_seeded_extended_key_gen(const Vec_uint8_t* seed) {  // extended_seeded_keygen (keygen.rs:72-94)
  return no_channel("ffi_seeded_extended_key_gen", kNoVecCFr, [&]() {
    ChaCha20Rng rng = seeded_rng(seed);
    CFr trapdoor = rng.next_fr();
    CFr nullifier = rng.next_fr();
    return extended_identity(trapdoor, nullifier);
  });
}

CResult_CFr_ptr_Vec_uint8_t ffi_compute_id_secret(const CFr_t* share1_x, const CFr_t* share1_y, const CFr_t* share2_x,
                                                  const CFr_t* share2_y) {
  return guard_ptr<CResult_CFr_ptr_Vec_uint8_t>(
      [&]() -> CFr_t* { return box_cfr(compute_id_secret(R(share1_x), R(share1_y), R(share2_x), R(share2_y))); });
}
CResult_CFr_ptr_Vec_uint8_t ffi_recover_id_secret(FFI_RLNProofValues_t* const* pv1, FFI_RLNProofValues_t* const* pv2) {
  return guard_ptr<CResult_CFr_ptr_Vec_uint8_t>([&]() -> CFr_t* {  // recover_id_secret (slashing.rs:43-100)
    const FFI_RLNProofValues &a = PV(pv1), &b = PV(pv2);
    if (cfr_cmp(a.external_nullifier, b.external_nullifier) != 0)
      throw Error("External nullifiers mismatch: " + cfr_dec(a.external_nullifier) + " != " +
                  cfr_dec(b.external_nullifier));
    if (!a.multi && !b.multi) return box_cfr(compute_id_secret(a.x, a.y, b.x, b.y));
    if (a.multi && b.multi)
      for (size_t i = 0; i < a.nullifiers.size(); i++) {
        if (!a.selector_used[i]) continue;
        for (size_t j = 0; j < b.nullifiers.size(); j++)
          if (b.selector_used[j] && cfr_cmp(a.nullifiers[i], b.nullifiers[j]) == 0)
            return box_cfr(compute_id_secret(a.x, a.ys[i], b.x, b.ys[j]));
      }
    throw Error("No matching nullifier found across the provided proof values");
  });
}

CResult_Vec_uint8_Vec_uint8_t ffi_rln_witness_to_bigint_json(FFI_RLNWitnessInput_t* const* wp) {
  // rln_witness_to_bigint_json (witness.rs:317-366); serde_json's default map is a BTreeMap: keys sorted, compact
  const FFI_RLNWitnessInput& w = W(wp);
  std::vector<std::string> pe, pi, ids, sel;
  for (auto& e : w.path_elements) pe.push_back(cfr_dec(e));
  for (uint8_t b : w.identity_path_index) pi.push_back(std::to_string((unsigned)b));
  for (auto& e : w.message_ids) ids.push_back(cfr_dec(e));
  for (uint8_t b : w.selector_used) sel.push_back(b ? "1" : "0");
  std::string j = "{\"externalNullifier\":\"" + cfr_dec(w.external_nullifier) + "\",\"identityPathIndex\":" +
                  json_str_array(pi) + ",\"identitySecret\":\"" + cfr_dec(w.identity_secret) + "\",\"messageId\":" +
                  (w.multi ? json_str_array(ids) : "\"" + cfr_dec(w.message_id) + "\"") + ",\"pathElements\":" +
                  json_str_array(pe) + (w.multi ? ",\"selectorUsed\":" + json_str_array(sel) : "") +
                  ",\"userMessageLimit\":\"" + cfr_dec(w.user_message_limit) + "\",\"x\":\"" + cfr_dec(w.x) + "\"}";
  return {make_str(j), no_err()};
}

CResult_FFI_RLNProof_ptr_Vec_uint8_t ffi_generate_rln_proof_with_witness(FFI_RLN_t* const* rln,
                                                                         const Vec_String_t* calculated_witness,
                                                                         FFI_RLNWitnessInput_t* const* witness) {
  return guard_ptr<CResult_FFI_RLNProof_ptr_Vec_uint8_t>([&]() -> FFI_RLNProof_t* {
    return (FFI_RLNProof_t*)prove_with_witness(*(FFI_RLN*)*rln, calculated_witness, W(witness));
  });
}
void ffi_c_string_free(Vec_uint8_t s) { free(s.ptr); }

}  // extern "C"

#include "ffi_v3.inc"
