// ffi_wire.h -- the byte formats of the zerokit C ABI: everything that parses bytes a caller (or the network behind the
// caller: a relay node hands received proofs straight to ffi_bytes_*_to_rln_proof) controls, and the serialisers that
// are their inverses.  V1 (protocol/proof.rs:192-236,413-572, protocol/witness.rs:369-760, protocol/mode.rs:9-75,
// rln/src/utils.rs:75-156) and V3 (protocol/serialize.rs:63-714; enum tags + arkworks field order).  Pure host code with
// no HIP call, in a header of its own (like tree_config.h) so that the CPU suite compiles it with AddressSanitizer /
// UBSan and fuzzes it (tests/host/sanitize_main.cpp); ffi.cpp and ffi_v3.inc include it and keep
// only the extern "C" wrappers.  The object structs the parsers fill live here too (they ARE the FFI's opaque types).
#pragma once
#include "../../include/rln.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "common.h"   // rlnamd::Error
#include "field.h"    // limbs_geq, FrParams
#include "zkey.h"     // G1Affine / G2Affine, point (de)compression, the subgroup check

using namespace rlnamd;

struct CFr {
  uint8_t le[32];
};

namespace {

bool is_canonical(const uint8_t* le) {
  uint32_t c[8];
  memcpy(c, le, 32);
  return !limbs_geq(c, FrParams::MOD);
}
bool cfr_is_zero(const CFr& a) {
  for (int i = 0; i < 32; i++)
    if (a.le[i]) return false;
  return true;
}
int cfr_cmp(const CFr& a, const CFr& b) {
  for (int i = 31; i >= 0; i--)
    if (a.le[i] != b.le[i]) return a.le[i] > b.le[i] ? 1 : -1;
  return 0;
}
CFr cfr_from_u64(uint64_t v) {
  CFr r;
  memset(r.le, 0, 32);
  memcpy(r.le, &v, 8);
  return r;
}
std::string cfr_dec(const CFr& a) {  // decimal, as arkworks' Display/Debug for Fp prints it
  uint32_t v[8];
  memcpy(v, a.le, 32);
  std::string out;
  bool nz = true;
  while (nz) {
    uint64_t rem = 0;
    nz = false;
    for (int i = 7; i >= 0; i--) {
      uint64_t cur = (rem << 32) | v[i];
      v[i] = (uint32_t)(cur / 10);
      rem = cur % 10;
      nz |= v[i] != 0;
    }
    out.push_back((char)('0' + rem));
  }
  std::reverse(out.begin(), out.end());
  return out;
}
void be32(const uint8_t le[32], uint8_t out[32]) {
  for (int i = 0; i < 32; i++) out[i] = le[31 - i];
}
void put_u64(std::vector<uint8_t>& b, uint64_t v, bool be) {
  for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (be ? 56 - 8 * i : 8 * i)));
}
uint64_t get_u64(const uint8_t* p, bool be) {
  uint64_t v = 0;
  for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (be ? 56 - 8 * i : 8 * i);
  return v;
}
void put_fr(std::vector<uint8_t>& b, const CFr& a, bool be) {
  uint8_t t[32];
  if (be) be32(a.le, t); else memcpy(t, a.le, 32);
  b.insert(b.end(), t, t + 32);
}

struct Cursor {  // bounds-checked reader with the reference's error texts
  const uint8_t* d;
  size_t n, o = 0;
  bool be;
  void need(size_t k) {
    if (o + k > n || o + k < o)
      throw Error("Input data too short: expected at least " + std::to_string(o + k) + " bytes, got " +
                  std::to_string(n) + " bytes");
  }
  CFr fr() {
    need(32);
    CFr r;
    if (be) be32(d + o, r.le); else memcpy(r.le, d + o, 32);
    o += 32;
    if (!is_canonical(r.le)) throw Error("Non-canonical field element: value is not in [0, r-1]");
    return r;
  }
  uint64_t len() {
    need(8);
    uint64_t v = get_u64(d + o, be);
    o += 8;
    return v;
  }
  std::vector<CFr> vec_fr() {
    uint64_t k = len();
    if (k > (n - o) / 32) need((size_t)-1 - o);
    std::vector<CFr> v;
    for (uint64_t i = 0; i < k; i++) v.push_back(fr());
    return v;
  }
  std::vector<uint8_t> vec_u8() {
    uint64_t k = len();
    if (k > n - o) need((size_t)-1 - o);
    std::vector<uint8_t> v(d + o, d + o + k);
    o += k;
    return v;
  }
};
}  // namespace

// the reference's IdSecret zeroises itself when dropped (rln/src/utils.rs:440-527: Zeroize + ZeroizeOnDrop); the stores
// go through a volatile pointer so that they cannot be elided as dead
static void secure_zero(void* p, size_t n) {
  volatile uint8_t* v = (volatile uint8_t*)p;
  for (size_t i = 0; i < n; i++) v[i] = 0;
}
struct ZeroOnExit {   // host staging copies of witness inputs / (r, s) / externally computed witnesses
  std::vector<uint8_t>& v;
  ~ZeroOnExit() { secure_zero(v.data(), v.size()); }
};
struct FFI_RLNWitnessInput {  // RLNWitnessInput (protocol/witness.rs:44-58): SingleV1 or MultiV1
  ~FFI_RLNWitnessInput() { secure_zero(&identity_secret, sizeof identity_secret); }
  CFr identity_secret, user_message_limit, message_id;
  std::vector<CFr> path_elements;
  std::vector<uint8_t> identity_path_index;
  CFr x, external_nullifier;
  bool multi = false;
  std::vector<CFr> message_ids;        // MultiV1
  std::vector<uint8_t> selector_used;  // MultiV1, 0/1
};
struct FFI_RLNProofValues {  // RLNProofValues (protocol/proof.rs:40-110): SingleV1 {y, nullifier} or MultiV1
  CFr root, x, external_nullifier, y, nullifier;
  bool multi = false;
  std::vector<CFr> ys, nullifiers;
  std::vector<uint8_t> selector_used;
};
struct FFI_RLNProof {
  uint8_t proof[128];
  FFI_RLNProofValues values;
};

struct FFI_RLNPartialWitnessInput {  // RLNPartialWitnessInput (protocol/witness.rs:62-73)
  ~FFI_RLNPartialWitnessInput() { secure_zero(&identity_secret, sizeof identity_secret); }
  CFr identity_secret, user_message_limit;
  std::vector<CFr> path_elements;
  std::vector<uint8_t> identity_path_index;
};
struct FFI_RLNPartialProof {  // PartialProof (partial_proof.rs:31-43): mask + four points (affine, canonical LE)
  std::vector<uint8_t> mask;  // per assignment entry (witness signals 1..)
  uint8_t coords[320];        // pi_a | rho | pi_b | pi_c
  // Beside the wire form, never inside it: the prover's cache entry with the values the partial witness fixed
  // (rln_amd.h: rlnamd_prover_collect_partial_cached).  0 for a partial proof that came in as bytes or whose prover had no
  // room: finishing it walks the whole graph.  The entry holds witness values: released (overwritten) with the object.
  uint64_t handle = 0;
  std::function<void(uint64_t)> release;
  FFI_RLNPartialProof() = default;
  FFI_RLNPartialProof(const FFI_RLNPartialProof& o) : mask(o.mask) { memcpy(coords, o.coords, 320); }   // a copy owns no entry
  FFI_RLNPartialProof& operator=(const FFI_RLNPartialProof&) = delete;
  ~FFI_RLNPartialProof() {
    if (handle && release) {
      try { release(handle); } catch (...) {}
    }
  }
};

namespace {

void validate_witness(const FFI_RLNWitnessInput& w) {  // witness.rs:78-108
  if (cfr_is_zero(w.user_message_limit)) throw Error("User message limit cannot be zero");
  if (w.path_elements.size() != w.identity_path_index.size())
    throw Error("Merkle proof length mismatch: expected " + std::to_string(w.path_elements.size()) + ", got " +
                std::to_string(w.identity_path_index.size()));
  if (!w.multi) {
    if (cfr_cmp(w.message_id, w.user_message_limit) >= 0)
      throw Error("Message id (" + cfr_dec(w.message_id) + ") is not within user_message_limit (" +
                  cfr_dec(w.user_message_limit) + ")");
    return;
  }
  if (w.message_ids.empty()) throw Error("The field message_ids must contain at least one message_id");
  if (w.selector_used.size() != w.message_ids.size())
    throw Error("The field message_ids has length " + std::to_string(w.message_ids.size()) +
                ", but the field selector_used has length " + std::to_string(w.selector_used.size()));
  bool any = false;
  for (uint8_t b : w.selector_used) any |= b != 0;
  if (!any) throw Error("At least one selector_used value must be true");
  for (size_t i = 0; i < w.message_ids.size(); i++) {
    if (!w.selector_used[i]) continue;
    for (size_t j = 0; j < i; j++)
      if (w.selector_used[j] && cfr_cmp(w.message_ids[i], w.message_ids[j]) == 0)
        throw Error("Duplicate message ID found in message_ids");
  }
  for (size_t i = 0; i < w.message_ids.size(); i++)
    if (w.selector_used[i] && cfr_cmp(w.message_ids[i], w.user_message_limit) >= 0)
      throw Error("Message id (" + cfr_dec(w.message_ids[i]) + ") is not within user_message_limit (" +
                  cfr_dec(w.user_message_limit) + ")");
}

void put_vec_fr(std::vector<uint8_t>& b, const std::vector<CFr>& v, bool be) {  // utils.rs:123-156
  put_u64(b, v.size(), be);
  for (auto& e : v) put_fr(b, e, be);
}
void put_vec_bool(std::vector<uint8_t>& b, const std::vector<uint8_t>& v, bool be) {  // utils.rs: vec_bool_to_bytes
  put_u64(b, v.size(), be);
  for (uint8_t x : v) b.push_back(x ? 1 : 0);
}
std::vector<uint8_t> read_vec_bool(Cursor& c) {
  std::vector<uint8_t> raw = c.vec_u8();
  for (uint8_t x : raw)
    if (x > 1) {
      char buf[8];
      snprintf(buf, sizeof buf, "%#04x", x);
      throw Error(std::string("Non-canonical bool byte: expected 0x00 or 0x01, got ") + buf);
    }
  return raw;
}
std::vector<uint8_t> partial_proof_bytes(const FFI_RLNPartialProof& pp) {  // proof.rs:535-547 (always LE)
  std::vector<uint8_t> b;
  b.push_back(0x00);
  put_u64(b, pp.mask.size(), false);
  b.insert(b.end(), pp.mask.begin(), pp.mask.end());
  auto fq = [&](int k) {
    uint32_t c[8];
    memcpy(c, pp.coords + 32 * k, 32);
    return Fq::from_canonical(c);
  };
  uint8_t buf[64];
  g1_compress(G1Affine{fq(0), fq(1)}, buf);
  b.insert(b.end(), buf, buf + 32);
  g1_compress(G1Affine{fq(2), fq(3)}, buf);
  b.insert(b.end(), buf, buf + 32);
  g2_compress(G2Affine{{fq(4), fq(5)}, {fq(6), fq(7)}}, buf);
  b.insert(b.end(), buf, buf + 64);
  g1_compress(G1Affine{fq(8), fq(9)}, buf);
  b.insert(b.end(), buf, buf + 32);
  return b;
}
FFI_RLNPartialProof* partial_proof_from_bytes(const Vec_uint8_t* bytes) {  // proof.rs:552-572
  if (!bytes || bytes->len == 0) throw Error("Expected to read 1 bytes but read 0 bytes");
  const uint8_t* d = bytes->ptr;
  if (d[0] > 1) {
    char buf[8];
    snprintf(buf, sizeof buf, "%#04x", d[0]);
    throw Error(std::string("Unknown message mode version byte: ") + buf);
  }
  Cursor c{d, bytes->len, 1, false};
  uint64_t k = c.len();
  if (k > bytes->len) c.need((size_t)-1 - c.o);
  c.need((size_t)k + 160);
  std::unique_ptr<FFI_RLNPartialProof> pp(new FFI_RLNPartialProof);
  pp->mask.assign(d + c.o, d + c.o + k);
  for (uint8_t m : pp->mask)
    if (m > 1) throw Error("Proof serialization error: the input buffer contained invalid data");
  c.o += k;
  G1Affine a, rho, pc;
  G2Affine pb;
  if (!g1_decompress(d + c.o, &a) || !g1_decompress(d + c.o + 32, &rho) || !g2_decompress(d + c.o + 64, &pb) ||
      !g1_decompress(d + c.o + 128, &pc) || !g2_in_subgroup(pb))
    throw Error("Proof serialization error: the input buffer contained invalid data");
  c.o += 160;
  if (c.o != bytes->len)
    throw Error("Expected to read " + std::to_string(c.o) + " bytes but read " + std::to_string(bytes->len) + " bytes");
  auto st = [&](int k2, const Fq& v) {
    uint32_t w[8];
    v.to_canonical(w);
    memcpy(pp->coords + 32 * k2, w, 32);
  };
  st(0, a.x); st(1, a.y); st(2, rho.x); st(3, rho.y);
  st(4, pb.x.c0); st(5, pb.x.c1); st(6, pb.y.c0); st(7, pb.y.c1);
  st(8, pc.x); st(9, pc.y);
  return pp.release();
}

std::vector<uint8_t> values_bytes(const FFI_RLNProofValues& v, bool be) {  // proof.rs:192-236 / :239-283
  std::vector<uint8_t> b;
  b.push_back(v.multi ? 0x01 : 0x00);
  put_fr(b, v.root, be);
  put_fr(b, v.external_nullifier, be);
  put_fr(b, v.x, be);
  if (!v.multi) {
    put_fr(b, v.y, be);
    put_fr(b, v.nullifier, be);
  } else {
    put_vec_fr(b, v.ys, be);
    put_vec_fr(b, v.nullifiers, be);
    put_vec_bool(b, v.selector_used, be);
  }
  return b;
}
FFI_RLNProofValues values_from(Cursor& c) {  // proof.rs:285-411
  c.need(1);
  uint8_t ver = c.d[c.o++];
  if (ver > 0x01) {
    char buf[8];
    snprintf(buf, sizeof buf, "%#04x", ver);
    throw Error(std::string("Unknown message mode version byte: ") + buf);
  }
  FFI_RLNProofValues v;
  v.root = c.fr();
  v.external_nullifier = c.fr();
  v.x = c.fr();
  if (ver == 0x00) {
    v.y = c.fr();
    v.nullifier = c.fr();
  } else {
    v.multi = true;
    v.ys = c.vec_fr();
    v.nullifiers = c.vec_fr();
    v.selector_used = read_vec_bool(c);
    if (v.selector_used.size() != v.ys.size())
      throw Error("The field ys has length " + std::to_string(v.ys.size()) + ", but the field selector_used has length " +
                  std::to_string(v.selector_used.size()));
    if (v.nullifiers.size() != v.ys.size())
      throw Error("The field ys has length " + std::to_string(v.ys.size()) + ", but the field nullifiers has length " +
                  std::to_string(v.nullifiers.size()));
  }
  return v;
}
std::vector<uint8_t> proof_bytes(const FFI_RLNProof& pr, bool be) {  // proof.rs:413-449
  std::vector<uint8_t> b;
  b.push_back(pr.values.multi ? 0x01 : 0x00);
  b.insert(b.end(), pr.proof, pr.proof + 128);
  auto v = values_bytes(pr.values, be);
  b.insert(b.end(), v.begin(), v.end());
  return b;
}
FFI_RLNProof* proof_from_bytes(const Vec_uint8_t* bytes, bool be) {  // proof.rs:455-530
  if (!bytes || bytes->len == 0) throw Error("Expected to read 1 bytes but read 0 bytes");
  const uint8_t* d = bytes->ptr;
  if (d[0] > 1) {
    char buf[8];
    snprintf(buf, sizeof buf, "%#04x", d[0]);
    throw Error(std::string("Unknown message mode version byte: ") + buf);
  }
  if (bytes->len < 129)
    throw Error("Expected to read 129 bytes but read " + std::to_string(bytes->len) + " bytes");
  std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
  memcpy(pr->proof, d + 1, 128);
  G1Affine A, C;
  G2Affine B;
  if (!g1_decompress(pr->proof, &A) || !g2_decompress(pr->proof + 32, &B) || !g1_decompress(pr->proof + 96, &C) ||
      !g2_in_subgroup(B))
    throw Error("Proof serialization error: the input buffer contained invalid data");
  Cursor c{d, bytes->len, 129, be};
  pr->values = values_from(c);
  if (c.o != bytes->len)
    throw Error("Expected to read " + std::to_string(c.o) + " bytes but read " + std::to_string(bytes->len) + " bytes");
  return pr.release();
}
std::vector<uint8_t> witness_bytes(const FFI_RLNWitnessInput& w, bool be) {  // witness.rs:369-468, mode.rs:27-35
  std::vector<uint8_t> b;
  b.push_back(w.multi ? 0x01 : 0x00);
  put_fr(b, w.identity_secret, be);
  put_fr(b, w.user_message_limit, be);
  if (!w.multi) put_fr(b, w.message_id, be);
  put_vec_fr(b, w.path_elements, be);
  put_u64(b, w.identity_path_index.size(), be);
  b.insert(b.end(), w.identity_path_index.begin(), w.identity_path_index.end());
  put_fr(b, w.x, be);
  put_fr(b, w.external_nullifier, be);
  if (w.multi) {
    put_vec_fr(b, w.message_ids, be);
    put_vec_bool(b, w.selector_used, be);
  }
  return b;
}
FFI_RLNWitnessInput* witness_from_bytes(const Vec_uint8_t* bytes, bool be) {  // witness.rs:470-620
  if (!bytes || bytes->len == 0) throw Error("Expected to read 1 bytes but read 0 bytes");
  Cursor c{bytes->ptr, bytes->len, 0, be};
  uint8_t ver = c.d[c.o++];
  if (ver > 0x01) {
    char buf[8];
    snprintf(buf, sizeof buf, "%#04x", ver);
    throw Error(std::string("Unknown message mode version byte: ") + buf);
  }
  std::unique_ptr<FFI_RLNWitnessInput> w(new FFI_RLNWitnessInput);
  w->multi = ver == 0x01;
  w->identity_secret = c.fr();
  w->user_message_limit = c.fr();
  if (!w->multi) w->message_id = c.fr();
  w->path_elements = c.vec_fr();
  w->identity_path_index = c.vec_u8();
  w->x = c.fr();
  w->external_nullifier = c.fr();
  if (w->multi) {
    w->message_ids = c.vec_fr();
    w->selector_used = read_vec_bool(c);
  }
  if (c.o != bytes->len)
    throw Error("Expected to read " + std::to_string(c.o) + " bytes but read " + std::to_string(bytes->len) + " bytes");
  validate_witness(*w);
  return w.release();
}

void validate_partial_witness(const FFI_RLNPartialWitnessInput& w) {  // witness.rs:253-270
  if (cfr_is_zero(w.user_message_limit)) throw Error("User message limit cannot be zero");
  if (w.path_elements.size() != w.identity_path_index.size())
    throw Error("Merkle proof length mismatch: expected " + std::to_string(w.path_elements.size()) + ", got " +
                std::to_string(w.identity_path_index.size()));
}
std::vector<uint8_t> partial_witness_bytes(const FFI_RLNPartialWitnessInput& w, bool be) {  // witness.rs:631-676
  std::vector<uint8_t> b;
  b.push_back(0x00);
  put_fr(b, w.identity_secret, be);
  put_fr(b, w.user_message_limit, be);
  put_vec_fr(b, w.path_elements, be);
  put_u64(b, w.identity_path_index.size(), be);
  b.insert(b.end(), w.identity_path_index.begin(), w.identity_path_index.end());
  return b;
}
FFI_RLNPartialWitnessInput* partial_witness_from_bytes(const Vec_uint8_t* bytes, bool be) {  // witness.rs:679-760
  if (!bytes || bytes->len == 0) throw Error("Expected to read 1 bytes but read 0 bytes");
  Cursor c{bytes->ptr, bytes->len, 0, be};
  uint8_t ver = c.d[c.o++];
  if (ver > 0x01) {
    char buf[8];
    snprintf(buf, sizeof buf, "%#04x", ver);
    throw Error(std::string("Unknown message mode version byte: ") + buf);
  }
  std::unique_ptr<FFI_RLNPartialWitnessInput> w(new FFI_RLNPartialWitnessInput);
  w->identity_secret = c.fr();
  w->user_message_limit = c.fr();
  w->path_elements = c.vec_fr();
  w->identity_path_index = c.vec_u8();
  if (c.o != bytes->len)
    throw Error("Expected to read " + std::to_string(c.o) + " bytes but read " + std::to_string(bytes->len) + " bytes");
  validate_partial_witness(*w);
  return w.release();
}

// =================================================================================== V3 forms (serialize.rs)
// ---- readers with the two error vocabularies
struct V3Reader {
  const uint8_t* d;
  size_t n, o;
  bool be;  // BE = CanonicalDeserializeBE (SerializationErrorV3 texts); LE = arkworks derive (SerializationError)
  [[noreturn]] void eof() const {
    throw Error(be ? "I/O error: failed to fill whole buffer"
                   : "I/O error: Error { kind: UnexpectedEof, message: \"failed to fill whole buffer\" }");
  }
  [[noreturn]] void invalid() const {
    throw Error(be ? "Arkworks canonical serialization error: the input buffer contained invalid data"
                   : "the input buffer contained invalid data");
  }
  void need(size_t k) const {
    if (k > n - o) eof();
  }
  uint8_t u8() {
    need(1);
    return d[o++];
  }
  uint64_t len() {
    need(8);
    uint64_t v = get_u64(d + o, be);
    o += 8;
    return v;
  }
  CFr fr() {
    need(32);
    CFr r;
    if (be) be32(d + o, r.le); else memcpy(r.le, d + o, 32);
    o += 32;
    if (!is_canonical(r.le)) {
      if (be) throw Error("Non-canonical field element: value is not in [0, r-1]");
      invalid();
    }
    return r;
  }
  std::vector<CFr> vec_fr() {
    uint64_t k = len();
    std::vector<CFr> v;
    for (uint64_t i = 0; i < k; i++) v.push_back(fr());
    return v;
  }
  std::vector<uint8_t> vec_u8() {
    uint64_t k = len();
    if (k > n - o) eof();
    std::vector<uint8_t> v(d + o, d + o + k);
    o += k;
    return v;
  }
  std::vector<uint8_t> vec_bool() {
    std::vector<uint8_t> v;
    if (be) {  // reads the whole run, then checks (serialize.rs:246-263)
      v = vec_u8();
      for (uint8_t b : v)
        if (b > 1) {
          char buf[8];
          snprintf(buf, sizeof buf, "%#04x", b);
          throw Error(std::string("Non-canonical bool byte: expected 0x00 or 0x01, got ") + buf);
        }
    } else {
      uint64_t k = len();
      for (uint64_t i = 0; i < k; i++) {
        uint8_t b = u8();
        if (b > 1) invalid();
        v.push_back(b);
      }
    }
    return v;
  }
};

// ---- witness (RLNWitnessInputV3: enum tag, then Single / Multi; LE follows the struct declaration order of
// witness.rs:1246-1267 (message_id last), BE the hand-written order of serialize.rs:349-445 (message_id third))
std::vector<uint8_t> v3_witness_bytes(const FFI_RLNWitnessInput& w, bool be) {
  std::vector<uint8_t> b;
  b.push_back(w.multi ? 1 : 0);
  put_fr(b, w.identity_secret, be);
  put_fr(b, w.user_message_limit, be);
  if (!w.multi && be) put_fr(b, w.message_id, be);
  put_vec_fr(b, w.path_elements, be);
  put_u64(b, w.identity_path_index.size(), be);
  b.insert(b.end(), w.identity_path_index.begin(), w.identity_path_index.end());
  put_fr(b, w.x, be);
  put_fr(b, w.external_nullifier, be);
  if (!w.multi && !be) put_fr(b, w.message_id, be);
  if (w.multi) {
    put_vec_fr(b, w.message_ids, be);
    put_vec_bool(b, w.selector_used, be);
  }
  return b;
}
FFI_RLNWitnessInput* v3_witness_from(const Vec_uint8_t* bytes, bool be) {  // trailing bytes are accepted
  V3Reader c{bytes ? bytes->ptr : nullptr, bytes ? bytes->len : 0, 0, be};
  uint8_t tag = c.u8();
  if (tag > 1) c.invalid();
  std::unique_ptr<FFI_RLNWitnessInput> w(new FFI_RLNWitnessInput);
  w->multi = tag == 1;
  w->message_id = cfr_from_u64(0);
  w->identity_secret = c.fr();
  w->user_message_limit = c.fr();
  if (!w->multi && be) w->message_id = c.fr();
  w->path_elements = c.vec_fr();
  w->identity_path_index = c.vec_u8();
  w->x = c.fr();
  w->external_nullifier = c.fr();
  if (!w->multi && !be) w->message_id = c.fr();
  if (w->multi) {
    w->message_ids = c.vec_fr();
    w->selector_used = c.vec_bool();
  }
  return w.release();  // no semantic validation on this path (Valid::check is field-wise only)
}

std::vector<uint8_t> v3_partial_witness_bytes(const FFI_RLNPartialWitnessInput& w, bool be) {  // no tag
  std::vector<uint8_t> b;
  put_fr(b, w.identity_secret, be);
  put_fr(b, w.user_message_limit, be);
  put_vec_fr(b, w.path_elements, be);
  put_u64(b, w.identity_path_index.size(), be);
  b.insert(b.end(), w.identity_path_index.begin(), w.identity_path_index.end());
  return b;
}
FFI_RLNPartialWitnessInput* v3_partial_witness_from(const Vec_uint8_t* bytes, bool be) {
  V3Reader c{bytes ? bytes->ptr : nullptr, bytes ? bytes->len : 0, 0, be};
  std::unique_ptr<FFI_RLNPartialWitnessInput> w(new FFI_RLNPartialWitnessInput);
  w->identity_secret = c.fr();
  w->user_message_limit = c.fr();
  w->path_elements = c.vec_fr();
  w->identity_path_index = c.vec_u8();
  return w.release();
}

// ---- proof values (RLNProofValuesV3: tag, then y root nullifier x ext | ys root nullifiers x ext selector_used,
// proof.rs:981-1060; the same order in LE and BE)
void v3_put_values(std::vector<uint8_t>& b, const FFI_RLNProofValues& v, bool be) {
  b.push_back(v.multi ? 1 : 0);
  if (!v.multi) {
    put_fr(b, v.y, be);
    put_fr(b, v.root, be);
    put_fr(b, v.nullifier, be);
    put_fr(b, v.x, be);
    put_fr(b, v.external_nullifier, be);
  } else {
    put_vec_fr(b, v.ys, be);
    put_fr(b, v.root, be);
    put_vec_fr(b, v.nullifiers, be);
    put_fr(b, v.x, be);
    put_fr(b, v.external_nullifier, be);
    put_vec_bool(b, v.selector_used, be);
  }
}
FFI_RLNProofValues v3_values_from(V3Reader& c) {
  uint8_t tag = c.u8();
  if (tag > 1) c.invalid();
  FFI_RLNProofValues v;
  v.multi = tag == 1;
  if (!v.multi) {
    v.y = c.fr();
    v.root = c.fr();
    v.nullifier = c.fr();
    v.x = c.fr();
    v.external_nullifier = c.fr();
  } else {
    v.ys = c.vec_fr();
    v.root = c.fr();
    v.nullifiers = c.vec_fr();
    v.x = c.fr();
    v.external_nullifier = c.fr();
    v.selector_used = c.vec_bool();
  }
  return v;
}

// ---- RLNProofV3 = Proof (128 B compressed, always LE) || values (LE, or BE in the "mixed" form)
std::vector<uint8_t> v3_proof_bytes(const FFI_RLNProof& pr, bool values_be) {
  std::vector<uint8_t> b(pr.proof, pr.proof + 128);
  v3_put_values(b, pr.values, values_be);
  return b;
}
FFI_RLNProof* v3_proof_from(const Vec_uint8_t* bytes, bool values_be) {
  V3Reader c{bytes ? bytes->ptr : nullptr, bytes ? bytes->len : 0, 0, false};
  const char* pre = values_be ? "Arkworks canonical serialization error: " : "";
  if (c.n < 128)
    throw Error(std::string(pre) + "I/O error: Error { kind: UnexpectedEof, message: \"failed to fill whole buffer\" }");
  std::unique_ptr<FFI_RLNProof> pr(new FFI_RLNProof);
  memcpy(pr->proof, c.d, 128);
  G1Affine A, C;
  G2Affine B;
  if (!g1_decompress(pr->proof, &A) || !g2_decompress(pr->proof + 32, &B) || !g1_decompress(pr->proof + 96, &C) ||
      !g2_in_subgroup(B))
    throw Error(std::string(pre) + "the input buffer contained invalid data");
  V3Reader vc{c.d, c.n, 128, values_be};
  pr->values = v3_values_from(vc);
  return pr.release();
}

// ---- PartialProof: the arkworks derive without the V1 version byte (partial_proof.rs:31-43)
std::vector<uint8_t> v3_partial_proof_bytes(const FFI_RLNPartialProof& pp) {
  std::vector<uint8_t> b = partial_proof_bytes(pp);
  b.erase(b.begin());
  return b;
}
FFI_RLNPartialProof* v3_partial_proof_from(const Vec_uint8_t* bytes) {
  std::vector<uint8_t> tmp(1, 0x00);
  if (bytes && bytes->len) tmp.insert(tmp.end(), bytes->ptr, bytes->ptr + bytes->len);
  // arkworks readers stop after the last field: cut trailing bytes before reusing the exact-length V1 parser
  if (tmp.size() >= 9) {
    uint64_t k = get_u64(tmp.data() + 1, false);
    if (k <= tmp.size() && 9 + k + 160 < tmp.size()) tmp.resize(9 + (size_t)k + 160);
  }
  Vec_uint8_t v{tmp.data(), tmp.size(), tmp.size()};
  try {
    return partial_proof_from_bytes(&v);
  } catch (const Error& e) {
    std::string m = e.what();
    if (m.find("invalid data") != std::string::npos) throw Error("the input buffer contained invalid data");
    throw Error("I/O error: Error { kind: UnexpectedEof, message: \"failed to fill whole buffer\" }");
  }
}

// ---- validation with the V3 texts
void v3_validate_common(const CFr& limit, size_t path_len, size_t index_len) {  // error.rs:125-165
  if (cfr_is_zero(limit)) throw Error("User message limit cannot be zero");
  if (path_len != index_len)
    throw Error("Field `path_elements` has length " + std::to_string(path_len) +
                ", but field `identity_path_index` has length " + std::to_string(index_len));
}
void v3_validate_witness(const FFI_RLNWitnessInput& w) {  // witness.rs:1016-1107
  v3_validate_common(w.user_message_limit, w.path_elements.size(), w.identity_path_index.size());
  auto bad_id = [&](const CFr& id) {
    return Error("Message id (" + cfr_dec(id) + ") is not within user_message_limit (" + cfr_dec(w.user_message_limit) + ")");
  };
  if (!w.multi) {
    if (cfr_cmp(w.message_id, w.user_message_limit) >= 0) throw bad_id(w.message_id);
    return;
  }
  if (w.message_ids.empty()) throw Error("The field `message_ids` must contain at least one message_id");
  if (w.selector_used.size() != w.message_ids.size())
    throw Error("Field `message_ids` has length " + std::to_string(w.message_ids.size()) +
                ", but field `selector_used` has length " + std::to_string(w.selector_used.size()));
  bool any = false;
  for (uint8_t b : w.selector_used) any |= b != 0;
  if (!any) throw Error("At least one value in `selector_used` must be true");
  for (size_t i = 0; i < w.message_ids.size(); i++)
    for (size_t j = 0; w.selector_used[i] && j < i; j++)
      if (w.selector_used[j] && cfr_cmp(w.message_ids[i], w.message_ids[j]) == 0)
        throw Error("Duplicate message ID found in `message_ids`");
  for (size_t i = 0; i < w.message_ids.size(); i++)
    if (w.selector_used[i] && cfr_cmp(w.message_ids[i], w.user_message_limit) >= 0) throw bad_id(w.message_ids[i]);
}
}  // namespace
