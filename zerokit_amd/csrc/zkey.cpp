#include "zkey.h"

#include <string.h>

#include "common.h"

namespace rlnamd {
namespace {
struct Reader {
  const uint8_t* d;
  size_t n, o = 0;
  void need(size_t k) {
    if (o + k > n || o + k < o) throw Error("SerializationError: unexpected end of data");
  }
  uint64_t u64() {
    need(8);
    uint64_t v;
    memcpy(&v, d + o, 8);
    o += 8;
    return v;
  }
  // 32-byte LE field element; flags (top two bits of the last byte) returned separately
  template <class F>
  F fp(uint8_t* flags = nullptr) {
    need(32);
    uint32_t c[8];
    memcpy(c, d + o, 32);
    o += 32;
    if (flags) {
      *flags = (uint8_t)((c[7] >> 24) & 0xC0);
      c[7] &= 0x3FFFFFFFu;
    }
    return F::from_canonical(c);
  }
  G1Affine g1() {
    Fq x = fp<Fq>();
    uint8_t fl;
    Fq y = fp<Fq>(&fl);
    if (fl & 0x40) return G1Affine::inf();
    return {x, y};
  }
  G2Affine g2() {
    Fq x0 = fp<Fq>(), x1 = fp<Fq>(), y0 = fp<Fq>();
    uint8_t fl;
    Fq y1 = fp<Fq>(&fl);
    if (fl & 0x40) return G2Affine::inf();
    return {{x0, x1}, {y0, y1}};
  }
  size_t count(size_t elem_bytes) {
    uint64_t k = u64();
    if (k > (n - o) / (elem_bytes ? elem_bytes : 1)) throw Error("SerializationError: vector length exceeds input");
    return (size_t)k;
  }
};
}  // namespace

Zkey parse_arkzkey(const uint8_t* data, size_t len) {
  if (!data || len == 0) throw Error("Empty zkey bytes");  // ZKeyReadError::EmptyBytes
  Reader r{data, len};
  Zkey z;
  // VerifyingKey { alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 }
  z.alpha_g1 = r.g1();
  z.beta_g2 = r.g2();
  z.gamma_g2 = r.g2();
  z.delta_g2 = r.g2();
  size_t k = r.count(64);
  for (size_t i = 0; i < k; i++) z.gamma_abc_g1.push_back(r.g1());
  // ProvingKey { vk, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query }
  z.beta_g1 = r.g1();
  z.delta_g1 = r.g1();
  k = r.count(64);
  z.a_query.reserve(k);
  for (size_t i = 0; i < k; i++) z.a_query.push_back(r.g1());
  k = r.count(64);
  z.b_g1_query.reserve(k);
  for (size_t i = 0; i < k; i++) z.b_g1_query.push_back(r.g1());
  k = r.count(128);
  z.b_g2_query.reserve(k);
  for (size_t i = 0; i < k; i++) z.b_g2_query.push_back(r.g2());
  k = r.count(64);
  z.h_query.reserve(k);
  for (size_t i = 0; i < k; i++) z.h_query.push_back(r.g1());
  k = r.count(64);
  z.l_query.reserve(k);
  for (size_t i = 0; i < k; i++) z.l_query.push_back(r.g1());
  // SerializableConstraintMatrices (circuit/mod.rs:259-270)
  z.num_instance_variables = r.u64();
  z.num_witness_variables = r.u64();
  z.num_constraints = r.u64();
  z.a_nnz = r.u64();
  z.b_nnz = r.u64();
  z.c_nnz = r.u64();
  auto matrix = [&](std::vector<SparseRow>& m) {
    size_t rows = r.count(8);
    m.resize(rows);
    for (size_t i = 0; i < rows; i++) {
      size_t e = r.count(40);
      m[i].coeff.reserve(e);
      m[i].col.reserve(e);
      for (size_t j = 0; j < e; j++) {
        m[i].coeff.push_back(r.fp<Fr>());
        uint64_t c = r.u64();
        if (c > 0xFFFFFFFFull) throw Error("SerializationError: column index out of range");
        m[i].col.push_back((uint32_t)c);
      }
    }
  };
  matrix(z.a);
  matrix(z.b);
  matrix(z.c);
  size_t nvars = z.a_query.size();
  for (auto* m : {&z.a, &z.b, &z.c})
    for (auto& row : *m)
      for (uint32_t c : row.col)
        if (c >= nvars) throw Error("SerializationError: matrix column beyond variable count");
  if (z.a.size() < z.num_constraints || z.b.size() < z.num_constraints)
    throw Error("SerializationError: fewer matrix rows than constraints");
  return z;
}

// ---------------------------------------------------------------- wtns.graph
namespace {
struct PB {
  const uint8_t* d;
  size_t n, o = 0;
  bool more() const { return o < n; }
  uint64_t varint() {
    uint64_t v = 0;
    int s = 0;
    for (;;) {
      if (o >= n || s > 63) throw Error("failed to decode Protobuf message: invalid varint");
      uint8_t c = d[o++];
      v |= (uint64_t)(c & 0x7F) << s;
      if (!(c & 0x80)) return v;
      s += 7;
    }
  }
  PB sub() {
    uint64_t ln = varint();
    if (ln > n - o) throw Error("failed to decode Protobuf message: buffer underflow");
    PB p{d + o, (size_t)ln};
    o += ln;
    return p;
  }
  void skip(int wt) {
    if (wt == 0) varint();
    else if (wt == 2) sub();
    else if (wt == 1) { if (n - o < 8) throw Error("protobuf underflow"); o += 8; }
    else if (wt == 5) { if (n - o < 4) throw Error("protobuf underflow"); o += 4; }
    else throw Error("failed to decode Protobuf message: invalid wire type");
  }
};
const char kMagic[] = "wtns.graph.001";
}  // namespace

Graph parse_graph(const uint8_t* data, size_t len) {
  if (!data || len == 0) throw Error("Empty graph bytes");  // GraphReadError::EmptyBytes
  const size_t ml = sizeof(kMagic) - 1;
  if (len < ml + 8 || memcmp(data, kMagic, ml) != 0) throw Error("Invalid magic");
  uint64_t nn;
  memcpy(&nn, data + ml, 8);
  PB r{data, len, ml + 8};
  if (nn > len) throw Error("graph node count exceeds input");
  Graph g;
  g.nodes.reserve(nn);
  for (uint64_t i = 0; i < nn; i++) {
    PB msg = r.sub();
    bool have = false;
    GNode nd{0, 0, 0, 0};
    while (msg.more()) {
      uint64_t key = msg.varint();
      int tag = (int)(key >> 3), wt = (int)(key & 7);
      if (wt != 2 || tag < 1 || tag > 5) {
        msg.skip(wt);
        continue;
      }
      PB body = msg.sub();
      have = true;
      uint32_t f[5] = {0, 0, 0, 0, 0};
      if (tag == 2) {  // ConstantNode { BigUInt value = 1 { bytes value_le = 1 } }
        uint32_t limbs[16];
        memset(limbs, 0, sizeof(limbs));
        bool has_value = false;
        while (body.more()) {
          uint64_t k2 = body.varint();
          if ((k2 >> 3) == 1 && (k2 & 7) == 2) {
            PB big = body.sub();
            has_value = true;
            while (big.more()) {
              uint64_t k3 = big.varint();
              if ((k3 >> 3) == 1 && (k3 & 7) == 2) {
                PB bytes = big.sub();
                if (bytes.n > 64) throw Error("constant wider than 512 bits");
                memcpy(limbs, bytes.d, bytes.n);
              } else {
                big.skip((int)(k3 & 7));
              }
            }
          } else {
            body.skip((int)(k2 & 7));
          }
        }
        if (!has_value) throw Error("Constant node must have a value");
        // from_le_bytes_mod_order (storage.rs:45-47): value = lo + hi * 2^256  (mod r)
        bool hi_nz = false;
        for (int q = 8; q < 16; q++) hi_nz |= limbs[q] != 0;
        Fr v;
        {
          // lo may be >= r: reduce by repeated subtraction (at most 5 times since 2^256 < 6r)
          uint32_t lo[8];
          memcpy(lo, limbs, 32);
          while (limbs_geq(lo, FrParams::MOD)) {
            uint32_t borrow = 0;
            for (int q = 0; q < 8; q++) {
              uint64_t s = (uint64_t)lo[q] - FrParams::MOD[q] - borrow;
              lo[q] = (uint32_t)s;
              borrow = (uint32_t)(s >> 63);
            }
          }
          v = Fr::from_canonical(lo);
        }
        if (hi_nz) {
          uint32_t hi[8];
          memcpy(hi, limbs + 8, 32);
          while (limbs_geq(hi, FrParams::MOD)) {
            uint32_t borrow = 0;
            for (int q = 0; q < 8; q++) {
              uint64_t s = (uint64_t)hi[q] - FrParams::MOD[q] - borrow;
              hi[q] = (uint32_t)s;
              borrow = (uint32_t)(s >> 63);
            }
          }
          // hi * 2^256 mod r: the Montgomery form of hi IS hi*2^256 mod r, read as a canonical value
          Fr hm = Fr::from_canonical(hi);      // Montgomery residue = hi * R mod r
          uint32_t as_canon[8];
          memcpy(as_canon, hm.v, 32);
          v = v + Fr::from_canonical(as_canon);
        }
        nd.op = G_CONST;
        nd.a = (uint32_t)g.constants.size();
        g.constants.push_back(v);
      } else {
        while (body.more()) {
          uint64_t k2 = body.varint();
          int t2 = (int)(k2 >> 3);
          if ((k2 & 7) == 0 && t2 >= 1 && t2 <= 4) f[t2] = (uint32_t)body.varint();
          else body.skip((int)(k2 & 7));
        }
        if (tag == 1) {
          nd = {G_INPUT, f[1], 0, 0};
        } else if (tag == 3) {
          if (f[1] > 1) throw Error("UnoOp must be valid enum value");
          nd = {f[1] == 0 ? (uint32_t)G_NEG : (uint32_t)G_ID, f[2], 0, 0};
        } else if (tag == 4) {
          if (f[1] > 19) throw Error("DuoOp must be valid enum value");
          nd = {(uint32_t)G_MUL + f[1], f[2], f[3], 0};
        } else {
          if (f[1] != 0) throw Error("TresOp must be valid enum value");
          nd = {G_TERN, f[2], f[3], f[4]};
        }
      }
    }
    if (!have) throw Error("Proto::Node must have a node field");
    // operands must refer to earlier nodes (straight-line program)
    if (nd.op >= G_MUL) {
      uint32_t lim = (uint32_t)g.nodes.size();
      bool bad = nd.a >= lim;
      if (nd.op <= G_BXOR || nd.op == G_TERN) bad |= nd.b >= lim;
      if (nd.op == G_TERN) bad |= nd.c >= lim;
      if (bad) throw Error("graph node refers to a later node");
    }
    g.nodes.push_back(nd);
  }
  // GraphMetadata { repeated uint32 witness_signals = 1; map<string, SignalDescription> inputs = 2 }
  PB md = r.sub();
  while (md.more()) {
    uint64_t key = md.varint();
    int tag = (int)(key >> 3), wt = (int)(key & 7);
    if (tag == 1 && wt == 2) {
      PB packed = md.sub();
      while (packed.more()) g.signals.push_back((uint32_t)packed.varint());
    } else if (tag == 1 && wt == 0) {
      g.signals.push_back((uint32_t)md.varint());
    } else if (tag == 2 && wt == 2) {
      PB ent = md.sub();
      std::string name;
      uint32_t off = 0, ln = 0;
      while (ent.more()) {
        uint64_t k2 = ent.varint();
        if ((k2 >> 3) == 1 && (k2 & 7) == 2) {
          PB s = ent.sub();
          name.assign((const char*)s.d, s.n);
        } else if ((k2 >> 3) == 2 && (k2 & 7) == 2) {
          PB sd = ent.sub();
          while (sd.more()) {
            uint64_t k3 = sd.varint();
            if ((k3 & 7) == 0) {
              uint32_t v = (uint32_t)sd.varint();
              if ((k3 >> 3) == 1) off = v;
              else if ((k3 >> 3) == 2) ln = v;
            } else {
              sd.skip((int)(k3 & 7));
            }
          }
        } else {
          ent.skip((int)(k2 & 7));
        }
      }
      g.input_mapping[name] = {off, ln};
    } else {
      md.skip(wt);
    }
  }
  for (uint32_t s : g.signals)
    if (s >= g.nodes.size()) throw Error("witness signal refers to a missing node");
  // get_inputs_size (iden3calc.rs:106-120)
  {
    bool start = false;
    uint32_t mx = 0;
    for (auto& nd : g.nodes) {
      if (nd.op == G_INPUT) {
        if (nd.a > mx) mx = nd.a;
        start = true;
      } else if (start) {
        break;
      }
    }
    g.inputs_size = mx + 1;
  }
  for (auto& nd : g.nodes)
    if (nd.op == G_INPUT && nd.a >= g.inputs_size) throw Error("graph input index beyond the inputs buffer");
  for (auto& kv : g.input_mapping)
    if ((uint64_t)kv.second.first + kv.second.second > g.inputs_size) throw Error("input signal beyond the inputs buffer");
  auto it = g.input_mapping.find("pathElements");          // circuit/mod.rs:163-179
  g.tree_depth = it == g.input_mapping.end() ? 0 : it->second.second;
  it = g.input_mapping.find("messageId");                  // circuit/mod.rs:181-194
  g.max_out = it == g.input_mapping.end() ? 1 : it->second.second;
  return g;
}

// ---------------------------------------------------------------- compressed points
static bool fq_is_neg(const Fq& y) {
  uint32_t c[8];
  y.to_canonical(c);
  return limbs_gt(c, FqParams::HALF);
}
static bool fq2_is_neg(const Fq2& y) {  // lexicographic, c1 first
  if (!y.c1.is_zero()) return fq_is_neg(y.c1);
  return fq_is_neg(y.c0);
}
void g1_compress(const G1Affine& p, uint8_t out[32]) {
  if (p.is_inf()) {
    memset(out, 0, 32);
    out[31] = 0x40;
    return;
  }
  uint32_t c[8];
  p.x.to_canonical(c);
  memcpy(out, c, 32);
  if (fq_is_neg(p.y)) out[31] |= 0x80;
}
void g2_compress(const G2Affine& p, uint8_t out[64]) {
  if (p.is_inf()) {
    memset(out, 0, 64);
    out[63] = 0x40;
    return;
  }
  uint32_t c[8];
  p.x.c0.to_canonical(c);
  memcpy(out, c, 32);
  p.x.c1.to_canonical(c);
  memcpy(out + 32, c, 32);
  if (fq2_is_neg(p.y)) out[63] |= 0x80;
}
static bool fq_sqrt(const Fq& a, Fq* r) {  // q = 3 mod 4: a^((q+1)/4)
  uint32_t e[8];
  // (q+1)/4
  uint32_t q1[8];
  uint64_t c = 1;
  for (int i = 0; i < 8; i++) {
    uint64_t s = (uint64_t)FqParams::MOD[i] + c;
    q1[i] = (uint32_t)s;
    c = s >> 32;
  }
  for (int i = 0; i < 8; i++) e[i] = (q1[i] >> 2) | (i < 7 ? q1[i + 1] << 30 : 0);
  Fq x = a.pow(e);
  if (x.sqr() != a) return false;
  *r = x;
  return true;
}
static bool load_fq_checked(const uint8_t* in, bool strip, Fq* out) {
  uint32_t c[8];
  memcpy(c, in, 32);
  if (strip) c[7] &= 0x3FFFFFFFu;
  if (limbs_geq(c, FqParams::MOD)) return false;
  *out = Fq::from_canonical(c);
  return true;
}
bool g1_decompress(const uint8_t in[32], G1Affine* out) {
  if (in[31] & 0x40) {
    *out = G1Affine::inf();
    return true;
  }
  Fq x;
  if (!load_fq_checked(in, true, &x)) return false;
  Fq rhs = x.sqr() * x + Fq::from_u32(3);
  Fq y;
  if (!fq_sqrt(rhs, &y)) return false;
  if (fq_is_neg(y) != ((in[31] & 0x80) != 0)) y = y.neg();
  *out = {x, y};
  return true;
}
static bool fq2_sqrt(const Fq2& a, Fq2* r) {
  // norm method: sqrt(a0 + a1 u); q = 3 mod 4
  if (a.c1.is_zero()) {
    Fq s;
    if (fq_sqrt(a.c0, &s)) {
      *r = {s, Fq::zero()};
      return true;
    }
    if (fq_sqrt(a.c0.neg(), &s)) {
      *r = {Fq::zero(), s};
      return true;
    }
    return false;
  }
  Fq n;
  if (!fq_sqrt(a.c0.sqr() + a.c1.sqr(), &n)) return false;
  Fq inv2 = Fq::from_u32(2).inv();
  for (int k = 0; k < 2; k++) {
    Fq t = (a.c0 + (k ? n.neg() : n)) * inv2;
    Fq x0;
    if (!fq_sqrt(t, &x0) || x0.is_zero()) continue;
    Fq x1 = a.c1 * x0.dbl().inv();
    Fq2 cand{x0, x1};
    if (cand.sqr() == a) {
      *r = cand;
      return true;
    }
  }
  return false;
}
bool g2_decompress(const uint8_t in[64], G2Affine* out) {
  if (in[63] & 0x40) {
    *out = G2Affine::inf();
    return true;
  }
  Fq x0, x1;
  if (!load_fq_checked(in, false, &x0) || !load_fq_checked(in + 32, true, &x1)) return false;
  Fq2 x{x0, x1};
  // b' = 3 / (9 + u)
  Fq2 xi{Fq::from_u32(9), Fq::one()};
  Fq2 b2 = xi.inv().mul_fq(Fq::from_u32(3));
  Fq2 rhs = x.sqr() * x + b2;
  Fq2 y;
  if (!fq2_sqrt(rhs, &y)) return false;
  if (fq2_is_neg(y) != ((in[63] & 0x80) != 0)) y = y.neg();
  *out = {x, y};
  return true;
}

// Membership in the order-r subgroup of the twist.  ark-ec's BN model (what `Validate::Yes` runs in the reference's
// deserialisers) tests psi(P) == [6 u^2] P, psi the untwist-Frobenius-twist endomorphism (Dai, Lin, Zhao, Zhou,
// "Fast subgroup membership testings for G1, G2 and GT on pairing-friendly curves", section 4.3): a 127-bit scalar
// instead of the 254-bit [r] P == O -- the same verdict on every point of the twist (tests/test_host_math.py checks
// both forms on subgroup and non-subgroup points).
bool g2_in_subgroup(const G2Affine& p) {
  if (p.is_inf()) return true;
  static const uint32_t SIX_U2[8] = {0xe87cfd46u, 0xf83e9682u, 0xeeb859fbu, 0x6f4d8248u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u};   // 6 u^2, u = 4965661367192848881
  auto fq2c = [](const uint32_t v[2][8]) { return Fq2{Fq::from_canonical(v[0]), Fq::from_canonical(v[1])}; };
  static const Fq2 g12 = fq2c(FROB_G12), g13 = fq2c(FROB_G13);
  const Fq2 px = p.x.conj() * g12, py = p.y.conj() * g13;   // psi(P)
  const G2XYZZ k = scalar_mul(p, SIX_U2);
  if (k.is_inf()) return false;
  return k.X == px * k.ZZ && k.Y == py * k.ZZZ;
}
bool g2_in_subgroup_by_order(const G2Affine& p) { return scalar_mul(p, FrParams::MOD).is_inf(); }

}  // namespace rlnamd
