// Host parsers for the two circuit resources the reference embeds (rln/src/circuit/mod.rs:29-42):
//   * `.arkzkey`  -- read_arkzkey_from_bytes_uncompressed, rln/src/circuit/mod.rs:277-305 (arkworks
//     uncompressed, unchecked ProvingKey<Bn254> + SerializableConstraintMatrices :256-275)
//   * `graph.bin` -- deserialize_witnesscalc_graph, rln/src/circuit/iden3calc/storage.rs:265-302
//     (schema iden3calc/proto.rs:7-117)
// Points and scalars come out as Montgomery residues ready to upload.
#pragma once
#include <stdint.h>

#include <map>
#include <string>
#include <memory>
#include <vector>

#include "curve.h"

namespace rlnamd {

struct SparseRow {
  std::vector<Fr> coeff;
  std::vector<uint32_t> col;
};

struct Zkey {
  G1Affine alpha_g1, beta_g1, delta_g1;
  G2Affine beta_g2, gamma_g2, delta_g2;
  std::vector<G1Affine> gamma_abc_g1, a_query, b_g1_query, h_query, l_query;
  std::vector<G2Affine> b_g2_query;
  uint64_t num_instance_variables = 0, num_witness_variables = 0, num_constraints = 0;
  uint64_t a_nnz = 0, b_nnz = 0, c_nnz = 0;
  std::vector<SparseRow> a, b, c;
  // verifier-side precomputation (pairing.h: prepare_vk), built on first use; copies of a key share it
  mutable std::shared_ptr<const void> prepared_vk;
};
// throws Error("...") on malformed input (ZKeyReadError in the reference, circuit/error.rs)
Zkey parse_arkzkey(const uint8_t* data, size_t len);

// iden3calc/graph.rs:36-60 + proto.rs:88-117, flattened to one opcode space
enum GOp : uint32_t {
  G_INPUT = 0, G_CONST = 1,
  G_MUL = 2, G_DIV, G_ADD, G_SUB, G_POW, G_IDIV, G_MOD, G_EQ, G_NEQ, G_LT, G_GT, G_LEQ, G_GEQ, G_LAND, G_LOR,
  G_SHL, G_SHR, G_BOR, G_BAND, G_BXOR,   // 2 + DuoOp value
  G_NEG = 22, G_ID = 23, G_TERN = 24,
};
struct GNode {
  uint32_t op, a, b, c;
};
struct Graph {
  std::vector<GNode> nodes;
  std::vector<Fr> constants;                // Montgomery; G_CONST.a indexes this
  std::vector<uint32_t> signals;            // witness_signals
  std::map<std::string, std::pair<uint32_t, uint32_t>> input_mapping;  // name -> (offset, len)
  uint32_t tree_depth = 0, max_out = 1, inputs_size = 0;
};
Graph parse_graph(const uint8_t* data, size_t len);

// arkworks compressed point encoding (flags in the top byte: 0x80 = y is the larger root, 0x40 = infinity;
// Fq2 ordered c1-first).  Used for the 128-byte proof (COMPRESS_PROOF_SIZE, circuit/mod.rs:82).
void g1_compress(const G1Affine& p, uint8_t out[32]);
void g2_compress(const G2Affine& p, uint8_t out[64]);
bool g1_decompress(const uint8_t in[32], G1Affine* out);   // false: x not on curve / non-canonical
bool g2_decompress(const uint8_t in[64], G2Affine* out);
// [r]P == 0: membership in the order-r subgroup of the twist (cofactor != 1 on G2).  ark-serialize Validate::Yes
// performs this check on every deserialised proof (protocol/proof.rs:413-449 -> Proof::deserialize_compressed);
// every entry that accepts proof bytes from outside calls it.  G1 has cofactor 1: on-curve is enough there.
bool g2_in_subgroup(const G2Affine& p);            // psi(P) == [6 u^2] P
bool g2_in_subgroup_by_order(const G2Affine& p);   // [r] P == O: the definition, kept as the cross-check of the tests

}  // namespace rlnamd
