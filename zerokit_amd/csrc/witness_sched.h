// Host-side scheduler of the lanes = nodes witness interpreter (witness_lanes.hip): pure C++, no device code, so that the
// CPU suite can run the schedule through a host emulator (tests/host/witsched.cpp) against the golden witness digests.
// Cuts the graph of /root/reference/rln/src/circuit/iden3calc/graph.rs:246-272 into dependency steps, assigns LDS slots
// from the liveness of the schedule and emits the 16-byte micro-op descriptors the kernel reads.
#pragma once
#include <stdint.h>

#include <vector>

#include "zkey.h"

namespace rlnamd {

constexpr uint32_t WL_W = 64;                 // descriptors per step: one per lane (lane-form steps), 16 copies per row (row form)
constexpr uint32_t WL_ROWS = 4;               // products per row-form step: one per DPP row of 16 lanes
constexpr uint32_t WL_SLOTS = 3200;           // LDS value slots of 48 bytes: 150 KiB
constexpr uint32_t WL_PF = 16;                // descriptors prefetched per lane (steps ahead)
constexpr double WL_BMAX = 7.5;
enum : uint32_t { WK_FMA = 0, WK_ADD = 1, WK_MISC = 2, WK_SQR = 3, WK_ROW = 4 };   // 3 bits in the descriptor
// WK_SQR: every lane computes a * a + c.  WK_ROW: a * b + c with ONE product per 16-lane DPP row, a limb per lane.
enum : uint32_t { WO_NOP = 0, WO_COMPUTE = 1, WO_INPUT = 2, WO_RARE = 3 };   // MISC steps: what the lane does
constexpr uint32_t WL_STORE = 1u << 8;
constexpr uint32_t WL_GROUP_ROWS = 1u << 15;   // on the first step of a group of WL_PF steps: every step of the group is a row step
// descriptor: x = lane op | WL_STORE | kind << 12 (3 bits) | WL_GROUP_ROWS | graph op << 16;  y = dst | a << 16;  z = b | c << 16;  w = V29 slot
//             (dst, a, b, c: LDS slots; WO_INPUT: a = index into the inputs buffer)
struct WlDesc {
  uint32_t x, y, z, w;
};
struct WlProgram {
  bool ok = false;            // false: the graph does not fit this form (too many constants / live values)
  uint32_t nsteps = 0, nrow = 0, nfma = 0, nsqr = 0, nadd = 0, nmisc = 0, peak_slots = 0, n_consts = 0;
  std::vector<WlDesc> img;    // [nsteps + 2 WL_PF][WL_W]
  std::vector<Fr> consts;     // the program's constants (LDS slots 0 .. n_consts - 1): the graph's, then the folded ones
};
// store_slot[n]: index of node n in the compact array of stored values, or 0xFFFFFFFF; trash_slot: a row nobody reads;
// rows: products in row form (WK_ROW, at most WL_ROWS per step) instead of lane form (WK_FMA / WK_SQR)
WlProgram wl_schedule(const Graph& graph, const std::vector<uint32_t>& store_slot, uint32_t trash_slot, bool rows);


// ---- finishing a partial proof without re-walking what the partial witness already fixed (round 6) ----------------------
// evaluate_partial (/root/reference/rln/src/circuit/iden3calc/graph.rs:274-312): a node is known iff all its operands
// are; the inputs the partial witness leaves open are the per-message ones (inputs_for_partial_witness_calculation,
// protocol/witness.rs:887-937).  finish_zk_proof_with_rs (protocol/proof.rs:822-849) calculates the WHOLE witness again;
// the values of the known nodes are the ones the partial run already produced.  wl_cone cuts the graph down to what a
// finish has to compute when those values are at hand: the unknown nodes, plus the few known nodes they read that are
// neither inputs nor constants (recomputed: in the shipped circuits 13 nodes, all shallow), plus the input / constant
// nodes either kind reads -- a graph of its own in the original order, to be scheduled by wl_schedule like the full one
// (depth-20 circuit: 1 934 of 23 414 nodes, multiplication depth 512 of 5 736: the 20-level Merkle chain is known).
struct WlCone {
  Graph graph;                        // inputs buffer, constants, input mapping as in the full graph; signals = the unknown ones
  std::vector<uint32_t> node_of;      // cone node -> node of the full graph
  std::vector<uint8_t> node_known;    // per node of the FULL graph: evaluate_partial's Some / None
  std::vector<uint8_t> recomputed;    // per cone node: a known non-input node that is computed again (never stored)
};
WlCone wl_cone(const Graph& graph);
// store slots of the cone's nodes from the full graph's: an unknown node keeps its slot (the same row of the stored
// values), everything else is not stored -- the known rows come from the partial run
std::vector<uint32_t> wl_cone_store_slots(const WlCone& cone, const std::vector<uint32_t>& store_slot_full);


// ---- independent segments of the graph given the values of a few nodes in advance (round 6) ------------------------------
// The depth-20 circuit is 22 Poseidon hashes in a row (identity commitment, rate commitment, 20 Merkle levels): 4 300 of
// its 4 800 interpreter steps are that one dependency chain, and a dependent 256-bit product costs a lone GPU wave 0.31 us
// against 0.02 us on a host core.  The chain's VALUES are cheap to obtain elsewhere (22 host hashes); what the proof needs
// from the device is every node of the graph -- the round states inside the hashes -- and those are independent of each
// other once the value BETWEEN two hashes is given.  wl_segments cuts the graph at such nodes ("cuts", each with a hint
// index): a node's key is the set of hints it reaches backwards without crossing another cut; nodes with one key form a
// segment; a segment's graph is its nodes plus whatever else they read down to cuts (-> inputs of index inputs_size +
// hint), inputs and constants (a node another segment owns is computed again: shallow in the shipped circuits).  The
// segments are scheduled and run as programs of their own, all at once; every cut node is still COMPUTED by its owner
// and compared with its hint afterwards, so a wrong hint is detected, never used (the caller then runs the whole graph).
struct WlSegments {
  uint32_t n_hints = 0;
  std::vector<Graph> graphs;                      // inputs_size = the full graph's + n_hints
  std::vector<std::vector<uint32_t>> node_of;     // per segment: segment node -> node of the full graph (NONE: a hint input)
  std::vector<std::vector<uint8_t>> owned;        // per segment node: this segment stores it (its key is the segment's)
  std::vector<uint32_t> cut_nodes, cut_hint;      // every cut node of the full graph and the hint it must equal
};
WlSegments wl_segments(const Graph& graph, const std::vector<std::vector<uint32_t>>& cuts);
std::vector<uint32_t> wl_segment_store_slots(const WlSegments& S, size_t k, const std::vector<uint32_t>& store_slot_full);
// plain evaluation of the graph on the host (graph.rs:246-272): inputs canonical LE, inputs_size x 32 bytes; err: WitnessErr
std::vector<Fr> wl_eval_host(const Graph& graph, const uint8_t* inputs_le, uint32_t* err);

}  // namespace rlnamd
