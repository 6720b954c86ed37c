// Witness-graph interpreter with lanes = independent nodes (witness_lanes.hip), for batches too small to fill a wave
// with proofs.  k_witness29 runs one proof per LANE: a single proof issues every instruction of its ~15 000 node chain
// with 63 lanes idle (11.1 ms on MI355X).  Here one proof owns a whole WAVE and the host cuts the graph
// (/root/reference/rln/src/circuit/iden3calc/graph.rs:246-272 evaluates it node by node) into dependency steps: every
// lane of a step evaluates a different node whose operands earlier steps produced, values live in LDS slots assigned
// by the host from the liveness of the schedule.  The shipped depth-20 circuit has a multiplication depth of 5 736
// against 13 972 products, and its additions ride along: ~7 000 steps instead of ~15 000 sequential nodes.
#pragma once
#include <stdint.h>

#include <vector>

#include "common.h"
#include "zkey.h"

namespace rlnamd {

struct WitLanes {
  bool ok = false;            // false: the graph does not fit this form (too many constants / live values): use k_witness29
  uint32_t nsteps = 0, nrow = 0, nfma = 0, nsqr = 0, nadd = 0, nmisc = 0, peak_slots = 0, n_consts = 0;
  DevBuf<uint4> prog;         // [nsteps + padding][WL_W] micro-op descriptors
  DevBuf<uint32_t> consts29;  // the PROGRAM's constants (the graph's, then the ones the scheduler folded) in the 9 x 29 form
  // store_slot[n]: index of node n in the compact array of stored values (V29), or 0xFFFFFFFF when it is not stored.
  // trash_slot: a row of V29 nobody reads (the kernel stores every value; values that are not kept go there).
  void build(const Graph& g, const std::vector<uint32_t>& store_slot, uint32_t trash_slot, hipStream_t s);
  // one wave per proof; V29 / err as for k_witness29: V29[(slot * B + proof) * 3 .. + 3), err[proof]
  void launch(hipStream_t s, const uint32_t* d_inputs, uint32_t n_inputs, uint4* V29, uint32_t* err, uint32_t B,
              uint32_t nb) const;
};

// The graph as independent segments behind hints (witness_sched.h: wl_segments): every segment a program of its own, all
// of them in ONE launch -- grid (proofs, segments), a wave and a CU's LDS each.  hints: n x n_hints x 32 bytes canonical LE
// (pinned host memory is fine: each is read once, by one lane); a segment's errors are OR-ed into err[p], which the
// caller zeroes first.
struct WlSegDesc {
  uint32_t prog_off, nsteps, const_off, n_consts;   // uint4 units into prog; words into consts29
};
struct WitSegs {
  bool ok = false;
  uint32_t nseg = 0, n_hints = 0, max_steps = 0, total_steps = 0;
  DevBuf<uint4> prog;
  DevBuf<uint32_t> consts29;
  DevBuf<WlSegDesc> descs;
  void build(const struct WlSegments& S, const std::vector<uint32_t>& store_slot_full, uint32_t trash_slot, hipStream_t s);
  void launch(hipStream_t s, const uint32_t* d_inputs, uint32_t n_inputs, const uint32_t* hints, uint4* V29, uint32_t* err,
              uint32_t B, uint32_t nb) const;
};

}  // namespace rlnamd
