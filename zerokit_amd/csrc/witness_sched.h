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

}  // namespace rlnamd
