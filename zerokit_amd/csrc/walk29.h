// walk29.h -- the fixed-base table walk of the prover in the 9 x 29-bit limb form (fq29.h): window schedule, chunk
// descriptor and the DECLARATION of the kernel.  The body lives in walk29_impl.h and is compiled once, by prover_walks.hip
// (explicit instantiations there); tools/asm_walk.hip includes the body to look at the ISA of one instantiation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fq29.h"

namespace rlnamd {

// Window schedule of the comb tables.  Window j covers cw[j] scalar bits starting at bit bo[j]; its table row holds
// the 2^(cw[j]-1) multiples d 2^bo[j] P (signed digits) at entry offset ro[j] inside the point's block of `stride`
// entries.  Uniform widths (c, c, ...) are the classical comb; with 288 GB of HBM the first `wide` windows take one
// more bit so that W drops from 20 to 19 at c = 13 (8 x 14 + 11 x 13 = 255 bits, table x 1.35).  Passed by value:
// the kernels index it with wave-uniform j (scalar loads from the kernarg segment).
struct WinSched {
  int W;
  uint32_t stride;
  uint8_t cw[32];
  uint16_t bo[32];
  uint32_t ro[32];
};

struct ChunkDesc {
  uint32_t pt_begin, pt_end;  // compact point range
};

// A row word (the `rows` lists of the walk plans) = table point index | flags.
//   ROW_HALF2   bit 31: the second GLV half of the scalar (k2; the sum goes through phi afterwards)
//   ROW_PAIRED  bit 30: the point is a member of a PAIR (two points whose rows are walked under the SAME scalar: A_i and
//               B1_i, or one of them and L_i).  The two members sit at consecutive point indices 2 q, 2 q + 1 and their
//               tables are interleaved entry by entry: entry x of member m at ((2 q) stride + 2 x + m) -- the two entries a
//               digit selects share one 128-byte line.  Any walk may read a paired point by itself (a strided row); the
//               pair chunks of the throughput plan read both with a lane pair and halve the HBM requests of those rows.
constexpr uint32_t ROW_HALF2 = 1u << 31, ROW_PAIRED = 1u << 30, ROW_INDEX = ROW_PAIRED - 1;
// first entry of point k's table and the entry stride inside it (in entries)
template <class Entry>
__device__ __forceinline__ const Entry* row_base(const Entry* __restrict__ table, uint32_t roww, uint32_t stride, uint32_t* shift) {
  const uint32_t k = roww & ROW_INDEX, il = (roww >> 30) & 1u;
  *shift = il;
  return table + (size_t)(il ? (k & ~1u) : k) * stride + (il ? (k & 1u) : 0u);
}

// Pair chunks of the throughput plan (null / 0: none).  A pair chunk is a range of `rows` entries, each the row word of the
// EVEN member of a pair (+ ROW_HALF2); a wave walks it for 32 proofs with lane pairs: lane 2 t works for member 0, lane
// 2 t + 1 for member 1 of the same proof and the same digit.  out[2 c + m] = the chunk slot of member m's partial sum.
struct PairPlan {
  const uint32_t* rows;
  const uint32_t* sid;
  const ChunkDesc* chunks;
  const uint32_t* out;
  uint32_t nchunks;
};

// The same walk for G1 in the 9 x 29-bit form of fq29.h (tables and accumulator): 16.0 G mixed additions/s against
// 12.6 G in the 8 x 32 form (tools/microbench29.hip).  Partial sums leave in the common XYZZ<Fq> form.
//
// LANECHUNK (small batches): the 64 lanes of a wave are 64 CHUNKS of one proof instead of one chunk of 64 proofs.  A
// proof's walk is issue-bound whatever the batch holds -- with lanes = proofs a single proof still issues every
// instruction of its 2 960 + 962 chunk waves, 63 lanes idle (4.4 + 5.6 ms) -- so below half a wave of proofs the walk
// runs transposed: 47 + 16 waves per proof, each alone on its SIMD.  Rows, scalar ids and table rows become per-lane
// (vector) loads; the partial sums land at part[chunk * pgroups + proof] (the parameter is the stride of the
// partial-sum array in this mode: the small-batch plans cut the walks into shorter chunks and keep [chunk][64]).
// `sid` is indexed by the position in `rows` (a plan may walk a table row with another scalar than the row's own: the
// small-batch plan walks the A and B1 rows a second time with s w_i and r w_i, see Prover::Prover).
// Between the two (RLNAMD_LANECHUNK_WALK < proofs <= RLNAMD_LANECHUNK): lanes = proofs over the small batches' SHORT chunks.
// With lanes = chunks every lane of every wave gathers from its own table row -- 64 proofs are 5 900 + 1 900 waves of 64
// scattered 72-byte reads per step over a 228 GiB table (6.9 ms for the G1 walk, TLB-bound); with lanes = proofs a wave
// reads ONE row region and the short chunks still give 5 900 waves of 72 additions.
template <class Acc, class Entry, class Out, int WAVES, bool LANECHUNK = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k_msm29(const Entry* __restrict__ table, const uint32_t* __restrict__ sid,
                                              const uint32_t* __restrict__ rows, const ChunkDesc* __restrict__ chunks,
                                              uint32_t nchunks, const int16_t* __restrict__ digits,
                                              Out* __restrict__ part, WinSched ws, uint32_t B, uint32_t pgroups,
                                              uint32_t nh, unsigned long long* __restrict__ clk,
                                              const uint32_t* __restrict__ chunk_ids = nullptr, uint32_t pstride = 0,
                                              PairPlan pairs = PairPlan{nullptr, nullptr, nullptr, nullptr, 0});

}  // namespace rlnamd
