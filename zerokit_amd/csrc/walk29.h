// walk29.h -- the fixed-base table walk of the prover in the 9 x 29-bit limb form (fq29.h): window schedule, chunk
// descriptor and the kernel itself.  A header of its own so that tools/asm_walk.hip can compile the two instantiations
// alone (seconds instead of the three minutes of prover.hip) when the instruction count of the walk is being worked on.
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
                                              const uint32_t* __restrict__ chunk_ids = nullptr, uint32_t pstride = 0) {
  // clock tap (clk may be null): every 64th workgroup adds its shader-clock cycles and its 100 MHz wall ticks; their
  // ratio is the clock the power management held under this kernel, which is what the issue-bound walk scales with
  const unsigned long long c0 = clk ? clock64() : 0, w0 = clk ? wall_clock64() : 0;
  uint32_t L = blockIdx.x;
  const int W = ws.W;
  if (LANECHUNK) {
    // chunk_ids (may be null): the launch covers a subset of the plan's chunks -- the rows that do not depend on the
    // quotient h are walked while the NTTs still run, the h rows afterwards; partial sums land at their chunk's index
    const uint32_t idx = L * 64 + threadIdx.x, p = blockIdx.y;   // one proof per grid row
    if (idx >= nchunks) return;
    const uint32_t chunk = chunk_ids ? chunk_ids[idx] : idx;
    const ChunkDesc cd = chunks[chunk];
    Acc acc = Acc::inf();
    // A lane here is a lone dependent chain: digit -> table entry -> addition.  With one or two waves per SIMD nothing hides
    // the two HBM round trips of every step (measured: 10 - 28 us per addition against 4.4 us of arithmetic), so the digit
    // of step t + 2 and the entry of step t + 1 are fetched before the addition of step t (the throughput form below
    // does not need this: four waves per SIMD hide the latency, and prefetching there only costs registers).
    struct Cur {
      uint32_t i, j;
    };
    auto adv = [&](Cur& c) {
      if (++c.j == (uint32_t)W) {
        c.j = 0;
        c.i++;
      }
    };
    auto digit_at = [&](const Cur& c) -> int {
      if (c.i >= cd.pt_end) return 0;
      return digits[((size_t)sid[c.i] * nh + (rows[c.i] >> 31)) * W * B + (size_t)c.j * B + p];
    };
    auto entry_at = [&](const Cur& c, int d) -> Entry {
      const uint32_t i = c.i < cd.pt_end ? c.i : cd.pt_begin;                  // past the end: any valid entry, ignored
      const uint32_t e = (uint32_t)(d < 0 ? -d : d) - (d != 0 ? 1u : 0u);     // d == 0: entry 0, loaded and ignored
      return table[(size_t)(rows[i] & 0x7FFFFFFFu) * ws.stride + ws.ro[c.j] + e];
    };
    if (cd.pt_end > cd.pt_begin) {
      Cur c0{cd.pt_begin, 0}, c1 = c0, c2;
      adv(c1);
      c2 = c1;
      adv(c2);
      int d0 = digit_at(c0), d1 = digit_at(c1);
      Entry e0 = entry_at(c0, d0);
#pragma unroll 1
      while (c0.i < cd.pt_end) {
        const int d2 = digit_at(c2);
        const Entry e1 = entry_at(c1, d1);
        if (d0 != 0) acc.madd(e0, d0 < 0);
        e0 = e1;
        d0 = d1;
        d1 = d2;
        adv(c0);
        adv(c1);
        adv(c2);
      }
    }
    part[(size_t)chunk * pgroups + p] = acc.to_xyzz();   // LANECHUNK: `pgroups` carries the stride of `part`
    return;
  }
  uint32_t xcd = L & 7, q = L >> 3;
  uint32_t chunk = (q / pgroups) * 8 + xcd, pg = q % pgroups;
  if (chunk >= nchunks) return;
  // chunk_ids / pstride (mid-size batches): the short-chunk plans of the small batches walked with lanes = proofs -- a
  // subset of the plan's chunks per launch, partial sums at part[chunk * pstride + proof] (0: the batch stride B)
  if (chunk_ids) chunk = chunk_ids[chunk];
  uint32_t p = pg * 64 + threadIdx.x;
  ChunkDesc cd = chunks[chunk];
  Acc acc = Acc::inf();
#pragma unroll 1
  for (uint32_t i = cd.pt_begin; i < cd.pt_end; i++) {
    const uint32_t kk = rows[i], k = kk & 0x7FFFFFFFu;  // bit 31: second GLV half (see k_msm)
    const int16_t* dg = digits + ((size_t)sid[i] * nh + (kk >> 31)) * W * B + p;
    const Entry* row = table + (size_t)k * ws.stride;
#pragma unroll 1
    for (int j = 0; j < W; j++) {  // (touching the next entry ahead of the addition was measured: 3 % slower)
      int d = dg[(size_t)j * B];
      if (d != 0) {
        uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        acc.madd(row[ws.ro[j] + e], d < 0);
      }
    }
  }
  part[(size_t)chunk * (pstride ? pstride : B) + p] = acc.to_xyzz();
  if (clk && threadIdx.x == 0 && (L & 63) == 0) {
    atomicAdd(clk, clock64() - c0);
    atomicAdd(clk + 1, wall_clock64() - w0);
  }
}

}  // namespace rlnamd
