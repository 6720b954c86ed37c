// walk29_impl.h -- body of k_msm29 (declared in walk29.h).  Include it only where the kernel is instantiated.
#pragma once
#include "walk29.h"

namespace rlnamd {

template <class Acc, class Entry, class Out, int WAVES, bool LANECHUNK>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k_msm29(const Entry* __restrict__ table, const uint32_t* __restrict__ sid,
                                              const uint32_t* __restrict__ rows, const ChunkDesc* __restrict__ chunks,
                                              uint32_t nchunks, const int16_t* __restrict__ digits,
                                              Out* __restrict__ part, WinSched ws, uint32_t B, uint32_t pgroups,
                                              uint32_t nh, unsigned long long* __restrict__ clk,
                                              const uint32_t* __restrict__ chunk_ids, uint32_t pstride, PairPlan pairs) {
  // clock tap (clk may be null): every 64th workgroup adds its shader-clock cycles and its 100 MHz wall ticks; their
  // ratio is the clock the power management held under this kernel, which is what the issue-bound walk scales with
  const unsigned long long c0 = clk ? clock64() : 0, w0 = clk ? wall_clock64() : 0;
  uint32_t L = blockIdx.x;
  const int W = ws.W;
  if (LANECHUNK) {
    // chunk_ids (may be null): the launch covers a subset of the plan's chunks -- the rows that do not depend on the
    // quotient h are walked while the NTTs still run, the h rows afterwards; partial sums land at their chunk's index
    const uint32_t idx = (L * 64 + threadIdx.x) / Acc::LPP, p = blockIdx.y;   // one proof per grid row; LPP lanes per chunk
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
      uint32_t sh;
      const Entry* rb = row_base(table, rows[i], ws.stride, &sh);
      return rb[(size_t)(ws.ro[c.j] + e) << sh];
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
    acc.store_xyzz(&part[(size_t)chunk * pgroups + p]);   // LANECHUNK: `pgroups` carries the stride of `part`
    return;
  }
  // Single chunks (a wave = one chunk x 64 proofs) come first in the grid, pair chunks (a wave = one pair chunk x 32
  // proofs x 2 members) behind them; both run the ONE loop below -- a second copy of the addition's 2 000 instructions
  // was measured to cost the walk 3 %, more than pairing gains (profiles/r4_rocprof_summary.md section 3).
  const uint32_t nsb = ((nchunks + 7) / 8) * 8 * pgroups;
  const bool pm = L >= nsb;                      // uniform
  if (pm) L -= nsb;
  const uint32_t pgr = pm ? 2 * pgroups : pgroups;
  uint32_t xcd = L & 7, q = L >> 3;
  uint32_t chunk = (q / pgr) * 8 + xcd, pg = q % pgr;
  if (chunk >= (pm ? pairs.nchunks : nchunks)) return;
  // chunk_ids / pstride (mid-size batches): the short-chunk plans of the small batches walked with lanes = proofs -- a
  // subset of the plan's chunks per launch, partial sums at part[chunk * pstride + proof] (0: the batch stride B)
  if (!pm && chunk_ids) chunk = chunk_ids[chunk];
  const uint32_t member = pm ? (threadIdx.x & 1u) : 0u;
  const uint32_t p = pm ? pg * 32 + (threadIdx.x >> 1) : pg * 64 + threadIdx.x;
  const uint32_t* const R = pm ? pairs.rows : rows;
  const uint32_t* const S = pm ? pairs.sid : sid;
  const ChunkDesc cd = pm ? pairs.chunks[chunk] : chunks[chunk];
  if (!pm && cd.pt_begin >= cd.pt_end) return;   // a slot that a pair chunk fills
  const uint32_t out_slot = pm ? pairs.out[2 * chunk + member] : chunk;
  Acc acc = Acc::inf();
#pragma unroll 1
  for (uint32_t i = cd.pt_begin; i < cd.pt_end; i++) {
    const uint32_t kk = R[i] + member;   // (pair chunks list the even member: + 1 is its partner)
    const int16_t* dg = digits + ((size_t)S[i] * nh + (kk >> 31)) * W * B + p;
    uint32_t sh;
    const Entry* row = row_base(table, kk, ws.stride, &sh);
#pragma unroll 1
    for (int j = 0; j < W; j++) {  // (touching the next entry ahead of the addition was measured: 3 % slower)
      int d = dg[(size_t)j * B];
      if (d != 0) {
        uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        acc.madd(row[(size_t)(ws.ro[j] + e) << sh], d < 0);
      }
    }
  }
  part[(size_t)out_slot * (pstride ? pstride : B) + p] = acc.to_xyzz();
  if (clk && threadIdx.x == 0 && (L & 63) == 0) {
    atomicAdd(clk, clock64() - c0);
    atomicAdd(clk + 1, wall_clock64() - w0);
  }
}

}  // namespace rlnamd
