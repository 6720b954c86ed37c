// msm_impl.h -- the variable-base MSM's kernels and host class, templated on the group (see msm.hip for the algorithm).
// Included by msm.hip (G1 + the fq29 self-test) and msm_g2.hip (G2): two translation units that compile in parallel; the
// group-blind kernels (digits, counting sort, scans) are internal to each.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>
#include <rccl/rccl.h>

#include "common.h"
#include "curve.h"
#include "fq29.h"
#include "msm.h"

namespace rlnamd {

// window width: 16 by measurement (profiles/r5_rocprof_summary.md section 6: c = 13 .. 16 at 2^21 and 2^24 points;
// RLN_MSM_C overrides at compile time, 11 .. 16 -- the digits are packed in 16 bits, the counters of a window live in LDS)
#ifndef RLN_MSM_C
#define RLN_MSM_C 16
#endif
constexpr int MSM_C = RLN_MSM_C;
static_assert(MSM_C >= 11 && MSM_C <= 16, "window width");
constexpr int MSM_W = (255 + MSM_C - 1) / MSM_C;   // 254-bit scalars plus the carry of the signed recoding
constexpr uint32_t MSM_NB = 1u << (MSM_C - 1);     // buckets per window (signed digits)
constexpr uint32_t MSM_CHUNK = 32;        // buckets per reduction chunk
constexpr uint32_t MSM_NCH = MSM_NB / MSM_CHUNK;

struct Range {
  uint32_t begin, end;
};

// The group the buckets live in.  Digits, counting sort and offsets are group-blind; everything that touches a point is
// templated on these (G2: VariableBaseMSM over the twist, `partial_proof.rs:98-104` is generic over the group).
struct MsmOpsG1 {
  typedef G1Affine Aff;
  typedef G1Affine29 Aff29;
  typedef G1Acc29 Acc;
  typedef G1XYZZ XY;
  static constexpr int AFF_WORDS = 16;   // canonical affine point in 32-bit words
};
struct MsmOpsG2 {
  typedef G2Affine Aff;
  typedef G2Affine29 Aff29;
  typedef G2Acc29 Acc;
  typedef G2XYZZ XY;
  static constexpr int AFF_WORDS = 32;
};
__device__ __forceinline__ bool entry_is_inf(const G1Affine29& e) { return e.is_inf(); }
__device__ __forceinline__ bool entry_is_inf(const G2Affine29& e) {
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) o |= e.x0[j] | e.x1[j] | e.y0[j] | e.y1[j];
  return o == 0;
}
// canonical LE coordinates of an affine point: x | y (G1), x.c0 | x.c1 | y.c0 | y.c1 (G2); infinity = all zero
static RLN_HD void affine_to_words(const G1Affine& a, uint32_t* o) {
  a.x.to_canonical(o);
  a.y.to_canonical(o + 8);
}
static RLN_HD void affine_to_words(const G2Affine& a, uint32_t* o) {
  a.x.c0.to_canonical(o);
  a.x.c1.to_canonical(o + 8);
  a.y.c0.to_canonical(o + 16);
  a.y.c1.to_canonical(o + 24);
}
static bool words_canonical_fq(const uint32_t* w, int coords) {
  for (int c = 0; c < coords; c++)
    if (limbs_geq(w + 8 * c, FqParams::MOD)) return false;
  return true;
}
static void affine_from_words(const uint32_t* w, G1Affine* a) { *a = {Fq::from_canonical(w), Fq::from_canonical(w + 8)}; }
static void affine_from_words(const uint32_t* w, G2Affine* a) {
  a->x = {Fq::from_canonical(w), Fq::from_canonical(w + 8)};
  a->y = {Fq::from_canonical(w + 16), Fq::from_canonical(w + 24)};
}

__device__ __forceinline__ uint64_t splitmix_at(uint64_t seed, uint64_t j) {
  uint64_t z = seed + (j + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// synthetic workload: P_i = k_i * G, scalar s_i; k_i, s_i 253-bit values of the stream.
// mode bit 0: every scalar is s_0 (one bucket per window receives every point); bit 1: k_i = k_(i mod 4) (four distinct
// bases).  The expected result of a run is NOT computed here: tests and bench.py take it from the oracle.
template <class O>
__global__ void __launch_bounds__(64) k_gen(uint64_t seed, uint64_t first, uint32_t n, uint32_t mode, typename O::Aff gen,
                                            typename O::Aff* __restrict__ pts, uint32_t* __restrict__ scal) {
  uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n) return;
  uint64_t i = first + t;
  const uint64_t ik = (mode & 2) ? (i & 3) : i, is = (mode & 1) ? 0 : i;
  uint32_t k[8], s[8];
  for (int q = 0; q < 4; q++) {
    uint64_t a = splitmix_at(seed, 8 * ik + q), b = splitmix_at(seed, 8 * is + 4 + q);
    k[2 * q] = (uint32_t)a;
    k[2 * q + 1] = (uint32_t)(a >> 32);
    s[2 * q] = (uint32_t)b;
    s[2 * q + 1] = (uint32_t)(b >> 32);
  }
  k[7] &= 0x1FFFFFFFu;  // < 2^253 < r
  s[7] &= 0x1FFFFFFFu;
  pts[t] = scalar_mul(gen, k).to_affine();
  for (int q = 0; q < 8; q++) scal[(size_t)t * 8 + q] = s[q];
}

// Signed MSM_C-bit digits, packed (sign << 15) | (|d| - 1); 0xFFFF marks a zero digit (|d| - 1 = 32767 never carries
// a sign: the most negative digit is -32767).  Coalesced [window][point] stores, no atomics.
constexpr uint16_t DIG_ZERO = 0xFFFFu;
static __global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scal, uint32_t n, uint16_t* __restrict__ dig) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t l[8];
#pragma unroll
  for (int q = 0; q < 8; q++) l[q] = scal[(size_t)i * 8 + q];
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < MSM_W; w++) {
    const int o = w * MSM_C, lo = o >> 5, sh = o & 31;
    const uint64_t two = (uint64_t)(lo < 8 ? l[lo] : 0u) | ((uint64_t)(lo + 1 < 8 ? l[lo + 1] : 0u) << 32);
    uint32_t raw = ((uint32_t)(two >> sh) & ((1u << MSM_C) - 1u)) + carry;
    uint32_t mag, sign;
    if (raw > MSM_NB) {
      mag = (1u << MSM_C) - raw;
      sign = 1;
      carry = 1;
    } else {
      mag = raw;
      sign = 0;
      carry = 0;
    }
    dig[(size_t)w * n + i] = mag ? (uint16_t)((sign << 15) | (mag - 1)) : DIG_ZERO;
  }
}

// Counting sort by (window, bucket) with the 2^15 counters of one window held in LDS (128 KiB of the CU's 160):
// workgroup (tile, window) histograms its slice of the points with LDS atomics and stores the 32 768 counts; after the
// prefix sums the same workgroup shape replays the slice, taking slots from LDS cursors preset to its global bases.
// 268 M global atomics per pass (9.8 + 11.9 ms measured) become LDS atomics plus plain stores.
constexpr uint32_t MSM_TILES = 16;
static __global__ void __launch_bounds__(1024) k_hist(const uint16_t* __restrict__ dig, uint32_t n, uint32_t tile_len,
                                               uint32_t* __restrict__ hist) {
  extern __shared__ uint32_t bins[];
  const uint32_t t = blockIdx.x, w = blockIdx.y;
  for (uint32_t b = threadIdx.x; b < MSM_NB; b += 1024) bins[b] = 0;
  __syncthreads();
  uint32_t lo = t * tile_len, hi = lo + tile_len < n ? lo + tile_len : n;
  const uint16_t* d = dig + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
    uint16_t v = d[i];
    if (v != DIG_ZERO) atomicAdd(&bins[v & 0x7FFFu], 1u);
  }
  __syncthreads();
  uint32_t* out = hist + ((size_t)w * MSM_TILES + t) * MSM_NB;
  for (uint32_t b = threadIdx.x; b < MSM_NB; b += 1024) out[b] = bins[b];
}
// per key: tile counts -> exclusive prefix over the tiles (in place) and the bucket total
static __global__ void __launch_bounds__(256) k_tile_prefix(uint32_t* __restrict__ hist, uint32_t* __restrict__ count) {
  uint32_t key = blockIdx.x * 256 + threadIdx.x;
  if (key >= MSM_W * MSM_NB) return;
  uint32_t w = key / MSM_NB, b = key % MSM_NB, run = 0;
  for (uint32_t t = 0; t < MSM_TILES; t++) {
    uint32_t* h = hist + ((size_t)w * MSM_TILES + t) * MSM_NB + b;
    uint32_t v = *h;
    *h = run;
    run += v;
  }
  count[key] = run;
}
static __global__ void __launch_bounds__(1024) k_scatter(const uint16_t* __restrict__ dig, uint32_t n, uint32_t tile_len,
                                                  const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                  uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t bins[];
  const uint32_t t = blockIdx.x, w = blockIdx.y;
  const uint32_t* base = hist + ((size_t)w * MSM_TILES + t) * MSM_NB;
  for (uint32_t b = threadIdx.x; b < MSM_NB; b += 1024) bins[b] = offs[w * MSM_NB + b] + base[b];
  __syncthreads();
  uint32_t lo = t * tile_len, hi = lo + tile_len < n ? lo + tile_len : n;
  const uint16_t* d = dig + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
    uint16_t v = d[i];
    if (v == DIG_ZERO) continue;
    uint32_t pos = atomicAdd(&bins[v & 0x7FFFu], 1u);
    sorted[pos] = i | ((uint32_t)(v >> 15) << 31);
  }
}

// Two-level placement (used when n <= 2^24, so that index + 6 bucket bits + sign fit one word).  Scattering straight
// into 32 768 buckets per window leaves ~1 GiB of partly written cache lines in flight (7.0 ms for 1 GiB of output);
// here a workgroup first appends its points to 512 coarse partitions of 64 buckets -- 512 open lines per workgroup,
// each filled front to back -- and a second pass places every partition (~128 KiB) into its buckets through LDS.
constexpr uint32_t MSM_PARTS = 512, MSM_PART_NB = MSM_NB / MSM_PARTS;  // 64 buckets per partition
constexpr uint32_t MSM_PART_CAP = 32768;                               // entries of a partition that fit LDS (128 KiB)
static __global__ void __launch_bounds__(1024) k_part1(const uint16_t* __restrict__ dig, uint32_t n, uint32_t tile_len,
                                                const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                uint32_t* __restrict__ tmp) {
  __shared__ uint32_t cur[MSM_PARTS];
  const uint32_t t = blockIdx.x, w = blockIdx.y;
  if (threadIdx.x < MSM_PARTS) {
    const uint32_t P = threadIdx.x;
    const uint32_t* h = hist + ((size_t)w * MSM_TILES + t) * MSM_NB + P * MSM_PART_NB;  // counts of the earlier tiles
    uint32_t sum = 0;
    for (uint32_t b = 0; b < MSM_PART_NB; b++) sum += h[b];
    cur[P] = offs[w * MSM_NB + P * MSM_PART_NB] + sum;
  }
  __syncthreads();
  uint32_t lo = t * tile_len, hi = lo + tile_len < n ? lo + tile_len : n;
  const uint16_t* d = dig + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
    uint32_t v = d[i];
    if (v == DIG_ZERO) continue;
    uint32_t b = v & 0x7FFFu;
    uint32_t pos = atomicAdd(&cur[b / MSM_PART_NB], 1u);
    tmp[pos] = i | ((b % MSM_PART_NB) << 24) | ((v >> 15) << 31);
  }
}
// A partition that fits LDS is placed there (positions relative to the partition start) and streamed out coalesced;
// larger ones (the top window of 254-bit scalars: 4x the average) are placed directly in global memory.
static __global__ void __launch_bounds__(1024) k_part2(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ offs,
                                                uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t lds[];  // [MSM_PART_CAP] staged output
  __shared__ uint32_t cur[MSM_PART_NB];
  const uint32_t P = blockIdx.x, w = blockIdx.y;
  const uint32_t k0 = w * MSM_NB + P * MSM_PART_NB;
  const uint32_t lo = offs[k0], hi = offs[k0 + MSM_PART_NB], len = hi - lo;
  const bool staged = len <= MSM_PART_CAP;
  if (threadIdx.x < MSM_PART_NB) cur[threadIdx.x] = offs[k0 + threadIdx.x] - (staged ? lo : 0);
  __syncthreads();
  for (uint32_t e = lo + threadIdx.x; e < hi; e += 1024) {
    uint32_t x = tmp[e];
    uint32_t pos = atomicAdd(&cur[(x >> 24) & (MSM_PART_NB - 1)], 1u);
    uint32_t val = x & 0x80FFFFFFu;
    if (staged) lds[pos] = val; else sorted[pos] = val;
  }
  if (!staged) return;
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < len; e += 1024) sorted[lo + e] = lds[e];
}

// exclusive scan of the 16 x 2^15 bucket counts in two steps: one workgroup per window scans its 32 768 counters
// (32 per lane) and records the window total; a second launch adds the totals of the earlier windows.
// (One 1024-lane workgroup over all 524 288 counters took 0.75 ms.)
static __global__ void __launch_bounds__(1024) k_scan_window(const uint32_t* __restrict__ count, uint32_t* __restrict__ offs,
                                                      uint32_t* __restrict__ wtotal) {
  __shared__ uint32_t part[1024];
  const uint32_t t = threadIdx.x, w = blockIdx.x, per = MSM_NB / 1024;
  const uint32_t* c = count + (size_t)w * MSM_NB + t * per;
  uint32_t s = 0;
  for (uint32_t i = 0; i < per; i++) s += c[i];
  part[t] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = t ? part[t - 1] : 0;
  uint32_t* o = offs + (size_t)w * MSM_NB + t * per;
  for (uint32_t i = 0; i < per; i++) {
    o[i] = run;
    run += c[i];
  }
  if (t == 1023) wtotal[w] = part[1023];
}
static __global__ void __launch_bounds__(256) k_scan_add(uint32_t* __restrict__ offs, const uint32_t* __restrict__ wtotal) {
  uint32_t key = blockIdx.x * 256 + threadIdx.x;
  if (key > MSM_W * MSM_NB) return;
  uint32_t w = key / MSM_NB, base = 0;
  for (uint32_t k = 0; k < w && k < MSM_W; k++) base += wtotal[k];
  if (key == MSM_W * MSM_NB) offs[key] = base; else offs[key] += base;
}

// Bucket accumulation over equal SLICES of the sorted list instead of one lane per bucket: bucket sizes are
// Poisson(512) (a wave waits for its largest, +11 %) and the top window of a 254-bit scalar has 4-8x larger buckets
// (a 64-wave tail: 37 ms for 21 ms of additions).  Lane s walks entries [s*L, (s+1)*L) -- exactly L additions -- and
// flushes at every bucket boundary it crosses: a bucket lying inside the slice goes straight to buckets[], the part of
// a bucket that began in an earlier slice to head[s], the part of one that continues into the next slice to tail[s];
// k_slice_fix then forms bucket = tail[s0] + head[s0+1] + ... + head[s1].  The next index is loaded during an addition
// (prefetching the point as well costs 16 VGPRs and the fourth wave per SIMD: slower).
constexpr uint32_t MSM_SLICE = 128;
// the bases in the packed 9 x 29-bit-limb form the accumulator works in (fq29.h): one pass per MSM, 0.3 ms at 2^24
template <class O>
__global__ void __launch_bounds__(256) k_pts_to29(const typename O::Aff* __restrict__ src, typename O::Aff29* __restrict__ dst, uint32_t n) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t < n) dst[t] = to_table29(src[t]);
}
template <class O>
__global__ void __launch_bounds__(64) k_slice_acc(const typename O::Aff29* __restrict__ pts, const uint32_t* __restrict__ offs,
                                                  const uint32_t* __restrict__ sorted, uint32_t nkeys,
                                                  typename O::XY* __restrict__ buckets, typename O::XY* __restrict__ head,
                                                  typename O::XY* __restrict__ tail) {
  typedef typename O::Acc Acc;
  const uint32_t s = blockIdx.x * 64 + threadIdx.x;
  const uint32_t total = offs[nkeys];
  const uint64_t lo64 = (uint64_t)s * MSM_SLICE;
  if (lo64 >= total) return;
  const uint32_t lo = (uint32_t)lo64, hi = total - lo > MSM_SLICE ? lo + MSM_SLICE : total;
  // first key whose range ends after lo (skips empty buckets)
  uint32_t a = 0, b = nkeys - 1;
  while (a < b) {
    uint32_t m = (a + b) >> 1;
    if (offs[m + 1] > lo) b = m; else a = m + 1;
  }
  uint32_t key = a, kend = offs[key + 1];
  bool from_before = offs[key] < lo;
  Acc acc = Acc::inf();
  uint32_t v = sorted[lo];
  for (uint32_t j = lo; j < hi; j++) {
    if (j == kend) {  // bucket `key` ends inside this slice
      if (from_before) head[s] = acc.to_xyzz(); else buckets[key] = acc.to_xyzz();
      acc = Acc::inf();
      from_before = false;
      do kend = offs[++key + 1]; while (kend == j);
    }
    typename O::Aff29 cur = pts[v & 0x7FFFFFFFu];
    uint32_t cv = v;
    if (j + 1 < hi) v = sorted[j + 1];
    if (!entry_is_inf(cur)) acc.madd(cur, (cv & 0x80000000u) != 0);
  }
  typename O::XY out = acc.to_xyzz();
  if (from_before) head[s] = out;          // started earlier (and may run on: the fix-up adds the later heads)
  else if (kend > hi) tail[s] = out;       // started here, continues in the next slice
  else buckets[key] = out;                 // ends exactly at the slice end
}
// A bucket cut into more than MSM_BIG_SLICES slices (skewed inputs: many equal scalars -- witness values that are 0 or 1,
// or the all-equal test distribution, where ONE bucket per window holds every point and is 131 072 slices at 2^24) is not
// joined by one lane's serial loop (1.8 s) but listed, and k_slice_fix_big gives it a 256-lane workgroup: every lane
// adds its share of the heads, then an eight-level tree through LDS (7 ms).  Uniform scalars list nothing.
constexpr uint32_t MSM_BIG_SLICES = 64, MSM_BIG_CAP = 4096, MSM_BIG_LANES = 256;
template <class O>
__global__ void __launch_bounds__(64) k_slice_fix(const uint32_t* __restrict__ offs, uint32_t nkeys,
                                                  const typename O::XY* __restrict__ head, const typename O::XY* __restrict__ tail,
                                                  typename O::XY* __restrict__ buckets, uint32_t* __restrict__ big) {
  typedef typename O::XY XY;
  uint32_t key = blockIdx.x * 64 + threadIdx.x;
  if (key >= nkeys) return;
  uint32_t start = offs[key], end = offs[key + 1];
  if (start == end) {
    buckets[key] = XY::inf();
    return;
  }
  uint32_t s0 = start / MSM_SLICE, s1 = (end - 1) / MSM_SLICE;
  if (s0 == s1) return;  // written by its slice
  if (s1 - s0 > MSM_BIG_SLICES) {
    const uint32_t k = atomicAdd(big, 1u);
    if (k < MSM_BIG_CAP) {   // (beyond the list's capacity: the serial join below -- correct, only slower)
      big[1 + k] = key;
      return;
    }
  }
  XY acc = tail[s0];  // slice s0 saw it start (from_before == false) and run past its end
  for (uint32_t s = s0 + 1; s <= s1; s++) acc.add(head[s]);
  buckets[key] = acc;
}
template <class O>
__global__ void __launch_bounds__(MSM_BIG_LANES) k_slice_fix_big(const uint32_t* __restrict__ offs,
                                                                 const typename O::XY* __restrict__ head, const typename O::XY* __restrict__ tail,
                                                                 typename O::XY* __restrict__ buckets, const uint32_t* __restrict__ big) {
  typedef typename O::XY XY;
  __shared__ XY sh[MSM_BIG_LANES / 2];
  const uint32_t count = big[0] < MSM_BIG_CAP ? big[0] : MSM_BIG_CAP, l = threadIdx.x;
  for (uint32_t b = blockIdx.x; b < count; b += gridDim.x) {   // (uniform for the workgroup)
    const uint32_t key = big[1 + b];
    const uint32_t s0 = offs[key] / MSM_SLICE, s1 = (offs[key + 1] - 1) / MSM_SLICE;
    XY acc = l == 0 ? tail[s0] : XY::inf();
    for (uint32_t s = s0 + 1 + l; s <= s1; s += MSM_BIG_LANES) acc.add(head[s]);
    for (uint32_t stride = MSM_BIG_LANES / 2; stride >= 1; stride >>= 1) {
      if (l >= stride && l < 2 * stride) sh[l - stride] = acc;
      __syncthreads();
      if (l < stride) acc.add(sh[l]);
      __syncthreads();
    }
    if (l == 0) buckets[key] = acc;
  }
}

// chunk [lo, lo+32) of one window: S = sum B_b, T = sum (b - lo + 1) B_b by the running-sum trick
template <class O>
__global__ void __launch_bounds__(64) k_bucket_red(const typename O::XY* __restrict__ buckets, typename O::XY* __restrict__ chunkS,
                                                   typename O::XY* __restrict__ chunkT, uint32_t nchunks) {
  typedef typename O::Acc Acc;
  uint32_t ch = blockIdx.x * 64 + threadIdx.x;
  if (ch >= nchunks) return;
  const typename O::XY* b = buckets + (size_t)ch * MSM_CHUNK;
  // general additions in the 9 x 29 form (0.55 x the instructions of the 8 x 32 law; the chain of 64 of them per lane is
  // pure latency at 2^21 points per device: 0.57 ms of a 6.9 ms shard)
  Acc run = Acc::inf(), wsum = Acc::inf();
  for (int k = MSM_CHUNK - 1; k >= 0; k--) {
    run.add(Acc::from_xyzz(b[k]));
    wsum.add(run);
  }
  chunkS[ch] = run.to_xyzz();
  chunkT[ch] = wsum.to_xyzz();
}

// X = T + lo * S with lo = 32 * (chunk index within its window): bucket b holds digit value b + 1
template <class O>
__global__ void __launch_bounds__(64) k_chunk_fix(const typename O::XY* __restrict__ chunkS, typename O::XY* __restrict__ chunkT,
                                                  uint32_t nchunks) {
  typedef typename O::XY XY;
  uint32_t ch = blockIdx.x * 64 + threadIdx.x;
  if (ch >= nchunks) return;
  uint32_t lo = (ch % MSM_NCH) * MSM_CHUNK;
  XY S = chunkS[ch], acc = XY::inf();
  for (int bit = 14; bit >= 0; bit--) {
    acc = acc.dbl();
    if ((lo >> bit) & 1) acc.add(S);
  }
  XY T = chunkT[ch];
  T.add(acc);
  chunkT[ch] = T;
}

// sums of ranges of 32 points: a wave per range, one point per lane of its first half, five levels of lane-to-lane
// additions through LDS instead of one lane's 32 dependent additions (two such launches were 0.6 ms of a 2^21 shard)
template <class O>
__global__ void __launch_bounds__(64) k_range_sum(const typename O::XY* __restrict__ src, const Range* __restrict__ ranges,
                                                  uint32_t nr, typename O::XY* __restrict__ dst) {
  typedef typename O::Acc Acc;
  __shared__ typename O::XY sh[32];
  const uint32_t r = blockIdx.x, l = threadIdx.x;
  if (r >= nr) return;
  const uint32_t b = ranges[r].begin, e = ranges[r].end;
  Acc acc = Acc::inf();
  for (uint32_t i = b + l; i < e; i += 32)
    if (l < 32) acc.add(Acc::from_xyzz(src[i]));
  for (uint32_t stride = 16; stride >= 1; stride >>= 1) {
    if (l >= stride && l < 2 * stride) sh[l - stride] = acc.to_xyzz();
    __syncthreads();
    if (l < stride) acc.add(Acc::from_xyzz(sh[l]));
    __syncthreads();
  }
  if (l == 0) dst[r] = acc.to_xyzz();
}

// The same fold on the HOST (curve.h is __host__ __device__): 16 x k additions and 240 dependent doublings of ONE point.
// On a lone GPU lane that chain is 1.75 ms (a doubling every ~7 us: k_combine); on a host core 0.1 ms -- and the result
// is headed for the host anyway.  Default of run_sharded / combine; RLNAMD_MSM_FOLD=device keeps the kernel (parity
// tests run both).
template <class O>
static void fold_windows_host(const typename O::XY* wsums, size_t k, uint8_t* out_le) {
  typename O::XY total = O::XY::inf();
  for (int w = MSM_W - 1; w >= 0; w--) {
    for (int d = 0; d < MSM_C; d++) total = total.dbl();
    for (size_t r = 0; r < k; r++) total.add(wsums[r * MSM_W + w]);
  }
  typename O::Aff a = total.to_affine();
  uint32_t c[O::AFF_WORDS];
  affine_to_words(a, c);
  memcpy(out_le, c, sizeof c);
}
static bool fold_on_device() {
  const char* v = getenv("RLNAMD_MSM_FOLD");
  return v && !strcmp(v, "device");
}

// window sums of `k` contributors ([k][W]) -> sum per window -> Horner over the windows -> affine
template <class O>
__global__ void k_combine(const typename O::XY* __restrict__ wsums, uint32_t k, uint32_t* __restrict__ out_words) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  typename O::XY total = O::XY::inf();
  for (int w = MSM_W - 1; w >= 0; w--) {
    for (int d = 0; d < MSM_C; d++) total = total.dbl();
    for (uint32_t r = 0; r < k; r++) total.add(wsums[(size_t)r * MSM_W + w]);
  }
  typename O::Aff a = total.to_affine();
  affine_to_words(a, out_words);
}

template <class O>
struct MsmImpl {
  typedef typename O::Aff Aff;
  typedef typename O::Aff29 Aff29;
  typedef typename O::XY XY;
  hipStream_t s = nullptr;
  size_t cap = 0, n = 0;
  DevBuf<Aff> pts;
  DevBuf<Aff29> pts29;
  DevBuf<uint32_t> scal, count, offs, sorted, tmp, wtotal;
  DevBuf<uint16_t> dig;
  DevBuf<uint32_t> hist;
  DevBuf<XY> buckets, chunkS, chunkT, grp, wsum, head, tail;
  DevBuf<uint32_t> big;         // k_slice_fix: count + keys of the buckets cut into many slices
  size_t max_slices = 0;
  DevBuf<Range> r1, r2;
  DevBuf<XY> gather;            // run_sharded: the window sums of every rank
  DevBuf<uint32_t> result;
  hipEvent_t e[6];
  Aff generator;                // of the synthetic workload (G1: (1, 2); G2: the twist's generator)

  explicit MsmImpl(size_t capacity);
  ~MsmImpl();
  void set_host(const uint8_t* points_le, const uint8_t* scalars_le, size_t n);
  void generate(uint64_t seed, uint64_t first_index, size_t n, uint32_t mode);
  void fetch(size_t first, size_t count, uint8_t* points_le, uint8_t* scalars_le);
  void run_windows(uint8_t* window_sums_out, float ms[3]);
  void combine(const uint8_t* window_sums, size_t contributors, uint8_t* out_le);
  void run_sharded(void* nccl_comm, int nranks, uint8_t* out_le, float ms[4]);
  void enqueue_windows();
};

static G1Affine msm_generator(const MsmOpsG1*) { return {Fq::from_u32(1), Fq::from_u32(2)}; }
static G2Affine msm_generator(const MsmOpsG2*) {
  // the generator of the order-r subgroup of the twist (ark-bn254 g2::G2_GENERATOR_X / _Y; EIP-197's P2), canonical LE words
  static const uint32_t g[32] = {
      0xd992f6ed, 0x46debd5c, 0xf75edadd, 0x674322d4, 0x5e5c4479, 0x426a0066, 0x121f1e76, 0x1800deef,   // x.c0
      0xaef312c2, 0x97e485b7, 0x35a9e712, 0xf1aa4933, 0x31fb5d25, 0x7260bfb7, 0x920d483a, 0x198e9393,   // x.c1
      0x66fa7daa, 0x4ce6cc01, 0x0c43d37b, 0xe3d1e769, 0x8dcb408f, 0x4aab7180, 0xdb8c6deb, 0x12c85ea5,   // y.c0
      0xd122975b, 0x55acdadc, 0x70b38ef3, 0xbc4b3133, 0x690c3395, 0xec9e99ad, 0x585ff075, 0x090689d0};  // y.c1
  G2Affine a;
  affine_from_words(g, &a);
  return a;
}

template <class O>
MsmImpl<O>::MsmImpl(size_t capacity) {
  require_gpu();
  MsmImpl& D = *this;
  D.cap = capacity;
  D.generator = msm_generator((const O*)nullptr);
  RLN_HIP(hipStreamCreateWithFlags(&D.s, hipStreamNonBlocking));
  for (auto& e : D.e) RLN_HIP(hipEventCreate(&e));
  RLN_HIP(hipFuncSetAttribute((const void*)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, MSM_NB * 4));
  RLN_HIP(hipFuncSetAttribute((const void*)k_part2, hipFuncAttributeMaxDynamicSharedMemorySize, MSM_PART_CAP * 4));
  RLN_HIP(hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, MSM_NB * 4));
  const uint32_t nkeys = MSM_W * MSM_NB, nch = MSM_W * MSM_NCH;
  D.pts.alloc(capacity);
  D.pts29.alloc(capacity);
  D.scal.alloc(capacity * 8);
  D.dig.alloc(capacity * MSM_W);
  D.sorted.alloc(capacity * MSM_W);
  if (capacity <= (1u << 24)) D.tmp.alloc(capacity * MSM_W);
  D.count.alloc(nkeys);
  D.wtotal.alloc(MSM_W);
  D.offs.alloc(nkeys + 1);
  D.hist.alloc((size_t)nkeys * MSM_TILES);
  D.buckets.alloc(nkeys);
  D.max_slices = capacity * MSM_W / MSM_SLICE + 1;
  D.big.alloc(1 + MSM_BIG_CAP);
  D.head.alloc(D.max_slices);
  D.tail.alloc(D.max_slices);
  D.chunkS.alloc(nch);
  D.chunkT.alloc(nch);
  // per window: 1024 chunks -> 32 groups of 32 -> 1
  std::vector<Range> r1, r2;
  for (uint32_t w = 0; w < MSM_W; w++) {
    for (uint32_t g = 0; g < MSM_NCH / 32; g++) r1.push_back({w * MSM_NCH + g * 32, w * MSM_NCH + g * 32 + 32});
    r2.push_back({w * (MSM_NCH / 32), (w + 1) * (MSM_NCH / 32)});
  }
  D.r1.alloc(r1.size());
  D.r2.alloc(r2.size());
  D.r1.upload(r1.data(), r1.size(), D.s);
  D.r2.upload(r2.data(), r2.size(), D.s);
  D.grp.alloc(r1.size());
  D.wsum.alloc(MSM_W);
  RLN_HIP(hipStreamSynchronize(D.s));
}

template <class O>
MsmImpl<O>::~MsmImpl() {
  if (s) {
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
  }
  for (auto& ev : e) (void)hipEventDestroy(ev);
}

template <class O>
void MsmImpl<O>::set_host(const uint8_t* points_le, const uint8_t* scalars_le, size_t n_) {
  MsmImpl& D = *this;
  if (n_ > D.cap) throw Error("MSM larger than the workspace");
  constexpr int AW = O::AFF_WORDS;
  std::vector<Aff> p(n_);
  for (size_t i = 0; i < n_; i++) {
    uint32_t w[AW];
    memcpy(w, points_le + 4 * AW * i, 4 * AW);
    if (!words_canonical_fq(w, AW / 8)) throw Error("Non-canonical field element");
    affine_from_words(w, &p[i]);  // all-zero coordinates encode infinity
  }
  for (size_t i = 0; i < n_; i++) {
    uint32_t sc[8];
    memcpy(sc, scalars_le + 32 * i, 32);
    if (limbs_geq(sc, FrParams::MOD)) throw Error("Non-canonical field element: value is not in [0, r-1]");
  }
  RLN_HIP(hipMemcpyAsync(D.pts.p, p.data(), n_ * sizeof(Aff), hipMemcpyHostToDevice, D.s));
  RLN_HIP(hipMemcpyAsync(D.scal.p, scalars_le, n_ * 32, hipMemcpyHostToDevice, D.s));
  RLN_HIP(hipStreamSynchronize(D.s));
  D.n = n_;
}

template <class O>
void MsmImpl<O>::generate(uint64_t seed, uint64_t first_index, size_t n_, uint32_t mode) {
  MsmImpl& D = *this;
  if (n_ > D.cap) throw Error("MSM larger than the workspace");
  hipLaunchKernelGGL(k_gen<O>, dim3(div_up(n_, 64)), dim3(64), 0, D.s, seed, first_index, (uint32_t)n_, mode, D.generator,
                     D.pts.p, D.scal.p);
  RLN_HIP(hipGetLastError());
  RLN_HIP(hipStreamSynchronize(D.s));
  D.n = n_;
}

template <class O>
void MsmImpl<O>::fetch(size_t first, size_t count_, uint8_t* points_le, uint8_t* scalars_le) {
  MsmImpl& D = *this;
  if (first + count_ > D.n) throw Error("MSM fetch: range outside the loaded points");
  constexpr int AW = O::AFF_WORDS;
  std::vector<Aff> p(count_);
  RLN_HIP(hipMemcpy(p.data(), D.pts.p + first, count_ * sizeof(Aff), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < count_; i++) {
    uint32_t c[AW];
    affine_to_words(p[i], c);
    memcpy(points_le + 4 * AW * i, c, 4 * AW);
  }
  RLN_HIP(hipMemcpy(scalars_le, D.scal.p + first * 8, count_ * 32, hipMemcpyDeviceToHost));
}

template <class O>
void MsmImpl<O>::run_windows(uint8_t* window_sums_out, float ms[3]) {
  MsmImpl& D = *this;
  enqueue_windows();
  RLN_HIP(hipMemcpyAsync(window_sums_out, D.wsum.p, MSM_W * sizeof(XY), hipMemcpyDeviceToHost, D.s));
  RLN_HIP(hipStreamSynchronize(D.s));
  if (ms)
    for (int i = 0; i < 3; i++) RLN_HIP(hipEventElapsedTime(&ms[i], D.e[i], D.e[i + 1]));
}

// One MSM over the points of ALL ranks of an RCCL communicator (BASELINE config 5): every rank reduces its
// slice to the 16 window sums, ONE ncclAllGather moves the 2 KiB (G2: 4 KiB) blocks over xGMI (RCCL has no elliptic-curve
// reduce op: "all-reduce of partials" = gather + local add), and every rank adds them and folds the windows.  Everything is
// enqueued on the object's stream; the only host wait is the final copy.
template <class O>
void MsmImpl<O>::run_sharded(void* nccl_comm, int nranks, uint8_t* out_le, float ms[4]) {
  MsmImpl& D = *this;
  if (nranks < 1) throw Error("run_sharded: empty communicator");
  enqueue_windows();
  if (D.gather.n < (size_t)nranks * MSM_W) D.gather.alloc((size_t)nranks * MSM_W);
  if (!D.result.p) D.result.alloc(O::AFF_WORDS);
  ncclResult_t r = ncclAllGather(D.wsum.p, D.gather.p, MSM_W * sizeof(XY), ncclUint8, (ncclComm_t)nccl_comm, D.s);
  if (r != ncclSuccess) throw Error(std::string("RCCL error: ") + ncclGetErrorString(r) + " (ncclAllGather of the window sums)");
  RLN_HIP(hipEventRecord(D.e[4], D.s));
  float host_fold_ms = -1.f;
  if (fold_on_device()) {
    hipLaunchKernelGGL(k_combine<O>, dim3(1), dim3(64), 0, D.s, D.gather.p, (uint32_t)nranks, D.result.p);
    RLN_HIP(hipGetLastError());
    RLN_HIP(hipEventRecord(D.e[5], D.s));
    RLN_HIP(hipMemcpyAsync(out_le, D.result.p, 4 * O::AFF_WORDS, hipMemcpyDeviceToHost, D.s));
    RLN_HIP(hipStreamSynchronize(D.s));
  } else {
    RLN_HIP(hipEventRecord(D.e[5], D.s));
    std::vector<XY> h((size_t)nranks * MSM_W);
    RLN_HIP(hipMemcpyAsync(h.data(), D.gather.p, h.size() * sizeof(XY), hipMemcpyDeviceToHost, D.s));
    RLN_HIP(hipStreamSynchronize(D.s));
    const auto t0 = std::chrono::steady_clock::now();
    fold_windows_host<O>(h.data(), (size_t)nranks, out_le);
    host_fold_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  if (ms) {
    RLN_HIP(hipEventElapsedTime(&ms[0], D.e[0], D.e[1]));
    RLN_HIP(hipEventElapsedTime(&ms[1], D.e[1], D.e[3]));
    RLN_HIP(hipEventElapsedTime(&ms[2], D.e[3], D.e[4]));
    if (host_fold_ms >= 0) ms[3] = host_fold_ms;   // the fold on the host (wall time of the loop itself)
    else RLN_HIP(hipEventElapsedTime(&ms[3], D.e[4], D.e[5]));
  }
}

template <class O>
void MsmImpl<O>::enqueue_windows() {
  MsmImpl& D = *this;
  const uint32_t n = (uint32_t)D.n, nkeys = MSM_W * MSM_NB, nch = MSM_W * MSM_NCH;
  hipStream_t s = D.s;
  RLN_HIP(hipEventRecord(D.e[0], s));
  const uint32_t tile_len = div_up(n ? n : 1, MSM_TILES);
  const size_t lds = MSM_NB * sizeof(uint32_t);
  if (n) hipLaunchKernelGGL(k_pts_to29<O>, dim3(div_up(n, 256)), dim3(256), 0, s, D.pts.p, D.pts29.p, n);
  if (n) hipLaunchKernelGGL(k_digits, dim3(div_up(n, 256)), dim3(256), 0, s, D.scal.p, n, D.dig.p);
  hipLaunchKernelGGL(k_hist, dim3(MSM_TILES, MSM_W), dim3(1024), lds, s, D.dig.p, n, tile_len, D.hist.p);
  hipLaunchKernelGGL(k_tile_prefix, dim3(div_up(nkeys, 256)), dim3(256), 0, s, D.hist.p, D.count.p);
  hipLaunchKernelGGL(k_scan_window, dim3(MSM_W), dim3(1024), 0, s, D.count.p, D.offs.p, D.wtotal.p);
  hipLaunchKernelGGL(k_scan_add, dim3(div_up(nkeys + 1, 256)), dim3(256), 0, s, D.offs.p, D.wtotal.p);
  if (D.tmp.p) {   // n <= 2^24: two-level placement; larger workspaces scatter in one pass
    hipLaunchKernelGGL(k_part1, dim3(MSM_TILES, MSM_W), dim3(1024), 0, s, D.dig.p, n, tile_len, D.offs.p, D.hist.p, D.tmp.p);
    hipLaunchKernelGGL(k_part2, dim3(MSM_PARTS, MSM_W), dim3(1024), MSM_PART_CAP * 4, s, D.tmp.p, D.offs.p, D.sorted.p);
  } else {
    hipLaunchKernelGGL(k_scatter, dim3(MSM_TILES, MSM_W), dim3(1024), lds, s, D.dig.p, n, tile_len, D.offs.p, D.hist.p,
                       D.sorted.p);
  }
  RLN_HIP(hipEventRecord(D.e[1], s));
  const uint32_t nslices = div_up((size_t)n * MSM_W, MSM_SLICE);
  if (nslices)
    hipLaunchKernelGGL(k_slice_acc<O>, dim3(div_up(nslices, 64)), dim3(64), 0, s, D.pts29.p, D.offs.p, D.sorted.p, nkeys,
                       D.buckets.p, D.head.p, D.tail.p);
  RLN_HIP(hipMemsetAsync(D.big.p, 0, 4, s));
  hipLaunchKernelGGL(k_slice_fix<O>, dim3(div_up(nkeys, 64)), dim3(64), 0, s, D.offs.p, nkeys, D.head.p, D.tail.p, D.buckets.p,
                     D.big.p);
  hipLaunchKernelGGL(k_slice_fix_big<O>, dim3(64), dim3(MSM_BIG_LANES), 0, s, D.offs.p, D.head.p, D.tail.p, D.buckets.p, D.big.p);
  RLN_HIP(hipEventRecord(D.e[2], s));
  hipLaunchKernelGGL(k_bucket_red<O>, dim3(div_up(nch, 64)), dim3(64), 0, s, D.buckets.p, D.chunkS.p, D.chunkT.p, nch);
  hipLaunchKernelGGL(k_chunk_fix<O>, dim3(div_up(nch, 64)), dim3(64), 0, s, D.chunkS.p, D.chunkT.p, nch);
  hipLaunchKernelGGL(k_range_sum<O>, dim3(D.r1.n), dim3(64), 0, s, D.chunkT.p, D.r1.p, (uint32_t)D.r1.n, D.grp.p);
  hipLaunchKernelGGL(k_range_sum<O>, dim3(D.r2.n), dim3(64), 0, s, D.grp.p, D.r2.p, (uint32_t)D.r2.n, D.wsum.p);
  RLN_HIP(hipGetLastError());
  RLN_HIP(hipEventRecord(D.e[3], s));
}

template <class O>
void MsmImpl<O>::combine(const uint8_t* window_sums, size_t contributors, uint8_t* out_le) {
  MsmImpl& D = *this;
  if (!fold_on_device()) {
    std::vector<XY> h(contributors * MSM_W);
    memcpy(h.data(), window_sums, h.size() * sizeof(XY));
    fold_windows_host<O>(h.data(), contributors, out_le);
    return;
  }
  DevBuf<XY> in(contributors * MSM_W);
  DevBuf<uint32_t> out(O::AFF_WORDS);
  RLN_HIP(hipMemcpyAsync(in.p, window_sums, contributors * MSM_W * sizeof(XY), hipMemcpyHostToDevice, D.s));
  hipLaunchKernelGGL(k_combine<O>, dim3(1), dim3(64), 0, D.s, in.p, (uint32_t)contributors, out.p);
  RLN_HIP(hipGetLastError());
  RLN_HIP(hipMemcpyAsync(out_le, out.p, 4 * O::AFF_WORDS, hipMemcpyDeviceToHost, D.s));
  RLN_HIP(hipStreamSynchronize(D.s));
}


// the host class of one group behind the interface of msm.h
#define RLN_MSM_WRAPPERS(T, O)                                                                                                \
  T::T(size_t capacity) : d_(new Impl(capacity)) {}                                                                          \
  T::~T() {}                                                                                                                 \
  void T::set_host(const uint8_t* p, const uint8_t* s, size_t n) { d_->set_host(p, s, n); }                                  \
  void T::generate(uint64_t seed, uint64_t first, size_t n, uint32_t mode) { d_->generate(seed, first, n, mode); }           \
  void T::fetch(size_t first, size_t count, uint8_t* p, uint8_t* s) { d_->fetch(first, count, p, s); }                       \
  void T::run_windows(uint8_t* out, float ms[3]) { d_->run_windows(out, ms); }                                               \
  void T::combine(const uint8_t* ws, size_t k, uint8_t* out) { d_->combine(ws, k, out); }                                    \
  void T::run_sharded(void* comm, int nranks, uint8_t* out, float ms[4]) { d_->run_sharded(comm, nranks, out, ms); }         \
  size_t T::window_sums_bytes() { return MSM_W * sizeof(O::XY); }

}  // namespace rlnamd
