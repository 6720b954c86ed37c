// Poseidon over BN254 Fr, bit-identical to the reference's un-optimised form
// (/root/reference/utils/src/poseidon/poseidon_hash.rs:97-135: per round add t constants, x^5 on all lanes
// (full) or lane 0 (partial), dense t x t MDS; output state[0]) with the Grain-LFSR constants of
// poseidon_constants.rs:15-261 and the (t, R_F, R_P) table of rln/src/hashers.rs:14-23.
//
// The device kernels evaluate the SAME function with the partial rounds in an equivalent sparse form (field
// arithmetic is exact, so the output bits cannot change; every KAT and the random parity tests run through it):
//   * constants: in a partial round only lane 0 passes the S-box, so the constants of the other lanes commute with it
//     and are pushed forward through M into the next round's constants: k_r = c_r + v_(r-1), v_r = M (0, k_r[1..]);
//     round r adds k_r[0] to lane 0 only and v_(R_P) joins the constants of the first full round that follows;
//   * matrices: M_r = M A_(r-1) (A_0 = I) is factored as A_r B_r with A_r = [[1, 0], [0, Mhat_r]] and
//     B_r = [[m00, v^T], [Mhat_r^-1 w, I]]; A_r leaves lane 0 alone, so it commutes with the next S-box and is merged
//     into M_(r+1).  Round r is then  x0 <- (x0 + k_r[0])^5;  (x0, x^) <- (row0_r . x,  x^ + u_r x0):
//     2 t - 1 products instead of t^2; after the last partial round the state takes the leftover A_(R_P) once.
// poseidon_derive_params computes (k, row0, u, A) on the host from the Grain constants.
#pragma once
#include <vector>

#include "common.h"
#include "field.h"
#if defined(__HIPCC__)
#include "fq29.h"
#endif

namespace rlnamd {

constexpr int POSEIDON_MAX_T = 9;  // every width of rln/src/hashers.rs:14-23; RLN itself uses t = 2, 3, 4

struct PoseidonParams {
  int t = 0, rf = 0, rp = 0;
  std::vector<Fr> ark;  // (rf+rp)*t, Montgomery
  std::vector<Fr> mds;  // t*t row-major, Montgomery
  // sparse form of the partial rounds (header comment): per round k0[r], row0[r][t], u[r][t-1]; leftover matrix
  // a_fin (t-1 x t-1, row-major); ark2 = constants of the last rf/2 full rounds with v_(R_P) added to the first
  std::vector<Fr> k0, row0, u, a_fin, ark2;
};

// host: derive the constants exactly as find_poseidon_ark_and_mds does (poseidon_constants.rs:207-261)
PoseidonParams poseidon_derive_params(int t);
// host: one hash with the dense rounds of the reference and one with the sparse partial rounds (parameter self-check)
void poseidon_params_eval_host(const PoseidonParams& P, const Fr* in, Fr* out_dense, Fr* out_sparse);

// host: ONE hash in the sparse form (the function of the kernels, on the host's 4 x 64-bit Montgomery product): the
// dependent chain of a single Merkle path is 20 hashes one after the other -- on a core that is ~0.4 ms, on the GPU
// 20 x 0.146 ms whatever is done to it (merkle.h: MerkleTreeDev::set_few).  in: t - 1 Montgomery residues.
Fr poseidon_hash_host(const PoseidonParams& P, const Fr* in);
const PoseidonParams& poseidon_host_params(int t);   // derived once per process

// device-resident constant tables for t = 2..9
struct PoseidonDev {
  int rf[POSEIDON_MAX_T + 1] = {0}, rp[POSEIDON_MAX_T + 1] = {0};
  DevBuf<Fr> ark[POSEIDON_MAX_T + 1];
  DevBuf<Fr> mds[POSEIDON_MAX_T + 1];
  DevBuf<uint32_t> ark29[POSEIDON_MAX_T + 1], mds29[POSEIDON_MAX_T + 1];  // the same constants as Fr29 (9 words each)
  DevBuf<uint32_t> opt29[POSEIDON_MAX_T + 1];  // k0 | row0 | u | a_fin | ark2 as Fr29
  size_t off_k0[POSEIDON_MAX_T + 1] = {0}, off_row0[POSEIDON_MAX_T + 1] = {0}, off_u[POSEIDON_MAX_T + 1] = {0},
         off_afin[POSEIDON_MAX_T + 1] = {0}, off_ark2[POSEIDON_MAX_T + 1] = {0};
  void init();
  bool ready = false;
};
PoseidonDev& poseidon_dev();  // lazily initialised, one per device (the device current on the calling thread)

// device pointers view handed to kernels
struct PoseidonView {
  const Fr* ark;
  const Fr* mds;
  int rf, rp;
  const uint32_t* ark29;  // Fr29 images of ark / mds (fq29.h), 9 words per element
  const uint32_t* mds29;
  const uint32_t *k0, *row0, *u, *a_fin, *ark2;  // sparse partial rounds (Fr29)
};
PoseidonView poseidon_view(int t);

#if defined(__HIPCC__)
// One hash on one lane.  `in` are Montgomery residues.
template <int T, bool FULL>
__device__ __forceinline__ void poseidon_round_dev(Fr* st, const Fr* __restrict__ ark, const Fr* __restrict__ mds) {
  static_assert(T >= 2 && T <= 4, "the 8 x 32 form is kept for t = 2..4 only");
#pragma unroll
  for (int j = 0; j < T; j++) st[j] = st[j] + ark[j];
#pragma unroll
  for (int j = 0; j < (FULL ? T : 1); j++) {
    Fr x2 = st[j].sqr();
    st[j] = x2.sqr() * st[j];
  }
  // each MDS row is one fused dot product: T products, one Montgomery reduction
  Fr nx[T];
#pragma unroll
  for (int i = 0; i < T; i++) {
    const Fr* row = mds + i * T;
    if constexpr (T == 2)
      nx[i] = Fr::dot2(row[0], st[0], row[1], st[1]);
    else if constexpr (T == 3)
      nx[i] = Fr::dot3(row[0], st[0], row[1], st[1], row[2], st[2]);
    else
      nx[i] = Fr::dot4(row[0], st[0], row[1], st[1], row[2], st[2], row[3], st[3]);
  }
#pragma unroll
  for (int j = 0; j < T; j++) st[j] = nx[j];
}

// One hash on one lane, 8 x 32-bit limb form.  `in` are Montgomery residues.  Rounds run as three loops (first
// R_F/2 full, R_P partial, last R_F/2 full) so the state stays in registers with no per-round branch.
template <int T>
__device__ __forceinline__ Fr poseidon_hash_dev_8x32(const Fr* in, const PoseidonView& pv) {
  Fr st[T];
  st[0] = Fr::zero();
#pragma unroll
  for (int j = 1; j < T; j++) st[j] = in[j - 1];
  const int half = pv.rf / 2;
  const Fr* ark = pv.ark;
#pragma unroll 1
  for (int r = 0; r < half; r++, ark += T) poseidon_round_dev<T, true>(st, ark, pv.mds);
#pragma unroll 1
  for (int r = 0; r < pv.rp; r++, ark += T) poseidon_round_dev<T, false>(st, ark, pv.mds);
#pragma unroll 1
  for (int r = 0; r < half; r++, ark += T) poseidon_round_dev<T, true>(st, ark, pv.mds);
  return st[0];
}

// The same permutation with the state in the 9 x 29-bit limb form of fq29.h (235 instead of ~375 instructions per
// product).  Bounds: state entries leave a round normalised and < 1.1 r; "+ ark" makes them lazy (limbs < 2^30,
// < 2.1 r); x^2 = lazy x lazy (9 (2^60 + 2^58) < 2^64), x^5 = x^4 x; an MDS row is one T-term dot product with at most
// two lazy operands (T = 4 normalises the pass-through entries of a partial round first).
// sum of N products of normalised operands with one reduction, N <= 5: 9 (5 2^58 + 2^58) < 2^64 per column; up to
// four products take the wide reduction rounds of fq29.h (tools/check_fq29_bounds.py: 14.8 2^60 at N = 4)
template <int N>
__device__ __forceinline__ Fr29 poseidon_dotn29(const Fr29* a, const Fr29* b) {
  static_assert(N >= 1 && N <= 5, "column sums must stay below 2^64");
  const Fr29* aa[N];
  const Fr29* bb[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    aa[k] = a + k;
    bb[k] = b + k;
  }
  return Fr29::redc_dot<N, (N <= 4)>(aa, bb);
}

template <int T, bool FULL>
__device__ __forceinline__ void poseidon_round29(Fr29* st, const uint32_t* __restrict__ ark, const uint32_t* __restrict__ mds) {
#pragma unroll
  for (int j = 0; j < T; j++) {
#pragma unroll
    for (int k = 0; k < 9; k++) st[j].v[k] += ark[j * 9 + k];
  }
#pragma unroll
  for (int j = 0; j < T; j++) {
    if (FULL || j == 0) {
      Fr29 x2 = Fr29::sqr(st[j]);
      Fr29 x4 = Fr29::sqr(x2);
      st[j] = Fr29::mul(x4, st[j]);
    } else if (T >= 4) {
      st[j].normalize();  // a 4-term dot product takes at most two lazy operands, the wider ones none
    }
  }
  Fr29 nx[T];
#pragma unroll
  for (int i = 0; i < T; i++) {
    Fr29 m[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
#pragma unroll
      for (int k = 0; k < 9; k++) m[j].v[k] = mds[(i * T + j) * 9 + k];
    }
    if constexpr (T == 2)
      nx[i] = Fr29::dot2(m[0], st[0], m[1], st[1]);
    else if constexpr (T == 3)
      nx[i] = Fr29::dot3<FULL>(m[0], st[0], m[1], st[1], m[2], st[2]);   // FULL: every entry left the S-box normalised
    else if constexpr (T == 4)
      nx[i] = Fr29::dot4<true>(m[0], st[0], m[1], st[1], m[2], st[2], m[3], st[3]);   // all normalised (see above)
    else if constexpr (T == 5)
      nx[i] = poseidon_dotn29<5>(m, st);
    else {  // t = 6..9: two groups, each reduced once; the sum (< 2.2 r, limbs < 2^30) is normalised
      nx[i] = poseidon_dotn29<5>(m, st);
      const Fr29 hi = poseidon_dotn29<T - 5>(m + 5, st + 5);
#pragma unroll
      for (int k = 0; k < 9; k++) nx[i].v[k] += hi.v[k];
      nx[i].normalize();
    }
  }
#pragma unroll
  for (int j = 0; j < T; j++) st[j] = nx[j];
}
// sum_j c[j] x[j], c = N constants at `c` (9 words each), any T <= 9; x normalised
template <int T>
__device__ __forceinline__ Fr29 poseidon_row29(const uint32_t* __restrict__ c, const Fr29* x) {
  Fr29 m[T];
#pragma unroll
  for (int j = 0; j < T; j++) {
#pragma unroll
    for (int k = 0; k < 9; k++) m[j].v[k] = c[j * 9 + k];
  }
  if constexpr (T <= 5) {
    return poseidon_dotn29<T>(m, x);
  } else {
    Fr29 lo = poseidon_dotn29<5>(m, x);
    const Fr29 hi = poseidon_dotn29<T - 5>(m + 5, x + 5);
#pragma unroll
    for (int k = 0; k < 9; k++) lo.v[k] += hi.v[k];
    lo.normalize();
    return lo;
  }
}
// One partial round in the sparse form: x0 <- (x0 + k)^5; x0' = row0 . x; x_i += u_i x0.  All entries stay normalised;
// lanes 1.. accumulate one product (< 1.1 r) per round, < 72 r after 64 rounds, inside the 261-bit range (169 r).
template <int T>
__device__ __forceinline__ void poseidon_partial29(Fr29* st, const uint32_t* __restrict__ k0,
                                                   const uint32_t* __restrict__ row0, const uint32_t* __restrict__ u) {
#pragma unroll
  for (int k = 0; k < 9; k++) st[0].v[k] += k0[k];
  const Fr29 x2 = Fr29::sqr(st[0]);
  const Fr29 x4 = Fr29::sqr(x2);
  st[0] = Fr29::mul(x4, st[0]);
  const Fr29 n0 = poseidon_row29<T>(row0, st);
#pragma unroll
  for (int i = 1; i < T; i++) {
    Fr29 ui;
#pragma unroll
    for (int k = 0; k < 9; k++) ui.v[k] = u[(i - 1) * 9 + k];
    st[i] = Fr29::mul_add(ui, st[0], st[i]);   // the addend rides the product's own carry chain (no second normalisation)
  }
  st[0] = n0;
}
template <int T>
__device__ __forceinline__ Fr poseidon_hash_dev(const Fr* in, const PoseidonView& pv) {
  Fr29 st[T];
  st[0] = Fr29::zero();
#pragma unroll
  for (int j = 1; j < T; j++) st[j] = Fr29::from_fq(in[j - 1]);
  const int half = pv.rf / 2;
  const uint32_t* ark = pv.ark29;
#pragma unroll 1
  for (int r = 0; r < half; r++, ark += T * 9) poseidon_round29<T, true>(st, ark, pv.mds29);
  const uint32_t *k0 = pv.k0, *row0 = pv.row0, *u = pv.u;
#pragma unroll 1
  for (int r = 0; r < pv.rp; r++, k0 += 9, row0 += T * 9, u += (T - 1) * 9) poseidon_partial29<T>(st, k0, row0, u);
  {  // leftover A_(R_P) on lanes 1..T-1
    Fr29 nx[T];
#pragma unroll
    for (int i = 1; i < T; i++) nx[i] = poseidon_row29<T - 1>(pv.a_fin + (i - 1) * (T - 1) * 9, st + 1);
#pragma unroll
    for (int i = 1; i < T; i++) st[i] = nx[i];
  }
  ark = pv.ark2;
#pragma unroll 1
  for (int r = 0; r < half; r++, ark += T * 9) poseidon_round29<T, true>(st, ark, pv.mds29);
  return st[0].to_fq();
}
// ---- t = 3 on FOUR lanes per hash (latency form).  A hash on one lane is a chain of ~520 products, and a lone wave
// issues them at 4.4 cycles per instruction whatever else the chip does: 0.2 ms per hash, which is what the top levels
// of a tree build (fewer nodes than lanes), a dirty-path pass and every other short chain of hashes cost.  Here lane j
// (0 .. 2) of a quadruple holds state element j, lane 3 helps, and the lanes meet in LDS:
//   full round     every lane: (s + ark_j)^5; exchange; lane j: row j of the MDS product          3 + 1 product slots
//   partial round  x = s_0 + k is known to all four lanes; x^5 cannot be had in fewer than three multiplicative levels,
//                  but nothing else has to wait for it:
//     slot 1   lane 0: x^2            lane 1: a_1 = u_1 x     lane 2: a_2 = u_2 x     lane 3: b = row0[0] x
//     slot 2   lane 0: x^4            lane 1: p_1 = row0[1] s_1                     lane 2: p_2 = row0[2] s_2
//     slot 3   lane 0: s_0' = b x^4 + (p_1 + p_2)       lanes 1, 2: s_i' = a_i x^4 + s_i
// -- the same field values as the sparse round on one lane (s_0' = row0 . (x^5, s_1, s_2), s_i' = s_i + u_i x^5), so the
// output is bit-identical; 203 dependent product slots per hash (the three-lane form of rounds 3 - 4 had 260: 0.140 ->
// 0.108 ms per level).  Lane 3 holds no state (it follows lane 2's code path on values nobody reads, so that the wave
// stays uniform); 16 hashes per wave.  Call with all 64 lanes of every wave of the workgroup (the waves meet at the same
// barriers); `in`: the input of lanes 1 and 2; the hash is returned on lane 0.  `sh`: 64 x 12 words of LDS per wave.
// Operand classes: tools/check_fq29_bounds.py ("poseidon 4-lane ...").
constexpr uint32_t POSEIDON_LANES_PER_HASH = 4, POSEIDON_HASHES_PER_WAVE = 64 / POSEIDON_LANES_PER_HASH;
__device__ __forceinline__ Fr poseidon_hash4_lanes(const Fr& in, const PoseidonView& pv, uint32_t* sh) {
  const uint32_t lane = threadIdx.x & 63, j = lane & 3, base = lane - j, jj = j == 3 ? 2 : j;
  auto put = [&](const Fr29& x) {
    uint32_t* d = sh + lane * 12;
    *(uint4*)d = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    *(uint4*)(d + 4) = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    d[8] = x.v[8];
  };
  auto get = [&](uint32_t from) {
    const uint32_t* d = sh + from * 12;
    const uint4 a = *(const uint4*)d, b = *(const uint4*)(d + 4);
    Fr29 x;
    x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w;
    x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
    x.v[8] = d[8];
    return x;
  };
  auto cst = [&](const uint32_t* c) {
    Fr29 x;
#pragma unroll
    for (int k = 0; k < 9; k++) x.v[k] = c[k];
    return x;
  };
  auto sel = [&](bool c, const Fr29& a, const Fr29& b) {
    Fr29 x;
#pragma unroll
    for (int k = 0; k < 9; k++) x.v[k] = c ? a.v[k] : b.v[k];
    return x;
  };
  Fr29 s = (j == 1 || j == 2) ? Fr29::from_fq(in) : Fr29::zero();
  const int half = pv.rf / 2;
  // A lone wave waits out every load it issues: the MDS row of this lane stays in registers, and the constants of round
  // r + 1 are fetched while round r computes (a latency kernel has the whole register file to itself).
  const Fr29 m0 = cst(pv.mds29 + jj * 27), m1 = cst(pv.mds29 + jj * 27 + 9), m2 = cst(pv.mds29 + jj * 27 + 18);
  auto full_rounds = [&](const uint32_t* ark) {
    Fr29 a = cst(ark + jj * 9);
#pragma unroll 1
    for (int r = 0; r < half; r++) {
#pragma unroll
      for (int k = 0; k < 9; k++) s.v[k] += a.v[k];
      ark += r + 1 < half ? 27 : 0;
      a = cst(ark + jj * 9);
      const Fr29 x2 = Fr29::sqr(s), x4 = Fr29::sqr(x2);
      s = Fr29::mul(x4, s);
      __syncthreads();
      put(s);
      __syncthreads();
      const Fr29 t0 = get(base), t1 = get(base + 1), t2 = get(base + 2);
      s = Fr29::dot3<true>(m0, t0, m1, t1, m2, t2);
    }
  };
  full_rounds(pv.ark29);
  __syncthreads();
  if (j == 0) put(s);
  __syncthreads();
  Fr29 s0 = get(base);
  const uint32_t *k0 = pv.k0, *row0 = pv.row0, *u = pv.u;
  Fr29 k = cst(k0), c1 = cst(j == 1 ? u : j == 2 ? u + 9 : row0), c2 = cst(row0 + jj * 9);   // (lane 0 loads row0[0] into c1 and does not use it)
#pragma unroll 1
  for (int r = 0; r < pv.rp; r++) {
    Fr29 x = s0;
#pragma unroll
    for (int q = 0; q < 9; q++) x.v[q] += k.v[q];
    const Fr29 c1r = c1, c2r = c2;
    if (r + 1 < pv.rp) {
      k0 += 9;
      row0 += 27;
      u += 18;
    }
    k = cst(k0);
    c1 = cst(j == 1 ? u : j == 2 ? u + 9 : row0);
    c2 = cst(row0 + jj * 9);
    // slot 1: x times { x, u_1, u_2, row0[0] }
    const Fr29 t1 = Fr29::mul(x, sel(j == 0, x, c1r));
    // slot 2: lane 0 squares x^2; lanes 1, 2: row0[j] s_j (lane 3 follows lane 2 on its own, unread, state)
    const Fr29 t2 = Fr29::mul(sel(j == 0, t1, c2r), sel(j == 0, t1, s));
    __syncthreads();
    put(sel(j == 3, t1, t2));   // lane 0: x^4; lanes 1, 2: p_i; lane 3: b
    __syncthreads();
    const Fr29 g0 = get(base), g1 = get(base + 1), g2 = get(base + 2), g3 = get(base + 3);
    Fr29 p12;
#pragma unroll
    for (int q = 0; q < 9; q++) p12.v[q] = g1.v[q] + g2.v[q];
    // slot 3: lane 0: b x^4 + p_1 + p_2; lanes 1, 2: a_i x^4 + s_i
    s = Fr29::mul_add(sel(j == 0, g3, t1), sel(j == 0, t2, g0), sel(j == 0, p12, s));
    __syncthreads();
    if (j == 0) put(s);
    __syncthreads();
    s0 = get(base);
  }
  {  // leftover A_(R_P) on lanes 1, 2: s_i <- a_fin[i-1][0] s_1 + a_fin[i-1][1] s_2
    __syncthreads();
    put(s);
    __syncthreads();
    if (j != 0) {
      const Fr29 t1 = get(base + 1), t2 = get(base + 2);
      const uint32_t* a = pv.a_fin + (jj - 1) * 2 * 9;
      s = Fr29::dot2(cst(a), t1, cst(a + 9), t2);
    }
  }
  full_rounds(pv.ark2);
  return s.to_fq();   // meaningful on lane 0 of the quadruple
}
#endif

// Batched hash: `n` hashes of `arity` inputs each.  in/out are canonical 32-byte LE on the device
// (in: [n][arity][32], out: [n][32]).
void poseidon_hash_batch_device(const uint8_t* d_in, size_t n, int arity, uint8_t* d_out, hipStream_t s);

}  // namespace rlnamd
