// prover_back.hip -- back end of the prover pipeline: partial-proof points in / out, affine conversion, s A and r B1,
// the compressed proof, the proof values by the Poseidon formulae, the parity taps, input staging and the wipes.
#include "prover_kernels.h"
#include "witness_ops.h"

#include "glv.h"
#include "pairing.h"

namespace rlnamd {

// =====================================================================================================
// 6. finalize: A, B affine; C = s*A + r*B1 + (L + H - rs*delta) ; compressed encoding
//    (partial_proof.rs:232-273; the alpha/beta/delta/query[0] terms are folded into the MSM segments)
// =====================================================================================================
__device__ __forceinline__ bool fq_is_neg_dev(const Fq& y) {
  uint32_t c[8];
  y.to_canonical(c);
  return limbs_gt(c, FqParams::HALF);
}
__device__ __forceinline__ void store_fq(uint32_t* dst, const Fq& x) { x.to_canonical(dst); }

// Partial proofs (partial_proof.rs:108-179, 182-274).  k_partial_out: the four sums of the "known" walk leave as
// canonical affine coordinates [pi_a | rho | pi_b | pi_c] (320 B).  k_add_partial: the same four points, given
// back with the full witness, are added to the sums of the "unknown + H + blinding" walk before finalize.
__global__ void __launch_bounds__(64) k_partial_out(const G1XYZZ* __restrict__ sums1, const G2XYZZ* __restrict__ sums2,
                                                    uint32_t* __restrict__ out, uint32_t B, uint32_t nb) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  uint32_t* o = out + (size_t)p * 80;
  const uint32_t t = blockIdx.y;  // 0 pi_a, 1 rho, 2 pi_c, 3 pi_b
  if (t < 3) {
    G1Affine a = sums1[(size_t)t * B + p].to_affine();
    uint32_t* d = o + (t == 0 ? 0 : t == 1 ? 16 : 64);
    a.x.to_canonical(d);
    a.y.to_canonical(d + 8);
  } else {
    G2Affine b = sums2[p].to_affine();
    b.x.c0.to_canonical(o + 32);
    b.x.c1.to_canonical(o + 40);
    b.y.c0.to_canonical(o + 48);
    b.y.c1.to_canonical(o + 56);
  }
}
__global__ void __launch_bounds__(64) k_add_partial(G1XYZZ* __restrict__ sums1, G2XYZZ* __restrict__ sums2,
                                                    const uint32_t* __restrict__ pp, uint32_t B, uint32_t nb, TaskSel sel,
                                                    const G1XYZZ* __restrict__ extra) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t* o = pp + (size_t)p * 80;
  // 0 pi_a -> A, 1 rho -> B1, 2 pi_c -> C, 3 pi_b -> B2 (the first GLV half's sums); 4: extra[p] -> C (s pi_a + r rho of the
  // fused finish, k_pp_smul)
  const uint32_t t = sel.id[blockIdx.y];
  if (t == 4) {
    G1XYZZ acc = sums1[2 * (size_t)B + p];
    acc.add(extra[p]);
    sums1[2 * (size_t)B + p] = acc;
  } else if (t < 3) {
    const uint32_t* d = o + (t == 0 ? 0 : t == 1 ? 16 : 64);
    G1Affine a{Fq::from_canonical(d), Fq::from_canonical(d + 8)};
    G1XYZZ acc = sums1[(size_t)t * B + p];
    acc.madd(a);
    sums1[(size_t)t * B + p] = acc;
  } else {
    G2Affine b{{Fq::from_canonical(o + 32), Fq::from_canonical(o + 40)}, {Fq::from_canonical(o + 48), Fq::from_canonical(o + 56)}};
    G2XYZZ acc = sums2[p];
    acc.madd(b);
    sums2[p] = acc;
  }
}

// F1: the three MSM sums that become proof elements go to affine form in parallel (one inversion each)
__global__ void __launch_bounds__(64) k_fin_affine(const G1XYZZ* __restrict__ sums1, const G2XYZZ* __restrict__ sums2,
                                                   G1Affine* __restrict__ affA, G1Affine* __restrict__ affB1,
                                                   G2Affine* __restrict__ affB2, uint32_t B, uint32_t nb, TaskSel sel) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t task = sel.id[blockIdx.y];
  if (task == 0)
    affA[p] = sums1[p].to_affine();
  else if (task == 1)
    affB1[p] = sums1[(size_t)B + p].to_affine();
  else
    affB2[p] = sums2[p].to_affine();
}

// F2: the two variable-base products s*A and r*B1 (partial_proof.rs:257-260), one lane each, fixed 4-bit
// windows over a 15-entry table parked in HBM: 252 doublings + 63 additions instead of a bit-serial
// double-and-add whose lanes diverge on every scalar bit.
__global__ void __launch_bounds__(64) k_fin_smul(const G1Affine* __restrict__ affA, const G1Affine* __restrict__ affB1,
                                                 const uint32_t* __restrict__ rs, G1XYZZ* __restrict__ tbl,
                                                 G1XYZZ* __restrict__ prod, uint32_t B, uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t task = blockIdx.y;  // 0: s*A, 1: r*B1
  const G1Affine P = task == 0 ? affA[p] : affB1[p];
  const uint32_t* k = rs + (size_t)p * 16 + (task == 0 ? 8 : 0);
  G1XYZZ* T = tbl + (size_t)task * 16 * B + p;  // T[d] at T[d * B]
  G1XYZZ cur = G1XYZZ::from_affine(P);
  T[(size_t)1 * B] = cur;
#pragma unroll 1
  for (int d = 2; d < 16; d++) {
    cur.madd(P);
    T[(size_t)d * B] = cur;
  }
  // k P = (+-k1) P + (+-k2) phi(P), |k1|, |k2| < 2^126 (glv.h): one ladder of 32 four-bit windows for both halves
  // (126 doublings instead of 252 on this latency path); phi(T[d]) = (beta X, Y, ZZ, ZZZ) is one product per use
  uint32_t kk[8], k1[4], k2[4], n1, n2;
#pragma unroll
  for (int i = 0; i < 8; i++) kk[i] = k[i];
  glv_split(kk, k1, &n1, k2, &n2);
  const Fq beta = Fq::from_canonical(GlvParams::BETA_G1);
  G1XYZZ acc = G1XYZZ::inf();
#pragma unroll 1
  for (int w = 31; w >= 0; w--) {
    if (w != 31) {
      acc = acc.dbl();
      acc = acc.dbl();
      acc = acc.dbl();
      acc = acc.dbl();
    }
    const uint32_t d1 = (k1[w >> 3] >> ((w & 7) * 4)) & 15, d2 = (k2[w >> 3] >> ((w & 7) * 4)) & 15;
    if (d1) {
      G1XYZZ t = T[(size_t)d1 * B];
      if (n1) t.Y = t.Y.neg();
      acc.add(t);
    }
    if (d2) {
      G1XYZZ t = T[(size_t)d2 * B];
      t.X = t.X * beta;
      if (n2) t.Y = t.Y.neg();
      acc.add(t);
    }
  }
  prod[(size_t)task * B + p] = acc;  // r == 0 gives infinity, matching g1_b = 0 (partial_proof.rs:242-248)
}

// F3: C = s*A + r*B1 + (L + H - rs*delta); canonical coordinates and the compressed encoding
__global__ void __launch_bounds__(64) k_fin_out(const G1XYZZ* __restrict__ sums1, const G1XYZZ* __restrict__ prod,
                                                const G1Affine* __restrict__ affA, const G2Affine* __restrict__ affB2,
                                                uint32_t* __restrict__ coords, uint8_t* __restrict__ comp, uint32_t B,
                                                uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  G1XYZZ Cacc = sums1[2 * (size_t)B + p];
  Cacc.add(prod[p]);
  Cacc.add(prod[(size_t)B + p]);
  G1Affine C = Cacc.to_affine();
  G1Affine A = affA[p];
  G2Affine B2 = affB2[p];
  uint32_t* o = coords + (size_t)p * 64;
  store_fq(o, A.x);
  store_fq(o + 8, A.y);
  store_fq(o + 16, B2.x.c0);
  store_fq(o + 24, B2.x.c1);
  store_fq(o + 32, B2.y.c0);
  store_fq(o + 40, B2.y.c1);
  store_fq(o + 48, C.x);
  store_fq(o + 56, C.y);
  // ark-serialize compressed Proof{a,b,c}: x with flags in the top byte (0x80: y > -y, 0x40: infinity)
  uint32_t w[32];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    w[i] = o[i];
    w[8 + i] = o[16 + i];
    w[16 + i] = o[24 + i];
    w[24 + i] = o[48 + i];
  }
  if (A.is_inf()) w[7] |= 0x40000000u; else if (fq_is_neg_dev(A.y)) w[7] |= 0x80000000u;
  if (B2.is_inf()) w[23] |= 0x40000000u;
  else if (B2.y.c1.is_zero() ? fq_is_neg_dev(B2.y.c0) : fq_is_neg_dev(B2.y.c1)) w[23] |= 0x80000000u;
  if (C.is_inf()) w[31] |= 0x40000000u; else if (fq_is_neg_dev(C.y)) w[31] |= 0x80000000u;
  uint32_t* cw = (uint32_t*)(comp + (size_t)p * 128);
#pragma unroll
  for (int i = 0; i < 32; i++) cw[i] = w[i];
}

// Small full proofs (split back end): the output is finished by TWO kernels on their own streams instead of a chain of
// four -- k_fin_out_b2 behind the G2 sum (GLV fold, the Fq2 inversion, B's coordinates and compressed bytes) and
// k_fin_out_ac behind the C sum (GLV fold of the C segment, + s A + r B1, the Fq inversion, A's and C's bytes).  They write
// disjoint bytes of the same records, so nothing joins them but the copy home; the two inversions of a proof, which
// k_fin_affine -> k_fin_out ran one after the other at the very end of the critical path, run side by side.  Same field
// operations, same bytes as k_glv_fold + k_fin_affine + k_fin_out.
__global__ void __launch_bounds__(64) k_fin_out_b2(const G2XYZZ* __restrict__ sums2, uint32_t* __restrict__ coords,
                                                   uint8_t* __restrict__ comp, uint32_t B, uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  G2XYZZ a = sums2[p], b = sums2[(size_t)B + p];   // first / second GLV halves (k_glv_fold)
  b.X = b.X.mul_fq(Fq::from_canonical(GlvParams::BETA_G2));
  a.add(b);
  const G2Affine B2 = a.to_affine();
  uint32_t* o = coords + (size_t)p * 64;
  store_fq(o + 16, B2.x.c0);
  store_fq(o + 24, B2.x.c1);
  store_fq(o + 32, B2.y.c0);
  store_fq(o + 40, B2.y.c1);
  uint32_t* cw = (uint32_t*)(comp + (size_t)p * 128);
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    w[i] = o[16 + i];
    w[8 + i] = o[24 + i];
  }
  if (B2.is_inf()) w[15] |= 0x40000000u;
  else if (B2.y.c1.is_zero() ? fq_is_neg_dev(B2.y.c0) : fq_is_neg_dev(B2.y.c1)) w[15] |= 0x80000000u;
#pragma unroll
  for (int i = 0; i < 16; i++) cw[8 + i] = w[i];
}
__global__ void __launch_bounds__(64) k_fin_out_ac(const G1XYZZ* __restrict__ sums1, const G1XYZZ* __restrict__ prod,
                                                   const G1Affine* __restrict__ affA, uint32_t* __restrict__ coords,
                                                   uint8_t* __restrict__ comp, uint32_t B, uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  G1XYZZ Cacc = sums1[2 * (size_t)B + p], h2 = sums1[5 * (size_t)B + p];   // C segment: first / second GLV halves
  h2.X = h2.X * Fq::from_canonical(GlvParams::BETA_G1);
  Cacc.add(h2);
  Cacc.add(prod[p]);
  Cacc.add(prod[(size_t)B + p]);
  const G1Affine C = Cacc.to_affine();
  const G1Affine A = affA[p];
  uint32_t* o = coords + (size_t)p * 64;
  store_fq(o, A.x);
  store_fq(o + 8, A.y);
  store_fq(o + 48, C.x);
  store_fq(o + 56, C.y);
  uint32_t* cw = (uint32_t*)(comp + (size_t)p * 128);
  uint32_t wa[8], wc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    wa[i] = o[i];
    wc[i] = o[48 + i];
  }
  if (A.is_inf()) wa[7] |= 0x40000000u; else if (fq_is_neg_dev(A.y)) wa[7] |= 0x80000000u;
  if (C.is_inf()) wc[7] |= 0x40000000u; else if (fq_is_neg_dev(C.y)) wc[7] |= 0x80000000u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    cw[i] = wa[i];
    cw[24 + i] = wc[i];
  }
}

// The same for the FUSED plan of a lone small proof (s A and r B1 are rows of the C segment: nothing to add to C, and A is
// needed for its bytes only): A's GLV fold and inversion move in here as well and the two inversions share one
// (Montgomery's trick: (a c)^-1, then two products) -- the chain sum -> fold -> inversion -> memset -> output that A's side
// hung in front of this kernel is gone.  Same values: 1 / ZZZ is the same field element however it is obtained.
__global__ void __launch_bounds__(64) k_fin_out_ac_fused(const G1XYZZ* __restrict__ sums1, G1Affine* __restrict__ affA,
                                                         uint32_t* __restrict__ coords, uint8_t* __restrict__ comp, uint32_t B,
                                                         uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const Fq beta = Fq::from_canonical(GlvParams::BETA_G1);
  G1XYZZ Aacc = sums1[p], a2 = sums1[3 * (size_t)B + p];
  a2.X = a2.X * beta;
  Aacc.add(a2);
  G1XYZZ Cacc = sums1[2 * (size_t)B + p], c2 = sums1[5 * (size_t)B + p];
  c2.X = c2.X * beta;
  Cacc.add(c2);
  G1Affine A, C;
  if (Aacc.is_inf() || Cacc.is_inf()) {
    A = Aacc.to_affine();
    C = Cacc.to_affine();
  } else {
    const Fq t = (Aacc.ZZZ * Cacc.ZZZ).inv();
    const Fq ia = t * Cacc.ZZZ, ic = t * Aacc.ZZZ;     // 1 / A.ZZZ, 1 / C.ZZZ
    const Fq iza = Aacc.ZZ * ia, izc = Cacc.ZZ * ic;   // 1 / Z = ZZ / ZZZ
    A = {Aacc.X * iza.sqr(), Aacc.Y * ia};
    C = {Cacc.X * izc.sqr(), Cacc.Y * ic};
  }
  affA[p] = A;
  uint32_t* o = coords + (size_t)p * 64;
  store_fq(o, A.x);
  store_fq(o + 8, A.y);
  store_fq(o + 48, C.x);
  store_fq(o + 56, C.y);
  uint32_t* cw = (uint32_t*)(comp + (size_t)p * 128);
  uint32_t wa[8], wc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    wa[i] = o[i];
    wc[i] = o[48 + i];
  }
  if (A.is_inf()) wa[7] |= 0x40000000u; else if (fq_is_neg_dev(A.y)) wa[7] |= 0x80000000u;
  if (C.is_inf()) wc[7] |= 0x40000000u; else if (fq_is_neg_dev(C.y)) wc[7] |= 0x80000000u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    cw[i] = wa[i];
    cw[24 + i] = wc[i];
  }
}

// =====================================================================================================
// 7. proof values by the Poseidon formulae (witness.rs:759-828): root, a1, y, nullifier
// =====================================================================================================
__global__ void __launch_bounds__(64) k_proof_values(const uint32_t* __restrict__ inputs, uint32_t n_inputs,
                                                     InputSlots sl, PoseidonView p2, PoseidonView p3, PoseidonView p4,
                                                     uint32_t* __restrict__ values, uint32_t nb) {
  __builtin_amdgcn_s_setprio(3);  // few, latency-bound waves: issue ahead of the MSM waves sharing the SIMD
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= nb) return;
  const uint32_t* in = inputs + (size_t)p * n_inputs * 8;
  auto ld = [&](uint32_t slot) { return Fr::from_canonical(in + (size_t)slot * 8); };
  Fr secret = ld(sl.secret), limit = ld(sl.limit), msg = ld(sl.msg_id), x = ld(sl.x), ext = ld(sl.ext);
  Fr h1[1] = {secret};
  Fr idc = poseidon_hash_dev<2>(h1, p2);
  Fr h2[2] = {idc, limit};
  Fr root = poseidon_hash_dev<3>(h2, p3);
#pragma unroll 1
  for (uint32_t i = 0; i < sl.depth; i++) {
    Fr e = ld(sl.path + i);
    const uint32_t* bi = in + (size_t)(sl.path_idx + i) * 8;
    uint32_t nz = 0;
    for (int q = 0; q < 8; q++) nz |= bi[q];
    if (nz == 0) {
      h2[0] = root;
      h2[1] = e;
    } else {
      h2[0] = e;
      h2[1] = root;
    }
    root = poseidon_hash_dev<3>(h2, p3);
  }
  Fr h3[3] = {secret, ext, msg};
  Fr a1 = poseidon_hash_dev<4>(h3, p4);
  Fr y = secret + x * a1;
  h1[0] = a1;
  Fr nullifier = poseidon_hash_dev<2>(h1, p2);
  uint32_t* o = values + (size_t)p * 40;
  y.to_canonical(o);
  root.to_canonical(o + 8);
  nullifier.to_canonical(o + 16);
  x.to_canonical(o + 24);
  ext.to_canonical(o + 32);
}

// public signals w[1..npub] of every proof straight from the witness (the circuit's own outputs; for the
// single-message circuit they equal k_proof_values' y, root, nullifier, x, external_nullifier)
__global__ void __launch_bounds__(256) k_public_signals(const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node,
                                                        uint32_t npub, uint32_t B, uint32_t nb, uint32_t* __restrict__ out) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  uint32_t k = blockIdx.y * blockDim.y + threadIdx.y;
  if (p >= nb || k >= npub) return;
  V[(size_t)sig2node[1 + k] * B + p].to_canonical(out + ((size_t)p * npub + k) * 8);
}

// Small batches: k_proof_values is a lone lane's chain of 24 Poseidon hashes (5.3 ms for one proof -- longer than the
// whole rest of the proof once the interpreter runs in 2.9 ms), and the interpreter has just computed the same five
// values as the circuit's outputs: take them from the witness (single-message circuit: w[1..5] = y, root, nullifier,
// x, external_nullifier, the order of k_proof_values; rln.circom's public signals, protocol/proof.rs:37-52)
__global__ void __launch_bounds__(64) k_values_from_witness(const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node,
                                                            uint32_t B, uint32_t nb, uint32_t* __restrict__ values) {
  const uint32_t p = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (p >= nb) return;
  V[(size_t)sig2node[1 + k] * B + p].to_canonical(values + (size_t)p * 40 + k * 8);
}

// gathers for the parity taps
__global__ void k_gather_col(const Fr* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t count, uint32_t B,
                             uint32_t p, uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint32_t row = idx ? idx[i] : i;
  src[(size_t)row * B + p].to_canonical(out + (size_t)i * 8);
}

// generate_zk_proof_with_witness (protocol/proof.rs:705-732): an externally calculated witness replaces the
// graph interpreter's.  given = [proof][signal] canonical LE; each signal is stored at the node it aliases.
__global__ void k_scatter_witness(const uint32_t* __restrict__ given, const uint32_t* __restrict__ sig2node,
                                  uint32_t NS, Fr* __restrict__ V, uint32_t* __restrict__ err, uint32_t B, uint32_t nb) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
  if (p >= nb || j >= NS) return;
  V[(size_t)sig2node[j] * B + p] = Fr::from_canonical(given + ((size_t)p * NS + j) * 8);
  if (j == 0) err[p] = WERR_NONE;
}

// Streamed inputs: the batch's inputs, (r, s) and partial points move from the slot's pinned staging buffer to its device
// buffers by a kernel on the batch's own front-end stream (the pinned pages are device-visible).  A hipMemcpyAsync
// here goes through the runtime's copy path (SDMA / blit + cross-queue signalling), which with the HIP runtime torch
// bundles (7.0) cost 8 ms per 1024-proof batch against 0 with ROCm 7.2's -- the same-box A/B is in profiles/r3_*.
// Single-wave workgroups: a 256-thread workgroup needs four free wave slots on one CU at the same instant, which the
// single-wave MSM workgroups streaming through the chip rarely leave (rocprofv3: 3.0 ms on average, 26.7 ms at worst
// for this 30 us copy when it was launched as 256-thread workgroups).
// every cut node's own value (V, Montgomery 8 x 32, after k_v29_to_fr) against the hint the segments were given (canonical
// LE words, pinned host memory): a difference raises WERR_HINT for the proof -- its batch is run again over the whole graph
__global__ void __launch_bounds__(64) k_hint_check(const Fr* __restrict__ V, const uint32_t* __restrict__ cut_node,
                                                   const uint32_t* __restrict__ cut_hint, uint32_t n_cut,
                                                   const uint32_t* __restrict__ hints, uint32_t n_hints, uint32_t B,
                                                   uint32_t* __restrict__ err) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x, p = blockIdx.y;
  if (i >= n_cut) return;
  uint32_t c[8];
  V[(size_t)cut_node[i] * B + p].to_canonical(c);
  const uint32_t* h = hints + ((size_t)p * n_hints + cut_hint[i]) * 8;
  uint32_t d = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) d |= c[k] ^ h[k];
  if (d) {
#ifdef RLN_HINT_DEBUG
    printf("hint mismatch: cut %u node %u hint %u computed %08x %08x hint %08x %08x\n", i, cut_node[i], cut_hint[i], c[0], c[7], h[0], h[7]);
#endif
    atomicOr(&err[p], (uint32_t)WERR_HINT);
  }
}
__global__ void __launch_bounds__(256) k_cone_save(const uint4* __restrict__ V29, const uint32_t* __restrict__ rows, uint32_t nk,
                                                   uint32_t B, const uint32_t* __restrict__ entry_of, uint4* __restrict__ cache, uint32_t stride16) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
  if (i >= nk * 3) return;
  const uint32_t k = i / 3, w = i % 3;
  cache[(size_t)entry_of[p] * stride16 + (size_t)k * 3 + w] = V29[((size_t)rows[k] * B + p) * 3 + w];
}
__global__ void __launch_bounds__(256) k_cone_restore(const uint4* __restrict__ cache, const uint32_t* __restrict__ rows, uint32_t nk,
                                                      uint32_t B, const uint32_t* __restrict__ entry_of, uint4* __restrict__ V29, uint32_t stride16) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
  if (i >= nk * 3) return;
  const uint32_t k = i / 3, w = i % 3;
  V29[((size_t)rows[k] * B + p) * 3 + w] = cache[(size_t)entry_of[p] * stride16 + (size_t)k * 3 + w];
}
__global__ void __launch_bounds__(64) k_stage_in(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16) {
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

// Zeroisation of what a finished batch leaves behind.  The reference zeroises the identity secret wherever it holds it
// (IdSecret: Zeroize + ZeroizeOnDrop, rln/src/utils.rs:440-527) and the witness calculator's inputs buffer
// (circuit/iden3calc.rs:45-56).  Here the secret sits in the inputs of the batch and in its witness values: columns
// [0, n) of the stored rows of V (rows == nullptr: every row) and of V29.
__global__ void __launch_bounds__(64) k_wipe_cols(Fr* __restrict__ V, const uint32_t* __restrict__ rows, uint32_t nrows,
                                                  uint32_t B, uint32_t n) {
  const uint32_t p = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y;
  if (p >= n || r >= nrows) return;
  V[(size_t)(rows ? rows[r] : r) * B + p] = Fr::zero();
}
// (a single wave per workgroup, four words per lane: a 4-wave workgroup waits for four free wave slots on ONE CU, which
// two provers' walks sharing the chip rarely leave -- the slot's next user waits for its wipe)
__global__ void __launch_bounds__(64) k_wipe_bytes(uint4* __restrict__ dst, uint32_t n16) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (uint32_t k = 0; k < 4; k++)
    if (i + 64 * k < n16) dst[i + 64 * k] = make_uint4(0, 0, 0, 0);
}
__global__ void __launch_bounds__(64) k_wipe_ranges(WipeRanges R) {
  const uint32_t b = blockIdx.x;
  uint32_t r = 0;
  while (r + 1 < R.count && b >= R.first[r + 1]) r++;
  const uint32_t i = (b - R.first[r]) * 256 + threadIdx.x;
  uint4* const dst = R.p[r];
  const uint32_t n16 = R.n16[r];
#pragma unroll
  for (uint32_t k = 0; k < 4; k++)
    if (i + 64 * k < n16) dst[i + 64 * k] = make_uint4(0, 0, 0, 0);
}
// rows of `stride16` 16-byte words each: the first n16 words of every row (the columns of a batch's proofs in the digit
// rows, the quotient's operands and the walks' partial sums)
__global__ void __launch_bounds__(64) k_wipe_rows16(uint4* __restrict__ base, uint32_t nrows, uint32_t stride16, uint32_t n16) {
  const uint32_t j = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y;
  if (j >= n16 || r >= nrows) return;
  base[(size_t)r * stride16 + j] = make_uint4(0, 0, 0, 0);
}
// parity tap of the wipes: how many 16-byte words of a buffer are not zero
__global__ void __launch_bounds__(256) k_count_nonzero16(const uint4* __restrict__ src, size_t n16, unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = src[i];
    c += (v.x | v.y | v.z | v.w) != 0;
  }
  if (c) atomicAdd(out, c);
}
__global__ void __launch_bounds__(64) k_wipe_v29(uint4* __restrict__ V29, uint32_t nrows, uint32_t B, uint32_t n) {
  const uint32_t j = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y;
  if (j >= 3 * n || r >= nrows) return;
  V29[(size_t)r * B * 3 + j] = make_uint4(0, 0, 0, 0);
}


}  // namespace rlnamd
