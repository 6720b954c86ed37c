// prover_kernels.h -- internal to the prover: what the host pipeline (prover.hip) and the kernel translation units
// (prover_front.hip: witness interpreters, mat-vec, NTT, recoding; prover_walks.hip: table build, table walks, partial-sum
// reductions; prover_back.hip: finalize, proof values, taps, staging, wipes) share -- descriptor structs, constants and the
// kernel DECLARATIONS.  The kernels are defined (and their templates explicitly instantiated) in exactly one unit each, so
// a change to one kernel recompiles one unit, and the units compile in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>

#include "fq29.h"
#include "poseidon.h"
#include "prover.h"
#include "walk29.h"
#include "witness_ops.h"

namespace rlnamd {

// ---- witness interpreters (prover_front.hip)
constexpr uint32_t OPK_RING = 0u << 30, OPK_CONST = 1u << 30, OPK_FAR = 2u << 30, OPK_MASK = 3u << 30;
constexpr uint32_t G_STORE = 1u << 31;  // flag on GNode.op: this node's value must reach HBM (witness signal, input, far operand)
constexpr uint32_t WIT_RING = 32;        // node values kept in LDS (64 KiB)
constexpr uint32_t WIT_LDS_CONSTS = 2048;  // constants kept in LDS (64 KiB)
constexpr uint32_t W29_STORE = 1u << 8, W29_RED = 1u << 9, W29_RARE = 1u << 10;  // flags in descriptor word 0
constexpr uint32_t W29_FMA = 25;             // program-only operation: a * b + c (an Add fused with its single-use product)
constexpr uint32_t WIT29_RING = 32;          // node values kept in LDS: 32 x 64 x 48 B = 96 KiB
constexpr uint32_t WIT29_LDS_CONSTS = 1024;  // constants kept in LDS: 48 KiB
constexpr uint32_t WIT29_CH = 256;           // descriptors per program chunk: 64 lanes x 64 B; two chunks in LDS (8 KiB)
constexpr uint32_t WIT29_LDS_BYTES = WIT29_RING * 64 * 48 + WIT29_LDS_CONSTS * 48 + 2 * WIT29_CH * 16;
constexpr double WIT29_BMAX = 7.5;
struct GNode29 {
  uint32_t w0;       // op | flags | slot << 16 (slot: index into the compact array of stored values)
  uint32_t a, b, c;  // operands as in GNode: OPK_RING | node, OPK_CONST | index, OPK_FAR | slot
};

// ---- mat-vec
struct CsrView {
  const uint32_t* ptr;
  const uint32_t* col;  // already mapped to graph node ids
  const Fr* coef;
};
constexpr uint32_t MV_LONG = 8;

// ---- partial-sum reductions and the back end
struct TaskSel {
  uint8_t id[8];
};
inline TaskSel task_sel(std::initializer_list<uint32_t> ids) {
  TaskSel t{};
  uint32_t k = 0;
  for (uint32_t v : ids) t.id[k++] = (uint8_t)v;
  return t;
}
constexpr uint32_t SUM_TREE_LANES = 512;
// dynamic LDS of k_sum_tree / k_sum_blocks: the upper half of a level, one point in the 9 x 29 form per lane
constexpr uint32_t SUM_TREE_LDS_G1 = SUM_TREE_LANES / 2 * 4 * 9 * 4, SUM_TREE_LDS_G2 = 2 * SUM_TREE_LDS_G1;
struct InputSlots {
  uint32_t secret, limit, msg_id, path, path_idx, x, ext, depth;
};

#ifndef RLN_NTT_WAVES
#define RLN_NTT_WAVES 1
#endif

// ---- kernels of prover_front.hip
__global__ void __launch_bounds__(64) k_witness(const GNode* __restrict__ nodes, uint32_t n_nodes,
                                                const Fr* __restrict__ consts, uint32_t n_consts,
                                                const uint32_t* __restrict__ inputs, uint32_t n_inputs,
                                                Fr* __restrict__ V, uint32_t* __restrict__ err, uint32_t B, uint32_t nb);
template <bool PROF>
__global__ void __launch_bounds__(64) k_witness29(const GNode29* __restrict__ nodes, uint32_t n_nodes,
                                                  const uint32_t* __restrict__ consts29, uint32_t n_consts,
                                                  const uint32_t* __restrict__ inputs, uint32_t n_inputs,
                                                  uint4* __restrict__ V29, uint32_t* __restrict__ err, uint32_t B,
                                                  uint32_t nb, unsigned long long* __restrict__ prof);
__global__ void __launch_bounds__(64) k_v29_to_fr(const uint4* __restrict__ V29, const uint32_t* __restrict__ slot2node,
                                                  uint32_t nslots, Fr* __restrict__ V, uint32_t B, uint32_t nb,
                                                  uint32_t lg = 0);
template <bool LG>
__global__ void __launch_bounds__(256) k_matvec(CsrView A, CsrView Bm, const Fr* __restrict__ V,
                                                const uint32_t* __restrict__ sig2node, uint32_t nc, uint32_t ni,
                                                uint32_t n, Fr* __restrict__ abc, uint32_t B, uint32_t nb,
                                                const uint32_t* __restrict__ long_rows = nullptr, uint32_t nshort = 0);
__global__ void __launch_bounds__(256) k_consts_to29(const Fr* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n);
template <int K, bool DIF>
__global__ void __launch_bounds__(256, RLN_NTT_WAVES) k_ntt_pass(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, int s0,
                                                  const Fr* __restrict__ scale, uint32_t B, uint32_t nb);
__global__ void __launch_bounds__(256) k_ntt_mid(Fr* __restrict__ data, const Fr* __restrict__ tw_i,
                                                 const Fr* __restrict__ tw_f, int logn, const Fr* __restrict__ scale,
                                                 uint32_t B, uint32_t nb);
template <bool DIF>
__global__ void __launch_bounds__(256) k_ntt_edge(Fr* __restrict__ data, const Fr* __restrict__ tw, int logn, uint32_t B,
                                                  uint32_t nb);
__global__ void __launch_bounds__(256) k_hquot(Fr* __restrict__ abc, uint32_t n, uint32_t B, uint32_t nb, uint32_t lg);
__global__ void __launch_bounds__(256) k_recode(const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node,
                                                uint32_t ns, const Fr* H, uint32_t n,
                                                const uint32_t* __restrict__ rs, WinSched ws1, WinSched ws2, uint32_t nh,
                                                int16_t* __restrict__ dig1, int16_t* __restrict__ dig2, uint32_t B,
                                                uint32_t nb, uint32_t part, uint32_t lg, uint32_t dB);

// ---- kernels of prover_walks.hip (k_msm29 itself is declared in walk29.h)
template <class A, class E>
__global__ void __launch_bounds__(256) k_table_to29(const A* __restrict__ src, E* __restrict__ dst, size_t n, uint32_t stride,
                                                    uint32_t k0, uint32_t npaired);
template <class F>
__global__ void __launch_bounds__(64) k_sum_ranges(const XYZZ<F>* __restrict__ src, const ChunkDesc* __restrict__ ranges,
                                                   uint32_t nranges, XYZZ<F>* __restrict__ dst, uint32_t B, uint32_t nb);
template <class F, class Acc>
__global__ void __launch_bounds__(SUM_TREE_LANES) k_sum_tree(const XYZZ<F>* __restrict__ part, const ChunkDesc* __restrict__ segchunks,
                                                  XYZZ<F>* __restrict__ dst, uint32_t B, uint32_t PB, TaskSel sel);
template <class F, class Acc>
__global__ void __launch_bounds__(SUM_TREE_LANES) k_sum_blocks(const XYZZ<F>* __restrict__ part, const ChunkDesc* __restrict__ segchunks,
                                                    const ChunkDesc* __restrict__ segblocks, XYZZ<F>* __restrict__ dst, uint32_t PB,
                                                    TaskSel sel);
__global__ void __launch_bounds__(64) k_glv_fold(G1XYZZ* __restrict__ sums1, G2XYZZ* __restrict__ sums2, uint32_t nseg1,
                                                 uint32_t B, uint32_t nb, TaskSel sel);
template <class F>
__global__ void __launch_bounds__(64) k_table_build(const Affine<F>* __restrict__ pts, uint32_t npts, WinSched ws,
                                                    Affine<F>* __restrict__ table, F* __restrict__ scratch);

// ---- kernels of prover_back.hip
__global__ void __launch_bounds__(64) k_partial_out(const G1XYZZ* __restrict__ sums1, const G2XYZZ* __restrict__ sums2,
                                                    uint32_t* __restrict__ out, uint32_t B, uint32_t nb);
__global__ void __launch_bounds__(64) k_add_partial(G1XYZZ* __restrict__ sums1, G2XYZZ* __restrict__ sums2,
                                                    const uint32_t* __restrict__ pp, uint32_t B, uint32_t nb, TaskSel sel,
                                                    const G1XYZZ* __restrict__ extra);
__global__ void __launch_bounds__(64) k_fin_affine(const G1XYZZ* __restrict__ sums1, const G2XYZZ* __restrict__ sums2,
                                                   G1Affine* __restrict__ affA, G1Affine* __restrict__ affB1,
                                                   G2Affine* __restrict__ affB2, uint32_t B, uint32_t nb, TaskSel sel);
__global__ void __launch_bounds__(64) k_fin_smul(const G1Affine* __restrict__ affA, const G1Affine* __restrict__ affB1,
                                                 const uint32_t* __restrict__ rs, G1XYZZ* __restrict__ tbl,
                                                 G1XYZZ* __restrict__ prod, uint32_t B, uint32_t nb);
__global__ void __launch_bounds__(64) k_fin_out(const G1XYZZ* __restrict__ sums1, const G1XYZZ* __restrict__ prod,
                                                const G1Affine* __restrict__ affA, const G2Affine* __restrict__ affB2,
                                                uint32_t* __restrict__ coords, uint8_t* __restrict__ comp, uint32_t B,
                                                uint32_t nb);
__global__ void __launch_bounds__(64) k_fin_out_b2(const G2XYZZ* __restrict__ sums2, uint32_t* __restrict__ coords,
                                                   uint8_t* __restrict__ comp, uint32_t B, uint32_t nb);
__global__ void __launch_bounds__(64) k_fin_out_ac(const G1XYZZ* __restrict__ sums1, const G1XYZZ* __restrict__ prod,
                                                   const G1Affine* __restrict__ affA, uint32_t* __restrict__ coords,
                                                   uint8_t* __restrict__ comp, uint32_t B, uint32_t nb);
__global__ void __launch_bounds__(64) k_fin_out_ac_fused(const G1XYZZ* __restrict__ sums1, G1Affine* __restrict__ affA,
                                                         uint32_t* __restrict__ coords, uint8_t* __restrict__ comp, uint32_t B,
                                                         uint32_t nb);
__global__ void __launch_bounds__(64) k_proof_values(const uint32_t* __restrict__ inputs, uint32_t n_inputs,
                                                     InputSlots sl, PoseidonView p2, PoseidonView p3, PoseidonView p4,
                                                     uint32_t* __restrict__ values, uint32_t nb);
__global__ void __launch_bounds__(256) k_public_signals(const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node,
                                                        uint32_t npub, uint32_t B, uint32_t nb, uint32_t* __restrict__ out);
__global__ void __launch_bounds__(64) k_values_from_witness(const Fr* __restrict__ V, const uint32_t* __restrict__ sig2node,
                                                            uint32_t B, uint32_t nb, uint32_t* __restrict__ values);
__global__ void k_gather_col(const Fr* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t count, uint32_t B,
                             uint32_t p, uint32_t* __restrict__ out);
__global__ void k_scatter_witness(const uint32_t* __restrict__ given, const uint32_t* __restrict__ sig2node,
                                  uint32_t NS, Fr* __restrict__ V, uint32_t* __restrict__ err, uint32_t B, uint32_t nb);
// partial-proof cache (Prover::collect_partial_cached / submit_finish): rows[k] = stored slot of the k-th known value;
// entry_of[p] (pinned host memory) = cache entry of proof p; an entry is [nk][3] uint4 in the 9 x 29 form of V29
__global__ void __launch_bounds__(64) k_hint_check(const Fr* __restrict__ V, const uint32_t* __restrict__ cut_node,
                                                   const uint32_t* __restrict__ cut_hint, uint32_t n_cut,
                                                   const uint32_t* __restrict__ hints, uint32_t n_hints, uint32_t B,
                                                   uint32_t* __restrict__ err);
__global__ void __launch_bounds__(256) k_cone_save(const uint4* __restrict__ V29, const uint32_t* __restrict__ rows, uint32_t nk,
                                                   uint32_t B, const uint32_t* __restrict__ entry_of, uint4* __restrict__ cache, uint32_t stride16);
__global__ void __launch_bounds__(256) k_cone_restore(const uint4* __restrict__ cache, const uint32_t* __restrict__ rows, uint32_t nk,
                                                      uint32_t B, const uint32_t* __restrict__ entry_of, uint4* __restrict__ V29, uint32_t stride16);
__global__ void __launch_bounds__(64) k_stage_in(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16);
__global__ void __launch_bounds__(64) k_wipe_cols(Fr* __restrict__ V, const uint32_t* __restrict__ rows, uint32_t nrows,
                                                  uint32_t B, uint32_t n);
// several contiguous ranges in ONE launch (a small batch's wipes are a dozen tiny buffers: a dozen launches on the calling
// thread before collect returns); block b belongs to the range r with first[r] <= b < first[r + 1], 256 words per block
struct WipeRanges {
  uint4* p[16];
  uint32_t n16[16];
  uint32_t first[17];
  uint32_t count;
};
__global__ void __launch_bounds__(64) k_wipe_ranges(WipeRanges R);
__global__ void __launch_bounds__(64) k_wipe_bytes(uint4* __restrict__ dst, uint32_t n16);   // grid: div_up(n16, 256) workgroups of 64 lanes
__global__ void __launch_bounds__(64) k_wipe_rows16(uint4* __restrict__ base, uint32_t nrows, uint32_t stride16, uint32_t n16);
__global__ void __launch_bounds__(256) k_count_nonzero16(const uint4* __restrict__ src, size_t n16, unsigned long long* __restrict__ out);
__global__ void __launch_bounds__(64) k_wipe_v29(uint4* __restrict__ V29, uint32_t nrows, uint32_t B, uint32_t n);

}  // namespace rlnamd
