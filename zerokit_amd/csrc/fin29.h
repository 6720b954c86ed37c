// Small-batch form of the two variable-base products of the Groth16 assembly, s A and r B1
// (/root/reference/rln/src/partial_proof.rs:257-260): fin29.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve.h"

namespace rlnamd {

// prod[task * B + p] = k_task P_task for task 0 (s A) and 1 (r B1); rs: n x (r | s) canonical LE words
void launch_fin_smul29(hipStream_t s, const G1Affine* affA, const G1Affine* affB1, const uint32_t* rs, G1XYZZ* prod,
                       uint32_t B, uint32_t nb);

// finish from a cached partial proof: the powers of pi_a and rho (partial time) and s pi_a + r rho from them (fin29.hip).
// PP_POWERS16: uint4 units the powers take in a cache entry (2 points x 2 GLV halves x 16 chunks x 64 bytes)
constexpr uint32_t PP_POWERS16 = 2 * 2 * 16 * 4;
void launch_pp_powers(hipStream_t s, const uint32_t* pp, const uint32_t* entry_of, uint4* cache, uint32_t stride16,
                      uint32_t off16, uint32_t n);
void launch_pp_smul(hipStream_t s, const uint4* cache, const uint32_t* entry_of, uint32_t stride16, uint32_t off16,
                    const uint32_t* rs, G1XYZZ* out, uint32_t n);

}  // namespace rlnamd
