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

}  // namespace rlnamd
