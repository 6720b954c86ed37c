// Compile-only harness for ISA inspection of the two throughput table-walk kernels (body: walk29_impl.h):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=400000 --cuda-device-only -S \
//         -I zerokit_amd/csrc tools/asm_walk.hip -o /tmp/walk.s
#include "walk29_impl.h"
namespace rlnamd {
template __global__ void k_msm29<G1Acc29, G1Affine29, G1XYZZ, 4, false>(const G1Affine29*, const uint32_t*, const uint32_t*,
                                                                        const ChunkDesc*, uint32_t, const int16_t*, G1XYZZ*,
                                                                        WinSched, uint32_t, uint32_t, uint32_t,
                                                                        unsigned long long*, const uint32_t*, uint32_t, PairPlan);
template __global__ void k_msm29<G2Acc29, G2Affine29, G2XYZZ, 2, false>(const G2Affine29*, const uint32_t*, const uint32_t*,
                                                                        const ChunkDesc*, uint32_t, const int16_t*, G2XYZZ*,
                                                                        WinSched, uint32_t, uint32_t, uint32_t,
                                                                        unsigned long long*, const uint32_t*, uint32_t, PairPlan);
}  // namespace rlnamd
