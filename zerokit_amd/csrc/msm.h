// Variable-base MSM on G1 and G2 (large n, bases not known in advance): see msm.hip.
#pragma once
#include <stdint.h>

#include <memory>

namespace rlnamd {

// points: n x (x || y) canonical LE, all-zero = infinity -- 64 bytes per G1 point, 128 per G2 point (x.c0 | x.c1 | y.c0 |
// y.c1); scalars: n x 32 canonical LE.  The two classes differ in the group only (the point type of msm.hip's kernels).
#define RLN_MSM_CLASS(T, POINT_BYTES_)                                                                                     \
  class T {                                                                                                                \
   public:                                                                                                                 \
    static constexpr size_t POINT_BYTES = POINT_BYTES_;                                                                    \
    explicit T(size_t capacity);                                                                                           \
    ~T();                                                                                                                  \
    void set_host(const uint8_t* points_le, const uint8_t* scalars_le, size_t n);                                          \
    /* synthetic workload generated in HBM: P_i = k_i G, s_i from SplitMix64(seed) at global index first_index + i;        \
     * mode bit 0: every scalar is s_0; bit 1: k_i = k_(i mod 4).  The expected sum is the oracle's business, not ours. */ \
    void generate(uint64_t seed, uint64_t first_index, size_t n, uint32_t mode = 0);                                       \
    /* reads loaded / generated points [first, first + count) back: affine coordinates and scalars, canonical LE */        \
    void fetch(size_t first, size_t count, uint8_t* points_le, uint8_t* scalars_le);                                       \
    /* Pippenger up to one point per window; `window_sums_out` receives window_sums_bytes() bytes.                         \
     * ms[0] = digits + counting sort, ms[1] = bucket accumulation, ms[2] = bucket reduction (HIP events) */               \
    void run_windows(uint8_t* window_sums_out, float ms[3]);                                                               \
    /* adds the window sums of `contributors` devices and folds the windows (Horner); affine canonical result */           \
    void combine(const uint8_t* window_sums, size_t contributors, uint8_t* out_le);                                        \
    static size_t window_sums_bytes();                                                                                     \
    /* the whole of config 5 on one rank of an RCCL communicator (ncclComm_t passed as void*): windows, ncclAllGather of   \
     * the window sums on the object's stream, local add + fold.  ms[0] digits + sort, ms[1] buckets (accumulate +         \
     * reduce), ms[2] all-gather, ms[3] the fold.  Collective: every rank of the communicator must call it. */             \
    void run_sharded(void* nccl_comm, int nranks, uint8_t* out_le, float ms[4]);                                           \
                                                                                                                           \
   private:                                                                                                                \
    struct Impl;                                                                                                           \
    std::unique_ptr<Impl> d_;                                                                                              \
  };
RLN_MSM_CLASS(MsmG1, 64)
RLN_MSM_CLASS(MsmG2, 128)

// device self-test of the 9 x 29-bit group law against the 8 x 32-bit one (msm.hip); group 1 = G1, 2 = G2 (pass the
// G2 generator as x.c0 | x.c1 | y.c0 | y.c1, canonical LE); returns the number of mismatching walks
uint32_t selftest_fq29(int group, uint32_t threads, uint32_t iters, const uint8_t* g2_gen_xy_le);

}  // namespace rlnamd
