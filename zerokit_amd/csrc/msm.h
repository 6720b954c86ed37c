// Variable-base G1 MSM (large n, bases not known in advance): see msm.hip.
#pragma once
#include <stdint.h>

#include <memory>

namespace rlnamd {

class MsmG1 {
 public:
  explicit MsmG1(size_t capacity);
  ~MsmG1();
  // points: n x (x || y) canonical LE, (0,0) = infinity; scalars: n x 32 canonical LE
  void set_host(const uint8_t* points_xy_le, const uint8_t* scalars_le, size_t n);
  // synthetic workload generated in HBM: P_i = k_i G, s_i from SplitMix64(seed) at global index first_index + i
  // mode bit 0: every scalar is s_0; bit 1: k_i = k_(i mod 4).  The expected sum is the oracle's business, not ours.
  void generate(uint64_t seed, uint64_t first_index, size_t n, uint32_t mode = 0);
  // reads loaded / generated points [first, first + count) back: affine x || y and scalars, canonical LE
  void fetch(size_t first, size_t count, uint8_t* points_xy_le, uint8_t* scalars_le);
  // Pippenger up to one point per window; `window_sums_out` receives window_sums_bytes() bytes.
  // ms[0] = digits + counting sort, ms[1] = bucket accumulation, ms[2] = bucket reduction (HIP events)
  void run_windows(uint8_t* window_sums_out, float ms[3]);
  // adds the window sums of `contributors` devices and folds the windows (Horner); affine canonical result
  void combine(const uint8_t* window_sums, size_t contributors, uint8_t out_xy_le[64]);
  static size_t window_sums_bytes();
  // the whole of config 5 on one rank of an RCCL communicator (ncclComm_t passed as void*): windows, ncclAllGather of
  // the window sums on the object's stream, local add + fold.  ms[0] digits + sort, ms[1] buckets (accumulate + reduce),
  // ms[2] all-gather, ms[3] combine.  Collective: every rank of the communicator must call it.
  void run_sharded(void* nccl_comm, int nranks, uint8_t out_xy_le[64], float ms[4]);

 private:
  void enqueue_windows();
  struct Impl;
  std::unique_ptr<Impl> d_;
};

// device self-test of the 9 x 29-bit group law against the 8 x 32-bit one (msm.hip); group 1 = G1, 2 = G2 (pass the
// G2 generator as x.c0 | x.c1 | y.c0 | y.c1, canonical LE); returns the number of mismatching walks
uint32_t selftest_fq29(int group, uint32_t threads, uint32_t iters, const uint8_t* g2_gen_xy_le);

}  // namespace rlnamd
