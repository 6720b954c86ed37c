#include "common.h"

#include <stdlib.h>

// The prover keeps seven HIP streams busy (two graph interpreters, mat-vec/NTT, MSM, back end, two proof-value
// streams).  ROCclr multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a
// queue serialise, which cost 25 % of the throughput when measured.  The variable is read when the HIP runtime
// initialises, so it is set when this library is loaded -- unless the host program already chose a value.
namespace {
struct HwQueueDefault {
  HwQueueDefault() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }
} hw_queue_default;
}  // namespace
namespace rlnamd {
void require_gpu() {
  static int checked = 0;
  if (checked == 1) return;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    throw Error("no HIP device available: this library has no CPU fallback (MI355X / gfx950 required)");
  hipDeviceProp_t prop;
  RLN_HIP(hipGetDeviceProperties(&prop, 0));
  checked = 1;
}
}  // namespace rlnamd
