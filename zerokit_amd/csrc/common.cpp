#include "common.h"
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
