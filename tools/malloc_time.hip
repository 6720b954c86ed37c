#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  double t0 = now();
  hipFree(0);
  printf("init %.3f s\n", now() - t0);
  size_t GiB = 1ull << 30;
  size_t sizes[] = {16, 64, 115, 113};
  void* p[4];
  for (int i = 0; i < 4; i++) {
    t0 = now();
    hipError_t e = hipMalloc(&p[i], sizes[i] * GiB);
    printf("hipMalloc %zu GiB: %.3f s (%s)\n", sizes[i], now() - t0, hipGetErrorString(e));
  }
  for (int i = 0; i < 4; i++) { t0 = now(); hipFree(p[i]); printf("hipFree %zu GiB: %.3f s\n", sizes[i], now() - t0); }
  for (int i = 2; i < 4; i++) {
    t0 = now();
    hipError_t e = hipMalloc(&p[i], sizes[i] * GiB);
    printf("again hipMalloc %zu GiB: %.3f s (%s)\n", sizes[i], now() - t0, hipGetErrorString(e));
  }
  // touch: a memset of the second
  t0 = now(); hipMemset(p[3], 0, 113 * GiB); hipDeviceSynchronize(); printf("memset 113 GiB: %.3f s\n", now() - t0);
  return 0;
}
