// Does the virtual-memory API (hipMemCreate + hipMemMap) pay the same VRAM scrub as hipMalloc?  The 228 GiB table allocation
// of the throughput profile is 1 ms on a fresh box and 5 - 7.5 s when the memory was used before (the driver clears it inside
// the call).  Prints the time of: hipMalloc 224 GiB, free, hipMalloc again, free, then the same 224 GiB
// as hipMemCreate handles of 2 GiB mapped into one reserved range.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  CK(hipFree(0));
  const size_t GiB = 1ull << 30, total = 224 * GiB;
  void* p = nullptr;
  for (int round = 0; round < 2; round++) {
    double t0 = now();
    CK(hipMalloc(&p, total));
    double t1 = now();
    CK(hipMemset(p, 0x5A, total));
    CK(hipDeviceSynchronize());
    double t2 = now();
    CK(hipFree(p));
    printf("round %d: hipMalloc %.3f s, memset %.3f s, hipFree %.3f s\n", round, t1 - t0, t2 - t1, now() - t2);
  }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity %zu\n", gran);
  for (size_t chunk : {2 * GiB, 16 * GiB, 224 * GiB}) {
    double t0 = now();
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (size_t off = 0; off < total; off += chunk) {
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, chunk, &prop, 0));
      CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
      hs.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    double t1 = now();
    CK(hipMemset(va, 0x33, total));
    CK(hipDeviceSynchronize());
    double t2 = now();
    CK(hipMemUnmap(va, total));
    for (auto h : hs) CK(hipMemRelease(h));
    CK(hipMemAddressFree(va, total));
    printf("VMM chunks of %zu GiB: create + map %.3f s, memset %.3f s, release %.3f s\n", chunk / GiB, t1 - t0, t2 - t1, now() - t2);
  }
  return 0;
}
