// Does the VRAM scrub behind hipMalloc (about 30 ms per GiB of memory that was used before) run in parallel when several
// host threads allocate at once?      hipcc --offload-arch=gfx950 -O2 tools/malloc_parallel.hip -o /tmp/malloc_parallel -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t GiB = 1ull << 30;
static void dirty(size_t gib) {
  void* p;
  if (hipMalloc(&p, gib * GiB) != hipSuccess) { printf("dirty: alloc failed\n"); return; }
  (void)hipMemset(p, 0x5a, gib * GiB);
  (void)hipDeviceSynchronize();
  (void)hipFree(p);
}
int main() {
  (void)hipFree(0);
  const size_t total = 224;
  for (int threads : {1, 2, 4, 8, 1}) {
    dirty(total);
    std::vector<void*> p(threads, nullptr);
    std::vector<std::thread> th;
    double t0 = now();
    for (int i = 0; i < threads; i++)
      th.emplace_back([&, i] {
        (void)hipSetDevice(0);
        if (hipMalloc(&p[i], total / threads * GiB) != hipSuccess) printf("alloc failed\n");
      });
    for (auto& t : th) t.join();
    double dt = now() - t0;
    printf("%d thread(s) x %zu GiB: %.3f s\n", threads, total / threads, dt);
    t0 = now();
    for (auto q : p) (void)hipFree(q);
    printf("  free: %.3f s\n", now() - t0);
  }
  return 0;
}
