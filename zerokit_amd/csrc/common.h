// Shared host-side plumbing for the HIP modules: error propagation (never abort across the C ABI --
// the reference stringifies errors into CResult.err, rln/src/ffi/ffi_rln.rs:52-55), device buffers,
// HIP-event stage timers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace rlnamd {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define RLN_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      throw ::rlnamd::Error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " __FILE__ ":" + \
                            std::to_string(__LINE__) + " (" #expr ")");                           \
  } while (0)

// The product has no CPU fallback: every entry point that computes calls this first.
void require_gpu();

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    if (count) RLN_HIP(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void upload(const T* src, size_t count, hipStream_t s = 0) {
    RLN_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(T* dst, size_t count, hipStream_t s = 0) const {
    RLN_HIP(hipMemcpyAsync(dst, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
  }
  size_t bytes() const { return n * sizeof(T); }
};

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace rlnamd
