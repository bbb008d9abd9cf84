// Internal helpers shared by the libdfx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/dfx.h"
#include "../../include/dfx_debug.h"   // test / tuning hooks (not part of the drop-in ABI)

namespace dfx {

// Thread-local last-error message (dfx_last_error()).
char *err_buf();
int set_error(int code, const char *fmt, ...);

inline hipStream_t as_stream(dfx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Check the launch that was just issued (launch-configuration errors surface here; execution
// errors surface at the caller's next sync, as with any stream-ordered API).
inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(DFX_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return DFX_OK;
}

#define DFX_HIP_TRY(expr)                                                                   \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) return dfx::set_error(DFX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define DFX_REQUIRE(cond, ...)                                           \
  do {                                                                   \
    if (!(cond)) return dfx::set_error(DFX_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

// Kernels that need more than 64 KiB of dynamic LDS carry a per-function attribute that HIP keeps PER DEVICE: a guard per launch
// site, keyed by the current device (one bit each), set before the first launch there.  Thread-safe without a lock: the
// attribute call is idempotent, so two host threads racing on a device's first call both make it and both then launch correctly.
struct PerDeviceOnce {
  std::atomic<unsigned long long> done[4] = {};   // 256 devices
  template <class F>
  hipError_t run(F &&set_attributes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<unsigned long long> &word = done[(dev >> 6) & 3];
    const unsigned long long bit = 1ull << (dev & 63);
    if (word.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = set_attributes();
    if (e == hipSuccess) word.fetch_or(bit, std::memory_order_release);
    return e;
  }
};
inline hipError_t set_max_lds(const void *kernel, int bytes) {
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// A second HIP stream per (device, caller's stream) for work that is off the caller's critical path (the training step's context branch and its
// parameter-gradient reductions: small-grid kernels that leave most of the chip idle when they run one after the other on the caller's
// stream).  Every entry point that uses it forks from and joins back into the caller's stream before it returns, so the caller sees plain
// stream-ordered behaviour (and a stream capture of the caller's stream captures the side work with it).  Events come from the pair's own
// ring, one per fork / join; calls on different caller streams (or host threads with streams of their own) share nothing — not the side stream, so not
// each other's capture state either.
struct SideStream {
  hipStream_t main = nullptr, side = nullptr;
  void *impl = nullptr;       // the (device, caller's stream) pair's side stream + event ring
  bool on = false;
  bool pending = false;       // forked and not joined since: the destructor joins (an entry point's early error return must not leave side work
                              // running against a workspace the caller may free, nor a stream capture with an unjoined fork — ADVICE r4)
  ~SideStream() { if (on && pending) (void)join(); }
  // `enable` false: every launch "on the side stream" goes to the caller's stream (side == main, fork / join do nothing)
  int open(hipStream_t caller, bool enable);
  int fork();                 // the side stream waits for everything enqueued on the caller's stream so far
  int join();                 // the caller's stream waits for everything enqueued on the side stream so far
};

// Optional HIP-event timing of the hot-path launches (bench.py's roofline leg).
struct EventTimer {
  hipEvent_t a = nullptr, b = nullptr;
  bool active = false;
  hipStream_t st = nullptr;
  void begin(hipStream_t s);
  void end();
};
extern bool g_event_timing;
extern float g_last_ms;
extern const char *g_last_variant;   // name of the denoiser kernel the last launch() took (dfx_last_kernel_variant)

}  // namespace dfx
