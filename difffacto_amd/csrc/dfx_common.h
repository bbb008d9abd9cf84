// Internal helpers shared by the libdfx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/dfx.h"

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

}  // namespace dfx
