// libdfx: error plumbing, version, event timing.
#include "dfx_common.h"

#include <cstring>

namespace dfx {

static thread_local char t_err[512] = "";

char *err_buf() { return t_err; }

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_err, sizeof(t_err), fmt, ap);
  va_end(ap);
  return code;
}

bool g_event_timing = false;
float g_last_ms = -1.0f;
const char *g_last_variant = "";

void EventTimer::begin(hipStream_t s) {
  active = g_event_timing;
  if (!active) return;
  st = s;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
    active = false;
    return;
  }
  (void)hipEventRecord(a, st);
}

void EventTimer::end() {
  if (!active) return;
  (void)hipEventRecord(b, st);
  (void)hipEventSynchronize(b);
  float ms = -1.0f;
  if (hipEventElapsedTime(&ms, a, b) == hipSuccess) g_last_ms = ms;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  active = false;
}

}  // namespace dfx

extern "C" {

int dfx_version(void) { return 100; }
int dfx_abi_version(void) { return DFX_ABI_VERSION; }

const char *dfx_last_error(void) { return dfx::err_buf(); }

void dfx_set_event_timing(int enable) { dfx::g_event_timing = enable != 0; }

float dfx_last_kernel_ms(void) { return dfx::g_last_ms; }

const char *dfx_last_kernel_variant(void) { return dfx::g_last_variant; }

}  // extern "C"
