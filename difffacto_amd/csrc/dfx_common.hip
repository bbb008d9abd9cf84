// libdfx: error plumbing, version, event timing.
#include "dfx_common.h"

#include <cstring>
#include <mutex>

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

namespace {
struct DevSide {
  hipStream_t st = nullptr;
  hipEvent_t ev[64] = {};
  std::atomic<unsigned> next{0};
};
std::mutex g_side_mu;
DevSide *g_side[256] = {};

DevSide *dev_side() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) return nullptr;
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (g_side[dev]) return g_side[dev];
  DevSide *d = new DevSide;
  if (hipStreamCreateWithFlags(&d->st, hipStreamNonBlocking) != hipSuccess) {
    delete d;
    return nullptr;
  }
  for (hipEvent_t &e : d->ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;   // (leaks a few handles on a broken device)
  g_side[dev] = d;
  return d;
}
hipEvent_t next_event(DevSide *d) { return d->ev[d->next.fetch_add(1, std::memory_order_relaxed) & 63]; }
}  // namespace

int SideStream::open(hipStream_t caller, bool enable) {
  main = side = caller;
  on = false;
  if (!enable) return DFX_OK;
  DevSide *d = dev_side();
  if (!d) return set_error(DFX_ERR_HIP, "side stream: cannot create the per-device stream / events");
  side = d->st;
  on = true;
  return DFX_OK;
}
int SideStream::fork() {
  if (!on) return DFX_OK;
  hipEvent_t e = next_event(dev_side());
  DFX_HIP_TRY(hipEventRecord(e, main));
  DFX_HIP_TRY(hipStreamWaitEvent(side, e, 0));
  return DFX_OK;
}
int SideStream::join() {
  if (!on) return DFX_OK;
  hipEvent_t e = next_event(dev_side());
  DFX_HIP_TRY(hipEventRecord(e, side));
  DFX_HIP_TRY(hipStreamWaitEvent(main, e, 0));
  return DFX_OK;
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
