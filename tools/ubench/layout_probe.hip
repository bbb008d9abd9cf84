// rows_to_acc (train_ff_fused.h) against a direct load in the accumulator layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../difffacto_amd/csrc/train_ff_fused.h"
using namespace dfx::ffused;
__global__ void k(const float *h, float *out) {
  const int lane = threadIdx.x & 63, hf = lane >> 5, pj = lane & 31;
  v8f x[4][2];
  load_rows(h + pj * C, hf, x);
  v16f d[4];
  rows_to_acc(x, d);
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) out[(lane * 4 + c) * 16 + r] = d[c][r] - h[pj * C + 32 * c + rho(r, hf)];
}
int main() {
  float *h, *o, hh[32 * 128], ho[64 * 64];
  for (int i = 0; i < 32 * 128; ++i) hh[i] = (float)i;
  hipMalloc(&h, sizeof(hh)), hipMalloc(&o, sizeof(ho));
  hipMemcpy(h, hh, sizeof(hh), hipMemcpyHostToDevice);
  k<<<1, 64>>>(h, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64 * 64; ++i) if (ho[i] != 0.f) { if (bad < 8) printf("lane %d c %d r %d diff %g\n", i / 64, (i / 16) & 3, i & 15, ho[i]); ++bad; }
  printf("bad %d\n", bad);
  return 0;
}
