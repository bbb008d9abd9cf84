// What v_permlane32_swap does, lane by lane:  hipcc --offload-arch=gfx950 -O2 permlane_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  unsigned x = 1000 + threadIdx.x, y = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[threadIdx.x] = r[0], out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned *d, h[128];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r0: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
  printf("r1: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
