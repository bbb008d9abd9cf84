// What the fp32 matrix pipe sustains on this box: bare v_mfma_f32_32x32x2_f32 streams (4 accumulators per wavefront, operands in
// registers: random bits or zeros), 1-4 wavefronts per SIMD on every CU.  hipcc --offload-arch=gfx950 -O3 mfma_f32_power.hip -o mfma_f32_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(1024) void k(float *out, int iters, float scale) {
  v16f acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // operands: pseudo-random per lane (or all zero when scale == 0)
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u, a[i] = scale * ((h >> 8) * (1.f / 16777216.f) - 0.5f);
    h = h * 1664525u + 1013904223u, b[i] = scale * ((h >> 8) * (1.f / 16777216.f) - 0.5f);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + t) & 7], acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}
int main() {
  float *d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (float scale : {1.f, 0.f})
    for (int threads : {256, 512, 1024}) {
      const int iters = 20000, blocks = 256;
      k<<<blocks, threads>>>(d, 100, scale);
      hipEventRecord(e0);
      k<<<blocks, threads>>>(d, iters, scale);
      hipEventRecord(e1), hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * (threads / 64) * iters * 32 * 4096.0;
      printf("%s operands, %d wavefront(s) per SIMD: %7.1f TFLOP/s (%.1f %% of 157.3) over %.0f ms\n", scale ? "random" : "zero  ", threads / 256,
             flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100, ms);
    }
  return 0;
}
