// Does feeding v_mfma_f32_32x32x2_f32 from LDS / global memory (the K loop of k_lin_wide_lds: per 16 MFMAs four ds_read_b128 of weight
// fragments and one global_load_dwordx4 of activations, one step ahead, two register sets) cost matrix-pipe time?  2 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_feed.hip -o tools/ubench/_build/mfma_f32_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: operands in registers; 1: weights from LDS; 2: + activations from global memory; 3: + the epilogue's stores; 4: + weights staged from global memory
__global__ __launch_bounds__(512) void k(const float *__restrict__ X, float *out, int U, int tiles, const float *__restrict__ W, float *__restrict__ Y) {
  extern __shared__ __align__(16) float smem[];
  v4f *wl = reinterpret_cast<v4f *>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 4 * U * 64; e += 512) {
    const int l = e & 63, tu = e >> 6, u = tu % U, t = tu / U;
    wl[e] = MODE >= 4 ? *reinterpret_cast<const v4f *>(W + (size_t)(32 * t + (l & 31)) * (8 * U) + 8 * u + 4 * (l >> 5)) : v4f{1.f + e, 2.f, 3.f, 4.f} * 1e-3f;
  }
  __syncthreads();
  const v4f *wlane = wl + lane;
  float s = 0.f;
  for (int tl = 0; tl < tiles; ++tl) {
    const float *xp = X + ((size_t)(blockIdx.x * tiles + tl) * 8 + wave) * 32 * (8 * U) + (size_t)(lane & 31) * (8 * U) + 4 * (lane >> 5);
    v16f acc[4];
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto operands = [&](int u, v4f &xv, v4f (&wv)[4]) {
      xv = MODE >= 2 ? *reinterpret_cast<const v4f *>(xp + 8 * u) : v4f{1.f, 2.f, 3.f, 4.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) wv[t] = MODE >= 1 ? wlane[(t * U + u) * 64] : v4f{1.f, 2.f, 3.f, 4.f};
    };
    auto step = [&](const v4f &xv, const v4f (&wv)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t][q], xv[q], acc[t], 0, 0, 0);
    };
    v4f xa, xb, wa[4], wb[4];
    operands(0, xa, wa);
    for (int u = 0; u + 1 < U; u += 2) {
      operands(u + 1, xb, wb);
      __builtin_amdgcn_sched_barrier(0);
      step(xa, wa);
      operands(min(u + 2, U - 1), xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      step(xb, wb);
    }
    if (MODE >= 3) {   // 128 channels of 32 rows, 16 bytes per lane and store, row stride 512 floats (the 256 -> 512 layer's output)
      float *yp = Y + ((size_t)(blockIdx.x * tiles + tl) * 256 + wave * 32 + (lane & 31)) * 512;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v4f v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[t][4 * q + e] + 1.f;
          *reinterpret_cast<v4f *>(yp + 32 * t + 8 * q + 4 * (lane >> 5)) = v;
        }
    } else {
      for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    }
  }
  if (s == 123.456f) out[0] = s;
}
template <int MODE> void run(const char *name, const float *X, float *d, int U, int tiles, const float *W, float *Y) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * U * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<MODE><<<256, 512, 4 * U * 1024>>>(X, d, U, 2, W, Y);
  hipEventRecord(e0);
  k<MODE><<<256, 512, 4 * U * 1024>>>(X, d, U, tiles, W, Y);
  hipEventRecord(e1), hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 * tiles * U * 16 * 4096.0;
  printf("%-44s %7.1f TFLOP/s (%.1f %% of 157.3) over %.1f ms\n", name, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100, ms);
}
int main() {
  const int U = 32, tiles = 16;   // K = 256, 16 row tiles of 256 per workgroup: the 256 -> 512 layer's shape per workgroup
  float *X, *d;
  hipMalloc(&X, (size_t)256 * tiles * 256 * 8 * U * 4), hipMalloc(&d, 4);
  hipMemset(X, 0, (size_t)256 * tiles * 256 * 8 * U * 4);
  float *W, *Y;
  hipMalloc(&W, (size_t)128 * 8 * U * 4), hipMalloc(&Y, (size_t)256 * tiles * 256 * 512 * 4);
  hipMemset(W, 0, (size_t)128 * 8 * U * 4);
  run<0>("operands in registers", X, d, U, tiles, W, Y);
  run<1>("weights from LDS", X, d, U, tiles, W, Y);
  run<2>("weights from LDS, activations from HBM", X, d, U, tiles, W, Y);
  run<3>("... + epilogue stores", X, d, U, tiles, W, Y);
  run<4>("... + weight block staged from memory", X, d, U, tiles, W, Y);
  return 0;
}
