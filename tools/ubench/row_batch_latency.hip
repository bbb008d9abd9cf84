// Micro-benchmark: how long does ONE batch of row loads take to land?  A wavefront requests 32 x 1 KiB (32 global_load_dwordx4, contiguous 32 KiB of fresh
// HBM-resident data — the prologue of the training kernels' k_ff<true> with tile-major rows) and waits for all of it; between two batches it sleeps
// `gap` x 64 cycles (duty cycle of the memory phases).  512 workgroups of 4 wavefronts (two per CU) as in the kernel.  Reports the mean batch latency in
// shader cycles (s_memtime runs at 100 MHz: __builtin_readcyclecounter is the shader clock on gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) k(const float *src, long long *out, int iters, int gap, size_t span_floats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wid = (size_t)blockIdx.x * 4 + wave;
  long long total = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float *p = src + ((wid * 8192 + (size_t)it * 2048 * 8192) % span_floats) + lane * 4;   // 32 KiB per wavefront and batch, fresh addresses
    v4f r[32];
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 32; ++i) r[i] = *reinterpret_cast<const v4f *>(p + i * 256);
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += r[i][0] + r[i][3];
    const long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
    for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(1);
  }
  if (lane == 0) out[wid] = total;
  if (acc == 12345.f) out[0] = 0;
}

int main() {
  const size_t span = (size_t)1 << 30;   // 4 GiB of floats
  float *src;
  long long *out;
  (void)hipMalloc(&src, span * 4);
  (void)hipMemset(src, 0, span * 4);
  (void)hipMalloc(&out, 2048 * 8);
  for (int gap : {0, 50, 150, 400, 1200}) {
    const int iters = 60;
    k<<<512, 256>>>(src, out, 4, gap, span);
    hipEvent_t a, b;
    (void)hipEventCreate(&a), (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    k<<<512, 256>>>(src, out, iters, gap, span);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    long long h[2048];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 2048; ++i) s += h[i];
    printf("gap %5d x 64 cycles between batches: batch of 32 KiB per wavefront lands in %8.0f cycles on average; chip %7.1f GB/s over the run\n", gap, s / 2048 / iters,
           2048.0 * 32768 * iters / ms / 1e6);
  }
  return 0;
}
