// Micro-benchmark: LDS read bandwidth per CU for the weight-fragment access pattern of k_denoise_pipe
// (every wave streams the same 24 KiB record with ds_read_b128, lane l reads 16 B at l*16 + unit*1024).
// Build: hipcc --offload-arch=gfx950 -O3 lds_bw.hip -o lds_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int WIDTH>   // 16 = ds_read_b128, 8 = ds_read_b64
__global__ void __launch_bounds__(512) k(float *out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  for (int i = threadIdx.x; i < 24 * 1024 / 4; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = (float)i;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + lane * WIDTH;
  v4f acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    v4f r0, r1, r2, r3, r4, r5, r6, r7;
    if (WIDTH == 16) {
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                   "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr));
    } else {
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f q0, q1, q2, q3, q4, q5, q6, q7;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                   "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr));
      r0 = {q0[0], q1[0], q2[0], q3[0]}; r1 = {q4[0], q5[0], q6[0], q7[0]}; r2 = r3 = r4 = r5 = r6 = r7 = r0;
    }
    acc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    addr ^= 8192;   // alternate between two 8 KiB windows
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int WIDTH>
void run(float *d, int threads) {
  const int iters = 100000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<WIDTH><<<256, threads, 32 * 1024>>>(d, 1000);
  (void)hipEventRecord(a);
  k<WIDTH><<<256, threads, 32 * 1024>>>(d, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)iters * 8 * 64 * WIDTH * (threads / 64);   // per CU (1 block per CU)
  printf("width %2d B, %d waves/CU: %.1f bytes/ns per CU = %.1f B/clk @2.4 GHz (%.1f @2.1 GHz)\n", WIDTH, threads / 64,
         bytes / (ms * 1e6), bytes / (ms * 1e6) / 2.4, bytes / (ms * 1e6) / 2.1);
}

int main() {
  float *d; (void)hipMalloc(&d, 64);
  for (int threads : {64, 128, 256, 512}) { run<16>(d, threads); run<8>(d, threads); }
  return 0;
}
