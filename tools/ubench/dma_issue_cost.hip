// Micro-benchmark: how long does a wavefront's instruction stream stall when it ISSUES LDS-DMA loads
// (global_load_lds_dwordx4) compared with plain register loads (global_load_dwordx4)?  8 waves per workgroup as in
// k_denoise_pipe; every wave issues 3 pieces of 1 KiB per iteration (24 KiB per workgroup), then spins on VALU work.
// Reports the clock ticks (s_memtime) around the issue sequence only — not the completion.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 = LDS-DMA, 1 = register loads (+ ds_write one iteration later), 2 = LDS-DMA with 8 ds_reads in flight
__global__ void __launch_bounds__(512) k(const char *src, long long *out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem);
  const unsigned voff = lane * 16;
  long long total = 0;
  float acc = lane;
  v4f r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, q0 = r0;
  for (int it = 0; it < iters; ++it) {
    const char *g = src + ((size_t)(blockIdx.x * 8 + wave) * 64 + (it & 63)) * 3072;
    __builtin_amdgcn_s_barrier();
    if (MODE == 2) {
      asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:1024\n ds_read_b128 %0, %1 offset:2048\n ds_read_b128 %0, %1 offset:3072\n"
                   "ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1 offset:5120\n ds_read_b128 %0, %1 offset:6144\n ds_read_b128 %0, %1 offset:7168\n"
                   : "=v"(q0) : "v"(lds0 + 32768 + lane * 16));
    }
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 0 || MODE == 2) {
      asm volatile("s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %0, %1\n global_load_lds_dwordx4 %0, %1 offset:1024\n"
                   "global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(voff), "s"(g), "s"(lds0 + wave * 3072) : "memory");
    } else {
      asm volatile("global_load_dwordx4 %0, %3, %4\n global_load_dwordx4 %1, %3, %4 offset:1024\n global_load_dwordx4 %2, %3, %4 offset:2048"
                   : "=v"(r0), "=v"(r1), "=v"(r2) : "v"(voff), "s"(g) : "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
    for (int i = 0; i < 200; ++i) acc = acc * 1.0001f + 0.5f;   // ~1000 cycles of VALU work: lets the loads complete
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (MODE == 1) {
      asm volatile("ds_write_b128 %3, %0\n ds_write_b128 %3, %1 offset:1024\n ds_write_b128 %3, %2 offset:2048"
                   :: "v"(r0), "v"(r1), "v"(r2), "v"(lds0 + wave * 3072 + lane * 16) : "memory");
    }
  }
  if (lane == 0) out[blockIdx.x * 8 + wave] = total;
  if (acc + q0[0] == 12345.f) out[0] = 0;
}

template <int MODE>
void run(const char *src, long long *out, const char *name) {
  const int iters = 2000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  k<MODE><<<256, 512, 64 * 1024>>>(src, out, 10);
  k<MODE><<<256, 512, 64 * 1024>>>(src, out, iters);
  (void)hipDeviceSynchronize();
  long long h[2048];
  (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 2048; ++i) s += h[i];
  printf("%-48s %8.1f ticks per 3-piece issue (s_memtime ticks; 100 MHz -> x ~21 shader cycles)\n", name, s / 2048 / iters);
}

int main() {
  char *src; long long *out;
  (void)hipMalloc(&src, (size_t)256 * 8 * 64 * 3072);
  (void)hipMemset(src, 1, (size_t)256 * 8 * 64 * 3072);
  (void)hipMalloc(&out, 2048 * 8);
  run<0>(src, out, "LDS-DMA (global_load_lds_dwordx4 x3)");
  run<1>(src, out, "register loads (global_load_dwordx4 x3)");
  run<2>(src, out, "LDS-DMA with 8 ds_read_b128 in flight");
  return 0;
}
