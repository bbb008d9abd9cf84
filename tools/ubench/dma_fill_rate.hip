// Micro-benchmark: what does a CU ingest through LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) from an L2-resident weight image,
// as a function of the pieces a wavefront keeps in flight?  Eight wavefronts per workgroup, one or two workgroups per CU (64 KiB of LDS each),
// every wavefront issues K pieces, waits for all of them (s_waitcnt vmcnt(0)) and repeats; the image is 640 KiB (the backward's weight records of
// one transformer block), read round-robin like the chunk loop does.  Reports bytes per shader cycle and CU (shader cycles from s_memtime x clock ratio
// measured with __builtin_readcyclecounter -> use wall time instead: GB/s per CU and the chip total).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int K>
__global__ void __launch_bounds__(512) k_fill(const char *src, int iters, int image_kib, float *sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem);
  const unsigned voff = lane * 16;
  int pos = (blockIdx.x * 37) % image_kib;   // start somewhere in the image (workgroups are not in lockstep in the real kernels)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < K; ++p) {
      const int kib = (pos + p * 8 + wave) % image_kib;
      const char *g = src + (size_t)kib * 1024;
      const unsigned dst = lds0 + ((p * 8 + wave) % 56) * 1024;
      asm volatile("s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(g), "s"(dst) : "memory");
    }
    pos = (pos + K * 8) % image_kib;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (sink && threadIdx.x == 0 && smem[lane] == 77) sink[0] = 1.f;
}

template <int K>
void run(const char *src, int wgs, int iters, float *sink) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fill<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  k_fill<K><<<wgs, 512, 64 * 1024>>>(src, 10, 640, sink);
  (void)hipEventRecord(a);
  k_fill<K><<<wgs, 512, 64 * 1024>>>(src, iters, 640, sink);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * 8 * K * 1024 * iters;
  printf("K = %2d pieces in flight per wavefront, %3d workgroups of 8 wavefronts: %7.3f ms  %8.1f GB/s chip  %6.1f GB/s per CU (%5.1f B per cycle at 2.05 GHz)\n", K, wgs, ms,
         bytes / ms / 1e6, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.05);
}

int main() {
  char *src;
  float *sink;
  (void)hipMalloc(&src, 1 << 20);
  (void)hipMemset(src, 1, 1 << 20);
  (void)hipMalloc(&sink, 16);
  for (int wgs : {256, 512}) {
    run<1>(src, wgs, 4000, sink);
    run<2>(src, wgs, 4000, sink);
    run<4>(src, wgs, 2000, sink);
    run<6>(src, wgs, 2000, sink);
    run<10>(src, wgs, 1000, sink);
    run<16>(src, wgs, 1000, sink);
    run<24>(src, wgs, 500, sink);
    run<48>(src, wgs, 500, sink);
  }
  return 0;
}
