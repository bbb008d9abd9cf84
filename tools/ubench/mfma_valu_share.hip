// Do VALU instructions of one wavefront run beside the MFMAs of the other wavefront of its SIMD?  512-thread workgroups (two wavefronts per
// SIMD): waves 0..3 issue NM MFMAs (fp32 32x32x2 or bf16 32x32x16, four independent accumulators), waves 4..7 issue NV v_fma_f32 (eight
// independent chains).  Each alone, then together: if the two streams overlapped, "together" = max(alone); if they share hardware, the sum.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_share.hip -o _build/mfma_valu_share
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool BF16>
__global__ void __launch_bounds__(512, 2) k(unsigned long long *out, int nm, int nv, float seed) {
  const int wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_readcyclecounter();
  float r = 0.f;
  if (wave < 4) {
    v16f c0 = {0}, c1 = c0, c2 = c0, c3 = c0;
    const float a = seed + threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const v4f av = {a, b, a, b}, bv = {b, a, b, a};
    for (int i = 0; i < nm; i += 4) {
      if (BF16) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv));
      } else {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n v_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n v_mfma_f32_32x32x2_f32 %3, %4, %5, %3"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
      }
    }
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
  } else {
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
    const float m = 1.0001f, d = 1e-3f;
#define F8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
           "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
    for (int i = 0; i < nv; i += 64)   // 64 per trip: the loop branch does not dominate
      asm volatile(F8 F8 F8 F8 F8 F8 F8 F8 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(d));
    r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
  if (r == 12345.678f) out[8] = 1;
}

template <bool BF16>
void run(const char *name, int nm, int nv) {
  unsigned long long *d, h[9];
  hipMalloc(&d, 9 * 8);
  k<BF16><<<1, 512>>>(d, 4, 8, 0.5f);
  k<BF16><<<1, 512>>>(d, nm, nv, 0.5f);
  hipDeviceSynchronize();
  hipMemcpy(h, d, 72, hipMemcpyDeviceToHost);
  printf("%-46s MFMA wave: %8llu cycles (%5.1f / MFMA)   VALU wave: %8llu cycles (%4.2f / v_fma)\n", name, h[0], nm ? (double)h[0] / nm : 0.0, h[4],
         nv ? (double)h[4] / nv : 0.0);
  hipFree(d);
}

int main() {
  const int NM = 4000, NV = 256000;
  run<false>("fp32 MFMAs alone", NM, 0);
  run<false>("v_fma_f32 alone", 0, NV);
  run<false>("fp32 MFMAs + v_fma_f32 on the same SIMD", NM, NV);
  run<true>("bf16 MFMAs alone", NM, 0);
  run<true>("bf16 MFMAs + v_fma_f32 on the same SIMD", NM, NV);
  return 0;
}
