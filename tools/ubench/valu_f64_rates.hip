// Issue cost (cycles per wavefront instruction, SIMD kept full by 4 wavefronts) of the instructions of the EMD auction's pair evaluation:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_f64_rates.hip -o tools/ubench/_build/valu_f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define F8(T) T a[8]; for (int i = 0; i < 8; ++i) a[i] = seed + i
#define BODY(NAME, DECL, ASM)                                                                                       \
  __global__ __launch_bounds__(1024) void NAME(long long *out, float seed) {                                         \
    DECL;                                                                                                            \
    long long t0 = clock64();                                                                                        \
    for (int it = 0; it < 256; ++it) {                                                                               \
      _Pragma("unroll") for (int r = 0; r < REP; ++r) { ASM; }                                                       \
    }                                                                                                                \
    long long t1 = clock64();                                                                                        \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                 \
  }
// eight independent chains per wave so that dependent-issue latency does not show
BODY(k_fma32, F8(float); float b = seed, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[r & 7]) : "v"(b)))
BODY(k_add64, F8(double); double b = seed, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[r & 7]) : "v"(b)))
BODY(k_cvt64_32, F8(float); double d[8], asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[r & 7]) : "v"(a[r & 7])))
BODY(k_cvt32_64, F8(double); float d[8], asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(d[r & 7]) : "v"(a[r & 7])))
BODY(k_sqrt32, F8(float), asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[r & 7])))
BODY(k_med3, F8(float); float b = seed, asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[r & 7]) : "v"(b)))
BODY(k_cndmask, F8(float); float b = seed, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[r & 7]) : "v"(b)))
BODY(k_cmp, F8(float); float b = seed, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[r & 7]), "v"(b) : "vcc"))
BODY(k_pkfma, F8(double); double b = seed, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[r & 7]) : "v"(b)))
BODY(k_cndmask64, F8(float); float b = seed, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[r & 7]) : "v"(b) : "s10", "s11"))
BODY(k_cmp_cnd, F8(float); float b = seed, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[r & 7]) : "v"(b) : "vcc"))
BODY(k_cmp64_cnd, F8(float); float b = seed, asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[r & 7]) : "v"(b) : "s10", "s11"))
BODY(k_addc, F8(int); int b = 3, asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[r & 7]) : "v"(b) : "vcc"))
BODY(k_addu32, F8(int); int b = 3, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[r & 7]) : "v"(b)))
BODY(k_max32, F8(float); float b = seed, asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[r & 7]) : "v"(b)))
template <class K> void run(const char *name, K k, int threads) {
  long long *d, h;
  hipMalloc(&d, 8);
  k<<<1, threads>>>(d, 1.5f);
  k<<<1, threads>>>(d, 1.5f);
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const int waves_per_simd = threads / 256;
  printf("%-14s %2d wavefronts / SIMD: %6.2f cycles per wavefront instruction\n", name, waves_per_simd, (double)h / (256.0 * REP * (waves_per_simd ? waves_per_simd : 1)));
  hipFree(d);
}
int main() {
  for (int threads : {1024}) {
    run("v_fma_f32", k_fma32, threads), run("v_pk_fma_f32", k_pkfma, threads), run("v_add_f64", k_add64, threads), run("v_cvt_f64_f32", k_cvt64_32, threads);
    run("v_cvt_f32_f64", k_cvt32_64, threads), run("v_sqrt_f32", k_sqrt32, threads), run("v_med3_f32", k_med3, threads), run("v_cndmask_b32", k_cndmask, threads);
    run("v_cmp_lt_f32", k_cmp, threads), run("cndmask sgpr", k_cndmask64, threads), run("cmp+cndmask vcc", k_cmp_cnd, threads);
    run("cmp+cnd sgpr", k_cmp64_cnd, threads), run("v_addc_co_u32", k_addc, threads), run("v_add_u32", k_addu32, threads), run("v_max_f32", k_max32, threads);
  }
  return 0;
}
