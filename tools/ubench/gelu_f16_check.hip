// Check of the packed-fp16 GELU helper against the exact erf form (standalone; mirrors gelu16_f16 of denoiser_kernel.hip)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 pk_f16(float lo, float hi) { return __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
__device__ __forceinline__ h2 exp2_h2(h2 y) {
  h2 e;
  asm("v_exp_f16_e32 %0, %1\n\ts_nop 1\n\tv_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 1" : "=&v"(e) : "v"(y));
  return e;
}
__device__ __forceinline__ h2 rcp_h2(h2 d) {
  h2 r;
  asm("v_rcp_f16_e32 %0, %1\n\ts_nop 1\n\tv_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 1" : "=&v"(r) : "v"(d));
  return r;
}
__global__ void k(const float *a, const float *g, float *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const h2 c1 = {(_Float16)-2.30876530f, (_Float16)-2.30876530f}, c3 = {(_Float16)-0.100125614f, (_Float16)-0.100125614f};
  const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
  const h2 gg = pk_f16(g[2 * i], g[2 * i + 1]);
  h2 y = gg * gg;
  y = __builtin_elementwise_fma(y, c3, c1);
  y = gg * y;
  const h2 ag = pk_f16(a[2 * i], a[2 * i + 1]) * gg;
  y = exp2_h2(y);
  y = one + y;
  y = rcp_h2(y);
  y = ag * y;
  out[2 * i] = (float)y[0];
  out[2 * i + 1] = (float)y[1];
}
int main() {
  const int n = 1 << 16;
  float *ha = new float[n], *hg = new float[n], *ho = new float[n];
  for (int i = 0; i < n; ++i) { hg[i] = -12.f + 24.f * i / n; ha[i] = 0.3f + (i % 7) * 0.2f; }
  hg[0] = -300.f; hg[1] = 300.f; hg[2] = 0.f; hg[3] = -1e4f; hg[5] = 1e4f;
  float *a, *g, *o;
  (void)hipMalloc(&a, n * 4); (void)hipMalloc(&g, n * 4); (void)hipMalloc(&o, n * 4);
  (void)hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(g, hg, n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(a, g, o, n);
  (void)hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxrel = 0; int nan = 0;
  for (int i = 0; i < n; ++i) {
    const double ref = ha[i] * 0.5 * hg[i] * (1.0 + erf(hg[i] / sqrt(2.0)));
    if (std::isnan(ho[i])) { ++nan; if (nan < 5) printf("NaN at %d g=%f a=%f\n", i, hg[i], ha[i]); continue; }
    if (fabs(hg[i]) > 250) { if (i < 8) printf("g=%g a=%g -> %g (ref %g)\n", hg[i], ha[i], ho[i], ref); continue; }
    const double e = fabs(ho[i] - ref);
    if (e > maxerr) maxerr = e;
    if (fabs(ref) > 1e-2 && e / fabs(ref) > maxrel) maxrel = e / fabs(ref);
  }
  printf("max abs err %.3e, max rel err (|ref|>1e-2) %.3e, NaNs %d\n", maxerr, maxrel, nan);
  return 0;
}
