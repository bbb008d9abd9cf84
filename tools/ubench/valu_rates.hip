// Micro-benchmark: issue cost of VALU / transcendental / packed / MFMA instructions on gfx950,
// alone and next to MFMAs, at 1 and 2 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(1024) k(float *out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v16f acc0 = {0}, acc1 = {0};
  v8bf A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(seed + i); B[i] = (__bf16)(seed - i); }
  h2 p0 = {(_Float16)seed, (_Float16)1.f}, p1 = p0, p2 = p0, p3 = p0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {  // v_fma_f32, 8 independent chains
      REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                        "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 1) {  // v_exp_f32
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 2) {  // v_rcp_f32
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 3) {  // v_pk_fma_f16
      REP8(asm volatile("v_pk_fma_f16 %0, %0, %0, %0\n v_pk_fma_f16 %1, %1, %1, %1\n v_pk_fma_f16 %2, %2, %2, %2\n v_pk_fma_f16 %3, %3, %3, %3\n"
                        "v_pk_fma_f16 %0, %0, %0, %0\n v_pk_fma_f16 %1, %1, %1, %1\n v_pk_fma_f16 %2, %2, %2, %2\n v_pk_fma_f16 %3, %3, %3, %3\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (KIND == 4) {  // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                        "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                        : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));)
    } else if (KIND == 5) {  // MFMA only, 2 accumulators: 8 MFMAs per REP
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0);
           acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0);
           acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0);
           acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0);)
    } else if (KIND == 6 || KIND == 7 || KIND == 8 || KIND == 9) {  // 1 MFMA + F fillers (v_fma_f32), F = 4 / 8 / 12 / 6 fma + 2 exp
      REP8(
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0);
          if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
          if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                                      "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
          if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                                      "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                                      "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
          if (KIND == 9) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_exp_f32 %3, %3\n"
                                      "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_rcp_f32 %7, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 10) {  // ds_read_b128 broadcast-free stream handled elsewhere
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[3] + (float)p0[0] + (float)p1[1] + (float)p2[0] + (float)p3[1];
  if (r == 12345.678f) out[0] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}


// role-specialised waves in one block: waves [0, mf) run MFMAs only, the others VALU only.
template <int VKIND>
__global__ void __launch_bounds__(1024) k_roles(float *out, int iters, float seed, int nv_waves_start) {
  const int wave = threadIdx.x >> 6;
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v16f acc0 = {0}, acc1 = {0};
  v8bf A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(seed + i); B[i] = (__bf16)(seed - i); }
  if (wave < nv_waves_start) {
    for (int it = 0; it < iters; ++it) {
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0);
           acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0);
           acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0);
           acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0);)
    }
  } else {
    for (int it = 0; it < iters * VKIND; ++it) {
      REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                        "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[3];
  if (r == 12345.678f) out[0] = r;
}
template <int VKIND>
void run_roles(int mf_waves, int v_waves) {
  float *d; hipMalloc(&d, 64);
  const int iters = 2000;
  k_roles<VKIND><<<256, 64 * (mf_waves + v_waves)>>>(d, 10, 1.0f, mf_waves);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k_roles<VKIND><<<256, 64 * (mf_waves + v_waves)>>>(d, iters, 1.0f, mf_waves);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("roles: %d MFMA waves + %d VALU waves per CU, VALU instr per wave = %d x MFMA count: %.3f ms  (alone: MFMA waves %.3f ms, VALU waves %.3f ms)\n",
         mf_waves, v_waves, VKIND, ms, mf_waves ? iters * 64 * 36.1 / 2.4e6 * ((mf_waves + 3) / 4) : 0.0,
         v_waves ? iters * VKIND * 64 * (v_waves > 4 ? 3.2 * ((v_waves + 3) / 4) : 5.5) / 2.4e6 : 0.0);
  hipFree(d);
}

template <int KIND>
void run(const char *name, int ops_per_iter, int wavesPerSimd) {
  float *d;
  hipMalloc(&d, 64);
  const int iters = 2000;
  const int blocks = 256, threads = 256 * wavesPerSimd;  // 1 block per CU, 4*w waves
  k<KIND><<<blocks, threads>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<KIND><<<blocks, threads>>>(d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  const double cyc = h[1];  // s_memtime ticks (100 MHz?) -> use wall time instead
  const double instr_per_wave = (double)iters * ops_per_iter;
  // cycles per instruction per SIMD at nominal 2.4 GHz: time * 2.4e9 / (instr_per_wave * wavesPerSimd)
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f SIMD-cycles@2.4GHz per instr (per wave: %.2f)  [memtime ticks %.0f]\n", name, wavesPerSimd, ms,
         ms * 1e-3 * 2.4e9 / (instr_per_wave * wavesPerSimd), ms * 1e-3 * 2.4e9 / instr_per_wave, cyc);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_fma_f32", 64, w);
    run<1>("v_exp_f32", 64, w);
    run<2>("v_rcp_f32", 64, w);
    run<3>("v_pk_fma_f16", 64, w);
    run<4>("v_pk_fma_f32", 64, w);
    run<5>("mfma_32x32x16_bf16", 64, w);
    run<6>("1 mfma + 4 fma  (per group)", 8, w);
    run<7>("1 mfma + 8 fma  (per group)", 8, w);
    run<8>("1 mfma + 12 fma (per group)", 8, w);
    run<9>("1 mfma + 6fma+exp+rcp", 8, w);
  }
  run_roles<1>(4, 0);
  run_roles<1>(0, 4);
  run_roles<1>(4, 4);
  run_roles<4>(4, 4);
  run_roles<7>(4, 4);
  run_roles<4>(4, 8);
  run_roles<4>(8, 8);
  run_roles<2>(4, 12);
  run<7>("1 mfma + 8 fma  (per group)", 8, 3);
  run<7>("1 mfma + 8 fma  (per group)", 8, 4);
  return 0;
}
