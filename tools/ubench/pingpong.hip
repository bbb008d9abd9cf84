// Micro-benchmark of the slot structure of k_denoise_pipe: 8 waves / WG, groups A (waves 0-3) and B (4-7) alternate a
// pure MFMA burst (24 x 32x32x16 bf16) and a pure VALU burst (NV plain + NT transcendental ops) with one
// s_barrier per slot; B is one slot behind A.  Reports cycles (at 2.4 GHz nominal) per slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int MODE>  // 0: anti-phase ping-pong, 1: lock-step (all waves same phase), 2: no barriers (anti-phase start), 3: M only, 4: V only
__global__ void __launch_bounds__(512, 2) k(float *out, int slots, float seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool grpA = wave < 4;
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
  v16f h0 = {0}, h1 = {0}, h2 = {0}, h3 = {0}, ag = {0}, gg = {0};
  v8bf A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(seed + i); B[i] = (__bf16)(seed - i); }
  if (MODE == 0 && !grpA) __builtin_amdgcn_s_barrier();
  for (int s = 0; s < slots; ++s) {
    const bool mslot = (MODE == 3) || (MODE != 4 && ((s & 1) == 0));
    if (MODE == 0 || MODE == 1) __builtin_amdgcn_s_barrier();
    if (MODE == 2 && s == 0 && !grpA) {  // start B with a V slot
      // fallthrough: B just begins with the V branch below by flipping parity
    }
    const bool m = (MODE == 2 && !grpA) ? !mslot : mslot;
    if (m) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, h1, 0, 0, 0);
        h2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, h2, 0, 0, 0);
        h3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, h3, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ag = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ag, 0, 0, 0);
        gg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, gg, 0, 0, 0);
      }
    } else {
      // 16 elements x (8 plain + exp + rcp)  =  128 plain + 32 transcendental, 2 independent chains at a time like hipcc emits
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        asm volatile(
            "v_med3_f32 %0, %0, %2, %2\n v_med3_f32 %1, %1, %2, %2\n"
            "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n"
            "v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n"
            "v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n"
            "v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n"
            "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n"
            "v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n"
            "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n"
            "v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n"
            "v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n"
            : "+v"(a[e & 7]), "+v"(a[(e + 1) & 7]) : "v"(seed));
      }
    }
  }
  if (MODE == 0 && grpA) __builtin_amdgcn_s_barrier();
  float r = h0[0] + h1[1] + h2[2] + h3[3] + ag[4] + gg[5];
  for (int i = 0; i < 8; ++i) r += a[i];
  if (r == 12345.678f) out[0] = r;
}

template <int MODE>
void run(const char *name) {
  float *d; hipMalloc(&d, 64);
  const int slots = 4000;
  k<MODE><<<256, 512>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<256, 512>>>(d, slots, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms  -> %.0f cycles@2.4GHz per slot\n", name, ms, ms * 1e-3 * 2.4e9 / slots);
  hipFree(d);
}

int main() {
  run<3>("M only (24 MFMA/slot, 2 waves/SIMD)");
  run<4>("V only (160 VALU/slot, 2 waves/SIMD)");
  run<1>("lock-step M,V alternating + barrier");
  run<0>("anti-phase ping-pong + barrier");
  run<2>("anti-phase start, no barriers");
  return 0;
}
