// Prototype of the in-wave software-pipelined feed-forward stage written with builtins (what the real kernel
// would contain), to check that hipcc + sched_group_barrier produce the interleaved stream of stage_mix.hip.
//   stage: GEMM1 of half-chunk k+1 (8 chained MFMAs -> accN), GELU of half-chunk k (accC -> hidO, 60 VALU),
//          GEMM2 of half-chunk k-1 (4 MFMAs, hidP -> h[0..3]); 12 fragments + 4 bias reads from LDS.
// Build: hipcc --offload-arch=gfx950 -O3 stage_cxx.hip -o stage_cxx  [-DNO_SGB to drop the sched_group_barriers]
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef PATTERN
#define PATTERN 0
#endif
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ v8bf as_bf(const uint4 &u) { return __builtin_bit_cast(v8bf, u); }

__device__ __forceinline__ float gelu_arg(float x) { return x * fmaf(-0.100125614f, x * x, -2.30876530f); }

// One record = two stages = 24 fragments of 1 KiB in MFMA order: [W1 half-chunk A (8) | W2 (4) | W1 half-chunk B (8) | W2 (4)].
// F is an 8-deep fragment FIFO that runs across stage and record boundaries: MFMA m consumes F[m & 7], then the
// slot is refilled with fragment m + 8 (from the next record when m + 8 >= 24).  Bv holds the bias tile of the
// next GEMM1 (C operand of its first MFMA), refilled right after it is consumed.
template <int S>
__device__ __forceinline__ void gelu8(const v16f &accC, v8bf &hidO) {
  float u[8], ag[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = gelu_arg(accC[8 + i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) ag[i] = accC[i] * accC[8 + i];
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = __builtin_amdgcn_exp2f(u[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = 1.0f + u[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = __builtin_amdgcn_rcpf(u[i]);
  v8f t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = ag[i] * u[i];
  hidO = __builtin_convertvector(t, v8bf);
}

__device__ __forceinline__ void load_bias(v16f &B, const float *bias) {
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const float4 t = *reinterpret_cast<const float4 *>(bias + r4 * 4);
    B[r4 * 4 + 0] = t.x; B[r4 * 4 + 1] = t.y; B[r4 * 4 + 2] = t.z; B[r4 * 4 + 3] = t.w;
  }
}

__device__ __forceinline__ void record(v16f (&h)[4], const v8bf (&xn)[8], v16f &acc0, v16f &acc1, v8bf &hid0, v8bf &hid1,
                                       uint4 (&F)[8], v16f &Bv, const uint4 *ck, const uint4 *ck_next, const float *bias, const v16f &indep0, const v16f &indep1) {
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    v16f &accC = st ? acc1 : acc0;   // GELU input (GEMM1 of the previous stage)
    v16f &accN = st ? acc0 : acc1;   // GEMM1 output
    v8bf &hidP = st ? hid0 : hid1;   // GEMM2 input (GELU of the previous stage)
    v8bf &hidO = st ? hid1 : hid0;
#ifdef NO_DEP
    { v16f fake = accC; asm volatile("" : "+v"(fake)); gelu8<0>(st ? indep1 : indep0, hidO); }
#else
    gelu8<0>(accC, hidO);
#endif
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int m = st * 12 + i;
      if (i == 0) accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(F[m & 7]), xn[0], Bv, 0, 0, 0);
      else if (i < 8) accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(F[m & 7]), xn[i], accN, 0, 0, 0);
      else h[i - 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(F[m & 7]), hidP, h[i - 8], 0, 0, 0);
      F[m & 7] = (m + 8 < 24) ? ck[(m + 8) * 64] : ck_next[(m + 8 - 24) * 64];
#ifndef NO_BIAS
      if (i == 0) load_bias(Bv, bias + (st + 1) * 32);
#endif
    }
  }
#ifndef NO_SGB
#pragma unroll
  for (int st = 0; st < 2; ++st) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#ifndef NO_BIAS
      if (i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
      else
#endif
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#if PATTERN == 0
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
#elif PATTERN == 1      // nothing in the first two slots, then 6 per slot
      if (i >= 2) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
#elif PATTERN == 2      // plain VALU and transcendentals as separate groups
      if (i >= 2) { __builtin_amdgcn_sched_group_barrier(0x002, 5, 0); __builtin_amdgcn_sched_group_barrier(0x400, 2, 0); }
#endif
    }
#ifdef STAGE_FENCE
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
#endif
}

__global__ void __launch_bounds__(512, 2) k(float *out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  for (int i = threadIdx.x; i < 50 * 1024 / 4; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 1e-3f / (float)(i + 1);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  v16f h[4], acc0, acc1;
  v8bf xn[8], hid0, hid1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.01f * i; acc1[i] = 0.02f * i; for (int t = 0; t < 4; ++t) h[t][i] = 0.f; }
  for (int i = 0; i < 8; ++i) { hid0[i] = (__bf16)0.5f; hid1[i] = (__bf16)0.25f; for (int t = 0; t < 8; ++t) xn[t][i] = (__bf16)(0.1f * (t + i + lane)); }
  const uint4 *ring = reinterpret_cast<const uint4 *>(smem) + lane;
  const float *bias = reinterpret_cast<const float *>(smem + 48 * 1024) + (lane >> 5) * 16;
  uint4 F[8];
  v16f ind0 = acc0 + 1.0f, ind1 = acc1 + 2.0f;
  asm volatile("" : "+v"(ind0), "+v"(ind1));
#pragma unroll
  for (int i = 0; i < 8; ++i) F[i] = ring[i * 64];
  v16f Bv;
  load_bias(Bv, bias);
  for (int it = 0; it < iters; ++it) {
    const uint4 *ck = ring + (it & 1) * (24 * 1024 / 16);
    const uint4 *ckn = ring + ((it + 1) & 1) * (24 * 1024 / 16);
    __builtin_amdgcn_sched_barrier(0);
    record(h, xn, acc0, acc1, hid0, hid1, F, Bv, ck, ckn, bias, ind0, ind1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  float r = 0;
  for (int t = 0; t < 4; ++t) r += h[t][3];
  if (r == 12345.678f) out[0] = r;
}

int main() {
  float *d; (void)hipMalloc(&d, 64);
  const int iters = 40000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 50 * 1024);
  k<<<256, 512, 50 * 1024>>>(d, 1000);
  (void)hipEventRecord(a);
  k<<<256, 512, 50 * 1024>>>(d, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double ns = ms * 1e6 / iters / 2;
  printf("C++ stage: %.1f ns per stage = %.0f cycles @2.4 GHz (x2 = %.0f per record)\n", ns, ns * 2.4, ns * 4.8);
  return 0;
}
