// Micro-benchmark: does a wavefront's own VALU work overlap with its MFMAs on gfx950, and does it depend on
// whether the MFMA accumulators live in VGPRs or AGPRs?  One wave per SIMD (256-thread blocks), 256 blocks.
// Build: hipcc --offload-arch=gfx950 -O3 agpr_overlap.hip -o agpr_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define FMA6 "v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
#define MIX6 "v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_rcp_f32 %7, %7\n"
#define PKH3 "v_pk_fma_f16 %2, %2, %2, %2\n v_pk_fma_f16 %3, %3, %3, %3\n v_pk_fma_f16 %4, %4, %4, %4\n"
#define MIXH6 "v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_exp_f16 %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_rcp_f16 %7, %7\n"
#define PK3 "v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n"

// MODE: 0 = MFMA only (VGPR acc), 1 = MFMA only (AGPR acc), 2 = MFMA(V) + 6 fma, 3 = MFMA(A) + 6 fma, 4 = 6 fma only,
//       5 = MFMA(A) + 3 pk_fma, 6 = 3 pk_fma only, 7 = MFMA(A)+12 fma, 8 = 12 fma only, 9 = MFMA(V)+12 fma
template <int MODE>
__global__ void __launch_bounds__(512) k(float *out, int iters, float seed) {
  float f0 = seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5;
  double d0 = seed, d1 = seed + 1, d2 = seed + 2;
  v16f acc0, acc1;
  for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
  __shared__ float lds[4096];
  lds[threadIdx.x] = seed;
  __syncthreads();
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f L = {0, 0, 0, 0};
  unsigned laddr = (threadIdx.x & 63) * 16;
  v8bf A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(seed + i); B[i] = (__bf16)(seed - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n" : "+v"(acc0), "+v"(acc1) : "v"(A), "v"(B));
      } else if (MODE == 1) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n" : "+a"(acc0), "+a"(acc1) : "v"(A), "v"(B));
      } else if (MODE == 2) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n" FMA6 "v_mfma_f32_32x32x16_bf16 %1, %8, %9, %1\n" FMA6
                     : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(A), "v"(B));
      } else if (MODE == 3) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n" FMA6 "v_mfma_f32_32x32x16_bf16 %1, %8, %9, %1\n" FMA6
                     : "+a"(acc0), "+a"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(A), "v"(B));
      } else if (MODE == 4) {
        asm volatile(FMA6 FMA6 : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5));
      } else if (MODE == 5) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %5, %6, %0\n" PK3 "v_mfma_f32_32x32x16_bf16 %1, %5, %6, %1\n" PK3
                     : "+a"(acc0), "+a"(acc1), "+v"(d0), "+v"(d1), "+v"(d2) : "v"(A), "v"(B));
      } else if (MODE == 6) {
        asm volatile(PK3 PK3 : "+v"(acc0), "+v"(acc1), "+v"(d0), "+v"(d1), "+v"(d2));
      } else if (MODE == 7) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n" FMA6 FMA6 "v_mfma_f32_32x32x16_bf16 %1, %8, %9, %1\n" FMA6 FMA6
                     : "+a"(acc0), "+a"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(A), "v"(B));
      } else if (MODE == 8) {
        asm volatile(FMA6 FMA6 FMA6 FMA6 : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5));
      } else if (MODE == 10) {   // single accumulator: every MFMA depends on the previous one
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n" FMA6 "v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n" FMA6
                     : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(A), "v"(B));
      } else if (MODE == 11) {   // 4 fma + exp + rcp
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n" MIX6 "v_mfma_f32_32x32x16_bf16 %1, %8, %9, %1\n" MIX6
                     : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(A), "v"(B));
      } else if (MODE == 12) {   // the same VALU mix alone
        asm volatile(MIX6 MIX6 : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5));
      } else if (MODE == 13) {   // single accumulator, MFMA only
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n" : "+v"(acc0), "+v"(acc1) : "v"(A), "v"(B));
      } else if (MODE == 14) {   // MFMA + ds_read_b128 + 5 fma
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n ds_read_b128 %10, %11\n" FMA6 "v_mfma_f32_32x32x16_bf16 %1, %8, %9, %1\n ds_read_b128 %10, %11 offset:4096\n" FMA6 "s_waitcnt lgkmcnt(0)\n"
                     : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(A), "v"(B), "v"(L), "v"(laddr));
      } else if (MODE == 15) {   // MFMA + 3 packed-f16 FMAs
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %5, %6, %0\n" PKH3 "v_mfma_f32_32x32x16_bf16 %1, %5, %6, %1\n" PKH3
                     : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2) : "v"(A), "v"(B));
      } else if (MODE == 16) {
        asm volatile(PKH3 PKH3 : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2));
      } else if (MODE == 17) {   // f16 transcendentals in the mix
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n" MIXH6 "v_mfma_f32_32x32x16_bf16 %1, %8, %9, %1\n" MIXH6
                     : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(A), "v"(B));
      } else if (MODE == 18) {
        asm volatile(MIXH6 MIXH6 : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5));
      } else if (MODE == 9) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n" FMA6 FMA6 "v_mfma_f32_32x32x16_bf16 %1, %8, %9, %1\n" FMA6 FMA6
                     : "+v"(acc0), "+v"(acc1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(A), "v"(B));
      }
    }
  }
  float r = L[0] + f0 + f1 + f2 + f3 + f4 + f5 + acc0[0] + acc1[3] + (float)(d0 + d1 + d2);
  if (r == 12345.678f) out[0] = r;
}

template <int MODE>
float run(float *d, int iters, int threads) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, threads>>>(d, 1000, 1.0f);
  hipEventRecord(a);
  k<MODE><<<256, threads>>>(d, iters, 1.0f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e6f / iters / 8.f;   // ns per (1 MFMA [+ fillers]) group; 8 groups per iteration
}

int main() {
  float *d; hipMalloc(&d, 64);
  const int it = 200000;
  const char *names[] = {"MFMA only (VGPR acc)", "MFMA only (AGPR acc)", "MFMA(V) + 6 fma", "MFMA(A) + 6 fma", "6 fma only",
                         "MFMA(A) + 3 pk_fma", "3 pk_fma only", "MFMA(A) + 12 fma", "12 fma only", "MFMA(V) + 12 fma",
                         "MFMA(1 acc) + 6 fma", "MFMA(V) + 4fma+exp+rcp", "4fma+exp+rcp only", "MFMA only (1 acc)", "MFMA(V)+ds_read+6fma",
                         "MFMA + 3 pk_fma_f16", "3 pk_fma_f16 only", "MFMA + 4fma+exp_f16+rcp_f16", "4fma+exp_f16+rcp_f16 only"};
  for (int threads : {256, 512}) {
    printf("== %d waves per SIMD\n", threads / 256);
    float r[19] = {run<0>(d, it, threads), run<1>(d, it, threads), run<2>(d, it, threads), run<3>(d, it, threads), run<4>(d, it, threads),
                   run<5>(d, it, threads), run<6>(d, it, threads), run<7>(d, it, threads), run<8>(d, it, threads), run<9>(d, it, threads),
                   run<10>(d, it, threads), run<11>(d, it, threads), run<12>(d, it, threads), run<13>(d, it, threads), run<14>(d, it, threads),
                   run<15>(d, it, threads), run<16>(d, it, threads), run<17>(d, it, threads), run<18>(d, it, threads)};
    for (int i = 0; i < 19; ++i) printf("%-24s %7.2f ns per group  (%.1f cycles @2.4GHz)\n", names[i], r[i], r[i] * 2.4f);
  }
  return 0;
}
