// Micro-benchmark for k_denoise_pipe2's MFMA stream: ONE wavefront per SIMD (256 threads, 512 registers) issuing pairs of
// v_mfma_f32_32x32x16 that share their A fragment (two point tiles), the fragment refilled in place from LDS behind the pair.
// Variants: C/D in the accumulator file or in VGPRs, B from the accumulator file or VGPRs, with / without the refill, NV packed
// fp16 VALU fillers per MFMA.  Prints shader cycles per MFMA (floor: 32).
// Build: hipcc --offload-arch=gfx950 -O3 pair_issue.hip -o _build/pair_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define FILL "v_pk_fma_f16 %[t0], %[g0], %[g1], %[t0]\n v_pk_mul_f16 %[t1], %[g0], %[t1]\n v_pk_fma_f16 %[t2], %[g1], %[g0], %[t2]\n v_pk_add_f16 %[t3], %[g1], %[t3]\n v_pk_fma_f16 %[t4], %[g0], %[g1], %[t4]\n"
#define FILL3 "v_pk_fma_f16 %[t0], %[g0], %[g1], %[t0]\n v_pk_mul_f16 %[t1], %[g0], %[t1]\n v_pk_fma_f16 %[t2], %[g1], %[g0], %[t2]\n"

// ACC: 0 = C/D in VGPRs, 1 = accumulator file.  BA: B operand from the accumulator file.  LDS: refill behind every pair.
// NV: 0, 3 or 5 fillers behind every MFMA.  PAIR: 1 = the two MFMAs of a pair share A; 0 = every MFMA its own A (+ its own refill)
template <int ACC, int BA, int LDS, int NV, int PAIR>
__global__ void __launch_bounds__(256, 1) k(unsigned long long *out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  for (int i = threadIdx.x; i < 32 * 1024 / 4; i += blockDim.x) {
    unsigned hsh = (i * 2654435761u) ^ (i >> 3) * 40503u;
    reinterpret_cast<unsigned *>(smem)[i] = (hsh & 0x807f807fu) | 0x3f003f00u | ((hsh >> 9) & 0x00800080u);
  }
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + lane * 16;
  v16f c0 = {0}, c1 = c0, c2 = c0, c3 = c0;
  const v4f *fs = reinterpret_cast<const v4f *>(smem) + lane;
  v4f f0 = fs[64], f1 = fs[128], f2 = fs[192], f3 = fs[256], f4 = fs[320], f5 = fs[384], f6 = fs[448], f7 = fs[512], b0 = fs[0], b1 = fs[576];
  unsigned g0 = 0x3c003c00u ^ (threadIdx.x * 0x01230123u & 0x03ff03ffu), g1 = 0x38003800u ^ (threadIdx.x * 0x04560457u & 0x03ff03ffu), t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
  const unsigned long long tA = __builtin_readcyclecounter();
#define OPSV : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4), [f5] "+v"(f5), \
               [f6] "+v"(f6), [f7] "+v"(f7), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [t4] "+v"(t4) : [la] "v"(la), [b0] "v"(b0), [b1] "v"(b1), [g0] "v"(g0), [g1] "v"(g1)
#define OPSA : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4), [f5] "+v"(f5), \
               [f6] "+v"(f6), [f7] "+v"(f7), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [t4] "+v"(t4) : [la] "v"(la), [b0] "v"(b0), [b1] "v"(b1), [g0] "v"(g0), [g1] "v"(g1)
#define OPSVB : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4), [f5] "+v"(f5), \
               [f6] "+v"(f6), [f7] "+v"(f7), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [t4] "+v"(t4) : [la] "v"(la), [b0] "a"(b0), [b1] "a"(b1), [g0] "v"(g0), [g1] "v"(g1)
#define EMIT(str) do { if (ACC) asm volatile(str OPSA); else if (BA) asm volatile(str OPSVB); else asm volatile(str OPSV); } while (0)
#define GRP(ca, cb, fr, fr2, off)                                                                     \
    EMIT("s_waitcnt lgkmcnt(7)\n v_mfma_f32_32x32x16_bf16 %[" ca "], %[" fr "], %[b0], %[" ca "]\n");  \
    if (NV == 5) EMIT(FILL); if (NV == 3) EMIT(FILL3);                                                \
    EMIT("v_mfma_f32_32x32x16_bf16 %[" cb "], %[" fr2 "], %[b1], %[" cb "]\n");                        \
    if (LDS) EMIT("ds_read_b128 %[" fr "], %[la] offset:" #off "\n");                                 \
    if (LDS && !PAIR) EMIT("ds_read_b128 %[" fr2 "], %[la] offset:" #off "+16384\n");                  \
    if (NV == 5) EMIT(FILL); if (NV == 3) EMIT(FILL3);
  for (int it = 0; it < iters; ++it) {
    if (PAIR) {
      GRP("c0", "c1", "f0", "f0", 1024) GRP("c2", "c3", "f1", "f1", 2048) GRP("c0", "c1", "f2", "f2", 3072) GRP("c2", "c3", "f3", "f3", 4096)
      GRP("c0", "c1", "f4", "f4", 5120) GRP("c2", "c3", "f5", "f5", 6144) GRP("c0", "c1", "f6", "f6", 7168) GRP("c2", "c3", "f7", "f7", 8192)
    } else {
      GRP("c0", "c1", "f0", "f1", 1024) GRP("c2", "c3", "f2", "f3", 2048) GRP("c0", "c1", "f4", "f5", 3072) GRP("c2", "c3", "f6", "f7", 4096)
      GRP("c0", "c1", "f0", "f1", 5120) GRP("c2", "c3", "f2", "f3", 6144) GRP("c0", "c1", "f4", "f5", 7168) GRP("c2", "c3", "f6", "f7", 8192)
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long tB = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  s += f0[0] + f1[0] + f2[0] + f3[0] + f4[0] + f5[0] + f6[0] + f7[0] + __uint_as_float(t0 ^ t1 ^ t2 ^ t3 ^ t4);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = tB - tA;
  if (s == 12345.678f) out[1] = 1;
}

template <int ACC, int BA, int LDS, int NV, int PAIR>
void run(const char *name, int blocks) {
  unsigned long long *d;
  hipMalloc(&d, 16);
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<ACC, BA, LDS, NV, PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  k<ACC, BA, LDS, NV, PAIR><<<blocks, 256, 64 * 1024>>>(d, 10);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  k<ACC, BA, LDS, NV, PAIR><<<blocks, 256, 64 * 1024>>>(d, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-58s blocks %4d: %6.1f cycles / MFMA   (%.3f ms -> %.0f TFLOP/s)\n", name, blocks, (double)h[0] / (iters * 16.0), ms,
         blocks * 4.0 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(d);
}

int main() {
  for (int blocks : {1, 256}) {
    run<0, 0, 0, 0, 1>("C/D VGPR, B VGPR, no refill, no filler", blocks);
    run<1, 0, 0, 0, 1>("C/D AGPR, B VGPR, no refill, no filler", blocks);
    run<0, 1, 0, 0, 1>("C/D VGPR, B AGPR, no refill, no filler", blocks);
    run<1, 0, 1, 0, 1>("C/D AGPR, refill per pair (shared A)", blocks);
    run<1, 0, 1, 0, 0>("C/D AGPR, refill per MFMA (own A)", blocks);
    run<0, 1, 1, 0, 1>("C/D VGPR, B AGPR, refill per pair", blocks);
    run<1, 0, 1, 3, 1>("C/D AGPR, refill per pair, 3 v_pk per MFMA", blocks);
    run<1, 0, 1, 5, 1>("C/D AGPR, refill per pair, 5 v_pk per MFMA", blocks);
    run<0, 1, 1, 5, 1>("C/D VGPR, B AGPR, refill per pair, 5 v_pk per MFMA", blocks);
    run<1, 0, 1, 5, 0>("C/D AGPR, refill per MFMA, 5 v_pk per MFMA", blocks);
  }
  return 0;
}
