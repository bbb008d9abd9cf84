// Micro-benchmark: does a wavefront's ds_read_b128 stream (one 1 KiB A fragment per MFMA, refilled in place, as in the M
// slots of k_denoise_pipe) slow its MFMA stream down?  Variants: MFMA only / reads only / 1:1 interleaved / 1 read per 2.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_lds_mix.hip -o _build/mfma_lds_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define M(P, ACC) "v_mfma_f32_32x32x16_bf16 %[" ACC "], %[" P "], %[b], %[" ACC "]\n"
#define R(P, OFF) "ds_read_b128 %[" P "], %[addr] offset:" #OFF "\n"
#define W7 "s_waitcnt lgkmcnt(7)\n"
#define W3 "s_waitcnt lgkmcnt(3)\n"

template <int MODE>
__global__ void __launch_bounds__(512) k(float *out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  for (int i = threadIdx.x; i < 32 * 1024 / 4; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 0.f;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + lane * 16;
  v4f p0 = {0, 0, 0, 0}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0, b = p0;
  v16f a0 = {0}, a1 = a0, a2 = a0, a3 = a0;
  for (int it = 0; it < iters; ++it) {
#define OPS : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4), [p5] "+v"(p5), [p6] "+v"(p6), [p7] "+v"(p7), \
              [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3) : [b] "v"(b), [addr] "v"(addr)
    if (MODE == 0) {   // 24 MFMAs
      asm volatile(M("p0","a0") M("p1","a1") M("p2","a2") M("p3","a3") M("p4","a0") M("p5","a1") M("p6","a2") M("p7","a3")
                   M("p0","a0") M("p1","a1") M("p2","a2") M("p3","a3") M("p4","a0") M("p5","a1") M("p6","a2") M("p7","a3")
                   M("p0","a0") M("p1","a1") M("p2","a2") M("p3","a3") M("p4","a0") M("p5","a1") M("p6","a2") M("p7","a3") OPS);
    } else if (MODE == 1) {   // 24 reads
      asm volatile(W7 R("p0",0) W7 R("p1",1024) W7 R("p2",2048) W7 R("p3",3072) W7 R("p4",4096) W7 R("p5",5120) W7 R("p6",6144) W7 R("p7",7168)
                   W7 R("p0",8192) W7 R("p1",9216) W7 R("p2",10240) W7 R("p3",11264) W7 R("p4",12288) W7 R("p5",13312) W7 R("p6",14336) W7 R("p7",15360)
                   W7 R("p0",16384) W7 R("p1",17408) W7 R("p2",18432) W7 R("p3",19456) W7 R("p4",20480) W7 R("p5",21504) W7 R("p6",22528) W7 R("p7",23552) OPS);
    } else if (MODE == 2) {   // 1:1
      asm volatile(W7 M("p0","a0") R("p0",0) W7 M("p1","a1") R("p1",1024) W7 M("p2","a2") R("p2",2048) W7 M("p3","a3") R("p3",3072)
                   W7 M("p4","a0") R("p4",4096) W7 M("p5","a1") R("p5",5120) W7 M("p6","a2") R("p6",6144) W7 M("p7","a3") R("p7",7168)
                   W7 M("p0","a0") R("p0",8192) W7 M("p1","a1") R("p1",9216) W7 M("p2","a2") R("p2",10240) W7 M("p3","a3") R("p3",11264)
                   W7 M("p4","a0") R("p4",12288) W7 M("p5","a1") R("p5",13312) W7 M("p6","a2") R("p6",14336) W7 M("p7","a3") R("p7",15360)
                   W7 M("p0","a0") R("p0",16384) W7 M("p1","a1") R("p1",17408) W7 M("p2","a2") R("p2",18432) W7 M("p3","a3") R("p3",19456)
                   W7 M("p4","a0") R("p4",20480) W7 M("p5","a1") R("p5",21504) W7 M("p6","a2") R("p6",22528) W7 M("p7","a3") R("p7",23552) OPS);
    } else if (MODE == 3) {   // 2 MFMAs per read (each fragment used for two B tiles)
      asm volatile(W7 M("p0","a0") M("p0","a2") R("p0",0) W7 M("p1","a1") M("p1","a3") R("p1",1024) W7 M("p2","a0") M("p2","a2") R("p2",2048) W7 M("p3","a1") M("p3","a3") R("p3",3072)
                   W7 M("p4","a0") M("p4","a2") R("p4",4096) W7 M("p5","a1") M("p5","a3") R("p5",5120) W7 M("p6","a0") M("p6","a2") R("p6",6144) W7 M("p7","a1") M("p7","a3") R("p7",7168)
                   W7 M("p0","a0") M("p0","a2") R("p0",8192) W7 M("p1","a1") M("p1","a3") R("p1",9216) W7 M("p2","a0") M("p2","a2") R("p2",10240) W7 M("p3","a1") M("p3","a3") R("p3",11264) OPS);
    } else if (MODE == 4) {   // 1:1 with two ds_read_b64 instead of one b128
#define R2(P, OFF) "ds_read_b64 %[" P "], %[addr] offset:" #OFF "\n"
      asm volatile(W7 M("p0","a0") R("p0",0) W7 M("p1","a1") R("p1",1024) W7 M("p2","a2") R("p2",2048) W7 M("p3","a3") R("p3",3072)
                   W7 M("p4","a0") R("p4",4096) W7 M("p5","a1") R("p5",5120) W7 M("p6","a2") R("p6",6144) W7 M("p7","a3") R("p7",7168) OPS);
    }
  }
  v16f s = a0 + a1 + a2 + a3;
  float t = p0[0] + p1[0] + p2[0] + p3[0] + p4[0] + p5[0] + p6[0] + p7[0];
  if (s[0] + t == 12345.678f) out[0] = s[0];
}

template <int MODE>
void run(float *d, int threads, int nm, const char *name) {
  const int iters = 20000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<MODE><<<256, threads, 32 * 1024>>>(d, 100);
  (void)hipEventRecord(a);
  k<MODE><<<256, threads, 32 * 1024>>>(d, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-34s %d waves/CU: %.1f ns per iteration, %.2f ns per slot item (%d items)\n", name, threads / 64, ms * 1e6 / iters, ms * 1e6 / iters / nm, nm);
}

int main() {
  float *d; (void)hipMalloc(&d, 64);
  for (int threads : {256, 512}) {
    run<0>(d, threads, 24, "24 MFMA");
    run<1>(d, threads, 24, "24 ds_read_b128");
    run<2>(d, threads, 24, "24 x (MFMA, ds_read_b128)");
    run<3>(d, threads, 24, "12 x (MFMA, MFMA, ds_read_b128)");
    run<4>(d, threads, 8, "8 x (MFMA, ds_read_b128)");
  }
  return 0;
}
