// Does the immediate offset of global_load_lds_dwordx4 apply to the LDS address as well as to the global address?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
__global__ void k(const unsigned *src, unsigned *out) {
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem);
  unsigned *s = (unsigned *)smem;
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = 0xdeadbeef;
  __syncthreads();
  const unsigned voff = threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(src), "s"(lds0) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = s[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, 16384); hipMalloc(&o, 16384);
  hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
  k<<<1, 64, 16384>>>(d, o);
  hipMemcpy(h.data(), o, 16384, hipMemcpyDeviceToHost);
  // find where data landed and what it is
  for (int i = 0; i < 4096; ++i) if (h[i] != 0xdeadbeef) { printf("first written LDS dword %d (byte %d) = source dword %u (byte %u)\n", i, i * 4, h[i], h[i] * 4); break; }
  int n = 0; for (int i = 0; i < 4096; ++i) n += h[i] != 0xdeadbeef;
  printf("dwords written: %d\n", n);
  return 0;
}
