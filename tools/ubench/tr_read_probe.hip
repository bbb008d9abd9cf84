// Probe of ds_read_b64_tr_b16's lane mapping (gfx950): which (source lane, element) does output (lane, element) receive?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
// Every lane supplies the address of ITS OWN 8 bytes (a scrambled, lane-unique slot), whose four 16-bit elements hold lane * 4 + e: the values that
// come back name their source.  Printed: the mapping, and whether it equals  out[lane = 16 G + q][j] = in[lane 16 G + 4 j + (q >> 2)][q & 3].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
extern __shared__ char smem[];
__global__ void k(unsigned short *out) {
  const int l = threadIdx.x, slot = (l * 37 + 11) & 63;   // a permutation of the 64 slots
  unsigned short *mine = reinterpret_cast<unsigned short *>(smem + slot * 8);
  for (int e = 0; e < 4; ++e) mine[e] = (unsigned short)(l * 4 + e);
  __syncthreads();
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)(smem + slot * 8));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main() {
  unsigned short *d, h[256];
  hipMalloc(&d, sizeof h);
  k<<<1, 64, 1024>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      const int s = h[l * 4 + j] >> 2, e = h[l * 4 + j] & 3, G = l >> 4, q = l & 15;
      printf("  [%d] <- lane %2d elem %d", j, s, e);
      bad += !(s == 16 * G + 4 * j + (q >> 2) && e == (q & 3));
    }
    printf("\n");
  }
  printf("hypothesis out[16 G + q][j] = in[16 G + 4 j + (q >> 2)][q & 3]: %s (%d mismatches)\n", bad ? "WRONG" : "holds", bad);
  return 0;
}
