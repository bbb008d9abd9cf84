// libdfx's dropout contract for the training path (nn.Dropout in train mode: attention.py:84 behind the GEGLU, :177 behind to_out).
//
//   factor(seed, site, element i) = draw(seed, site, i) < thr ? 0 : 1 / (1 - p),      thr = round(p 2^16), capped at 2^16 - 1
//   draw = 16 bits of Philox4x32-7 keyed by the step's 64-bit seed, counter = (i >> 3 = group of EIGHT consecutive elements of the site's
//          row-major tensor, site, 0xD20F0): word e >> 1 of the output, half e & 1 (low half first) for element e = i & 7 of the group
//
// Round 5: eight elements per call (16-bit draws; the probability resolution 2^-16 biases E[factor] by < 8e-6) and seven rounds (Philox4x32-7 is
// the smallest variant Salmon et al. report as passing BigCrush; -10 is their safety-margin default) instead of four 32-bit draws of Philox4x32-10:
// 2.9x fewer rounds per element.  On gfx950 a round is two quarter-rate 32 x 32 -> 64 multiplies plus two xor3: the factors of a 32-point tile
// of one block (20 480 elements) cost a wavefront 40 calls.  ONE definition for the layer-by-layer kernels, the fused kernels and
// dfx_debug_dropout_factors (what the tests replay into the torch oracle).  torch's own CUDA dropout stream depends on its launch geometry and
// cannot be reproduced; the contract here is (seed, site, element index).
#pragma once
#include <hip/hip_runtime.h>

namespace dfx {

struct DropKey {
  unsigned k0, k1;   // the step's seed
  unsigned thr;      // drop when the 16-bit draw is below
  float keep;        // 1 / (1 - p)
};
__host__ __device__ inline DropKey drop_key(unsigned long long seed, float p) {
  unsigned thr = (unsigned)((double)p * 65536.0 + 0.5);
  if (thr > 65535u) thr = 65535u;
  return DropKey{(unsigned)seed, (unsigned)(seed >> 32), thr, 1.0f / (1.0f - p)};
}

__device__ __forceinline__ uint4 philox4x32_7(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;   // (v_mad_u64_u32: lo and hi in one)
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    c0 = hi1 ^ c1 ^ k0, c1 = lo1, c2 = hi0 ^ c3 ^ k1, c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
// the four words of group g8 (eight consecutive elements) of `site`
__device__ __forceinline__ uint4 drop_group(const DropKey &k, unsigned site, unsigned long long g8) {
  return philox4x32_7((unsigned)g8, (unsigned)(g8 >> 32), site, 0xD20F0u, k.k0, k.k1);
}
// keep-flags of the two elements a word holds: low half = element 2 w, high half = element 2 w + 1
__device__ __forceinline__ bool drop_keep_lo(unsigned word, unsigned thr) { return (word & 0xffffu) >= thr; }
__device__ __forceinline__ bool drop_keep_hi(unsigned word, unsigned thr) { return word >= (thr << 16); }

}  // namespace dfx
