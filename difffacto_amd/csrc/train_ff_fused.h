// Fused GEGLU feed-forward of a transformer block for the TRAINING path (train_kernels.hip, bf16 matrix products, dropout off):
//
//   forward    h2 = h1 + W2 (a * gelu(g)) + b2,   [a | g] = W1 xn3 + b1                      (attention.py:50-57, 77-94)
//   backward   given dh2:  hid = a * gelu(g),  d[a | g],  dxn3 = W1^T d[a | g]               (a, g recomputed from xn3)
//
// Same structure as the sampling kernel: one wavefront = 32 points, channels on the MFMA M axis, points on the lanes, the
// 1024-wide [a | g] and the 512-wide hidden activation live in accumulator registers one 32-unit chunk at a time and never
// touch HBM in the forward.  The layer-by-layer path wrote [a | g] (2 KB / point), hid (1 KB), d hid (2 KB) and d[a | g]
// (2 KB) per block and read each of them back once or twice: ~22 KB per point and block; this path moves 1.3 KB (forward)
// + 4.3 KB (backward: hid and d[a | g] are still written once, as bf16, for the weight-gradient products) = 4x less.
//
// Weights: per optimiser step the block's W1 / W2 are re-packed (k_ff_pack) as bf16 MFMA A-fragments, 24 tiles of 32 x 32 per
// hidden chunk (48 KiB): W1a, W1g (K = channels, natural order: the B operand comes from memory), W2 (K = hidden units in the
// accumulator-register order of the GELU output), W2^T (rows = hidden units, K = channels) and W1a^T, W1g^T (rows = channels,
// K = hidden units in register order).  A workgroup (8 wavefronts, 256 points) streams the chunks L2 -> LDS with LDS-DMA through
// three buffers, two chunks ahead of the compute, counted s_waitcnt vmcnt and one barrier per chunk.
//
// GELU: g * sigmoid(g (c1 + c3 g^2)) in fp32 with the hardware exp2 / rcp (max abs error 2.7e-4 against the erf form, the same
// form as the direct sampling kernel) and its exact derivative  s + g s (1 - s) (c1 + 3 c3 g^2).
#pragma once
#include "dfx_common.h"

namespace dfx {
namespace ffused {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));

constexpr int C = 128, FH = 512, NCHUNK = FH / 32;
constexpr int TILES = 24;                         // tiles per chunk in the pack
constexpr int TILE_U4 = 128;                      // uint4 per tile (2 units x 64 lanes)
constexpr int CHUNK_U4 = TILES * TILE_U4;         // 3072 uint4 = 48 KiB
constexpr int FWD_TILES = 12, BWD_TILES = 20;     // forward: tiles 0..11; backward: tiles 0..7 and 12..23
enum { T_W1A = 0, T_W1G = 4, T_W2 = 8, T_W2T = 12, T_W1AT = 16, T_W1GT = 20 };

__host__ __device__ inline int rho(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
// K index (0..31) held by unit u, element e of a lane in half hf
__host__ __device__ inline int k_nat(int u, int hf, int e) { return 16 * u + 8 * hf + e; }            // operand loaded from memory
__host__ __device__ inline int k_reg(int u, int hf, int e) { return rho(8 * u + e, hf); }              // operand built from C/D registers

struct PackArgs {
  const float *w1;   // (1024, 128): rows 0..511 = a, 512..1023 = g
  const float *b1;   // (1024)
  const float *w2;   // (128, 512)
  const float *b2;   // (128)
  uint4 *frags;      // [NCHUNK][TILES][2][64]
  float *b1p;        // [NCHUNK][2 parts][2 hf][16]
  float *b2p;        // [2 hf][4 c][16]
};

// one thread per (chunk, tile, unit, lane)
__global__ void k_ff_pack(PackArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < NCHUNK * 2 * 2 * 16) {   // b1p
    const int r = idx & 15, hf = (idx >> 4) & 1, p = (idx >> 5) & 1, j = idx >> 6;
    a.b1p[idx] = a.b1[p * FH + 32 * j + rho(r, hf)];
  }
  if (idx < 2 * 4 * 16) {
    const int r = idx & 15, c = (idx >> 4) & 3, hf = idx >> 6;
    a.b2p[idx] = a.b2[32 * c + rho(r, hf)];
  }
  if (idx >= NCHUNK * TILES * 2 * 64) return;
  const int lane = idx & 63, u = (idx >> 6) & 1, t = (idx >> 7) % TILES, j = idx / (TILES * 128);
  const int i = lane & 31, hf = lane >> 5;
  __bf16 v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float x;
    if (t < T_W2) {                      // W1a / W1g, k-tile c, natural K
      const int p = t >> 2, c = t & 3;
      x = a.w1[(size_t)(p * FH + 32 * j + i) * C + 32 * c + k_nat(u, hf, e)];
    } else if (t < T_W2T) {              // W2 row tile ct, K = hidden units in register order
      const int ct = t - T_W2;
      x = a.w2[(size_t)(32 * ct + i) * FH + 32 * j + k_reg(u, hf, e)];
    } else if (t < T_W1AT) {             // W2^T: rows = hidden units, k-tile c over the channels, natural K
      const int c = t - T_W2T;
      x = a.w2[(size_t)(32 * c + k_nat(u, hf, e)) * FH + 32 * j + i];
    } else {                             // W1a^T / W1g^T: rows = channels 32 ct + i, K = hidden units in register order
      const int p = (t - T_W1AT) >> 2, ct = (t - T_W1AT) & 3;
      x = a.w1[(size_t)(p * FH + 32 * j + k_reg(u, hf, e)) * C + 32 * ct + i];
    }
    v[e] = (__bf16)x;
  }
  a.frags[idx] = *reinterpret_cast<const uint4 *>(v);
}

struct FfArgs {
  const uint4 *frags;
  const float *b1p, *b2p;
  const __bf16 *xn3;     // (R, 128) bf16
  const float *h1;       // forward: (R, 128) residual input
  float *h2;             // forward: (R, 128) output (may alias h1)
  const float *dh;       // backward: (R, 128) gradient at the block output
  __bf16 *hid;           // backward: (R, 512) bf16
  __bf16 *dag;           // backward: (R, 1024) bf16, [da | dg]
  float *dxn;            // backward: (R, 128) fp32
  long long R;           // multiple of 32
};

__device__ __forceinline__ void dma1k(const void *gbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ v8bf as_bf(const uint4 &u) { return __builtin_bit_cast(v8bf, u); }
__device__ __forceinline__ v16f mfma(const uint4 &a, const uint4 &b, v16f c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a), as_bf(b), c, 0, 0, 0);
}
__device__ __forceinline__ uint4 pack8(const v16f &x, int u) {   // registers 8u .. 8u+7 -> 8 bf16
  v8f t = u == 0 ? __builtin_shufflevector(x, x, 0, 1, 2, 3, 4, 5, 6, 7) : __builtin_shufflevector(x, x, 8, 9, 10, 11, 12, 13, 14, 15);
  return __builtin_bit_cast(uint4, __builtin_convertvector(t, v8bf));
}
__device__ __forceinline__ void load16(v16f &v, const float *src) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const v4f t = *reinterpret_cast<const v4f *>(src + 4 * q);
    v[4 * q + 0] = t[0], v[4 * q + 1] = t[1], v[4 * q + 2] = t[2], v[4 * q + 3] = t[3];
  }
}
// x * sigmoid(k(x)), k(x) = x (c1 + c3 x^2); returns gelu and d gelu / dx
__device__ __forceinline__ void gelu_fd(float x, float &f, float &d) {
  const float x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(-0.100125614f, x2, -2.30876530f));   // exp(-k)
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  f = x * s;
  d = fmaf(f * (1.0f - s), fmaf(0.20820537f, x2, 1.60031416f), s);   // s + x s (1 - s) (c1 + 3 c3 x^2)
}
__device__ __forceinline__ float gelu_f(float x) {
  const float e = __builtin_amdgcn_exp2f(x * fmaf(-0.100125614f, x * x, -2.30876530f));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

constexpr int B1P_FLOATS = NCHUNK * 64, B2P_FLOATS = 128;
constexpr int TILE_ROW = 80, TILE_BYTES = 32 * TILE_ROW;   // per-wave transposition tile of the backward's stores
constexpr int NW = 8;     // wavefronts per workgroup (256 points, one workgroup per CU)
constexpr int NBUF = 3;   // LDS chunk buffers: the stream runs two chunks ahead of the compute
// (measured, B = 128 x 2048, backward / forward per block: 8 waves x 3 buffers 441 / 161 us; 4 waves x 2 buffers, two workgroups
// per CU, 513 / 161 us; 8 x 2: 662 / 182 us)

// stage `ntile_groups` runs of tiles of chunk j into LDS buffer `buf` (each wave copies every NW-th KiB)
template <bool BWD>
__device__ __forceinline__ void stage_chunk(const uint4 *frags, int j, unsigned lds_buf, int wave, unsigned voff) {
  const char *src = reinterpret_cast<const char *>(frags + (size_t)j * CHUNK_U4);
  constexpr int NK = (BWD ? BWD_TILES : FWD_TILES) * 2;   // KiB to copy
#pragma unroll
  for (int k = 0; k < NK / NW; ++k) {
    const int piece = k * NW + wave;                                     // destination KiB
    const int spiece = BWD && piece >= 16 ? piece + 8 : piece;           // backward skips tiles 8..11 (W2)
    dma1k(src + spiece * 1024, voff, lds_buf + piece * 1024);
  }
}

// LDS tile index of pack tile t
template <bool BWD>
__device__ __forceinline__ constexpr int lt(int t) { return BWD && t >= 12 ? t - 4 : t; }

template <bool BWD>
__global__ __launch_bounds__(NW * 64, 2) void k_ff(FfArgs a) {
  constexpr int BUF_BYTES = (BWD ? BWD_TILES : FWD_TILES) * 2048;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ff_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ff_smem);
  const unsigned voff = lane * 16;
  long long row = ((long long)blockIdx.x * NW + wave) * 32;
  const bool live = row < a.R;
  if (!live) row = a.R - 32;   // a workgroup's trailing wavefronts past the end recompute the last tile and store nothing
  row += pj;

  constexpr int PIECES = (BWD ? BWD_TILES : FWD_TILES) * 2 / NW;   // LDS-DMA instructions per wave and chunk
  stage_chunk<BWD>(a.frags, 0, lds0, wave, voff);
  if (NBUF > 2) stage_chunk<BWD>(a.frags, 1, lds0 + BUF_BYTES, wave, voff);

  // B operand of the products over the channels: xn3 (bf16, natural K order) and, backward, dh rounded to bf16
  uint4 xn[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) xn[c][u] = *reinterpret_cast<const uint4 *>(a.xn3 + row * C + 32 * c + k_nat(u, hf, 0));
  uint4 dhb[4][2];
  v16f acc[4];   // forward: the residual stream h; backward: dxn3
  if (BWD) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float *p = a.dh + row * C + 32 * c + k_nat(u, hf, 0);
        const v4f lo = *reinterpret_cast<const v4f *>(p), hi = *reinterpret_cast<const v4f *>(p + 4);
        const v8f t = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        dhb[c][u] = __builtin_bit_cast(uint4, __builtin_convertvector(t, v8bf));
      }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v16f b2;
      load16(b2, a.b2p + hf * 64 + c * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f t = *reinterpret_cast<const v4f *>(a.h1 + row * C + 32 * c + 8 * q + 4 * hf);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[c][4 * q + m] = t[m] + b2[4 * q + m];
      }
    }
  }
  // b1 of all chunks -> LDS: inside the loop every operand must come from LDS — vector-memory loads complete in order, so a
  // global load issued behind a chunk's LDS-DMA pieces could only be consumed after those pieces had landed as well, and the
  // stream would never run ahead
  // b1 of all chunks -> LDS: inside the loop every operand must come from LDS — vector-memory loads complete in order, so a
  // global load issued behind a chunk's LDS-DMA pieces could only be consumed after those pieces had landed as well, and the
  // stream would never run ahead
  float *b1s = reinterpret_cast<float *>(ff_smem + NBUF * BUF_BYTES);
  for (int i = threadIdx.x; i < B1P_FLOATS; i += NW * 64) b1s[i] = a.b1p[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first NBUF - 1 chunks have landed (one-off)
  __syncthreads();

  // backward: hid / d[a | g] leave through a per-wave LDS tile (32 points x 32 units bf16, rows padded to 80 B) that turns the
  // accumulator layout (a lane holds four scattered 8-byte pieces of its point's row) into row-major 16-byte pieces: four
  // consecutive lanes then write one point's 64 contiguous bytes and a store instruction covers 16 whole row segments.  Written
  // straight from the accumulator layout — twelve 8-byte stores per chunk, 16 contiguous bytes per point and instruction — these
  // stores were HALF of the kernel's time (timing ablation without them: 434 -> 225 us per block).
  unsigned char *tile = ff_smem + NBUF * BUF_BYTES + B1P_FLOATS * 4 + wave * TILE_BYTES;
  auto store_tile = [&](const v16f &x, __bf16 *dst, int ld, int col0) {   // dst[(row) * ld + col0 + unit]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f t = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
      *reinterpret_cast<v4bf *>(tile + pj * TILE_ROW + (8 * q + 4 * hf) * 2) = __builtin_convertvector(t, v4bf);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = (lane >> 2) + 16 * k, sg = lane & 3;
      const uint4 v = *reinterpret_cast<const uint4 *>(tile + r * TILE_ROW + sg * 16);
      if (live) *reinterpret_cast<uint4 *>(dst + (row - pj + r) * ld + col0 + sg * 8) = v;
    }
  };
#pragma unroll 1
  for (int j = 0; j < NCHUNK; ++j) {
    if (j + NBUF - 1 < NCHUNK) stage_chunk<BWD>(a.frags, j + NBUF - 1, lds0 + ((j + NBUF - 1) % NBUF) * BUF_BYTES, wave, voff);   // slot of chunk j - 1: every wave is past it
    const uint4 *fr = reinterpret_cast<const uint4 *>(ff_smem + (j % NBUF) * BUF_BYTES) + lane;
    auto frag = [&](int t, int u) -> uint4 { return fr[(lt<BWD>(t) * 2 + u) * 64]; };
    // ---- [a | g] = b1 + W1 xn3 ----
    v16f av, gv;
    load16(av, b1s + ((j * 2 + 0) * 2 + hf) * 16);
    load16(gv, b1s + ((j * 2 + 1) * 2 + hf) * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        av = mfma(frag(T_W1A + c, u), xn[c][u], av);
        gv = mfma(frag(T_W1G + c, u), xn[c][u], gv);
      }
    if (!BWD) {
      v16f hv;
#pragma unroll
      for (int r = 0; r < 16; ++r) hv[r] = av[r] * gelu_f(gv[r]);
      const uint4 h0 = pack8(hv, 0), h1 = pack8(hv, 1);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        acc[ct] = mfma(frag(T_W2 + ct, 0), h0, acc[ct]);
        acc[ct] = mfma(frag(T_W2 + ct, 1), h1, acc[ct]);
      }
    } else {
      // ---- d hid = W2^T dh ----
      v16f dhid;
#pragma unroll
      for (int r = 0; r < 16; ++r) dhid[r] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) dhid = mfma(frag(T_W2T + c, u), dhb[c][u], dhid);
      // ---- GEGLU backward on the registers; hid and d[a | g] go out as bf16 for the weight-gradient products ----
      v16f hv, da, dg;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float f, d;
        gelu_fd(gv[r], f, d);
        hv[r] = av[r] * f;
        da[r] = dhid[r] * f;
        dg[r] = dhid[r] * av[r] * d;
      }
      store_tile(hv, a.hid, FH, 32 * j);
      store_tile(da, a.dag, 2 * FH, 32 * j);
      store_tile(dg, a.dag, 2 * FH, FH + 32 * j);
      // ---- dxn3 += W1a^T da + W1g^T dg ----
      const uint4 a0 = pack8(da, 0), a1 = pack8(da, 1), g0 = pack8(dg, 0), g1 = pack8(dg, 1);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        acc[ct] = mfma(frag(T_W1AT + ct, 0), a0, acc[ct]);
        acc[ct] = mfma(frag(T_W1AT + ct, 1), a1, acc[ct]);
        acc[ct] = mfma(frag(T_W1GT + ct, 0), g0, acc[ct]);
        acc[ct] = mfma(frag(T_W1GT + ct, 1), g1, acc[ct]);
      }
    }
    // chunk j + 1 must have landed: loads complete in order, so "at most PIECES outstanding" leaves only chunk j + 2's pieces
    // (whatever the order between loads and the backward's stores); no new pieces in the last two iterations -> drain
    if (NBUF > 2 && j + 2 < NCHUNK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (!live) return;
  float *out = (BWD ? a.dxn : a.h2) + row * C;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<v4f *>(out + 32 * c + 8 * q + 4 * hf) = v4f{acc[c][4 * q], acc[c][4 * q + 1], acc[c][4 * q + 2], acc[c][4 * q + 3]};
}

inline size_t pack_bytes_frags() { return (size_t)NCHUNK * CHUNK_U4 * sizeof(uint4); }

inline void launch_pack(hipStream_t st, const PackArgs &a) {
  const int total = NCHUNK * TILES * 128;
  k_ff_pack<<<(total + 255) / 256, 256, 0, st>>>(a);
}
template <bool BWD>
inline int launch_ff(hipStream_t st, const FfArgs &a) {
  constexpr int LDS = NBUF * (BWD ? BWD_TILES : FWD_TILES) * 2048 + B1P_FLOATS * 4 + (BWD ? NW * TILE_BYTES : 0);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ff<BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return -1;
    attr_set = true;
  }
  const long long groups = (a.R / 32 + NW - 1) / NW;
  k_ff<BWD><<<(int)groups, NW * 64, LDS, st>>>(a);
  return 0;
}

}  // namespace ffused
}  // namespace dfx
