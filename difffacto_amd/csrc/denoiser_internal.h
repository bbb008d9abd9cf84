// Internal layout of the packed denoiser (shared by denoiser_setup.hip and denoiser_kernel.hip).
//
// Orientation.  The network is evaluated TRANSPOSED: channels run along the MFMA M dimension
// (accumulator registers), points along the N dimension (lanes).  One wavefront owns 32 points;
// its residual stream h^T (128 channels x 32 points, fp32) lives in 4 accumulator tiles of the
// 32x32 MFMA C/D layout:
//     lane l = (hf = l>>5, j = l&31)  holds point j;   tile c, register r  <->  channel
//     ch(c, r, hf) = 32c + rho(r, hf),    rho(r, hf) = (r&3) + 8(r>>2) + 4hf.
// Weights are always the A operand (32 rows x K), activations the B operand (K x 32 points), so the
// output of one GEMM is *already* laid out as the B operand of the next one up to a fixed
// permutation of the K index, which is folded into the weight packing ("kperm" below).  No
// activation ever goes through LDS or HBM between GEMMs.
//
// A-operand storage: a 32(rows) x 32(k) weight tile is stored as UNITS x 64 lanes x 16 bytes, so a
// wavefront fetches one unit with a single fully coalesced 1 KiB load (16 B per lane):
//     bf16 (v_mfma_f32_32x32x16_bf16): 2 units (q), lane (i, hf), element e<8:
//           W[row0 + i][k0 + (e&3) + 16q + 8(e>>2) + 4hf]
//     f32  (v_mfma_f32_32x32x2_f32)  : 4 units (r4), element e<4  (register r = 4 r4 + e):
//           W[row0 + i][k0 + e + 8 r4 + 4hf]
//
// "cvec" order: a 128-vector indexed by channel stored as [hf][c][r] so that a lane reads its 64
// values contiguously (offset hf*64).
#pragma once
#include "dfx_common.h"

namespace dfx {

constexpr int INNER = 128;    // n_heads * d_head
constexpr int HEADS = 8;
constexpr int DHEAD = 16;
constexpr int NCLS = 4;       // part tokens
constexpr int ZDIM = 256;
constexpr int CTX_STATIC = ZDIM + 6 + NCLS;  // 266: [part_code | mean | var | eye]
constexpr int CTX_DIM = CTX_STATIC + 256;    // 522
constexpr int IN_CH = 13;
constexpr int FF_HID = 512;
constexpr int FF_CHUNKS = FF_HID / 32;       // 16 hidden chunks of 32 units
constexpr int TEMB = 256;

__host__ __device__ constexpr int tile_units(int prec) { return prec == DFX_PREC_BF16 ? 2 : 4; }
// bytes of one 32x32 weight tile
__host__ __device__ constexpr int tile_bytes(int prec) { return tile_units(prec) * 64 * 16; }

__host__ __device__ inline int rho(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
__host__ __device__ inline int cvec_index(int ch) {  // channel -> position in cvec order
  const int c = ch >> 5, w = ch & 31;
  const int hf = (w >> 2) & 1, r = (w & 3) + 4 * (w >> 3);
  return hf * 64 + c * 16 + r;
}

// Per-block packed weights (device pointers into one allocation).
struct BlockPack {
  const uint4 *w1;      // [FF_CHUNKS][2 (a,g)][4 (c)] tiles, LN3 affine folded in
  const float *b1;      // [FF_CHUNKS][2][hf][16]  (b1 + W1 beta3), C-layout per chunk
  const uint4 *w2;      // [FF_CHUNKS][4 (ct)] tiles
  const float *b2;      // cvec
  const float *ct;      // [T][128] cvec: to_out(W_v[:,266:] t_embed(t)) + to_out.bias
};

struct DenoiserDev {
  int depth, T, prec;
  BlockPack blk[DFX_MAX_DEPTH];
  const float4 *win_x;   // cvec order, {W_in[ch][0], W_in[ch][1], W_in[ch][2], 0}
  const float2 *pre_gb;  // cvec order {gamma, beta} of pre_norm
  const float4 *wout;    // cvec order {W_out[0][ch] g, W_out[1][ch] g, W_out[2][ch] g, 0}, g = post_norm gamma
  float bout[4];         // proj_out bias + W_out beta_post
  const float *tab;      // [T][8]: sra, srm1, c1, c2, c3, sqrt(post_var), 0, 0
};

// Per-batch shape context (regions inside the caller's buffer).
struct ShapeCtxView {
  float *part;    // [B][32]: mean[3][4], var[3][4], valid[4], pad
  float *cpart;   // [B][4][128] cvec: proj_in of [anchors|variances|onehot] + bias, per part
  float *sbias;   // [B][depth][32]  C-layout [hf][16]: beta2 . A_s rows
  uint4 *as_ms;   // [B][depth][8 tiles]: 4 A_s tiles (k-tile c) then 4 M_s tiles (row-tile ct)
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

inline size_t shape_ctx_view(ShapeCtxView *v, void *base, int B, int depth, int prec) {
  size_t off = 0;
  char *p = static_cast<char *>(base);
  if (v) v->part = reinterpret_cast<float *>(p + off);
  off += align256(sizeof(float) * 32 * (size_t)B);
  if (v) v->cpart = reinterpret_cast<float *>(p + off);
  off += align256(sizeof(float) * 4 * 128 * (size_t)B);
  if (v) v->sbias = reinterpret_cast<float *>(p + off);
  off += align256(sizeof(float) * 32 * (size_t)B * depth);
  if (v) v->as_ms = reinterpret_cast<uint4 *>(p + off);
  off += align256((size_t)tile_bytes(prec) * 8 * (size_t)B * depth);
  return off;
}

}  // namespace dfx

// The opaque handle of the C-ABI.
struct dfx_denoiser {
  dfx::DenoiserDev dev;     // what the kernels receive (by value)
  void *pool = nullptr;     // one device allocation holding every packed array
  size_t pool_bytes = 0;
  // raw fp32 copies needed again at shape-prepare time (owned, inside pool)
  const float *wq[DFX_MAX_DEPTH], *wk[DFX_MAX_DEPTH], *wv[DFX_MAX_DEPTH], *wo[DFX_MAX_DEPTH];
  const float *g2[DFX_MAX_DEPTH], *be2[DFX_MAX_DEPTH];
  const float *win, *bin;   // proj_in weight (128,13), bias
  const float *const *wptrs_dev = nullptr;  // device array [depth][6] = {wq, wk, wv, wo, g2, be2}
  float *host_tables = nullptr;             // [8][T] fp32, order of dfx_denoiser_get_tables
};
