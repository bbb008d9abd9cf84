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

// Streaming records.  Everything the hot kernel reads per stage is laid out as whole 1 KiB pieces
// so that a stage is copied HBM/L2 -> LDS by a fixed number of 1 KiB LDS-DMA instructions.
//   FF stage record j = 0..16 (per block): 12 weight tiles, skewed by FF_SKEW = 1 chunk so that ONE record feeds
//   one uninterrupted MFMA burst (GEMM2 of hidden chunk j-1 followed by GEMM1 of chunk j):
//        tiles 0..7  = W1 rows {a: 32j.., g: 512+32j..} x k-tiles c  (index part*4 + c), norm3 affine folded in
//                      (zero for j = 16)
//        tiles 8..11 = W2 rows 32ct.. x hidden k-tile (j-1)          (index 8 + ct; zero for j = 0)
//   block-constant record (per block): b1' [u][part][hf][16] fp32 (4 KiB) | b2 cvec (512 B) | pad (512 B)
//   c_t rows (per block): [T] x 1 KiB, first 512 B = cvec of to_out(W_v[:,266:] t_embed(t)) + to_out.bias
//   attention record (per shape, per block): 4 A_s tiles (k-tile c) | 4 M_s tiles (row-tile ct) |
//        1 KiB: sbias [hf][16] fp32 (beta2 . A_s rows) + pad
constexpr int CHUNK_TILES = 12;
constexpr int FF_SKEW = 1;
constexpr int FF_STAGES = FF_CHUNKS + FF_SKEW;
__host__ __device__ constexpr int chunk_bytes(int prec) { return CHUNK_TILES * tile_bytes(prec); }
constexpr int BCONST_BYTES = 5 * 1024;
constexpr int BCONST_B2_OFF = 1024;   // float offset of b2 inside the block-constant record
constexpr int CT_ROW = 256;           // floats per c_t row (1 KiB)
__host__ __device__ constexpr int asms_bytes(int prec) { return 8 * tile_bytes(prec) + 1024; }

struct BlockPack {
  const uint4 *chunks;   // [FF_STAGES] stage records
  const float *bconst;   // block-constant record
  const float *ct;       // [T][CT_ROW]
};

// bf16 path, feed-forward: the GELU runs in packed fp16 (two hidden values per VALU instruction: v_pk_*_f16 does not
// contend with the matrix pipe, unlike v_pk_*_f32) as a transcendental-free polynomial gelu(g) = g Phi(g),
// Phi(g) ~ 1/2 + u R(u^2 - m) with u = g / 2 (denoiser_kernel.hip), and its output is the fp16 B operand of GEMM2
// (v_mfma_f32_32x32x16_f16), so W2 is packed as fp16.  To keep a * g inside the fp16 range the `a` half of W1 / b1 is
// pre-scaled by FF_A_SCALE, the `g` half by FF_G_SCALE (the polynomial works on u = g / 2), and W2 by the inverse of
// their product (exact powers of two).
constexpr float FF_A_SCALE = 0.0625f, FF_G_SCALE = 0.5f;

struct DenoiserDev {
  int depth, T, prec;
  int w1_fold;           // bf16: b1' rides in channel 127's K slot of W1 (denoiser_setup.hip: k_pack_w1); 0 = plain W1' + accumulator initialisers
  BlockPack blk[DFX_MAX_DEPTH];
  long long blk_stride;  // bytes from block b's chunks / bconst / ct to block b+1's (one carve per block: uniform)
  const float4 *win_x;   // cvec order, {W_in[ch][0], W_in[ch][1], W_in[ch][2], 0}
  const float2 *pre_gb;  // cvec order {gamma, beta} of pre_norm
  const float4 *wout;    // cvec order {W_out[0][ch] g, W_out[1][ch] g, W_out[2][ch] g, 0}, g = post_norm gamma
  float bout[4];         // proj_out bias + W_out beta_post
  const float *tab;      // [T][8]: sra, srm1, c1, c2, c3, posterior_variance, alphas_cumprod_prev, 0
  const float *qtab;     // [T][2]: sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod (q_sample)
};

// Per-batch shape context (regions inside the caller's buffer).
struct ShapeCtxView {
  float *part;    // [B][32]: mean[3][4], var[3][4], valid[4], pad
  float *cpart;   // [B][4][128] cvec: proj_in of [anchors|variances|onehot] + bias, per part
  uint4 *as_ms;   // [B][depth] attention records (asms_bytes each)
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

inline size_t shape_ctx_view(ShapeCtxView *v, void *base, int B, int depth, int prec) {
  size_t off = 0;
  char *p = static_cast<char *>(base);
  if (v) v->part = reinterpret_cast<float *>(p + off);
  off += align256(sizeof(float) * 32 * (size_t)B);
  if (v) v->cpart = reinterpret_cast<float *>(p + off);
  off += align256(sizeof(float) * 4 * 128 * (size_t)B);
  if (v) v->as_ms = reinterpret_cast<uint4 *>(p + off);
  off += align256((size_t)asms_bytes(prec) * (size_t)B * depth);
  return off;
}

}  // namespace dfx

// The opaque handle of the C-ABI.
struct dfx_denoiser {
  dfx::DenoiserDev dev;     // what the kernels receive (by value)
  void *pool = nullptr;     // one device allocation holding every packed array
  size_t pool_bytes = 0;
  // raw fp32 copies needed again at shape-prepare time (owned, inside pool)
  const float *wq[DFX_MAX_DEPTH], *wk[DFX_MAX_DEPTH], *wv[DFX_MAX_DEPTH], *wo[DFX_MAX_DEPTH];   // wk / wv: static columns, transposed [266][128]
  const float *g2[DFX_MAX_DEPTH], *be2[DFX_MAX_DEPTH];
  const float *win, *bin;   // proj_in weight (128,13), bias
  const float *const *wptrs_dev = nullptr;  // device array [depth][7] = {wq, wk^T (static columns), wv^T, wo^T, g2, be2, Wq be2}
  float *host_tables = nullptr;             // [8][T] fp32, order of dfx_denoiser_get_tables
  double *host_ac_pv = nullptr;             // [2][T] float64: alphas_cumprod, posterior_variance (DDIM coefficients per call)
  float w1_fold_ratio = 0.f;                // bf16: max_r |W1'[r][k]| / mean_c |W1'[r][c]| over the blocks for the folded channel k (127's when not folded)
  int w1_fold_channel = -1;                 // bf16: the hidden channel whose K slot carries b1' (exchanged with 127 at pack time when != 127); -1 = no fold
};
