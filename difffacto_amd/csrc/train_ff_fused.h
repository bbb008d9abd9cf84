// Fused GEGLU feed-forward of a transformer block for the TRAINING path (train_kernels.hip, bf16 matrix products, dropout on or off), with the
// block's LayerNorms and — optionally, FfArgs::at_frags — its attention sub-block in the same kernels:
//
//   forward    [h1 = hin + M_s softmax(A_s LN2(hin)) + b_o]   h2 = h1 + W2 (a * gelu(g)) + b2,   [a | g] = W1' xhat3(h1) + b1'   (LayerNorm3's affine
//              rides on W1' = W1 diag(gamma3), b1' = b1 + W1 beta3: PackArgs)  [last block: eps = W_out post_norm(h2) + b_out, TL_HEAD]
//   backward   given dh2:  d[a | g], d xhat3 = W1'^T d[a | g],  dh1 = dh2 + LN3'(d xhat3)   [dh_in = dh1 + LN2'(A_s^T dsim)]      (attention.py:50-57, 77-94,
//              column sums for the bias / LayerNorm2 gradients; xn2 / dh1 as bf16 fragments for k_attn_bwd_param                    179-204, 296-306)
//   k_ff_wgrad dW1, db1, dW2 (and through them d gamma3 / d beta3) by a weight-stationary recompute of hid and d[a | g]
//
// Same structure as the sampling kernel: one wavefront = 32 points, channels on the MFMA M axis, points on the lanes, the
// 1024-wide [a | g] and the 512-wide hidden activation live in accumulator registers one 32-unit chunk at a time and never
// touch HBM.  The layer-by-layer path wrote [a | g] (2 KB / point), hid (1 KB), d hid (2 KB) and d[a | g] (2 KB) per block and read
// each of them back once or twice: ~22 KB per point and block.  These kernels (round 5) move per point and block: forward 512 B in (hin), 772 B out
// (h2; the xhat3 fragments + 1 / std that are all the backward wants of h1, TL_H1_FRAG); backward 1284 B in (those fragments, the incoming gradient as a
// bf16 pair hi + lo, TL_DH_HL, hin) and 1024 B out (xn2 / dh1 fragments, the outgoing gradient pair); k_ff_wgrad reads the xhat3 fragments and the
// gradient's hi halves in place.  Round 4: 1.5 KB + 4.1 KB (h1 and dh as fp32, read twice each by the backward, fragments written for k_ff_wgrad).
// A sum over the points (weight gradients) needs the points on the K axis of the MFMA, i.e. along the registers, and here they are on
// the lanes: hence the second kernel, which recomputes hid and d[a | g] in the transposed orientation from the fragments.
//
// Weights: per optimiser step the block's W1 / W2 are re-packed (k_ff_pack) as bf16 MFMA A-fragments, 24 tiles of 32 x 32 per
// hidden chunk (48 KiB): W1a, W1g (K = channels, natural order: the B operand comes from memory), W2 (K = hidden units in the
// accumulator-register order of the GELU output), W2^T (rows = hidden units, K = channels) and W1a^T, W1g^T (rows = channels,
// K = hidden units in register order).  A workgroup (4 wavefronts = 128 points, two workgroups per CU: one's row loads and stores run
// under the other's chunk loop) streams the chunks L2 -> LDS with LDS-DMA through three buffers — forward: whole chunks (24 KiB), two
// ahead of the compute, one barrier per chunk; backward: the same since round 6 (BWD_TR: one 24 KiB record [W1a | W1g | W2^T] per chunk, the W1a^T / W1g^T
// operands read out of it with the LDS transpose read; until then two items per chunk, stage_item, and two barriers) — with counted s_waitcnt vmcnt.
//
// A workgroup's four tiles belong to ONE shape (its folded attention fragments sit in LDS once); every memory phase issues all of its loads
// before it consumes any; both kernels are spill-free (a scratch reload behind output stores waits for their acknowledgements); and between two
// of these kernels the row-shaped tensors are TILE-MAJOR (RowMap below) so that a wavefront's row accesses cover whole cache lines — see the
// notes at RowMap and in front of k_ff, and DESIGN.md 5.6.
//
// GELU: g * sigmoid(g (c1 + c3 g^2)) in fp32 with the hardware exp2 / rcp (max abs error 2.7e-4 against the erf form, the same
// form as the direct sampling kernel) and its exact derivative  s + g s (1 - s) (c1 + 3 c3 g^2).
#pragma once
#include "dfx_common.h"
#include "dfx_dropout.h"
#include <type_traits>

namespace dfx {
namespace ffused {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));

constexpr int C = 128, FH = 512, NCHUNK = FH / 32;
// Round 6: the FORWARD kernel computes hid = a gelu(g) with the sampling kernel's transcendental-free packed-fp16 polynomial (gelu16_f16 below) and feeds
// GEMM2 as an fp16 product (v_mfma_f32_32x32x16_f16) — the fp32 sigmoid form cost 6 packed-fp32 + 4 quarter-rate instructions per PAIR of values, and
// packed fp32 VALU serialises against the matrix pipe the SIMD's other wavefront is feeding (denoiser_kernel.hip's notes; ablation profiles/r06_ab_train.txt:
// the forward without any GELU arithmetic runs 716 -> 607 us).  It reads its OWN twelve tiles of a chunk (T_F*): W1a scaled by FWD_A_SCALE, W1g by
// FWD_G_SCALE (the polynomial works on u = g / 2; a g / 32 stays inside the fp16 range) and W2 as fp16 times the inverse of their product — exact
// powers of two.  The backward kernels keep the bf16 tiles 0..23 and the fp32 sigmoid form (their gradients multiply fp32 values that fp16 cannot hold).
#ifndef DFX_FF_FWD_F16
#define DFX_FF_FWD_F16 1
#endif
constexpr bool FWD_F16 = DFX_FF_FWD_F16 != 0;
#ifndef DFX_FF_BWD_TR
#define DFX_FF_BWD_TR 1   // the backward's second product reads W1a / W1g turned around in LDS instead of transposed tiles of its own (see BWD_SPREAD's neighbour below)
#endif
constexpr bool BWD_TR = DFX_FF_BWD_TR != 0;
constexpr float FWD_A_SCALE = 0.0625f, FWD_G_SCALE = 0.5f;
constexpr int TILES = FWD_F16 ? 36 : 24;          // tiles per chunk in the pack
constexpr int TILE_U4 = 128;                      // uint4 per tile (2 units x 64 lanes)
constexpr int CHUNK_U4 = TILES * TILE_U4;         // 4608 uint4 = 72 KiB (48 KiB without the forward's own tiles)
constexpr int FWD_TILES = 12, BWD_TILES = 12;     // LDS slot size in tiles.  forward: tiles FWD_TILE0 .. + 11 of a chunk; backward: see stage_item
enum { T_W1A = 0, T_W1G = 4, T_W2 = 8, T_W2T = 12, T_W1AT = 16, T_W1GT = 20, T_FW1A = 24, T_FW1G = 28, T_FW2 = 32 };
constexpr int FWD_TILE0 = FWD_F16 ? T_FW1A : 0;   // first tile of the forward's slot image (within it: W1a 0..3, W1g 4..7, W2 8..11 either way)
constexpr int PACK_RECORDS = FWD_F16 ? NCHUNK + 1 : NCHUNK;   // chunk records in the pack (the forward's skew needs a 17th for the last W2)

__host__ __device__ inline int rho(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
// Round 6: where lane l's 16 bytes of a W1a / W1g fragment (tiles 0 .. 7, unit u) sit inside the unit's 1 KiB: slot l ^ (4 u + 8 (l >> 5)).  Every reader of
// those tiles indexes with it (k_ff's first product, k_ff_wgrad's operands; ds_read_b128 stays conflict-free: an XOR with a constant inside each 16-lane
// access group).  It is there for the backward's SECOND product, which reads the same tiles turned around (ds_read_b64_tr_b16, see BWD_TR): the 32 lanes of
// an access group address rows 4 x (unit, k half) apart, which in the lane-linear image are 1024 / 512 bytes apart = the same banks, four ways; with the two
// bits folded into the row they are 32 distinct 8-byte slots of a 256-byte bank row (SQ_LDS_BANK_CONFLICT 2.5e7 -> 0 per launch).
__host__ __device__ inline int w1_slot(int lane, int u) { return lane ^ ((u << 2) | ((lane >> 5) << 3)); }
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
  float keep_a;      // dropout behind the GEGLU (attention.py:84): its scale 1 / (1 - p) rides on the `a` half of W1 / b1 (hid = a gelu(g) is linear in a),
                     // i.e. on the tiles W1a, W1a^T and b1a; 1 when dropout is off.  The kernels then only SELECT (k_ff_wgrad_finish scales dW1a, db1a back)
  // LayerNorm3's affine is folded into the product behind it (round 5):  W1 (xhat g3 + b3) + b1 = (W1 diag(g3)) xhat + (b1 + W1 b3).  The kernels'
  // B operand is the plain normalised row xhat3 (bf16) — which is also what LayerNorm3's backward needs, so the backward kernel does not read h1 a
  // second time — and their dxn3 accumulator holds d xhat3 = g3 o dxn3 directly.  The parameter side follows from G = d[a | g]^T xhat3 (k_ff_wgrad):
  // dW1 = G diag(g3) + db1 (x) b3,  d gamma3 = sum_o W1[o][.] G[o][.],  d beta3 = sum_o W1[o][.] db1[o]   (k_ff_wgrad_finish, k_ln3_param)
  const float *g3, *b3;   // (128) each
  float *b1f;             // (1024) the folded bias in natural order, `a` half times keep_a: k_ff_wgrad's
  float *b1ps;            // [NCHUNK][2 parts][2 hf][16] = b1p with the forward's scales (FWD_A_SCALE / FWD_G_SCALE): the forward kernel's table
};

struct PackBatch {
  PackArgs blk[DFX_MAX_DEPTH];   // blockIdx.y = transformer block: every block's weights in one launch
  // blockIdx.y == depth (optional, head_tab != nullptr): the head for the last block's forward kernel (TL_HEAD) — proj_out with post_norm's affine
  // folded in: tab[k][ch] = W_out[k][ch] gamma[ch] (3 x 128), tab[384 + k] = b_out[k] + sum_ch W_out[k][ch] beta[ch]
  int depth;
  const float *head_w, *head_b, *head_g, *head_be;
  float *head_tab;   // HEAD_TAB_FLOATS
};
constexpr int HEAD_TAB_FLOATS = 3 * C + 4;

// one thread per (chunk, tile, unit, lane)
__global__ void k_ff_pack(PackBatch batch) {
  if ((int)blockIdx.y == batch.depth) {   // the head's table: one workgroup
    if (blockIdx.x != 0) return;
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) batch.head_tab[i] = batch.head_w[i] * batch.head_g[i % C];
    if (threadIdx.x < 3) {
      float t = batch.head_b[threadIdx.x];
      for (int ch = 0; ch < C; ++ch) t = fmaf(batch.head_w[threadIdx.x * C + ch], batch.head_be[ch], t);
      batch.head_tab[3 * C + threadIdx.x] = t;
    }
    return;
  }
  const PackArgs &a = batch.blk[blockIdx.y];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < NCHUNK * 2 * 2 * 16) {   // b1p, b1f: b1 + W1 beta3
    const int r = idx & 15, hf = (idx >> 4) & 1, p = (idx >> 5) & 1, j = idx >> 6;
    const int row = p * FH + 32 * j + rho(r, hf);
    float t = a.b1[row];
    for (int ch = 0; ch < C; ++ch) t = fmaf(a.w1[(size_t)row * C + ch], a.b3[ch], t);
    t *= p == 0 ? a.keep_a : 1.0f;
    a.b1p[idx] = t, a.b1f[row] = t;
    a.b1ps[idx] = t * (p == 0 ? FWD_A_SCALE : FWD_G_SCALE);
  }
  if (idx < 2 * 4 * 16) {
    const int r = idx & 15, c = (idx >> 4) & 3, hf = idx >> 6;
    a.b2p[idx] = a.b2[32 * c + rho(r, hf)];
  }
  if (idx >= PACK_RECORDS * TILES * 2 * 64) return;
  const int lane = idx & 63, u = (idx >> 6) & 1, t = (idx >> 7) % TILES, j = idx / (TILES * 128);
  if (j == NCHUNK && t < T_FW2) return;   // the forward's 17th record: only its W2 tiles (hidden chunk 15) are read
  if (BWD_TR && t >= T_W1AT && t < T_FW1A) return;   // the transposed W1 tiles: nobody reads them (k_ff<true> turns tiles 0 .. 7 around in LDS)
  const int i = lane & 31, hf = lane >> 5;
  __bf16 v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float x;
    if (t < T_W2) {                      // W1a / W1g, k-tile c, natural K
      const int p = t >> 2, c = t & 3, ch = 32 * c + k_nat(u, hf, e);
      x = a.w1[(size_t)(p * FH + 32 * j + i) * C + ch] * a.g3[ch] * (p == 0 ? a.keep_a : 1.0f);
    } else if (t < T_W2T) {              // W2 row tile ct, K = hidden units in register order
      const int ct = t - T_W2;
      x = a.w2[(size_t)(32 * ct + i) * FH + 32 * j + k_reg(u, hf, e)];
    } else if (t < T_W1AT) {             // W2^T: rows = hidden units, k-tile c over the channels, natural K
      const int c = t - T_W2T;
      x = a.w2[(size_t)(32 * c + k_nat(u, hf, e)) * FH + 32 * j + i];
    } else if (t < T_FW1A) {             // W1a^T / W1g^T: rows = channels 32 ct + i, K = hidden units in register order
      const int p = (t - T_W1AT) >> 2, ct = (t - T_W1AT) & 3;
      x = a.w1[(size_t)(p * FH + 32 * j + k_reg(u, hf, e)) * C + 32 * ct + i] * a.g3[32 * ct + i] * (p == 0 ? a.keep_a : 1.0f);
    } else if (t < T_FW2) {              // the forward's W1a / W1g: tiles 0..7 times the fp16 scales (bf16, exact)
      const int p = (t - T_FW1A) >> 2, c = (t - T_FW1A) & 3, ch = 32 * c + k_nat(u, hf, e);
      x = a.w1[(size_t)(p * FH + 32 * j + i) * C + ch] * a.g3[ch] * (p == 0 ? a.keep_a * FWD_A_SCALE : FWD_G_SCALE);
    } else {                             // the forward's W2: tile 8 + ct as FP16, times 1 / (FWD_A_SCALE FWD_G_SCALE) — of hidden chunk j - 1: the forward's
      const int ct = t - T_FW2;          // records are SKEWED (record j = W1 of chunk j + W2 of chunk j - 1: one uninterrupted MFMA burst per record, ff_fwd)
      x = j == 0 ? 0.f : a.w2[(size_t)(32 * ct + i) * FH + 32 * (j - 1) + k_reg(u, hf, e)] * (1.0f / (FWD_A_SCALE * FWD_G_SCALE));
      const _Float16 h = (_Float16)x;
      v[e] = __builtin_bit_cast(__bf16, h);
      continue;
    }
    v[e] = (__bf16)x;
  }
  a.frags[t < T_W2 ? (idx & ~63) | w1_slot(lane, u) : idx] = *reinterpret_cast<const uint4 *>(v);
}

struct FfArgs {
  const uint4 *frags;
  const float *b1p, *b2p;
  const float *g3, *b3;  // LayerNorm3 affine: xn3 = LN3(h1) is computed here, forward and backward
  const float *h1;       // (R, 128) input of the sub-block (behind the attention)
  float *h2;             // forward: (R, 128) output (may alias h1)
  const float *dh;       // backward: (R, 128) gradient at the block output
  uint4 *pk;             // backward: [R / 32][2][4][2][64] the tile's xn3 / dh as MFMA fragments, for k_ff_wgrad
  float *dh1;            // backward: (R, 128) gradient at h1 = dh + LN3'(W1^T d[a | g])
  float *cpart;          // backward: [workgroups][3][128] column sums for d gamma3, d beta3, d b2
  long long R;           // B * N
  int B, N;              // shapes, points per shape (multiple of 32).  A workgroup's NW tiles belong to ONE shape (grid = B x ceil(N / 32 / NW);
                         // wavefronts past a shape's last tile recompute it and store nothing), so that the shape's folded attention
                         // fragments can sit in LDS once per workgroup
  // forward with the attention sub-block in front (train_attn_fused.h; at_frags != nullptr): h1 = hin + M_s softmax(A_s LN2(hin)) + b_o is
  // computed here from hin and written out (the backward reads it), instead of being read back from a kernel of its own
  const uint4 *at_frags;   // [B][4 sets][4][2][64] folded (A_s, M_s) fragments of this block
  const float *valid;      // (B, 4)
  const float *g2, *b2n, *bo;
  const float *hin;        // (R, 128)
  float *h1_out;           // (R, 128)
  // backward with the attention's input gradient behind it (at_frags != nullptr): dh_in = dh1 + LN2'(A_s^T dsim) leaves through dh_in
  // (may alias dh: a wavefront reads its rows of dh before it writes them) and cpart gets three more rows (d gamma2, d beta2, d b_o)
  float *dh_in;
  // Layout of the row-shaped tensors only these kernels touch (bit 1 = tile-major, see tile_b_off): the residual stream between two blocks and the
  // gradient between two blocks' backward kernels; the ends (written by the stem, read by the head, and back) stay row-major
  unsigned tiled;          // TL_* bits
  uint4 *pk2;              // with at_frags: [R / 32][2][4][2][64] the tile's xn2 and dh1 as bf16 fragments for k_attn_bwd_param (which
                           // needs nothing else of them); dh1 itself is then not written
  // dropout (k_ff<*, true>; dfx_dropout.h): the step's key, this block's two sites and ONE BIT per element of the two dropped tensors.
  // The forward draws the factors and leaves the bits; the backward kernels read them (k_ff_wgrad needs them in the transposed orientation,
  // where a regeneration would cost four Philox calls per group).  dmask [R / 32][DM_WORDS][64 lanes]: word w < 8, half h = hidden chunk 2 w + h,
  // bit r of the half = register r of the chunk's accumulator (unit 32 j + rho(r, hf) of point pj); words 8, 9 = the to_out site, half h =
  // channel tile 2 (w - 8) + h.  The GEGLU site's scale rides on W1a (PackArgs::keep_a); the to_out site's is applied here (dk.keep).
  DropKey dk;
  unsigned site_att, site_ff;
  unsigned *dmask;
  // the head inside the LAST block's forward kernel (TL_HEAD; round 5): eps = W_out post_norm(h2) + b_out from the accumulators (table: PackBatch::head_tab),
  // and h2 leaves only as what k_head_bwd wants of it — the bf16 fragments of post_norm's normalised row + 1 / std, in h2's own tiles (TL_H1_FRAG's format)
  const float *head_tab;
  float *eps;            // (B, 3, N)
#ifdef DFX_TRACE_FF
  unsigned long long *trace;
#endif
};
constexpr int DM_WORDS = 10, DM_TILE = DM_WORDS * 64;   // dwords per 32-point tile (2.5 KiB)

// The factors of 4 x 8 consecutive elements of one row — groups g8 .. g8 + 3 — for the two half-waves of a point together: lane (pj, hf) holds elements
// 4 hf .. 4 hf + 3 of every group (accumulator layout: register 4 q + m = element 8 q + 4 hf + m), so the pair computes each group ONCE (half hf
// takes groups hf and hf + 2) and trades the halves with four v_permlane32_swap.  w[q] = the two words (4 draws) of this lane's elements of group q.
__device__ __forceinline__ void drop_words(const DropKey &k, unsigned site, unsigned long long g8, int hf, unsigned (&w)[4][2]) {
#ifdef DFX_ABL_DROP_RNG   // (ablation builds only: wrong factors, no Philox rounds)
  const unsigned q = (unsigned)g8 * 2654435761u + site;
  uint4 A = make_uint4(q, q ^ k.k0, q + k.k1, q ^ 0x9E3779B9u), B = make_uint4(q + 1, q ^ k.k1, q + k.k0, q ^ 0xBB67AE85u);
#else
  uint4 A = drop_group(k, site, g8 + hf), B = drop_group(k, site, g8 + hf + 2);
#endif
  auto swap = [](unsigned &lo, unsigned &hi) {   // -> lo = [lo of the low half-wave | hi of the low half-wave], hi = [lo of the high | hi of the high]
    const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
    lo = r[0], hi = r[1];
  };
  swap(A.x, A.z), swap(A.y, A.w), swap(B.x, B.z), swap(B.y, B.w);
  w[0][0] = A.x, w[0][1] = A.y, w[1][0] = A.z, w[1][1] = A.w;
  w[2][0] = B.x, w[2][1] = B.y, w[3][0] = B.z, w[3][1] = B.w;
}
// select in place + the 16 keep bits (bit r = register r)
__device__ __forceinline__ unsigned drop_apply(v16f &v, const unsigned (&w)[4][2], unsigned thr) {
  unsigned bits = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned word = w[q][m >> 1];
      const bool keep = (m & 1) ? drop_keep_hi(word, thr) : drop_keep_lo(word, thr);
      v[4 * q + m] = keep ? v[4 * q + m] : 0.f;
      bits |= keep ? 1u << (4 * q + m) : 0u;
    }
  return bits;
}
// the same selection from stored bits (bit r of `bits`)
__device__ __forceinline__ void drop_select(v16f &v, unsigned bits) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = (bits >> r) & 1u ? v[r] : 0.f;
}

// fragment sets of the folded attention per shape (written by afused::k_attn_fold): tile t (4), unit u (2), lane (64) uint4 each
enum { F_AS = 0, F_MS = 1, F_MST = 2, F_AST = 3, NSETS = 4 };
constexpr int SET_U4 = 4 * 2 * 64, SHAPE_U4 = NSETS * SET_U4;   // 8 KiB per set, 32 KiB per shape

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }
constexpr float LN_EPS = 1e-5f;

__device__ __forceinline__ void dma1k(const void *gbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}

// 256 B per wavefront (one dword per lane) through the same path
__device__ __forceinline__ void dma256(const void *gbase, unsigned voff4, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff4), "s"(gbase), "s"(lds_addr) : "memory");
}

// Phase stamps of a few workgroups (tools/experiments/trace_train_ff.py builds with -DDFX_TRACE_FF; never in the shipped library): wave 0 of the
// workgroups whose id is a multiple of FFT_EVERY writes (tag, shader clock) pairs into a host-visible buffer dumped at process exit.
#ifdef DFX_TRACE_FF
constexpr int FFT_CAP = 320, FFT_WGS = 64, FFT_EVERY = 29;
struct FfTrace {
  unsigned long long *buf;
  int n;
  // `append` > 0: a later block of the forward chain keeps stamping behind the previous blocks' entries (their count); the first block clears the row
  __device__ __forceinline__ void init(unsigned long long *base, int kernel_slot, int thread = 0, int append = 0) {
    const int wg = blockIdx.x;
    buf = (base && wg % FFT_EVERY == 0 && wg / FFT_EVERY < FFT_WGS && (int)threadIdx.x == thread) ? base + ((size_t)kernel_slot * FFT_WGS + wg / FFT_EVERY) * FFT_CAP : nullptr;
    n = append;
    if (buf && !append)
      for (int i = 0; i < FFT_CAP; ++i) buf[i] = 0;
  }
  __device__ __forceinline__ void stamp(int tag) {
    if (buf && n < FFT_CAP) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      buf[n++] = ((unsigned long long)tag << 56) | ((unsigned long long)(hw & 0xffff) << 40) | (__builtin_readcyclecounter() & 0xffffffffffull);
    }
  }
};
inline unsigned long long *ff_trace_buffer() {
  static unsigned long long *buf = [] {
    unsigned long long *b = nullptr;
    const size_t n = (size_t)4 * FFT_WGS * FFT_CAP;
    if (hipHostMalloc(reinterpret_cast<void **>(&b), n * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) return (unsigned long long *)nullptr;
    for (size_t i = 0; i < n; ++i) b[i] = 0;
    static unsigned long long *keep = b;
    atexit([] {
      (void)hipDeviceSynchronize();
      const char *path = getenv("DFX_TRACE_FF_OUT");
      FILE *f = fopen(path ? path : "/tmp/ff_trace.txt", "w");
      if (!f) return;
      for (int k = 0; k < 4; ++k)
        for (int w = 0; w < FFT_WGS; ++w) {
          const unsigned long long *r = keep + ((size_t)k * FFT_WGS + w) * FFT_CAP;
          if (!r[0]) continue;
          fprintf(f, "kernel %d wg %d:", k, w * FFT_EVERY);
          for (int i = 0; i < FFT_CAP && r[i]; ++i) fprintf(f, " %d@%llx:%llu", (int)(r[i] >> 56), (r[i] >> 40) & 0xffff, r[i] & 0xffffffffffull);
          fprintf(f, "\n");
        }
      fclose(f);
    });
    return b;
  }();
  return buf;
}
#define FFT_INIT(args, slot) FfTrace fft; fft.init((args).trace, slot)
#define FFT_INIT_CHAIN(args, slot, append) FfTrace fft; fft.init((args).trace, slot, 0, append)
#define FFT_COUNT(var) var = fft.n
#define FFT_INIT2(args, slot, thread) FfTrace fft; fft.init((args).trace, slot, thread)
#define FFT(tag) fft.stamp(tag)
#else
#define FFT_INIT(args, slot)
#define FFT_INIT_CHAIN(args, slot, append)
#define FFT_COUNT(var)
#define FFT_INIT2(args, slot, thread)
#define FFT(tag)
#endif

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
// This lane's row of h in the B-operand layout (8 consecutive channels 32 c + 16 u + 8 hf .. per (c, u)), LayerNorm statistics
// (two passes like torch) and the normalised, affine row as bf16 fragments.  gb = LDS table [gamma(128) | beta(128)].
__device__ __forceinline__ void load_rows(const float *__restrict__ hrow, int hf, v8f (&x)[4][2]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float *p = hrow + 32 * c + k_nat(u, hf, 0);
      const v4f lo = *reinterpret_cast<const v4f *>(p), hi = *reinterpret_cast<const v4f *>(p + 4);
      x[c][u] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
// The same row in the accumulator layout (register r of tile c = channel 32 c + rho(r, hf)) without reading it again: the two
// half-waves of a point hold complementary channel sets in either layout, and v_permlane32_swap trades the halves of two registers
// in one instruction: (x[u][m], x[u][4 + m]) -> (d[8 u + m], d[8 u + 4 + m]).  The same call converts back.
__device__ __forceinline__ void rows_to_acc(const v8f (&x)[4][2], v16f (&d)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float lo = x[c][u][m], hi = x[c][u][4 + m];   // (copies: a bit_cast applied to a vector-element expression reads element 0)
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        d[c][8 * u + m] = __builtin_bit_cast(float, r0);
        d[c][8 * u + 4 + m] = __builtin_bit_cast(float, r1);
      }
}
__device__ __forceinline__ void acc_to_rows(const v16f (&d)[4], v8f (&x)[4][2]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float lo = d[c][8 * u + m], hi = d[c][8 * u + 4 + m];
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        x[c][u][m] = __builtin_bit_cast(float, r0);
        x[c][u][4 + m] = __builtin_bit_cast(float, r1);
      }
}
__device__ __forceinline__ v16f zero16() {
  v16f z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// masked softmax over the four keys of each head, accumulator layout: registers 4 g .. 4 g + 3 = keys of head 2 g + hf
__device__ __forceinline__ void softmax_regs(v16f &sim, unsigned vmask) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float sj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sj[j] = (vmask >> j) & 1u ? sim[4 * g + j] : -3.402823466e38f;   // masked_fill_(~mask, -finfo.max)
    const float m = fmaxf(fmaxf(sj[0], sj[1]), fmaxf(sj[2], sj[3]));
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) sj[j] = __builtin_amdgcn_exp2f((sj[j] - m) * 1.44269504088896341f), den += sj[j];   // hardware exp2 / rcp (these kernels
    const float inv = __builtin_amdgcn_rcpf(den);                                                                    // run with bf16 products only: P is rounded to bf16 next)
#pragma unroll
    for (int j = 0; j < 4; ++j) sim[4 * g + j] = sj[j] * inv;
  }
}
// softmax backward in the same layout: dsim = P (dP - sum_j P dP)
__device__ __forceinline__ void softmax_bwd_regs(const v16f &P, v16f &dP) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) dot = fmaf(P[4 * g + j], dP[4 * g + j], dot);
#pragma unroll
    for (int j = 0; j < 4; ++j) dP[4 * g + j] = P[4 * g + j] * (dP[4 * g + j] - dot);
  }
}
__device__ __forceinline__ void ln_rows(const v8f (&x)[4][2], int hf, const float *gb, uint4 (&xn)[4][2], float &mu, float &rstd) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[c][u][e];
  s += xhalf(s);
  mu = s * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = x[c][u][e] - mu;
        q = fmaf(d, d, q);
      }
  q += xhalf(q);
  rstd = 1.0f / sqrtf(q * (1.0f / C) + LN_EPS);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ch = 32 * c + k_nat(u, hf, 0);
      const v4f g0 = *reinterpret_cast<const v4f *>(gb + ch), g1 = *reinterpret_cast<const v4f *>(gb + ch + 4);
      const v4f b0 = *reinterpret_cast<const v4f *>(gb + C + ch), b1 = *reinterpret_cast<const v4f *>(gb + C + ch + 4);
      v8f y;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = fmaf((x[c][u][e] - mu) * rstd, g0[e], b0[e]);
        y[4 + e] = fmaf((x[c][u][4 + e] - mu) * rstd, g1[e], b1[e]);
      }
      xn[c][u] = __builtin_bit_cast(uint4, __builtin_convertvector(y, v8bf));
    }
}
__device__ __forceinline__ void ln_rows(const float *__restrict__ hrow, int hf, const float *gb, uint4 (&xn)[4][2], float &mu, float &rstd) {
  v8f x[4][2];
  load_rows(hrow, hf, x);
  ln_rows(x, hf, gb, xn, mu, rstd);
}

// ln_rows in two pieces — the statistics, and ONE affine bf16 fragment — for callers that consume the fragments as they are made
__device__ __forceinline__ void ln_stats(const v8f (&x)[4][2], float &mu, float &rstd) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[c][u][e];
  s += xhalf(s);
  mu = s * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = x[c][u][e] - mu;
        q = fmaf(d, d, q);
      }
  q += xhalf(q);
  rstd = 1.0f / sqrtf(q * (1.0f / C) + LN_EPS);
}
__device__ __forceinline__ uint4 ln_frag(const v8f &xcu, int c, int u, int hf, const float *gb, float mu, float rstd) {
  const int ch = 32 * c + k_nat(u, hf, 0);
  const v4f g0 = *reinterpret_cast<const v4f *>(gb + ch), g1 = *reinterpret_cast<const v4f *>(gb + ch + 4);
  const v4f b0 = *reinterpret_cast<const v4f *>(gb + C + ch), b1 = *reinterpret_cast<const v4f *>(gb + C + ch + 4);
  v8f y;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y[e] = fmaf((xcu[e] - mu) * rstd, g0[e], b0[e]);
    y[4 + e] = fmaf((xcu[4 + e] - mu) * rstd, g1[e], b1[e]);
  }
  return __builtin_bit_cast(uint4, __builtin_convertvector(y, v8bf));
}

// the plain normalised row as a bf16 fragment (LayerNorm3: the affine part rides on W1 / b1, PackArgs)
__device__ __forceinline__ uint4 xhat_frag(const v8f &xcu, float mu, float rstd) {
  v8f y;
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = (xcu[e] - mu) * rstd;
  return __builtin_bit_cast(uint4, __builtin_convertvector(y, v8bf));
}
// bf16 B-operand fragments of a row -> the same values as fp32 in the accumulator layout (rows_to_acc on packed pairs: dwords (0, 2) and (1, 3) of
// a fragment trade half-waves, i.e. elements (m, 4 + m) -> registers (8 u + m, 8 u + 4 + m), two elements per swap)
__device__ __forceinline__ void frags_to_acc(const uint4 (&f)[4][2], v16f (&d)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned w[4] = {f[c][u].x, f[c][u].y, f[c][u].z, f[c][u].w};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const auto r = __builtin_amdgcn_permlane32_swap(w[h], w[2 + h], false, false);
        const unsigned r0 = r[0], r1 = r[1];
        d[c][8 * u + 2 * h] = __builtin_bit_cast(float, r0 << 16), d[c][8 * u + 2 * h + 1] = __builtin_bit_cast(float, r0 & 0xffff0000u);
        d[c][8 * u + 4 + 2 * h] = __builtin_bit_cast(float, r1 << 16), d[c][8 * u + 4 + 2 * h + 1] = __builtin_bit_cast(float, r1 & 0xffff0000u);
      }
    }
}

constexpr int HL_LO_U4 = 8 * 64;   // bf16-pair tiles (TL_DH_HL below): uint4 offset of the lo half within a tile
// One channel tile (accumulator layout, fp32) of a row-shaped tensor as a bf16 pair, see TL_DH_HL: hi in the B-operand layout (frags_to_acc backwards:
// the packed pairs (8 u + 2 h, +1) and (8 u + 4 + 2 h, +1) trade half-waves), lo = bf16(v - hi) in place
// one channel tile (accumulator layout, fp32) -> its two bf16 B-operand fragments (frags_to_acc backwards)
__device__ __forceinline__ void acc_to_frags(const v16f &v, uint4 (&hb)[2]) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
  unsigned hp[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const v2f x = {v[2 * d], v[2 * d + 1]};
    hp[d] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, v2bf));
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    unsigned w[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const auto r = __builtin_amdgcn_permlane32_swap(hp[4 * u + h], hp[4 * u + 2 + h], false, false);
      w[h] = r[0], w[2 + h] = r[1];
    }
    hb[u] = uint4{w[0], w[1], w[2], w[3]};
  }
}
__device__ __forceinline__ void store_hl(uint4 *tile_lane, int c, const v16f &v, bool live) {
  unsigned hp[8], lp[8];   // packed pairs (2 d, 2 d + 1)
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    const v2f x = {v[2 * d], v[2 * d + 1]};
    hp[d] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, v2bf));
    const v2f r = {x[0] - __builtin_bit_cast(float, hp[d] << 16), x[1] - __builtin_bit_cast(float, hp[d] & 0xffff0000u)};
    lp[d] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, v2bf));
  }
  uint4 hb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    unsigned w[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const auto r = __builtin_amdgcn_permlane32_swap(hp[4 * u + h], hp[4 * u + 2 + h], false, false);
      w[h] = r[0], w[2 + h] = r[1];
    }
    hb[u] = uint4{w[0], w[1], w[2], w[3]};
  }
  if (live) {
    tile_lane[(c * 2 + 0) * 64] = hb[0], tile_lane[(c * 2 + 1) * 64] = hb[1];
    tile_lane[HL_LO_U4 + (c * 2 + 0) * 64] = uint4{lp[0], lp[1], lp[2], lp[3]};
    tile_lane[HL_LO_U4 + (c * 2 + 1) * 64] = uint4{lp[4], lp[5], lp[6], lp[7]};
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
// The same arithmetic on PAIRS of accumulator registers (round 5): gfx950's packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) do two
// lanes' worth of an fp32 operation per issue slot — same IEEE results, half the VALU slots; only exp2 / rcp stay one element per instruction.  The loops
// of the three feed-forward kernels spend as many SIMD cycles on this arithmetic as on their MFMAs.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float c) { return v2f{c, c}; }
__device__ __forceinline__ v2f exp2_2(v2f x) { return v2f{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])}; }
__device__ __forceinline__ v2f rcp_2(v2f x) { return v2f{__builtin_amdgcn_rcpf(x[0]), __builtin_amdgcn_rcpf(x[1])}; }
__device__ __forceinline__ void gelu_fd2(v2f x, v2f &f, v2f &d) {
  const v2f x2 = x * x;
  const v2f e = exp2_2(x * __builtin_elementwise_fma(splat2(-0.100125614f), x2, splat2(-2.30876530f)));
  const v2f s = rcp_2(splat2(1.0f) + e);
  f = x * s;
  d = __builtin_elementwise_fma(f * (splat2(1.0f) - s), __builtin_elementwise_fma(splat2(0.20820537f), x2, splat2(1.60031416f)), s);
}
__device__ __forceinline__ v2f gelu_f2(v2f x) {
  const v2f e = exp2_2(x * __builtin_elementwise_fma(splat2(-0.100125614f), x * x, splat2(-2.30876530f)));
  return x * rcp_2(splat2(1.0f) + e);
}
__device__ __forceinline__ v2f pair(const v16f &v, int i) { return v2f{v[2 * i], v[2 * i + 1]}; }
__device__ __forceinline__ void set_pair(v16f &v, int i, v2f x) { v[2 * i] = x[0], v[2 * i + 1] = x[1]; }

// ---- the sampling kernel's packed-fp16 GEGLU (denoiser_kernel.hip: gelu16_f16_cvt / gelu16_f16_math, fit: tools/experiments/fit_gelu_poly.py), for the
// FORWARD kernel: gelu(g) = g Phi(g), Phi(g) ~ 1/2 + u R(z), u = g / 2, z = min(u^2 - m, L^2 - m), R of degree 5 (3.9e-4 in exact arithmetic on [-6, 6];
// rms 8.5e-4 on a gelu(g) in fp16).  Ten packed instructions per pair of values, no transcendental, and v_pk_*_f16 does not contend with the matrix pipe.
// Inputs: a / 16 and g / 2 (the forward's tiles are scaled); output hid / 32 as the two fp16 B fragments of GEMM2 (pack8's element order: registers 8 u ..).
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
__device__ __forceinline__ h2 pk_f16(float lo, float hi) { return __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
__device__ __forceinline__ h2 h2c(float v) { return h2{(_Float16)v, (_Float16)v}; }
// (a, g) -> packed fp16 first: after these sixteen conversions the accumulators are dead and their next initialisers can be fetched underneath the arithmetic
__device__ __forceinline__ void geglu16_f16_cvt(const v16f &a, const v16f &g, h2 (&aa)[8], h2 (&gg)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) gg[i] = pk_f16(g[2 * i], g[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 8; ++i) aa[i] = pk_f16(a[2 * i], a[2 * i + 1]);
}
__device__ __forceinline__ void geglu16_f16_math(const h2 (&aa)[8], const h2 (&gg)[8], uint4 (&hid)[2]) {
  h2 y[8], z[8], r[8];
  // stage by stage over the eight pairs: eight independent instructions between two dependent ones (a dependent v_pk_* costs a wait state on gfx950)
#define DFX_STAGE(expr)                                   \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) { expr; } \
  __builtin_amdgcn_sched_barrier(0)
  __builtin_amdgcn_sched_barrier(0);
  DFX_STAGE(z[i] = __builtin_elementwise_fma(gg[i], gg[i], h2c(-1.62f)));
  DFX_STAGE(y[i] = aa[i] * gg[i]);
  DFX_STAGE(z[i] = __builtin_elementwise_min(z[i], h2c(1.62f)));
  DFX_STAGE(r[i] = __builtin_elementwise_fma(z[i], h2c(-0.0011402554f), h2c(0.0057853916f)));
  DFX_STAGE(r[i] = __builtin_elementwise_fma(r[i], z[i], h2c(-0.0158536041f)));
  DFX_STAGE(r[i] = __builtin_elementwise_fma(r[i], z[i], h2c(0.0409006897f)));
  DFX_STAGE(r[i] = __builtin_elementwise_fma(r[i], z[i], h2c(-0.1098130657f)));
  DFX_STAGE(r[i] = __builtin_elementwise_fma(r[i], z[i], h2c(0.3885767652f)));
  DFX_STAGE(asm("v_pk_fma_f16 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(z[i]) : "v"(gg[i]), "v"(r[i])));   // Phi
  DFX_STAGE(y[i] = y[i] * z[i]);
#undef DFX_STAGE
  hid[0] = uint4{__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1]), __builtin_bit_cast(unsigned, y[2]), __builtin_bit_cast(unsigned, y[3])};
  hid[1] = uint4{__builtin_bit_cast(unsigned, y[4]), __builtin_bit_cast(unsigned, y[5]), __builtin_bit_cast(unsigned, y[6]), __builtin_bit_cast(unsigned, y[7])};
}
__device__ __forceinline__ v16f mfma_f16(const uint4 &a, const uint4 &b, v16f c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
}

constexpr int B1P_FLOATS = NCHUNK * 64, B2P_FLOATS = 128;
// per 32-point tile, for k_ff_wgrad: two sets of fragments [4 c][2 u][64 lanes] (8 KiB each) in memory; k_ff_wgrad derives the
// transposed sets (PK_XNT, PK_DHT: channels on the lanes, points along the registers) in LDS
enum { PK_XN = 0, PK_DH = 1, PK_XNT = 2, PK_DHT = 3 };
constexpr int PK_TILE_U4 = 2 * 8 * 64;   // 16 KiB
constexpr int NW_BWD = 4;   // wavefronts per workgroup, backward: 128 points; three 24 KiB slots (79 KiB: two workgroups per CU) — one record per chunk (BWD_TR) or two items
#ifndef DFX_FF_NW_FWD
#define DFX_FF_NW_FWD 4
#endif
constexpr int NW_FWD = DFX_FF_NW_FWD;   // forward: 128 points and 77 KiB of LDS, TWO workgroups per CU — one's row loads / stores run under the other's chunk loop
template <bool BWD> constexpr int nw_of() { return BWD ? NW_BWD : NW_FWD; }
constexpr int NBUF = 3;   // LDS chunk buffers: the stream runs two chunks ahead of the compute
#ifndef DFX_FF_BWD_SPREAD
#define DFX_FF_BWD_SPREAD 0   // (measured round 6, same box: 279.3 vs 278.0 us per block — no gain in the backward; the forward keeps its spread)
#endif
// s_setprio experiments (round 6, same-box A/Bs, profiles/r06_ab_train_prio.txt): raising a wavefront's priority inside its VALU stretch (the sampling
// kernel runs its V slots at 3) helps the forward (652 -> 637 us) and HURTS the backward (278 -> 283 per block) and the weight-gradient kernel's
// producer (224 -> 244): there the other wavefront of the SIMD is the one feeding the matrix pipe
#ifndef DFX_FF_VPRIO_FWD
#define DFX_FF_VPRIO_FWD 3
#endif
#ifndef DFX_FF_VPRIO_BWD
#define DFX_FF_VPRIO_BWD 0
#endif
#ifndef DFX_FF_MPRIO_BWD
#define DFX_FF_MPRIO_BWD 0   // priority inside the backward's MFMA bursts
#endif
#ifndef DFX_WG_PPRIO
#define DFX_WG_PPRIO 0       // k_ff_wgrad: producer's GEGLU arithmetic
#endif
#ifndef DFX_WG_CPRIO
#define DFX_WG_CPRIO 0       // k_ff_wgrad: consumer's 24-MFMA stretch
#endif
constexpr bool BWD_SPREAD = DFX_FF_BWD_SPREAD != 0;   // backward: ring pieces issued inside the MFMA bursts (see the loop)
// Round 6: the backward's SECOND product (dxn3 += W1a^T da + W1g^T dg) takes its A operands out of the W1a / W1g tiles of the FIRST product with the LDS
// transpose read (ds_read_b64_tr_b16: a 16-lane group reads 16 x 4 elements and gets them back turned around — tools/ubench/tr_read_probe.hip pins the lane
// mapping), instead of from transposed tiles of their own: a chunk is ONE 24 KiB record (not two items, 40 KiB), the ring runs two chunks ahead behind ONE
// barrier per chunk, and the wave issues six ring pieces per chunk instead of ten.  Same bf16 values, same MFMA operand order: bit-identical gradients.
static_assert(!(BWD_TR && BWD_SPREAD), "the spread variant belongs to the two-item ring");
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v4s lds_tr16(unsigned lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) v4s *>(lds_addr));
}
// (measured, B = 128 x 2048, backward / forward per block: 8 waves x 3 buffers 441 / 161 us; 4 waves x 2 buffers, two workgroups
// per CU, 513 / 161 us; 8 x 2: 662 / 182 us)

// stage `ntile_groups` runs of tiles of chunk j into LDS buffer `buf` (each wave copies every NW-th KiB)
template <bool BWD>
__device__ __forceinline__ void stage_chunk(const uint4 *frags, int j, unsigned lds_buf, int wave, unsigned voff) {
  const char *src = reinterpret_cast<const char *>(frags + (size_t)j * CHUNK_U4);
  constexpr int NK = (BWD ? BWD_TILES : FWD_TILES) * 2;   // KiB to copy
  constexpr int NW = nw_of<BWD>();
#pragma unroll
  for (int k = 0; k < NK / NW; ++k) {
    const int piece = k * NW + wave;                                     // destination KiB
    const int spiece = BWD && piece >= 16 ? piece + 8 : piece;           // backward skips tiles 8..11 (W2)
    dma1k(src + spiece * 1024, voff, lds_buf + piece * 1024);
  }
}
// ff_fwd's record j: the forward's own twelve tiles of pack record j (T_FW1A ..: scaled W1a | W1g of chunk j, fp16 W2 of chunk j - 1)
__device__ __forceinline__ void stage_record_fwd(const uint4 *frags, int j, unsigned lds_buf, int wave, unsigned voff) {
  const char *src = reinterpret_cast<const char *>(frags + (size_t)j * CHUNK_U4) + FWD_TILE0 * 2048;
#pragma unroll
  for (int k = 0; k < FWD_TILES * 2 / NW_FWD; ++k) dma1k(src + (k * NW_FWD + wave) * 1024, voff, lds_buf + (k * NW_FWD + wave) * 1024);
}

// Backward (BWD_TR = 0; with BWD_TR only the even items exist: item 2 j IS chunk j's record): a chunk streams as two items through a ring of three 24 KiB slots — item 2 j = [W1a | W1g | W2^T] of chunk j (tiles 0..7 and
// 12..15: GEMM1 and the d hid product), item 2 j + 1 = [W1a^T | W1g^T] (tiles 16..23: the dxn3 product) — so that a workgroup fits into
// half a CU's LDS.  Wave w copies pieces w, w + 4, ..: six per even item, four per odd one.
__device__ __forceinline__ void stage_item(const uint4 *frags, int item, unsigned lds_slot, int wave, unsigned voff) {
  const char *src = reinterpret_cast<const char *>(frags + (size_t)(item >> 1) * CHUNK_U4);
  if (item & 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = k * NW_BWD + wave;
      dma1k(src + (32 + p) * 1024, voff, lds_slot + p * 1024);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int p = k * NW_BWD + wave;
      dma1k(src + (p < 16 ? p : p + 8) * 1024, voff, lds_slot + p * 1024);
    }
  }
}
// LDS tile index of pack tile t (backward: within its item's slot)
template <bool BWD>
__device__ __forceinline__ constexpr int lt(int t) { return !BWD ? t : t >= 16 ? t - 16 : t >= 12 ? t - 4 : t; }

// The same row in the accumulator layout straight from memory: register 4 q + m of tile c = channel 32 c + 8 q + 4 hf + m (= 32 c + rho(4 q + m, hf))
__device__ __forceinline__ void load_rows_acc(const float *__restrict__ hrow, int hf, v16f (&d)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f t = *reinterpret_cast<const v4f *>(hrow + 32 * c + 8 * q + 4 * hf);
      d[c][4 * q + 0] = t[0], d[c][4 * q + 1] = t[1], d[c][4 * q + 2] = t[2], d[c][4 * q + 3] = t[3];
    }
}

// ---- Tile-major rows (round 4).  With row-major (R, 128) fp32 tensors every row access of a wavefront — lane = point — touches 32 cache lines for 1 KiB
// (16 bytes per row), and each line is requested by four different instructions: the memory phases ran at the rate of their L1 transactions.  Between two
// of these kernels the layout is free: a 32-point tile (16 KiB, the same bytes as its 32 rows) is stored as 16 blocks (c, u, half) of 1 KiB = 64 lanes x 4
// floats, the B-operand layout itself; a load or store instruction then covers whole lines — in BOTH register layouts the kernels use:
//   B-operand layout, block (c, u), element e of lane (pj, hf) = channel 32 c + 16 u + 8 hf + e     -> ((c 2 + u) 2 + e / 4) 256 + lane 4 + e % 4
//   accumulator layout, register 4 q + m of tile c, lane (pj, hf) = channel 32 c + 8 q + 4 hf + m  -> ((c 2 + q / 2) 2 + hf) 256 + (pj + 32 (q & 1)) 4 + m
// (the same element either way).  Measured: k_ff<true> 356 -> 334 us, k_ff<false> 180 -> 165 us per block.
enum { TL_HIN = 1, TL_H1 = 2, TL_H2 = 4, TL_DH = 8, TL_DHIN = 16, TL_DH_HL = 32, TL_DHIN_HL = 64, TL_H1_FRAG = 128, TL_HEAD = 256 };
// ---- h1 as the backward wants it (round 5, TL_H1_FRAG; forward with the attention sub-block inside).  The backward kernels need two things of h1: the bf16
// fragments of xhat3 = LayerNorm3's normalised row (k_ff<true>'s B operand and, turned around, LayerNorm3's backward; k_ff_wgrad's tiles) and the row's
// 1 / std.  So the forward stores exactly those in the tile's 16 KiB — [xhat3: 8 blocks (c, u) of 1 KiB = the PK_XN set][rstd: 32 floats] — instead of
// the fp32 rows: 260 B per point written and read instead of 512, bit-identical operands (the backward rounded the same fp32 values to bf16), no
// LayerNorm statistics in the backward's prologue, and k_ff<true> no longer writes the xhat3 half of FfArgs::pk (k_ff_wgrad reads these tiles).
constexpr int H1F_RSTD = 2048;   // float offset of the tile's 32 rstd values
// ---- The gradient between two blocks' backward kernels as a bf16 PAIR (round 5, TL_DH_HL / TL_DHIN_HL): hi = bf16(dh) and lo = bf16(dh - hi), i.e. dh to
// 2^-17 relative, in the tile's own 16 KiB: [hi: 8 blocks (c, u) of 1 KiB, the bf16 B-operand fragments themselves][lo: 8 blocks (c, k) of 1 KiB in the
// accumulator layout, registers 8 k .. 8 k + 7 of tile c as 8 bf16 per lane].  The backward kernel wants dh twice — rounded to bf16 as the operand of
// d hid = W2^T dh in front of the chunk loop, and in full behind it for dh1 = dh + ... — and cannot hold it across the loop: as fp32 that was 2 x 512 B
// per point, now 256 B (hi, which goes straight into the MFMA and is turned into the accumulator layout with permlane swaps behind the loop) + 256 B
// (lo).  k_ff_wgrad reads the same hi blocks as its dh fragments (FwArgs::dhf), so k_ff<true> writes only the xhat3 half of FfArgs::pk: 768 B per
// point and block less traffic.  The ends of the stack (the head's output, the stem's input) stay fp32 row-major.

struct RowMap {   // per tensor: one lane offset for each register layout (floats), tiled or not
  bool tiled;
  unsigned lb, la;
  __host__ __device__ __forceinline__ RowMap(bool t, int lane, int pj, int hf) : tiled(t), lb(t ? lane * 4 : pj * C + 8 * hf), la(t ? hf * 256 + pj * 4 : pj * C + 4 * hf) {}
  __host__ __device__ __forceinline__ unsigned b(int c, int u, int half) const { return lb + (tiled ? ((c * 2 + u) * 2 + half) * 256 : 32 * c + 16 * u + 4 * half); }
  __host__ __device__ __forceinline__ unsigned a(int c, int q) const { return la + (tiled ? (c * 2 + (q >> 1)) * 512 + 128 * (q & 1) : 32 * c + 8 * q); }
};
__device__ __forceinline__ void load_rows(const float *__restrict__ tile, const RowMap &m, v8f (&x)[4][2]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const v4f lo = *reinterpret_cast<const v4f *>(tile + m.b(c, u, 0)), hi = *reinterpret_cast<const v4f *>(tile + m.b(c, u, 1));
      x[c][u] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
__device__ __forceinline__ void load_rows_acc(const float *__restrict__ tile, const RowMap &m, v16f (&d)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f t = *reinterpret_cast<const v4f *>(tile + m.a(c, q));
      d[c][4 * q + 0] = t[0], d[c][4 * q + 1] = t[1], d[c][4 * q + 2] = t[2], d[c][4 * q + 3] = t[3];
    }
}

// LDS map of k_ff: [ring: 3 slots x 24 KiB][b1 of all chunks 4 KiB][gamma3 | beta3 1 KiB][gamma2 | beta2 | b_o 1.5 KiB][b2 0.5 KiB] = 79 KiB, two
// workgroups per CU.  The folded attention fragments of the workgroup's shape borrow ring space while the ring is idle: forward, slot 2
// during the prologue ([A_s | M_s], 16 KiB; chunk 2 is only requested behind the prologue's last barrier); backward, 24 KiB at AT_OFF_BWD
// behind the loop ([A_s | M_s^T | A_s^T]; the column-sum tiles of the epilogue use the 30 KiB in front of it).
constexpr int RING_BYTES = NBUF * FWD_TILES * 2048;
constexpr int TAB_B1 = RING_BYTES, TAB_GB3 = TAB_B1 + B1P_FLOATS * 4, TAB_GB2 = TAB_GB3 + 2 * C * 4, TAB_B2 = TAB_GB2 + 3 * C * 4, FF_LDS = TAB_B2 + C * 4;
constexpr int AT_OFF_BWD = 32768;

// Memory phases (round 4): every phase that touches HBM issues ALL of its loads before it consumes any of them.  Left to hipcc, the folded
// attention's 32 fragment loads sat one s_waitcnt vmcnt(0) apart (each behind a store it could alias), the staging loops of the tables were
// three dependent round trips, and the rows were requested only behind the weight ring's first wait: 14 serial HBM round trips in the
// forward's prologue, ~6 in the backward's and ~45 in its epilogue — 58 % of a backward wavefront's life.  Now: rows first, tables and ring
// behind them, ONE wait; attention fragments through LDS (LDS-DMA, no registers); the epilogue's second reads as two batches.  Vector-memory
// operations complete in issue order (loads, stores and LDS-DMA alike), so stores may stay in flight across the counted waits of the loop.
// The body of k_ff / k_ff_fwd_chain.  `first`: the lane's rows come from memory; otherwise (forward chain, blocks behind the first) they are `hc`, the
// previous block's output, still in the accumulator registers it was computed in.  `hc` leaves with this block's output (forward).
template <bool BWD, bool DROP>
__device__ __forceinline__ void ff_run(const FfArgs &a, const bool first, v16f (&hc)[4], int &trace_n) {
  constexpr int NW = nw_of<BWD>();
  static_assert(NW == 4 && NW * 64 == 2 * C && FWD_TILES == BWD_TILES && B1P_FLOATS * 4 == NW * 1024, "table staging below assumes 256 threads");
  constexpr int BUF_BYTES = (BWD ? BWD_TILES : FWD_TILES) * 2048;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ff_smem[];
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));   // (opaque per call: in the chain hipcc otherwise hoists every lane-derived offset out of the block loop and keeps ~40 registers alive)
  const int lane = lane_, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ff_smem);
  const unsigned voff = lane * 16;
  // one shape per workgroup: tiles (blockIdx.x % gps) * NW .. + NW - 1 of shape blockIdx.x / gps
  const int tps = a.N / 32, gps = (tps + NW - 1) / NW;
  const int s = __builtin_amdgcn_readfirstlane((int)blockIdx.x / gps);
  int ti = ((int)blockIdx.x - s * gps) * NW + wave;
  const bool live = ti < tps;
  if (!live) ti = tps - 1;   // a workgroup's trailing wavefronts past the shape's end recompute its last tile and store nothing
  // rows: a wave-uniform base (scalar registers) + ONE lane offset for every row-shaped tensor
  const long long rowbase = ((long long)s * a.N + ti * 32) * C;
  const RowMap m_hin(a.tiled & TL_HIN, lane, pj, hf), m_h1(a.tiled & TL_H1, lane, pj, hf), m_h2(a.tiled & TL_H2, lane, pj, hf);
  const RowMap m_dh(a.tiled & TL_DH, lane, pj, hf), m_dhin(a.tiled & TL_DHIN, lane, pj, hf);
  const bool at = a.at_frags != nullptr;

  constexpr int PIECES = FWD_TILES * 2 / NW;   // forward: LDS-DMA instructions per wave and chunk
  FFT_INIT_CHAIN(a, BWD ? 1 : 0, !BWD && !first ? trace_n : 0);   // (trace builds: the chain's later blocks append)
  FFT(1);
  // ---- prologue, memory side: this lane's rows first (the oldest requests come back first), then the tables, then the ring ----
  v8f x[4][2], xd[4][2];
  const bool h1f = a.tiled & TL_H1_FRAG;
  uint4 xn[4][2];
  float mu, rstd = 0.f;
  if (BWD && h1f) {   // what the forward left: the xhat3 fragments and 1 / std of this lane's row
    const uint4 *hp = reinterpret_cast<const uint4 *>(a.h1 + rowbase) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) xn[c][0] = hp[(c * 2 + 0) * 64], xn[c][1] = hp[(c * 2 + 1) * 64];
    rstd = (a.h1 + rowbase)[H1F_RSTD + pj];
  } else if (BWD || first) {
    load_rows((BWD || !at ? a.h1 : a.hin) + rowbase, BWD || !at ? m_h1 : m_hin, x);   // (one load site: selected pointer and map, no branch)
  }
  const bool hl_in = BWD && (a.tiled & TL_DH_HL), hl_out = BWD && (a.tiled & TL_DHIN_HL);
  uint4 dhb[4][2];
  if (BWD) {
    if (hl_in) {   // the bf16 fragments themselves
      const uint4 *hp = reinterpret_cast<const uint4 *>(a.dh + rowbase) + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c) dhb[c][0] = hp[(c * 2 + 0) * 64], dhb[c][1] = hp[(c * 2 + 1) * 64];
    } else {
      load_rows(a.dh + rowbase, m_dh, xd);
    }
  }
  // dropout: this tile's bit words (one lane = one point half, the forward's mapping); the backward holds the eight feed-forward words across the loop
  unsigned *dmk = DROP ? a.dmask + (size_t)(rowbase / (32 * C)) * DM_TILE + lane : nullptr;
  const unsigned long long prow = (unsigned long long)(rowbase / C) + pj;   // this lane's row of the (R, .) tensors
  unsigned dmw = 0;   // forward: the word being assembled; backward: the word of chunks (j & ~1, + 1), requested at the top of every second chunk
  float *b1s = reinterpret_cast<float *>(ff_smem + TAB_B1);
  float *dump = reinterpret_cast<float *>(ff_smem + TAB_GB3);   // 1 KiB nobody reads
  float *gb2 = reinterpret_cast<float *>(ff_smem + TAB_GB2);   // LayerNorm2 affine | to_out bias (attention sub-block fused in)
  float *b2s = reinterpret_cast<float *>(ff_smem + TAB_B2);
  // tables: every thread fetches gamma2 | beta2 of its channel, threads 0 .. 127 (waves 0, 1) b_o, the others b2 — unconditional loads
  // through selected pointers (a branch here costs a wait for the rows: hipcc resolves the write-after-write on the other path with vmcnt(0))
  const int tx = threadIdx.x, tc = tx & (C - 1);
  const bool lo_half = wave < NW / 2;
  // (LayerNorm3's affine rides on W1 / b1: no table for it.  Without the attention sub-block the loads go to a valid dummy, b1p)
  const float *tp0 = at ? a.g2 : a.b1p, *tp1 = at ? a.b2n : a.b1p;
  const float *tp2 = lo_half ? (at ? a.bo : a.b1p) : (!BWD ? a.b2p : a.b1p);
  const float tv0 = tp0[tc], tv1 = tp1[tc], tv2 = tp2[tc];
  dma1k(reinterpret_cast<const char *>(a.b1p) + wave * 1024, voff, lds0 + TAB_B1 + wave * 1024);
  constexpr bool TR = BWD && BWD_TR;
  // (TR, dropout) the bit word of chunks (0, 1) -> this wave's 256 bytes of the mask buffer: LDS-DMA like the ring's pieces, so that the loop's counted waits
  // cover it (loads of one kind complete in order; a load that returns to a register does not keep that order against them — HISTORY round 6 #11)
  const unsigned *dmt = DROP ? a.dmask + (size_t)(rowbase / (32 * C)) * DM_TILE : nullptr;   // wave-uniform: this tile's words
  if (TR && DROP) dma256(dmt, lane * 4, lds0 + TAB_GB3 + wave * 256);   // (as sixteen lanes x 16 bytes instead: measured, no difference)
  if (BWD) {
    stage_item(a.frags, 0, lds0, wave, voff);
    stage_item(a.frags, TR ? 2 : 1, lds0 + BUF_BYTES, wave, voff);
  } else {
    if (at) {   // [A_s | M_s] of this shape -> slot 2
      const char *src = reinterpret_cast<const char *>(a.at_frags + (size_t)s * SHAPE_U4);
#pragma unroll
      for (int k = 0; k < 4; ++k) dma1k(src + (k * NW + wave) * 1024, voff, lds0 + 2 * BUF_BYTES + (k * NW + wave) * 1024);
    }
    stage_chunk<BWD>(a.frags, 0, lds0, wave, voff);
    stage_chunk<BWD>(a.frags, 1, lds0 + BUF_BYTES, wave, voff);
  }
  unsigned vmask = 0;
  if (at) {
#pragma unroll
    for (int j = 0; j < 4; ++j) vmask |= (a.valid[s * 4 + j] != 0.f ? 1u : 0u) << j;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ONE round trip: rows, tables, attention fragments, the first two chunks / items
  {
    // waves 0, 1: gamma2 | beta2 | b_o; waves 2, 3: b2 — their copies of gamma2 | beta2 go to the dump, or (TR: the dump is the mask buffer) on top of waves 0, 1's: the same values
    float *d0 = lo_half || TR ? gb2 : dump, *d2 = lo_half ? gb2 + 2 * C : b2s;
    d0[tc] = tv0, d0[C + tc] = tv1, d2[tc] = tv2;
  }
  __syncthreads();
  FFT(16);
  // B operand of the products over the channels: xhat3 of LN3(h1) (bf16, natural K order) and, backward, dh rounded to bf16
  v16f acc[4];   // forward: the residual stream h; backward: dxn3
  if (!BWD && !first) acc_to_rows(hc, x);   // (chained forward: this block's input rows = the previous block's output, never re-read)
  if (!BWD && at) {
    // attention sub-block in registers: h1 = hin + M_s softmax(A_s LN2(hin)) + b_o, then straight on to LayerNorm3
    const uint4 *fl = reinterpret_cast<const uint4 *>(ff_smem + 2 * BUF_BYTES) + lane;
    v16f sim = zero16();
    {
      float mu2, rstd2;
      ln_stats(x, mu2, rstd2);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) sim = mfma(fl[F_AS * SET_U4 + (c * 2 + u) * 64], ln_frag(x[c][u], c, u, hf, gb2, mu2, rstd2), sim);
    }
    softmax_regs(sim, vmask);
    const uint4 p0 = pack8(sim, 0), p1 = pack8(sim, 1);
    rows_to_acc(x, acc);   // (also in the chain: keeping hc alive through the attention prologue beside x costs 64 registers)
    unsigned attw = 0;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      if (DROP) {
        // h1 = hin + dropout(M_s P + b_o)  (attention.py:177: to_out = Sequential(Linear, Dropout)): the sub-block's output in a fresh accumulator,
        // selected and scaled, then added to the residual; element (row, channel 32 ct + 8 q + 4 hf + m) = group row * 16 + 4 ct + q of the site
        v16f t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f b = *reinterpret_cast<const v4f *>(gb2 + 2 * C + 32 * ct + 8 * q + 4 * hf);
#pragma unroll
          for (int m = 0; m < 4; ++m) t[4 * q + m] = b[m];
        }
        t = mfma(fl[F_MS * SET_U4 + (ct * 2 + 0) * 64], p0, t);
        t = mfma(fl[F_MS * SET_U4 + (ct * 2 + 1) * 64], p1, t);
        unsigned w[4][2];
        drop_words(a.dk, a.site_att, prow * (C / 8) + 4 * ct, hf, w);
        const unsigned bits = drop_apply(t, w, a.dk.thr);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = fmaf(a.dk.keep, t[r], acc[ct][r]);
        attw |= bits << (16 * (ct & 1));
        if (ct & 1) {
          if (live) dmk[(8 + (ct >> 1)) * 64] = attw;
          attw = 0;
        }
      } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f b = *reinterpret_cast<const v4f *>(gb2 + 2 * C + 32 * ct + 8 * q + 4 * hf);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[ct][4 * q + m] += b[m];
      }
      acc[ct] = mfma(fl[F_MS * SET_U4 + (ct * 2 + 0) * 64], p0, acc[ct]);
      acc[ct] = mfma(fl[F_MS * SET_U4 + (ct * 2 + 1) * 64], p1, acc[ct]);
      }
      if (live && !h1f) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<v4f *>(a.h1_out + rowbase + m_h1.a(ct, q)) = v4f{acc[ct][4 * q], acc[ct][4 * q + 1], acc[ct][4 * q + 2], acc[ct][4 * q + 3]};
      }
    }
    acc_to_rows(acc, x);   // h1 in the B-operand layout for LayerNorm3
    FFT(17);
    __syncthreads();       // everybody is done with the attention fragments: slot 2 takes chunk 2 at the top of the loop
    FFT(18);
  } else if (!BWD) {
    rows_to_acc(x, acc);   // h1 in the accumulator layout, from the same read
  }
  // LayerNorm3 on this lane's row of h1: statistics and the plain normalised row as bf16 fragments (gamma3 / beta3 ride on W1 / b1, PackArgs)
  if (!(BWD && h1f)) {
    ln_stats(x, mu, rstd);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) xn[c][u] = xhat_frag(x[c][u], mu, rstd);
  }
  if (!BWD && at && h1f && live) {   // h1 for the backward kernels: these fragments and 1 / std (TL_H1_FRAG)
    uint4 *hp = reinterpret_cast<uint4 *>(a.h1_out + rowbase) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) hp[(c * 2 + 0) * 64] = xn[c][0], hp[(c * 2 + 1) * 64] = xn[c][1];
    if (hf == 0) (a.h1_out + rowbase)[H1F_RSTD + pj] = rstd;
  }
  if (BWD) {
    if (!hl_in) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) dhb[c][u] = __builtin_bit_cast(uint4, __builtin_convertvector(xd[c][u], v8bf));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    // the tile for k_ff_wgrad: xhat3 and dh as bf16 B-operand fragments (points on the lanes), whole 1 KiB stores (left in flight); a bf16-pair
    // gradient's hi half IS the second set: k_ff_wgrad reads it in place
    if (live) {
      uint4 *pk = a.pk + (size_t)(rowbase / (32 * C)) * PK_TILE_U4 + lane;
      if (!h1f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) pk[(PK_XN * 8 + c * 2 + 0) * 64] = xn[c][0], pk[(PK_XN * 8 + c * 2 + 1) * 64] = xn[c][1];
      }
      if (!hl_in) {
#pragma unroll
        for (int c = 0; c < 4; ++c) pk[(PK_DH * 8 + c * 2 + 0) * 64] = dhb[c][0], pk[(PK_DH * 8 + c * 2 + 1) * 64] = dhb[c][1];
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v16f b2;
      load16(b2, b2s + hf * 64 + c * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] += b2[r];
    }
  }
  FFT(2);

#pragma unroll 1
  for (int j = 0; j < NCHUNK; ++j) {
    // forward: chunk j + 2 -> the slot of chunk j - 1 (every wave is past it); backward: item 2 j + 2 -> the slot of item 2 j - 1
    if (!BWD) {
      if (j + NBUF - 1 < NCHUNK) stage_chunk<BWD>(a.frags, j + NBUF - 1, lds0 + ((j + NBUF - 1) % NBUF) * BUF_BYTES, wave, voff);
    } else {
      // dropout: this lane's bit word of chunks (j, j + 1), requested in front of the item's pieces; one register instead of the tile's eight words
      // (inline asm: hipcc's own wait for a load it knows would be vmcnt(0) at the first use — behind the NEXT item's pieces).  ROUND 6 FIX: the wait that
      // covers it is vmcnt(0) at this chunk's item boundary, NOT the counted wait of the other chunks: a load that returns to a register and the LDS-DMA
      // pieces behind it do not complete in issue order under load — with vmcnt(6) the word was occasionally stale at B = 128 x 2048 (run-to-run
      // differences of 2e-3 in every gradient below the first block's; tools/soak_train_streams.py with dropout; rounds 5's code had it too).  LDS-DMA
      // pieces among themselves, and stores against them, keep the counted waits valid (loads of one kind complete in order).
      if (!TR) {
        if (DROP && !(j & 1)) asm volatile("global_load_dword %0, %1, off" : "=v"(dmw) : "v"(dmk + (j >> 1) * 64) : "memory");
        if (!BWD_SPREAD && 2 * j + 2 < 2 * NCHUNK) stage_item(a.frags, 2 * j + 2, lds0 + ((2 * j + 2) % 3) * BUF_BYTES, wave, voff);
      } else {
        // one record per chunk, two ahead: chunk j + 2 -> the slot of chunk j - 1 (every wave is past its last read: the barrier below).  Odd chunks first
        // request the bit word of the NEXT pair (its predecessor was read into a register in chunk j - 1): in front of the pieces, so "at most six
        // outstanding" at this chunk's end covers it
        if (DROP && (j & 1) && j + 1 < NCHUNK) dma256(dmt + ((j + 1) >> 1) * 64, lane * 4, lds0 + TAB_GB3 + wave * 256);
        if (j + 2 < NCHUNK) stage_item(a.frags, 2 * (j + 2), lds0 + ((j + 2) % 3) * BUF_BYTES, wave, voff);
      }
    }
    // Round 6 (BWD_SPREAD): the wave's pieces of the next chunk's two items are issued from INSIDE the two MFMA bursts, one behind every fourth MFMA (six
    // of item 2 j + 2 in the first burst, four of item 2 j + 3 in the second) instead of back to back in front of them, where each LDS-DMA instruction
    // cost the wave ~100 cycles of issue (MI355X_MICROARCH.md).  Same slots, same counted waits.  The last chunk has nothing to request: its pieces go
    // to a 1 KiB dump nobody reads (no branch in the bursts).
    const bool nxt = j + 1 < NCHUNK;
    const char *isrc = reinterpret_cast<const char *>(a.frags + (size_t)(nxt ? j + 1 : j) * CHUNK_U4) + wave * 1024;
    const unsigned idst0 = nxt ? lds0 + ((2 * j + 2) % 3) * BUF_BYTES + wave * 1024 : lds0 + TAB_GB3;
    const unsigned idst1 = nxt ? lds0 + ((2 * j + 3) % 3) * BUF_BYTES + wave * 1024 : lds0 + TAB_GB3;
    const unsigned istep = nxt ? 4096u : 0u;
    const uint4 *fr = reinterpret_cast<const uint4 *>(ff_smem + (BWD && !TR ? (2 * j) % 3 : j % NBUF) * BUF_BYTES) + lane;
    const uint4 *fs0 = fr - lane + w1_slot(lane, 0), *fs1 = fr - lane + w1_slot(lane, 1);
    auto frag = [&](int t, int u) -> uint4 { return (t < T_W2 ? (u ? fs1 : fs0) : fr)[(lt<BWD>(t) * 2 + u) * 64]; };   // (W1a / W1g: w1_slot)
    // ---- [a | g] = b1 + W1 xn3 ----
    v16f av, gv;
    load16(av, b1s + ((j * 2 + 0) * 2 + hf) * 16);
    load16(gv, b1s + ((j * 2 + 1) * 2 + hf) * 16);
    if (!BWD) {
      // Forward: the chunk's 24 A fragments go through a ring of eight registers (the sampling kernel's scheme): MFMA m takes P[m & 7],
      // which is refilled in place with fragment m + 8 — GEMM2's eight W2 fragments are requested during GEMM1's second half and have
      // landed long before the GELU is done.  One exposed LDS round trip per chunk; left to itself hipcc requested every fragment one
      // or two MFMAs ahead of its use and waited for it (lgkmcnt(0 / 1) in front of nearly every MFMA).
      // GEMM1 MFMA m: k-tile c = m >> 2, unit u = (m >> 1) & 1, a / g = m & 1;  GEMM2 MFMA i: row tile ct = i & 3, unit u = i >> 2
      auto f1 = [&](int m) -> uint4 { return frag(((m & 1) ? T_W1G : T_W1A) + (m >> 2), (m >> 1) & 1); };
      auto f2 = [&](int i) -> uint4 { return frag(T_W2 + (i & 3), i >> 2); };
      uint4 P[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) P[i] = f1(i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (m & 1) gv = mfma(P[m & 7], xn[m >> 2][(m >> 1) & 1], gv);
        else av = mfma(P[m & 7], xn[m >> 2][(m >> 1) & 1], av);
        P[m & 7] = m + 8 < 16 ? f1(m + 8) : f2(m + 8 - 16);
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      v16f hv;
#pragma unroll
#ifdef DFX_ABL_FWD_GELU   // (ablation builds only, tools/experiments/ab_train_variants.sh: wrong numbers, the loop without its GELU arithmetic)
      for (int i = 0; i < 8; ++i) set_pair(hv, i, pair(av, i) * pair(gv, i));
#else
      for (int i = 0; i < 8; ++i) set_pair(hv, i, pair(av, i) * gelu_f2(pair(gv, i)));
#endif
      if (DROP) {
        // dropout behind the GEGLU (attention.py:84): element (row, unit 32 j + 8 q + 4 hf + m) = group row * 64 + 4 j + q of the site; the scale rides on `a`
        unsigned w[4][2];
        drop_words(a.dk, a.site_ff, prow * (FH / 8) + 4 * j, hf, w);
        const unsigned bits = drop_apply(hv, w, a.dk.thr);
        dmw = (j & 1) ? dmw | (bits << 16) : bits;
        if ((j & 1) && live) dmk[(j >> 1) * 64] = dmw;   // (a store in flight only tightens the counted waits below: operations complete in order)
      }
      const uint4 h0 = pack8(hv, 0), h1 = pack8(hv, 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i & 3] = mfma(P[i], (i >> 2) ? h1 : h0, acc[i & 3]);
    } else {
      // Backward, first burst: GEMM1 recompute (16 MFMAs) and d hid = W2^T dh (8) are independent of each other — one stream of 24
      // MFMAs through the same ring of eight fragment registers (one exposed LDS round trip).
      auto f1 = [&](int m) -> uint4 {
        return m < 16 ? frag(((m & 1) ? T_W1G : T_W1A) + (m >> 2), (m >> 1) & 1) : frag(T_W2T + ((m - 16) >> 1), (m - 16) & 1);
      };
      uint4 P[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) P[i] = f1(i);
      v16f dhid;
#pragma unroll
      for (int r = 0; r < 16; ++r) dhid[r] = 0.f;
      __builtin_amdgcn_sched_barrier(0);
      if (DFX_FF_MPRIO_BWD) __builtin_amdgcn_s_setprio(DFX_FF_MPRIO_BWD);
#pragma unroll
      for (int m = 0; m < 24; ++m) {
        if (m >= 16) dhid = mfma(P[m & 7], dhb[(m - 16) >> 1][(m - 16) & 1], dhid);
        else if (m & 1) gv = mfma(P[m & 7], xn[m >> 2][(m >> 1) & 1], gv);
        else av = mfma(P[m & 7], xn[m >> 2][(m >> 1) & 1], av);
        if (BWD_SPREAD && m % 4 == 0) {   // piece k of item 2 j + 2 (tiles 0..7, 12..15 of the next chunk): destination KiB 4 k + wave, source KiB + 8 from 16 on
          const int k = m / 4;
          dma1k(isrc + (k < 4 ? k * 4 : k * 4 + 8) * 1024, voff, idst0 + k * istep);
        }
        if (m + 8 < 24) P[m & 7] = f1(m + 8);
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- item boundary: item 2 j + 1 must have landed (loads complete in order: only the six pieces of item 2 j + 2, requested at the
      // top of this chunk, may still be out), every wave is done with item 2 j -> its slot takes item 2 j + 3.  The boundary sits in FRONT
      // of the GEGLU arithmetic now, so that the second burst's first eight fragments travel while the VALU works ----
      FFT(10);
      if (DFX_FF_MPRIO_BWD) __builtin_amdgcn_s_setprio(0);
      if (!TR) {
      if (2 * j + 2 < 2 * NCHUNK && !(DROP && !(j & 1))) asm volatile("s_waitcnt vmcnt(6)" : "+v"(dmw)::"memory");   // (dmw: its readers stay behind the wait)
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(dmw)::"memory");   // (the last chunk; and with dropout every even chunk: its mask word must have landed)
      FFT(11);
#ifndef DFX_ABL_BWD_NOBAR   // (ablation builds only: racy, wrong numbers — what do the loop's two barriers cost?)
      __syncthreads();
#endif
      FFT(12);
      if (!BWD_SPREAD && 2 * j + 3 < 2 * NCHUNK) stage_item(a.frags, 2 * j + 3, lds0 + ((2 * j + 3) % 3) * BUF_BYTES, wave, voff);
      }
      const uint4 *fr2 = reinterpret_cast<const uint4 *>(ff_smem + ((2 * j + 1) % 3) * BUF_BYTES) + lane;
      // TR: the lane's address inside a 2 KiB W1a / W1g tile for the transpose read.  Lane 16 G + s of group G supplies row (hidden unit) 4 (G >> 1) + (s >> 2)
      // (+ 16 u + 8 r by the immediate), channels 16 (G & 1) + 4 (s & 3) .. + 3 — in the tile's image: unit G & 1, lane slot row + 32 ((s & 3) >> 1), byte
      // 8 (s & 1) — and receives hidden units 4 (G >> 1) + 0 .. 3 of channel 16 (G & 1) + s: elements 4 r .. 4 r + 3 of the A operand (K in register order)
      // (the row's slot through w1_slot: bits 2, 3 of the row carry the unit and the k half, so r = 0 / 1 need a base each)
      const int tu = (lane >> 4) & 1, thf = (lane & 3) >> 1, trow = 4 * (lane >> 5) + ((lane & 15) >> 2) + 32 * thf;
      const unsigned trb0 = lds0 + (j % 3) * BUF_BYTES + (tu * 64 + w1_slot(trow, tu)) * 16 + 8 * (lane & 1);
      const unsigned trb1 = lds0 + (j % 3) * BUF_BYTES + (tu * 64 + w1_slot(trow + 8, tu)) * 16 + 8 * (lane & 1);
      // second burst, MFMA m: row tile ct = m & 3, operand q = m >> 2 (W1a^T unit 0, unit 1, W1g^T unit 0, unit 1: the order per accumulator)
      auto f2 = [&](int m) -> uint4 {
        if (TR) {
          const unsigned o = ((m >> 3) * 4 + (m & 3)) * 2048 + ((m >> 2) & 1) * 256;
          return __builtin_bit_cast(uint4, __builtin_shufflevector(lds_tr16(trb0 + o), lds_tr16(trb1 + o), 0, 1, 2, 3, 4, 5, 6, 7));
        }
        return fr2[(lt<BWD>(((m >> 2) < 2 ? T_W1AT : T_W1GT) + (m & 3)) * 2 + ((m >> 2) & 1)) * 64];
      };
#pragma unroll
      for (int i = 0; i < 8; ++i) P[i] = f2(i);
      __builtin_amdgcn_sched_barrier(0);
      if (DROP) {   // d hid in front of the dropout = selected d hid behind it (the scale rides on `a` and on W1a^T)
        if (TR && !(j & 1)) dmw = reinterpret_cast<const unsigned *>(ff_smem + TAB_GB3)[wave * 64 + lane];   // (landed: the previous chunk's end, or the prologue)
        drop_select(dhid, (j & 1) ? dmw >> 16 : dmw);
      }
      // ---- GEGLU backward on the registers ----
      if (DFX_FF_VPRIO_BWD) __builtin_amdgcn_s_setprio(DFX_FF_VPRIO_BWD);
      v16f da, dg;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v2f f, d;
#ifdef DFX_ABL_BWD_GELU
        f = pair(gv, i), d = splat2(1.0f);
#else
        gelu_fd2(pair(gv, i), f, d);
#endif
        const v2f dh2 = pair(dhid, i);
        set_pair(da, i, dh2 * f);
        set_pair(dg, i, dh2 * pair(av, i) * d);
      }
      const uint4 a0 = pack8(da, 0), a1 = pack8(da, 1), g0 = pack8(dg, 0), g1 = pack8(dg, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (DFX_FF_VPRIO_BWD) __builtin_amdgcn_s_setprio(0);
      if (DFX_FF_MPRIO_BWD) __builtin_amdgcn_s_setprio(DFX_FF_MPRIO_BWD);
      // ---- dxn3 += W1a^T da + W1g^T dg ----
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int q = m >> 2;
        acc[m & 3] = mfma(P[m & 7], q == 0 ? a0 : q == 1 ? a1 : q == 2 ? g0 : g1, acc[m & 3]);
        if (BWD_SPREAD && m % 4 == 0) dma1k(isrc + (32 + m) * 1024, voff, idst1 + (m / 4) * istep);   // piece m / 4 of item 2 j + 3 (tiles 16..23)
        if (m + 8 < 16) P[m & 7] = f2(m + 8);
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BWD && DFX_FF_MPRIO_BWD) __builtin_amdgcn_s_setprio(0);
    // chunk j + 1 must have landed: loads complete in order, so "at most PIECES outstanding" leaves only chunk j + 2's pieces
    // (whatever the order between loads and the backward's stores); no new pieces in the last two iterations -> drain
    FFT(13);
    if (TR) {    // chunk j + 1 (and an odd chunk's bit word) must have landed; the six pieces of chunk j + 2 may still be out
      if (j + 2 < NCHUNK) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (BWD) {   // item 2 j + 2 must have landed; the four pieces of item 2 j + 3 may still be out
      if (2 * j + 3 < 2 * NCHUNK) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (NBUF > 2 && j + 2 < NCHUNK) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FFT(14);
#ifdef DFX_ABL_BWD_NOBAR
    if (!BWD) __syncthreads();
#else
    __syncthreads();
#endif
    FFT(15);
  }
  FFT(3);
#ifdef DFX_ABL_BWD_NOBAR
  __syncthreads();
#endif
  if (!BWD) {
#pragma unroll
    for (int c = 0; c < 4; ++c) hc[c] = acc[c];   // (dead in the single-block kernel)
    if (a.tiled & TL_HEAD) {
      // post_norm + proj_out on the accumulators (two-pass statistics like k_head_fwd's ln_row); xhat replaces h2 in acc
      float sm = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += acc[c][r];
      sm += xhalf(sm);
      const float mup = sm * (1.0f / C);
      float qs = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[c][r] - mup;
          qs = fmaf(d, d, qs);
        }
      qs += xhalf(qs);
      const float rstdp = 1.0f / sqrtf(qs * (1.0f / C) + LN_EPS);
      float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = 32 * c + 8 * q + 4 * hf;
          const v4f w0 = *reinterpret_cast<const v4f *>(a.head_tab + ch), w1 = *reinterpret_cast<const v4f *>(a.head_tab + C + ch);
          const v4f w2 = *reinterpret_cast<const v4f *>(a.head_tab + 2 * C + ch);
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float xh = (acc[c][4 * q + m] - mup) * rstdp;
            acc[c][4 * q + m] = xh;
            e0 = fmaf(xh, w0[m], e0), e1 = fmaf(xh, w1[m], e1), e2 = fmaf(xh, w2[m], e2);
          }
        }
      e0 += xhalf(e0), e1 += xhalf(e1), e2 += xhalf(e2);
      if (!live) return;
      if (hf == 0) {
        float *ep = a.eps + (size_t)s * 3 * a.N + ti * 32 + pj;
        ep[0] = e0 + a.head_tab[3 * C], ep[a.N] = e1 + a.head_tab[3 * C + 1], ep[2 * (size_t)a.N] = e2 + a.head_tab[3 * C + 2];
        (a.h2 + rowbase)[H1F_RSTD + pj] = rstdp;
      }
      uint4 *hp = reinterpret_cast<uint4 *>(a.h2 + rowbase) + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint4 hb[2];
        acc_to_frags(acc[c], hb);
        hp[(c * 2 + 0) * 64] = hb[0], hp[(c * 2 + 1) * 64] = hb[1];
      }
      FFT(9);
      FFT_COUNT(trace_n);
      return;
    }
    FFT_COUNT(trace_n);
    if (!live) return;
    float *out = a.h2 + rowbase;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<v4f *>(out + m_h2.a(c, q)) = v4f{acc[c][4 * q], acc[c][4 * q + 1], acc[c][4 * q + 2], acc[c][4 * q + 3]};
    FFT(9);
    FFT_COUNT(trace_n);
    return;
  }
  // ---- backward epilogue.  The ring is idle (every wave is behind the loop's last barrier): the shape's [A_s | M_s^T | A_s^T] fragments are
  // requested into it first, then the second reads of h1 and dh (accumulator layout) as one batch ----
  if (at) {
    const char *src = reinterpret_cast<const char *>(a.at_frags + (size_t)s * SHAPE_U4);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int p = k * NW + wave;   // destination KiB; the source skips the M_s set (KiB 8 .. 15)
      dma1k(src + (p < 8 ? p : p + 8) * 1024, voff, lds0 + AT_OFF_BWD + p * 1024);
    }
  }
  v16f xh[4], dv[4];
  if (hl_in) {   // dh = hi (the loop's own operand, turned into the accumulator layout) + lo (8 x 16 bytes per lane)
    const uint4 *lp = reinterpret_cast<const uint4 *>(a.dh + rowbase) + HL_LO_U4 + lane;
    uint4 lo[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) lo[c][0] = lp[(c * 2 + 0) * 64], lo[c][1] = lp[(c * 2 + 1) * 64];
    frags_to_acc(dhb, dv);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const unsigned w[4] = {lo[c][k].x, lo[c][k].y, lo[c][k].z, lo[c][k].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          dv[c][8 * k + 2 * d] += __builtin_bit_cast(float, w[d] << 16);
          dv[c][8 * k + 2 * d + 1] += __builtin_bit_cast(float, w[d] & 0xffff0000u);
        }
      }
  } else {
    load_rows_acc(a.dh + rowbase, m_dh, dv);
  }
  unsigned attw[2] = {0, 0};
  if (DROP && at) attw[0] = dmk[8 * 64], attw[1] = dmk[9 * 64];   // the to_out site's bits of this tile (same batch of loads)
  // gradient in front of the to_out dropout = selected, scaled gradient at h1 (channel tile c)
  auto att_drop = [&](v16f v, int c) {
    if (DROP) {
      drop_select(v, attw[c >> 1] >> (16 * (c & 1)));
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] *= a.dk.keep;
    }
    return v;
  };
  // ---- LayerNorm3 backward on the accumulators (register r of tile c = channel 32 c + rho(r, hf)).  acc holds d xhat3 (gamma3 rides on W1^T), and
  // xhat3 is the loop's own B operand turned into the accumulator layout (bf16: no second read of h1):
  //     dh1 = dh + rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat))
  // d gamma3 / d beta3 come from the weight-gradient side (k_ln3_param); the column sum over the workgroup's points left here is d b2 ----
  frags_to_acc(xn, xh);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s1 += acc[c][r];
      s2 = fmaf(acc[c][r], xh[c][r], s2);
    }
  s1 += xhalf(s1), s2 += xhalf(s2);
  s1 *= (1.0f / C), s2 *= (1.0f / C);
  FFT(4);
  // The sums run over the lanes; a per-wave LDS tile (the chunk buffers are free now) turns 32 points x 32 channels around: written
  // [channel][point] from the accumulator layout, read back 16 points of one channel per lane.
  constexpr int TROW = 36;   // floats per tile row: 16-byte aligned rows, conflict-free column reads
  float *tt = reinterpret_cast<float *>(ff_smem) + wave * 32 * TROW;
  float *cred = reinterpret_cast<float *>(ff_smem) + NW * 32 * TROW;   // [NW][NQ][128]; rows 0, 1 (d gamma3, d beta3 until round 5) are unused
  static_assert((NW * 32 * TROW + NW * 6 * C) * 4 <= AT_OFF_BWD, "column-sum tiles overlap the attention fragments");
  const int NQ = at ? 6 : 3;
  const float keep = live ? 1.f : 0.f;
  auto colsum = [&](const v16f &v, int which, int c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tt[rho(r, hf) * TROW + pj] = v[r];
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const v4f x4 = *reinterpret_cast<const v4f *>(tt + pj * TROW + 16 * hf + 4 * k);
      t += (x4[0] + x4[1]) + (x4[2] + x4[3]);
    }
    t += xhalf(t);
    if (hf == 0) cred[(wave * NQ + which) * C + 32 * c + pj] = t * keep;
  };
  // dh1 replaces dh in dv, tile by tile
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    colsum(dv[c], 2, c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v4f o;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int r = 4 * q + m;
        o[m] = dv[c][r] + rstd * (acc[c][r] - s1 - xh[c][r] * s2);
        dv[c][r] = o[m];
      }
      if (live && !at) *reinterpret_cast<v4f *>(a.dh1 + rowbase + (pj * C + 32 * c + 8 * q + 4 * hf)) = o;   // (with the attention fused in, dh1 leaves as fragments: pk2)
    }
  }
  FFT(5);
  if (at) {
    // ---- the attention sub-block's input gradient on the same rows (afused::k_attn_bwd_dx's arithmetic): recompute LN2 / sim / P from hin,
    // dP = M_s^T dh1, softmax backward, dxn2 = A_s^T dsim, LayerNorm2 backward: dh_in = dh1 + ... ; column sums for d gamma2, d beta2, d b_o ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the attention fragments (requested a phase ago)
    __syncthreads();
    const uint4 *fl = reinterpret_cast<const uint4 *>(ff_smem + AT_OFF_BWD) + lane;
    constexpr int L_AS = 0, L_MST = SET_U4, L_AST = 2 * SET_U4;
    load_rows(a.hin + rowbase, m_hin, x);   // travels while dP is computed
    float mu2, rstd2;
    v16f P = zero16(), ds = zero16();
    // this tile's xn2 | dh1 fragments for k_attn_bwd_param: a wave-uniform base (scalar registers) + the lane
    uint4 *pk2p = a.pk2 + (size_t)(rowbase / (32 * C)) * (2 * 8 * 64);
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // dh1 (behind the to_out dropout: what reaches M_s P + b_o) as the B operand, a channel tile at a time
      v8f t8[2];
      const v16f dvc = att_drop(dv[c], c);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float lo = dvc[8 * u + m], hi = dvc[8 * u + 4 + m];
          const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi), false, false);
          const unsigned r0 = r[0], r1 = r[1];
          t8[u][m] = __builtin_bit_cast(float, r0);
          t8[u][4 + m] = __builtin_bit_cast(float, r1);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint4 db = __builtin_bit_cast(uint4, __builtin_convertvector(t8[u], v8bf));
        ds = mfma(fl[L_MST + (c * 2 + u) * 64], db, ds);
        if (live) pk2p[(8 + c * 2 + u) * 64 + lane] = db;
      }
    }
    FFT(6);
    {
      // LayerNorm2 of hin (ln_rows' arithmetic), a fragment at a time: xhat2 replaces the row in place, the affine bf16 fragment goes to
      // k_attn_bwd_param and into sim = A_s xn2 straight away (no 32-register array of fragments beside the rows)
      float sm = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) sm += x[c][u][e];
      sm += xhalf(sm);
      mu2 = sm * (1.0f / C);
      float qs = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = x[c][u][e] - mu2;
            qs = fmaf(d, d, qs);
          }
      qs += xhalf(qs);
      rstd2 = 1.0f / sqrtf(qs * (1.0f / C) + LN_EPS);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int ch = 32 * c + k_nat(u, hf, 0);
          const v4f g0 = *reinterpret_cast<const v4f *>(gb2 + ch), g1 = *reinterpret_cast<const v4f *>(gb2 + ch + 4);
          const v4f b0 = *reinterpret_cast<const v4f *>(gb2 + C + ch), b1 = *reinterpret_cast<const v4f *>(gb2 + C + ch + 4);
          v8f y;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[c][u][e] = (x[c][u][e] - mu2) * rstd2, x[c][u][4 + e] = (x[c][u][4 + e] - mu2) * rstd2;
            y[e] = fmaf(x[c][u][e], g0[e], b0[e]);
            y[4 + e] = fmaf(x[c][u][4 + e], g1[e], b1[e]);
          }
          const uint4 xn2 = __builtin_bit_cast(uint4, __builtin_convertvector(y, v8bf));
          if (live) pk2p[(c * 2 + u) * 64 + lane] = xn2;
          P = mfma(fl[L_AS + (c * 2 + u) * 64], xn2, P);
        }
      rows_to_acc(x, xh);   // xhat2 in the accumulator layout (xhat3 is done with)
    }
    softmax_regs(P, vmask);
    softmax_bwd_regs(P, ds);
    uint4 q0 = pack8(ds, 0), q1 = pack8(ds, 1);
    // dxn2 = A_s^T dsim, a channel tile at a time and TWICE (2 MFMAs per tile): once for the row sums of LayerNorm2's backward, once for the
    // outputs — holding the four tiles across both passes is what used to push d1 / xhat2 into scratch memory, and every reload sat behind the
    // acknowledgement of the dh_in stores in flight (vector-memory operations complete in order)
    auto dxn2 = [&](int c) { return mfma(fl[L_AST + (c * 2 + 1) * 64], q1, mfma(fl[L_AST + (c * 2 + 0) * 64], q0, zero16())); };
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const v16f dx = dxn2(c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f g = *reinterpret_cast<const v4f *>(gb2 + 32 * c + 8 * q + 4 * hf);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float dg = dx[4 * q + m] * g[m];
          t1 += dg;
          t2 = fmaf(dg, xh[c][4 * q + m], t2);
        }
      }
    }
    t1 += xhalf(t1), t2 += xhalf(t2);
    t1 *= (1.0f / C), t2 *= (1.0f / C);
    // (opaque to the optimiser: the second pass must not be merged with the first)
    asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const v16f dx = dxn2(c);
      v16f gx, ov;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * c + 8 * q + 4 * hf;
        const v4f g = *reinterpret_cast<const v4f *>(gb2 + ch);
        v4f o;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int r = 4 * q + m;
          gx[r] = dx[r] * xh[c][r];
          o[m] = dv[c][r] + rstd2 * (dx[r] * g[m] - t1 - xh[c][r] * t2);
        }
        if (hl_out) {
#pragma unroll
          for (int m = 0; m < 4; ++m) ov[4 * q + m] = o[m];
        } else if (live) {
          *reinterpret_cast<v4f *>(a.dh_in + rowbase + m_dhin.a(c, q)) = o;
        }
      }
      if (hl_out) store_hl(reinterpret_cast<uint4 *>(a.dh_in + rowbase) + lane, c, ov, live);
      colsum(gx, 3, c);
      colsum(dx, 4, c);
      colsum(att_drop(dv[c], c), 5, c);   // d b_o
    }
  }
  FFT(7);
  __syncthreads();
  for (int i = 2 * C + threadIdx.x; i < NQ * C; i += NW * 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += cred[(w * NQ + i / C) * C + i % C];
    a.cpart[(size_t)blockIdx.x * NQ * C + i] = t;
  }
#ifdef DFX_TRACE_FF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  FFT(8);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The forward of one block, round 6 (FWD_F16): the sampling kernel's slot structure on the training kernel's rows.
//   * records are SKEWED: record j = [W1a | W1g of hidden chunk j | W2 of chunk j - 1] (k_ff_pack), so an iteration is ONE uninterrupted burst of 24 MFMAs
//     (GEMM2 of chunk j - 1, GEMM1 of chunk j) behind ONE stretch of VALU (the GEGLU of chunk j - 1): 17 iterations, the first without GEMM2, the last
//     without GEMM1;
//   * the burst's first eight A fragments are requested at the TOP of the iteration, in front of the GEGLU arithmetic, which hides their LDS round trip
//     (the unskewed loop paid it in front of every GEMM1: ~15 % of a chunk); the rest go through the ring of eight registers, refilled in place;
//   * the GEGLU is the packed-fp16 polynomial (geglu16_f16) and GEMM2 an fp16 product; b1 of chunk j is fetched underneath it (the accumulators
//     are dead once converted);
//   * in the chain, the NEXT block's tables, attention fragments and first two records are requested behind this block's last barrier, in front of
//     its row stores (`next` / `has_next` / `staged`): the next block's first wait finds most of them landed.
// Same rows, same tile layouts, same dropout bits as ff_run<false>; the attention sub-block and LayerNorm3 are its code.
template <bool DROP>
__device__ __forceinline__ void ff_fwd(const FfArgs &a, const FfArgs &next, const bool has_next, const bool first, const bool staged, v16f (&hc)[4], float (&tv)[3],
                                       int &trace_n) {
  constexpr int NW = NW_FWD, NREC = NCHUNK + 1;
  static_assert(NW == 4 && B1P_FLOATS * 4 == NW * 1024, "table staging below assumes 256 threads");
  constexpr int BUF_BYTES = FWD_TILES * 2048, PIECES = FWD_TILES * 2 / NW;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ff_smem[];
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));   // (opaque per call: see ff_run)
  const int lane = lane_, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ff_smem);
  const unsigned voff = lane * 16;
  const int tps = a.N / 32, gps = (tps + NW - 1) / NW;
  const int s = __builtin_amdgcn_readfirstlane((int)blockIdx.x / gps);
  int ti = ((int)blockIdx.x - s * gps) * NW + wave;
  const bool live = ti < tps;
  if (!live) ti = tps - 1;
  const long long rowbase = ((long long)s * a.N + ti * 32) * C;
  const RowMap m_hin(a.tiled & TL_HIN, lane, pj, hf), m_h1(a.tiled & TL_H1, lane, pj, hf), m_h2(a.tiled & TL_H2, lane, pj, hf);
  const bool h1f = a.tiled & TL_H1_FRAG;
  FFT_INIT_CHAIN(a, 0, !first ? trace_n : 0);
  FFT(1);
  // what a block's prologue requests through LDS-DMA: b1 table, [A_s | M_s] of the shape -> slot 2, records 0 and 1 -> slots 0 and 1
  auto stage_prologue = [&](const FfArgs &b) {
    dma1k(reinterpret_cast<const char *>(b.b1p) + wave * 1024, voff, lds0 + TAB_B1 + wave * 1024);
    const char *src = reinterpret_cast<const char *>(b.at_frags + (size_t)s * SHAPE_U4);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma1k(src + (k * NW + wave) * 1024, voff, lds0 + 2 * BUF_BYTES + (k * NW + wave) * 1024);
    stage_record_fwd(b.frags, 0, lds0, wave, voff);
    stage_record_fwd(b.frags, 1, lds0 + BUF_BYTES, wave, voff);
  };
  v8f x[4][2];
  if (first) load_rows(a.hin + rowbase, m_hin, x);
  const unsigned long long prow = (unsigned long long)(rowbase / C) + pj;
  unsigned *dmk = DROP ? a.dmask + (size_t)(rowbase / (32 * C)) * DM_TILE + lane : nullptr;
  float *b1s = reinterpret_cast<float *>(ff_smem + TAB_B1);
  float *dump = reinterpret_cast<float *>(ff_smem + TAB_GB3);
  float *gb2 = reinterpret_cast<float *>(ff_smem + TAB_GB2);
  float *b2s = reinterpret_cast<float *>(ff_smem + TAB_B2);
  const int tc = threadIdx.x & (C - 1);
  const bool lo_half = wave < NW / 2;
  // tables: gamma2 | beta2 of the thread's channel, b_o (waves 0, 1) or b2 (waves 2, 3) — in `tv`, fetched here or by the previous block's tail
  auto load_tables = [&](const FfArgs &b) {
    const float *tp2 = lo_half ? b.bo : b.b2p;
    tv[0] = b.g2[tc], tv[1] = b.b2n[tc], tv[2] = tp2[tc];
  };
  if (!staged) {
    load_tables(a);
    stage_prologue(a);
  }
  unsigned vmask = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) vmask |= (a.valid[s * 4 + j] != 0.f ? 1u : 0u) << j;
  // ONE round trip (first block: rows, tables, attention fragments, two records; behind a previous block the requests were made in front of its row stores and
  // have mostly landed).  vmcnt(0), not a count that leaves the sixteen row stores in flight: loads and stores do not complete in order against each other
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2])::"memory");
  {
    float *d0 = lo_half ? gb2 : dump, *d2 = lo_half ? gb2 + 2 * C : b2s;
    d0[tc] = tv[0], d0[C + tc] = tv[1], d2[tc] = tv[2];
  }
  __syncthreads();
  FFT(16);
  v16f acc[4];
  if (!first) acc_to_rows(hc, x);
  uint4 xn[4][2];
  float mu, rstd;
  {
    // ---- attention sub-block in registers (ff_run<false>'s code): h1 = hin + [dropout] (M_s softmax(A_s LN2(hin)) + b_o) ----
    const uint4 *fl = reinterpret_cast<const uint4 *>(ff_smem + 2 * BUF_BYTES) + lane;
    v16f sim = zero16();
    {
      float mu2, rstd2;
      ln_stats(x, mu2, rstd2);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) sim = mfma(fl[F_AS * SET_U4 + (c * 2 + u) * 64], ln_frag(x[c][u], c, u, hf, gb2, mu2, rstd2), sim);
    }
    softmax_regs(sim, vmask);
    const uint4 p0 = pack8(sim, 0), p1 = pack8(sim, 1);
    rows_to_acc(x, acc);
    unsigned attw = 0;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      if (DROP) {
        v16f t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f b = *reinterpret_cast<const v4f *>(gb2 + 2 * C + 32 * ct + 8 * q + 4 * hf);
#pragma unroll
          for (int m = 0; m < 4; ++m) t[4 * q + m] = b[m];
        }
        t = mfma(fl[F_MS * SET_U4 + (ct * 2 + 0) * 64], p0, t);
        t = mfma(fl[F_MS * SET_U4 + (ct * 2 + 1) * 64], p1, t);
        unsigned w[4][2];
        drop_words(a.dk, a.site_att, prow * (C / 8) + 4 * ct, hf, w);
        const unsigned bits = drop_apply(t, w, a.dk.thr);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = fmaf(a.dk.keep, t[r], acc[ct][r]);
        attw |= bits << (16 * (ct & 1));
        if (ct & 1) {
          if (live) dmk[(8 + (ct >> 1)) * 64] = attw;
          attw = 0;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f b = *reinterpret_cast<const v4f *>(gb2 + 2 * C + 32 * ct + 8 * q + 4 * hf);
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[ct][4 * q + m] += b[m];
        }
        acc[ct] = mfma(fl[F_MS * SET_U4 + (ct * 2 + 0) * 64], p0, acc[ct]);
        acc[ct] = mfma(fl[F_MS * SET_U4 + (ct * 2 + 1) * 64], p1, acc[ct]);
      }
      if (live && !h1f) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<v4f *>(a.h1_out + rowbase + m_h1.a(ct, q)) = v4f{acc[ct][4 * q], acc[ct][4 * q + 1], acc[ct][4 * q + 2], acc[ct][4 * q + 3]};
      }
    }
    acc_to_rows(acc, x);
    FFT(17);
    __syncthreads();   // everybody is done with the attention fragments: slot 2 takes record 2 at the top of the loop
    FFT(18);
  }
  ln_stats(x, mu, rstd);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) xn[c][u] = xhat_frag(x[c][u], mu, rstd);
  if (h1f && live) {
    uint4 *hp = reinterpret_cast<uint4 *>(a.h1_out + rowbase) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) hp[(c * 2 + 0) * 64] = xn[c][0], hp[(c * 2 + 1) * 64] = xn[c][1];
    if (hf == 0) (a.h1_out + rowbase)[H1F_RSTD + pj] = rstd;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v16f b2;
    load16(b2, b2s + hf * 64 + c * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] += b2[r];
  }
  FFT(2);

  // ---- the skewed loop: iteration j = [top: record j + 2 requested, first eight fragments of the burst requested] [V: GEGLU of chunk j - 1, b1 of
  // chunk j] [M: GEMM2 of chunk j - 1 (8 MFMAs), GEMM1 of chunk j (16)] [record j + 1 landed, barrier] ----
  v16f av, gv;
  unsigned dmw = 0;
  auto step = [&](const int j, auto s2_, auto s1_, auto stage_) {
    constexpr bool S2 = decltype(s2_)::value, S1 = decltype(s1_)::value, STAGE = decltype(stage_)::value;   // STAGE: record j + 2 exists
    constexpr int NM = (S2 ? 8 : 0) + (S1 ? 16 : 0);
    // this wave's six pieces of record j + 2 (-> the slot of record j - 1: everybody is past it) are issued from INSIDE the burst, one behind every fourth
    // MFMA, where issue slots are free — six back-to-back LDS-DMA instructions at the top cost the wave ~100 cycles each (MI355X_MICROARCH.md)
    const char *dsrc = reinterpret_cast<const char *>(a.frags + (size_t)(j + 2) * CHUNK_U4) + FWD_TILE0 * 2048 + wave * 1024;
    const unsigned ddst = lds0 + ((j + 2) % NBUF) * BUF_BYTES + wave * 1024;
    const uint4 *fr = reinterpret_cast<const uint4 *>(ff_smem + (j % NBUF) * BUF_BYTES) + lane;
    // A fragment of MFMA m of the burst: GEMM2 (row tile m & 3, unit m >> 2) first, then GEMM1 MFMA e (k-tile e >> 2, unit (e >> 1) & 1, a / g = e & 1)
    auto fm = [&](int m) -> uint4 {
      if (S2 && m < 8) return fr[((T_W2 + (m & 3)) * 2 + (m >> 2)) * 64];
      const int e = m - (S2 ? 8 : 0);
      return fr[((((e & 1) ? T_W1G : T_W1A) + (e >> 2)) * 2 + ((e >> 1) & 1)) * 64];
    };
    uint4 P[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) P[i] = fm(i);
    __builtin_amdgcn_sched_barrier(0);
    uint4 hh[2];
    h2 aa[8], gg[8];
    if (DFX_FF_VPRIO_FWD) __builtin_amdgcn_s_setprio(DFX_FF_VPRIO_FWD);
    if (S2) {
      if (DROP) {   // dropout behind the GEGLU of chunk j - 1 (attention.py:84), selected on `a`: element (row, unit 32 (j - 1) + 8 q + 4 hf + m) = group row * 64 + 4 (j - 1) + q
        const int jj = j - 1;
        unsigned w[4][2];
        drop_words(a.dk, a.site_ff, prow * (FH / 8) + 4 * jj, hf, w);
        const unsigned bits = drop_apply(av, w, a.dk.thr);
        dmw = (jj & 1) ? dmw | (bits << 16) : bits;
        if ((jj & 1) && live) dmk[(jj >> 1) * 64] = dmw;
      }
      geglu16_f16_cvt(av, gv, aa, gg);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (S1) {
      load16(av, b1s + ((j * 2 + 0) * 2 + hf) * 16);
      load16(gv, b1s + ((j * 2 + 1) * 2 + hf) * 16);
    }
    if (S2) {   // (pinned here: LLVM otherwise sinks the arithmetic into the burst)
      geglu16_f16_math(aa, gg, hh);
      asm volatile("" : "+v"(hh[0].x), "+v"(hh[0].y), "+v"(hh[0].z), "+v"(hh[0].w), "+v"(hh[1].x), "+v"(hh[1].y), "+v"(hh[1].z), "+v"(hh[1].w));
    }
    if (DFX_FF_VPRIO_FWD) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (S2 && m < 8) {
        acc[m & 3] = mfma_f16(P[m & 7], hh[m >> 2], acc[m & 3]);
      } else {
        const int e = m - (S2 ? 8 : 0);
        if (e & 1) gv = mfma(P[m & 7], xn[e >> 2][(e >> 1) & 1], gv);
        else av = mfma(P[m & 7], xn[e >> 2][(e >> 1) & 1], av);
      }
      if (STAGE) {
#pragma unroll
        for (int k = 0; k < PIECES; ++k)
          if (m == (k * NM) / PIECES) dma1k(dsrc + k * NW * 1024, voff, ddst + k * NW * 1024);
      }
      if (m + 8 < NM) P[m & 7] = fm(m + 8);
    }
#pragma unroll
    for (int m = 0; m < NM - 8; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    FFT(13);
    if (STAGE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");   // record j + 1 has landed; record j + 2's pieces may be out
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FFT(14);
    __syncthreads();
    FFT(15);
  };
  static_assert(NREC == NCHUNK + 1, "records 0 .. NCHUNK");
  step(0, std::false_type{}, std::true_type{}, std::true_type{});
#pragma unroll 1
  for (int j = 1; j < NCHUNK - 1; ++j) step(j, std::true_type{}, std::true_type{}, std::true_type{});
  step(NCHUNK - 1, std::true_type{}, std::true_type{}, std::false_type{});
  step(NCHUNK, std::true_type{}, std::false_type{}, std::false_type{});
  FFT(3);
  // the ring and the tables are idle from here: the next block's prologue requests travel under this block's row stores
  if (has_next) {
    load_tables(next);
    stage_prologue(next);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) hc[c] = acc[c];
  if (a.tiled & TL_HEAD) {
    // post_norm + proj_out on the accumulators (ff_run<false>'s code)
    float sm = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) sm += acc[c][r];
    sm += xhalf(sm);
    const float mup = sm * (1.0f / C);
    float qs = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[c][r] - mup;
        qs = fmaf(d, d, qs);
      }
    qs += xhalf(qs);
    const float rstdp = 1.0f / sqrtf(qs * (1.0f / C) + LN_EPS);
    float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * c + 8 * q + 4 * hf;
        const v4f w0 = *reinterpret_cast<const v4f *>(a.head_tab + ch), w1 = *reinterpret_cast<const v4f *>(a.head_tab + C + ch);
        const v4f w2 = *reinterpret_cast<const v4f *>(a.head_tab + 2 * C + ch);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float xh = (acc[c][4 * q + m] - mup) * rstdp;
          acc[c][4 * q + m] = xh;
          e0 = fmaf(xh, w0[m], e0), e1 = fmaf(xh, w1[m], e1), e2 = fmaf(xh, w2[m], e2);
        }
      }
    e0 += xhalf(e0), e1 += xhalf(e1), e2 += xhalf(e2);
    if (live) {
      if (hf == 0) {
        float *ep = a.eps + (size_t)s * 3 * a.N + ti * 32 + pj;
        ep[0] = e0 + a.head_tab[3 * C], ep[a.N] = e1 + a.head_tab[3 * C + 1], ep[2 * (size_t)a.N] = e2 + a.head_tab[3 * C + 2];
        (a.h2 + rowbase)[H1F_RSTD + pj] = rstdp;
      }
      uint4 *hp = reinterpret_cast<uint4 *>(a.h2 + rowbase) + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint4 hb[2];
        acc_to_frags(acc[c], hb);
        hp[(c * 2 + 0) * 64] = hb[0], hp[(c * 2 + 1) * 64] = hb[1];
      }
    }
    FFT(9);
    FFT_COUNT(trace_n);
    return;
  }
  if (live) {
    float *out = a.h2 + rowbase;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<v4f *>(out + m_h2.a(c, q)) = v4f{acc[c][4 * q], acc[c][4 * q + 1], acc[c][4 * q + 2], acc[c][4 * q + 3]};
  }
  FFT(9);
  FFT_COUNT(trace_n);
}

template <bool BWD, bool DROP>
__global__ __launch_bounds__(nw_of<BWD>() * 64, 2) void k_ff(FfArgs a) {
  v16f hc[4];
  int tn = 0;
  if constexpr (!BWD && FWD_F16) {
    if (a.at_frags) {   // (the attention sub-block inside: every shipped launch; dfx_debug_train_fused(2) keeps the round-5 body)
      float tv[3];
      ff_fwd<DROP>(a, a, false, true, false, hc, tv, tn);
      return;
    }
  }
  ff_run<BWD, DROP>(a, true, hc, tn);
}
// The forward of ALL blocks in one launch (round 5): a point's way through the network depends on no other point (the attention's keys / values are the
// four context tokens, folded into A_s / M_s), so a wavefront takes its 32 points through block after block with the residual stream in its accumulator
// registers — as the sampling kernel does.  Every block still WRITES its output rows (the backward recomputes LayerNorm2 from them), but nobody reads them
// back in the forward: 512 B per point and block less traffic, and the prologue's wait for the rows (a quarter of a single-block workgroup's life) is
// paid once.  Same arithmetic on the same values as five launches of k_ff<false>: bit-identical.
struct FfChain {
  FfArgs blk[DFX_MAX_DEPTH];
  int n;
};
template <bool DROP>
__global__ __launch_bounds__(NW_FWD * 64, 2) void k_ff_fwd_chain(FfChain ch) {
  v16f hc[4];
  int tn = 0;   // (trace builds only)
  // (Round 6, measured and dropped: the second workgroup of a CU — HW_REG_LDS_ALLOC base != 0 — waiting 16 / 24 / 32 k cycles once at the start, so that the two
  // co-resident workgroups' VALU-only stretches do not coincide: 661 / 672 / 672 us against 646-651 — they are not in step to begin with, the wait is pure cost)
  if constexpr (FWD_F16) {
    float tv[3];
    for (int b = 0; b < ch.n; ++b) ff_fwd<DROP>(ch.blk[b], ch.blk[b + 1 < ch.n ? b + 1 : b], b + 1 < ch.n, b == 0, b > 0, hc, tv, tn);
  } else {
    for (int b = 0; b < ch.n; ++b) ff_run<false, DROP>(ch.blk[b], b == 0, hc, tn);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradients of the feed-forward, weight-stationary:  dW1 = d[a | g]^T xn3 (1024 x 128),  dW2 = dh^T hid (128 x 512),
// d b1 = column sums of d[a | g].  A 32-unit chunk of the hidden layer for a slab of rows belongs to TWO wavefronts on one SIMD: the
// producer keeps the chunk's W1a / W1g / W2 fragments (96 registers), the consumer its twelve 32 x 32 gradient tiles (192 accumulator
// registers), while the slab's 32-point tiles stream by (16 KiB each, written by k_ff<true>, L2 -> LDS with LDS-DMA, shared by the
// workgroup's four chunks).  Everything with the chunk's units on the lanes and the 32 points along the registers (the transposed
// orientation: activations as the A operand, weights as B):
//     producer   a^T, g^T = xn3 W1^T + b1 (16 MFMAs)   d hid^T = dh W2 (8)   GEGLU forward / backward on the registers -> 6 fragments in LDS
//     consumer   xn3^T, dh^T of the tile by selection-matrix MFMAs (4 per consumer)   dW2^T chunk += hid^T dh (8)   dW1a += da^T xn3 (8)
//                dW1g += dg^T xn3 (8)            K = the 32 points
// Partials per (slab, chunk) are summed by k_ff_wgrad_finish in slab order.
struct FwArgs {
  const uint4 *frags;    // the k_ff_pack fragments: tiles T_W1A, T_W1G, T_W2T of each chunk are this kernel's B operands
  const float *b1;       // (1024) the FOLDED bias PackArgs::b1f (b1 + W1 beta3, `a` half times keep_a)
  const uint4 *pk;       // [R / 32][2][4][2][64]: set PK_XN = the xhat3 fragments (k_ff<true>'s pk, or the forward's h1 tiles, TL_H1_FRAG: same place in the tile)
  float *part;           // [nslab][NCHUNK][12][16][64] fp32 gradient tiles in accumulator layout
  float *bpart;          // [nslab][NCHUNK][2][32] column sums of da, dg
  long long ntiles;      // R / 32
  int nslab;
  const unsigned *dmask; // k_ff_wgrad<true>: the forward's dropout bits (FfArgs::dmask); words 0 .. 7 of a tile (2 KiB) travel with the tile
  const uint4 *dhf;      // the dh fragments: tiles of 16 KiB whose FIRST 8 KiB they are — the block's incoming gradient as bf16-pair tiles (TL_DH_HL: the hi
                         // halves), or k_ff<true>'s pk + the offset of its PK_DH set
#ifdef DFX_TRACE_FF
  unsigned long long *trace;
#endif
};
constexpr int WG_CHUNKS = 4, WG_NW = 2 * WG_CHUNKS;   // two wavefronts per chunk, on the same SIMD
// LDS: the tiles ([xn3 | dh] fragments, 16 KiB) travel in a ring of three slots, two tiles ahead of the producers (L2 latency under load
// is longer than one tile's arithmetic); the consumers turn tile k around (two MFMAs with a 0/1 selection matrix per 32 x 32 tile, exact)
// into one of two 16 KiB slots while they multiply tile k - 1
// Round 6 (WG_TR, measured and left OFF): the consumers read the turned operands straight out of the ring's tile with the LDS transpose read (ds_read_b64_tr_b16,
// as k_ff<true>'s second product does; the tile's 1 KiB fragments land w1_slot-swizzled — the LDS-DMA lanes fetch each other's 16 bytes — so that the reads are
// conflict-free): no selection MFMAs, no packs and LDS stores of a turned copy, no turned slots; the ring has four slots (tile k - 1 is still being read while
// tile k + 2 lands).  Bit-identical, 0 bank conflicts — and 2.3 % SLOWER (alternating same-box runs 225.0 / 226.0 against 230.0 / 231.7 us per block,
// profiles/r06_ab_train_wgrad_tr.txt): the turning is shared by the four consumers (each turns a quarter, all read 16 whole fragments back with ds_read_b128),
// the transpose read doubles every consumer's read instructions in front of its MFMAs, and the producer, not the consumer, is this kernel's critical wavefront.
#ifndef DFX_WG_TR
#define DFX_WG_TR 0
#endif
#ifndef DFX_WG_CSTAGE
#define DFX_WG_CSTAGE 1   // the consumers issue the ring's LDS-DMA pieces (see stage)
#endif
constexpr bool WG_CSTAGE = DFX_WG_CSTAGE != 0;
constexpr bool WG_TR = DFX_WG_TR != 0;
constexpr int WG_SLOTS = WG_TR ? 4 : 3, WG_AHEAD = 2;
constexpr int WG_RING_A = 0, WG_RING_T = WG_SLOTS * 16384, WG_RING = (WG_SLOTS + (WG_TR ? 0 : 2)) * 16384, WG_PACKS = 2 * WG_CHUNKS * 6 * 1024, WG_LDS = WG_RING + WG_PACKS;
constexpr int WG_MASK = WG_LDS, WG_LDS_DROP = WG_LDS + WG_SLOTS * 2048;   // dropout: the tiles' bit words (2 KiB each) in a ring of their own, same slots
template <bool DROP>
__global__ __launch_bounds__(WG_NW * 64, 2) void k_ff_wgrad(FwArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char fw_smem[];   // 3 tiles x 32 KiB | 2 x 4 chunks x 6 KiB of hid / da / dg fragments
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pj = lane & 31;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)fw_smem);
  const unsigned voff = lane * 16;
#ifdef DFX_WG_PAIR_ADJACENT
  const int cl = wave >> 1;
  const bool consumer = wave & 1;
#else
  const int cl = wave & (WG_CHUNKS - 1);
  const bool consumer = wave >= WG_CHUNKS;
#endif
  // workgroup ids are dealt round-robin over the 8 XCDs: with the slab index in the low bits the four workgroups that stream the same
  // slab (one per chunk group) share an XCD, i.e. one L2 — the slab leaves HBM once instead of four times (nslab is a multiple of 8
  // for all but tiny inputs)
  const int cg = blockIdx.x / a.nslab, slab = blockIdx.x % a.nslab, j = cg * WG_CHUNKS + cl;
  const long long per = (a.ntiles + a.nslab - 1) / a.nslab, t0 = (long long)slab * per, t1 = t0 + per < a.ntiles ? t0 + per : a.ntiles;
  const int nt = t1 > t0 ? (int)(t1 - t0) : 0;
  // iteration k requests tile k + 2: two 1 KiB pieces per wavefront
  constexpr int AHEAD = WG_AHEAD;
  const unsigned voffq[2] = {WG_TR ? w1_slot(lane, 0) * 16u : voff, WG_TR ? w1_slot(lane, 1) * 16u : voff};   // (piece 2 wave + q is a fragment of unit q)
  auto stage = [&](int k) {
    if (WG_CSTAGE) {
      // the CONSUMERS request the whole tile (consumer cl: pieces cl, cl + 4, cl + 8, cl + 12; with dropout consumers 0 and 1 one KiB of the bit words each):
      // the producers are this kernel's critical wavefronts and keep their issue slots for the MFMAs and the GEGLU arithmetic
      if (consumer && k + AHEAD < nt) {
        const char *pk = reinterpret_cast<const char *>(a.pk + (size_t)(t0 + k + AHEAD) * PK_TILE_U4);
        const char *dh = reinterpret_cast<const char *>(a.dhf + (size_t)(t0 + k + AHEAD) * PK_TILE_U4) - 8192;
        const unsigned slot = __builtin_amdgcn_readfirstlane((unsigned)((k + AHEAD) % WG_SLOTS));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int p = cl + 4 * q;
          dma1k((q < 2 ? pk : dh) + p * 1024, voffq[p & 1], lds0 + WG_RING_A + slot * 16384 + p * 1024);
        }
        if (DROP && cl < 2)
          dma1k(reinterpret_cast<const char *>(a.dmask + (size_t)(t0 + k + AHEAD) * DM_TILE) + cl * 1024, voff, lds0 + WG_MASK + slot * 2048 + cl * 1024);
      }
      return;
    }
    if (k + AHEAD < nt) {
      // pieces 0 .. 7 (waves 0 .. 3): the xhat3 fragments of pk; 8 .. 15: the dh fragments — pk's second set, or the hi half of the gradient tile itself
      const char *src = wave >= WG_NW / 2 ? reinterpret_cast<const char *>(a.dhf + (size_t)(t0 + k + AHEAD) * PK_TILE_U4) - 8192
                                          : reinterpret_cast<const char *>(a.pk + (size_t)(t0 + k + AHEAD) * PK_TILE_U4);
      const unsigned slot = __builtin_amdgcn_readfirstlane((unsigned)((k + AHEAD) % WG_SLOTS));   // (wave-uniform: the LDS address goes through m0)
#pragma unroll
      for (int q = 0; q < 2; ++q) dma1k(src + (wave * 2 + q) * 1024, voffq[q], lds0 + WG_RING_A + slot * 16384 + (wave * 2 + q) * 1024);
      if (DROP)   // + this tile's eight feed-forward bit words: 256 B per wavefront
        dma256(reinterpret_cast<const char *>(a.dmask + (size_t)(t0 + k + AHEAD) * DM_TILE) + wave * 256, lane * 4,
               lds0 + WG_MASK + slot * 2048 + wave * 256);
    }
  };
  // top of iteration k: everything requested before iteration k - 1 has landed, i.e. tiles <= k (loads complete in order; the last iterations
  // request nothing: drain)
  auto arrive = [&](int k) {
    if (WG_CSTAGE) {   // (only the consumers have pieces in flight: the newest tile's four — five with a KiB of bit words — may stay out)
      if (consumer) {
        if (k + AHEAD - 1 >= nt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DROP && cl < 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      __syncthreads();
      return;
    }
    if (k + AHEAD - 1 < nt) {
      if (DROP) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  uint4 *packs = reinterpret_cast<uint4 *>(fw_smem + WG_RING);
  // The two wavefronts of a chunk work one tile apart (one barrier per tile): the producer turns tile k into the chunk's hid / da / dg
  // fragments (24 MFMAs + the GEGLU arithmetic, weights in registers), the consumer multiplies tile k - 1's fragments with the
  // transposed tile (24 MFMAs into its 192 accumulator registers) — so the matrix pipe of the SIMD works through the producer's VALU
  // stretch, and neither role needs more than 256 registers.
  FFT_INIT2(a, consumer ? 3 : 2, consumer ? WG_CHUNKS * 64 : 0);
  FFT(1);
  for (int k = -AHEAD; k < 0; ++k) stage(k);   // tiles 0, 1
  if (!consumer) {
    uint4 w1a[4][2], w1g[4][2], w2t[4][2];
    {
      const uint4 *fr = a.frags + (size_t)j * CHUNK_U4 + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          w1a[c][u] = (fr - lane + w1_slot(lane, u))[((T_W1A + c) * 2 + u) * 64];
          w1g[c][u] = (fr - lane + w1_slot(lane, u))[((T_W1G + c) * 2 + u) * 64];
          w2t[c][u] = fr[((T_W2T + c) * 2 + u) * 64];
        }
    }
    const float ba = a.b1[32 * j + pj], bg = a.b1[FH + 32 * j + pj];
    float sa = 0.f, sg = 0.f;
    // dropout bits in this orientation (unit pj of chunk j on the lane, point 8 q + 4 half + m in register 4 q + m): the forward's lane of that point is
    // (point, (pj >> 2) & 1), its bit 16 (j & 1) + 4 (pj >> 3) + (pj & 3) of word j >> 1 — four consecutive lanes' words per 16-byte LDS read
    const int dm_off = (j >> 1) * 64 + 32 * ((pj >> 2) & 1) + 4 * (lane >> 5), dm_bit = 16 * (j & 1) + 4 * (pj >> 3) + (pj & 3);
    // 24 MFMAs of tile k: [a | g]^T = xhat3 W1^T + b1, d hid^T = dh W2
    auto mm = [&](int k, v16f &av, v16f &gv, v16f &dv) {
      const uint4 *tb = reinterpret_cast<const uint4 *>(fw_smem + WG_RING_A + (k % WG_SLOTS) * 16384);
      const uint4 *tlu[2] = {tb + (WG_TR ? w1_slot(lane, 0) : lane), tb + (WG_TR ? w1_slot(lane, 1) : lane)};
#pragma unroll
      for (int r = 0; r < 16; ++r) av[r] = ba, gv[r] = bg, dv[r] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint4 x = tlu[u][(PK_XN * 8 + c * 2 + u) * 64];
          av = mfma(x, w1a[c][u], av);
          gv = mfma(x, w1g[c][u], gv);
          dv = mfma(tlu[u][(PK_DH * 8 + c * 2 + u) * 64], w2t[c][u], dv);
        }
    };
    // GEGLU forward / backward on tile k's accumulators -> six fragments in LDS
    auto act = [&](int k, const v16f &av, const v16f &gv, v16f &dv) {
      unsigned keepm[16];
      if (DROP) {
        const unsigned *mk = reinterpret_cast<const unsigned *>(fw_smem + WG_MASK + (k % WG_SLOTS) * 2048) + dm_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 mq = *reinterpret_cast<const uint4 *>(mk + 8 * q);
          const unsigned wq[4] = {mq.x, mq.y, mq.z, mq.w};
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            keepm[4 * q + m] = (unsigned)__builtin_amdgcn_sbfe((int)wq[m], dm_bit, 1);   // v_bfe_i32: all ones = kept
            const float dvr = dv[4 * q + m];   // (copies: a bit_cast applied to a vector-element expression reads element 0)
            dv[4 * q + m] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, dvr) & keepm[4 * q + m]);
          }
        }
      }
      uint4 *po = packs + ((k & 1) * WG_CHUNKS + cl) * 6 * 64 + lane;
#pragma unroll
      for (int u = 0; u < 2; ++u) {   // (half a tile at a time: eight values of each of the three products alive, not sixteen)
        v8f hv, da, dg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // (scalar fp32 here: the packed forms of gelu_fd2 made this kernel 5 % SLOWER — 236 -> 249 us)
          const int r = 8 * u + e;
          float f, d;
#ifdef DFX_ABL_WG_ACT
          f = gv[r], d = 1.0f;
#else
          gelu_fd(gv[r], f, d);
#endif
          const float hr = av[r] * f;
          hv[e] = DROP ? __builtin_bit_cast(float, __builtin_bit_cast(unsigned, hr) & keepm[r]) : hr;
          da[e] = dv[r] * f;
          dg[e] = dv[r] * av[r] * d;
          sa += da[e], sg += dg[e];
        }
        po[(0 + u) * 64] = __builtin_bit_cast(uint4, __builtin_convertvector(hv, v8bf));
        po[(2 + u) * 64] = __builtin_bit_cast(uint4, __builtin_convertvector(da, v8bf));
        po[(4 + u) * 64] = __builtin_bit_cast(uint4, __builtin_convertvector(dg, v8bf));
      }
    };
    // (Round 5, measured and dropped: running the 24 MFMAs — or just GEMM1's 16 — of tile k + 1 beside the GEGLU arithmetic of tile k in the SAME wavefront,
    // two accumulator sets, four ring slots: 230 -> 249 us per block left to hipcc's order, 292-297 us with a forced MFMA : VALU interleave — VALU
    // beside MFMAs of one wavefront is an anti-lever on this part; the producer / consumer split across two wavefronts is what overlaps them.)
    {
      for (int k = 0; k <= nt; ++k) {
        if (k < 24) FFT(10);
        arrive(k);   // tile k has landed; everybody is done with tile k - 1's producer half and tile k - 2's consumer half and fragments
        if (k < 24) FFT(11);
        stage(k);
        if (k == nt) break;
        v16f av, gv, dv;
        mm(k, av, gv, dv);
        if (k < 12) FFT(14);
        if (DFX_WG_PPRIO) __builtin_amdgcn_s_setprio(DFX_WG_PPRIO);
        act(k, av, gv, dv);
        if (DFX_WG_PPRIO) __builtin_amdgcn_s_setprio(0);
        if (k < 12) FFT(13);
      }
    }
    FFT(3);
    sa += xhalf(sa), sg += xhalf(sg);   // the two half-waves hold different points of the same unit
    if (lane < 32) {
      float *bo = a.bpart + ((size_t)slab * NCHUNK + j) * 64;
      bo[pj] = sa, bo[32 + pj] = sg;
    }
    return;
  }
  v16f dW2[4], dWa[4], dWg[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) dW2[c][r] = 0.f, dWa[c][r] = 0.f, dWg[c][r] = 0.f;
  uint4 sel[2];   // selection matrices: B[k][n] = (n == 16 u + k), this lane's eight k = 8 hf .. 8 hf + 7 of column n = lane & 31
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    __bf16 o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(pj == 16 * u + 8 * (lane >> 5) + e ? 1.0f : 0.0f);
    sel[u] = *reinterpret_cast<const uint4 *>(o);
  }
  const int ctu = (lane >> 4) & 1, ctrow = 4 * (lane >> 5) + ((lane & 15) >> 2) + 32 * ((lane & 3) >> 1);
  const unsigned ctr0 = (ctu * 64 + w1_slot(ctrow, ctu)) * 16 + 8 * (lane & 1), ctr1 = (ctu * 64 + w1_slot(ctrow + 8, ctu)) * 16 + 8 * (lane & 1);
  for (int k = 0; k <= nt; ++k) {
    if (k < 12) FFT(20);
    arrive(k);
    if (k < 12) FFT(21);
    stage(k);
    // (Round 6, measured and dropped: issuing the selection MFMAs first, then tile k - 1's fragment reads and multiplications one gradient at a time, and the
    // turned tile's four LDS stores last — so that the reads do not queue behind the read -> MFMA -> MFMA -> pack -> store chain — keeps 32 more registers
    // alive beside the 192 accumulators and reads xn3^T twice: 243 against 231 us per block, profiles/r06_ab_train_wgrad_reorder.txt)
    if (!WG_TR && k < nt) {   // tile k turned around for the next iteration: consumer cl takes channel tile cl of xn3 and of dh
      const uint4 *tl = reinterpret_cast<const uint4 *>(fw_smem + WG_RING_A + (k % WG_SLOTS) * 16384) + lane;
      uint4 *to = reinterpret_cast<uint4 *>(fw_smem + WG_RING_T + (k & 1) * 16384) + lane;
      v16f z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
      for (int kind = 0; kind < 2; ++kind) {
        const v16f t = mfma(tl[(kind * 8 + cl * 2 + 1) * 64], sel[1], mfma(tl[(kind * 8 + cl * 2 + 0) * 64], sel[0], z));
        to[(kind * 8 + cl * 2 + 0) * 64] = pack8(t, 0), to[(kind * 8 + cl * 2 + 1) * 64] = pack8(t, 1);
      }
    }
    if (k < 12) FFT(22);
    if (k == 0) continue;
    const uint4 *tl = reinterpret_cast<const uint4 *>(fw_smem + WG_RING_T + ((k - 1) & 1) * 16384) - 16 * 64 + lane;   // (index the slot by PK_XNT, PK_DHT)
    const uint4 *pi = packs + (((k - 1) & 1) * WG_CHUNKS + cl) * 6 * 64 + lane;
    const uint4 h0 = pi[0 * 64], h1 = pi[1 * 64], a0 = pi[2 * 64], a1 = pi[3 * 64], g0 = pi[4 * 64], g1 = pi[5 * 64];
    // WG_TR: operand (kind, channel tile c, K unit uk) = channels on the lanes, points k_reg(uk, hf, e) along the registers, turned out of tile k - 1's own
    // fragments: lane 16 G + s supplies point row 16 uk + 8 r + 4 (G >> 1) + (s >> 2), channels 16 (G & 1) + 4 (s & 3) .. + 3 (see k_ff<true>'s trb0 / trb1)
    const unsigned tsb = lds0 + WG_RING_A + ((k - 1) % WG_SLOTS) * 16384;
    auto turned = [&](int kind, int c, int uk) -> uint4 {
      if (!WG_TR) return tl[((kind ? PK_DHT : PK_XNT) * 8 + c * 2 + uk) * 64];
      const unsigned o = tsb + (kind * 8 + c * 2) * 1024 + uk * 256;
      return __builtin_bit_cast(uint4, __builtin_shufflevector(lds_tr16(o + ctr0), lds_tr16(o + ctr1), 0, 1, 2, 3, 4, 5, 6, 7));
    };
    if (DFX_WG_CPRIO) __builtin_amdgcn_s_setprio(DFX_WG_CPRIO);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint4 x0 = turned(0, c, 0), x1 = turned(0, c, 1);
      const uint4 d0 = turned(1, c, 0), d1 = turned(1, c, 1);
      dW2[c] = mfma(h1, d1, mfma(h0, d0, dW2[c]));
      dWa[c] = mfma(a1, x1, mfma(a0, x0, dWa[c]));
      dWg[c] = mfma(g1, x1, mfma(g0, x0, dWg[c]));
    }
    if (DFX_WG_CPRIO) __builtin_amdgcn_s_setprio(0);
    if (k < 12) FFT(23);
  }
  float *out = a.part + ((size_t)slab * NCHUNK + j) * 12 * 1024 + lane;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      out[((0 + c) * 16 + r) * 64] = dW2[c][r];
      out[((4 + c) * 16 + r) * 64] = dWa[c][r];
      out[((8 + c) * 16 + r) * 64] = dWg[c][r];
    }
}
// The slab partials of k_ff_wgrad -> the gradients (three launches for all blocks, launch_ff_wgrad_finish)
struct FwFinishArgs {
  const float *part, *bpart;
  float *dw1, *db1, *dw2;   // (1024, 128), (1024), (128, 512)
  int nslab;
  float keep_a;             // dropout: the `a` half's gradients were accumulated against keep_a a (PackArgs::keep_a): G_a, d b1a get the factor back; 1 otherwise
  const float *g3, *b3;     // LayerNorm3's affine, folded into W1 / b1 by k_ff_pack: dW1 = G diag(g3) + db1 (x) b3
  const float *w1;          // (1024, 128) fp32 master weights: d gamma3 = sum_o W1[o][.] G[o][.],  d beta3 = sum_o W1[o][.] db1[o]
  float *lnpart;            // [NCHUNK * 8 * 4 workgroups of k_ff_wgrad_finish][2][32] their partial sums of the two, summed by k_ln3_param
  float *dg3, *db3;         // (128) each
};
struct FwFinishBatch {   // one launch for all transformer blocks (blockIdx.y): block i's partials wait in block i's own buffers
  FwFinishArgs blk[DFX_MAX_DEPTH];
};
constexpr int LNPART_FLOATS = NCHUNK * 8 * 4 * 64;
// d b1 first (k_ff_wgrad_finish needs a row's value for every element of the row): workgroup = chunk, thread = (quarter of the slabs, a / g, unit);
// the quarters are summed in order
__global__ __launch_bounds__(256) void k_ff_bsum(FwFinishBatch batch) {
  const FwFinishArgs &a = batch.blk[blockIdx.y];
  __shared__ float red[4][64];
  const int j = blockIdx.x, k = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int per = (a.nslab + 3) / 4, s1 = (q + 1) * per < a.nslab ? (q + 1) * per : a.nslab;
  float t = 0.f;
#pragma unroll 8
  for (int s = q * per; s < s1; ++s) t += a.bpart[(size_t)s * NCHUNK * 64 + j * 64 + k];
  red[q][k] = t;
  __syncthreads();
  if (q == 0) {
    const int p = k >> 5, i = k & 31;
    t = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    a.db1[p * FH + 32 * j + i] = p == 0 ? t * a.keep_a : t;
  }
}
// sums the slabs in order and scatters the tiles: gradient tile (which, c) of chunk j, register r, lane (i, hf) = unit rho(r, hf) of the
// chunk, channel 32 c + i.  A workgroup = (chunk, tile, four registers): 8 rows x 32 channels
__global__ __launch_bounds__(256) void k_ff_wgrad_finish(FwFinishBatch batch) {
  const FwFinishArgs &a = batch.blk[blockIdx.y];
  __shared__ float red[2][8][32];
  const int tile = (blockIdx.x >> 2) % 12, j = blockIdx.x / 48;   // (wave-uniform)
  const int idx = blockIdx.x * 256 + threadIdx.x;                 // over NCHUNK * 12 * 1024 tile elements
  constexpr int NT = NCHUNK * 12 * 1024;
  float t = 0.f;
  for (int s = 0; s < a.nslab; ++s) t += a.part[(size_t)s * NT + idx];
  const int lane = idx & 63, r = (idx >> 6) & 15;
  const int unit = 32 * j + rho(r, lane >> 5), ch = 32 * (tile & 3) + (lane & 31);
  if (tile < 4) {
    a.dw2[(size_t)ch * FH + unit] = t;
    return;
  }
  const int row = (tile < 8 ? 0 : FH) + unit;
  const float g = tile < 8 ? t * a.keep_a : t, b = a.db1[row], w = a.w1[(size_t)row * C + ch];
  a.dw1[(size_t)row * C + ch] = fmaf(g, a.g3[ch], b * a.b3[ch]);
  const int q = threadIdx.x >> 5;   // (register, half-wave) of the workgroup's eight rows
  red[0][q][lane & 31] = w * g, red[1][q][lane & 31] = w * b;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5, i = threadIdx.x & 31;
    float u = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) u += red[which][k][i];
    a.lnpart[((size_t)(j * 8 + tile - 4) * 4 + (blockIdx.x & 3)) * 64 + threadIdx.x] = u;
  }
}
// LayerNorm3's parameter gradients: channel tile c gets the partials of the workgroups of tiles 4 + c and 8 + c, in order
__global__ __launch_bounds__(64) void k_ln3_param(FwFinishBatch batch) {
  const FwFinishArgs &a = batch.blk[blockIdx.y];
  const int c = blockIdx.x, which = threadIdx.x >> 5, i = threadIdx.x & 31;
  float t = 0.f;
  for (int j = 0; j < NCHUNK; ++j)
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) t += a.lnpart[((size_t)(j * 8 + 4 * p + c) * 4 + q) * 64 + threadIdx.x];
  (which ? a.db3 : a.dg3)[32 * c + i] = t;
}
inline int wgrad_slabs(long long ntiles) { return (int)(ntiles < 64 ? ntiles : 64); }
inline void launch_ff_wgrad_finish(hipStream_t st, const FwFinishBatch &f, int depth) {
  k_ff_bsum<<<dim3(NCHUNK, depth), 256, 0, st>>>(f);
  k_ff_wgrad_finish<<<dim3(NCHUNK * 12 * 4, depth), 256, 0, st>>>(f);
  k_ln3_param<<<dim3(C / 32, depth), 64, 0, st>>>(f);
}
template <bool DROP>
inline int launch_ff_wgrad_t(hipStream_t st, const FwArgs &a) {
  constexpr int LDS = DROP ? WG_LDS_DROP : WG_LDS;
  static PerDeviceOnce attrs;
  if (attrs.run([] { return set_max_lds(reinterpret_cast<const void *>(k_ff_wgrad<DROP>), LDS); }) != hipSuccess) return -1;
#ifdef DFX_TRACE_FF
  FwArgs at = a;
  at.trace = ff_trace_buffer();
  k_ff_wgrad<DROP><<<a.nslab * (NCHUNK / WG_CHUNKS), WG_NW * 64, LDS, st>>>(at);
#else
  k_ff_wgrad<DROP><<<a.nslab * (NCHUNK / WG_CHUNKS), WG_NW * 64, LDS, st>>>(a);
#endif
  return 0;
}
inline int launch_ff_wgrad(hipStream_t st, const FwArgs &a) { return a.dmask ? launch_ff_wgrad_t<true>(st, a) : launch_ff_wgrad_t<false>(st, a); }

inline size_t pack_bytes_frags() { return (size_t)PACK_RECORDS * CHUNK_U4 * sizeof(uint4); }

inline void launch_pack(hipStream_t st, PackBatch &b, int depth) {
  const int total = PACK_RECORDS * TILES * 128;
  b.depth = depth;
  k_ff_pack<<<dim3((total + 255) / 256, depth + (b.head_tab ? 1 : 0)), 256, 0, st>>>(b);
}
// workgroups of k_ff<*> (= rows of the backward's column-sum partials): one shape per workgroup
inline long long ff_groups(int B, int N) { return (long long)B * ((N / 32 + NW_BWD - 1) / NW_BWD); }
template <bool BWD, bool DROP>
inline int launch_ff_t(hipStream_t st, const FfArgs &a) {
  constexpr int LDS = FF_LDS;
  static PerDeviceOnce attrs;
  if (attrs.run([] { return set_max_lds(reinterpret_cast<const void *>(k_ff<BWD, DROP>), LDS); }) != hipSuccess) return -1;
  constexpr int NW = nw_of<BWD>();
  const long long groups = ff_groups(a.B, a.N);
#ifdef DFX_TRACE_FF
  FfArgs at = a;
  at.trace = ff_trace_buffer();
  k_ff<BWD, DROP><<<(int)groups, NW * 64, LDS, st>>>(at);
#else
  k_ff<BWD, DROP><<<(int)groups, NW * 64, LDS, st>>>(a);
#endif
  return 0;
}
template <bool DROP>
inline int launch_ff_chain_t(hipStream_t st, const FfChain &c) {
  static PerDeviceOnce attrs;
  if (attrs.run([] { return set_max_lds(reinterpret_cast<const void *>(k_ff_fwd_chain<DROP>), FF_LDS); }) != hipSuccess) return -1;
  const long long groups = ff_groups(c.blk[0].B, c.blk[0].N);
#ifdef DFX_TRACE_FF
  FfChain ct = c;
  for (int b = 0; b < c.n; ++b) ct.blk[b].trace = ff_trace_buffer();
  k_ff_fwd_chain<DROP><<<(int)groups, NW_FWD * 64, FF_LDS, st>>>(ct);
#else
  k_ff_fwd_chain<DROP><<<(int)groups, NW_FWD * 64, FF_LDS, st>>>(c);
#endif
  return 0;
}
inline int launch_ff_chain(hipStream_t st, const FfChain &c) { return c.blk[0].dmask ? launch_ff_chain_t<true>(st, c) : launch_ff_chain_t<false>(st, c); }
// dropout (FfArgs::dmask set) takes the DROP instantiations; the p = 0 kernels are the ones of round 4, untouched
template <bool BWD>
inline int launch_ff(hipStream_t st, const FfArgs &a) { return a.dmask ? launch_ff_t<BWD, true>(st, a) : launch_ff_t<BWD, false>(st, a); }

}  // namespace ffused
}  // namespace dfx
