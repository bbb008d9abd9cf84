// Fused cross-attention sub-block of a transformer block for the TRAINING path (train_kernels.hip, bf16 matrix products,
// dropout off):
//
//   forward    h1 = h + Wo att + bo,  att = softmax_j(mask(q k_j / 4)) v_j,  q = Wq LN2(h)                (attention.py:179-204, 296-306)
//   backward   given dh1:  dh = dh1 + LN2'(...),  d Wq, d Wo, d bo, d k_j, d v_j, d gamma2, d beta2
//
// The four keys / values of a shape are the same for all of its points, so — exactly as in the sampling kernel
// (denoiser_setup.hip) — q, k, v never exist per point: per shape and block
//     A_s[(head, j)][c] = 1/4 sum_{d in head} k_j[d] Wq[d][c]        (32 x 128)     sim = A_s xn2
//     M_s[c][(head, j)] = sum_{d in head} Wo[c][d] v_j[d]            (128 x 32)     h1  = h + M_s P + bo
// and the per-point work is two 32-row products on the MFMA with the softmax over the four keys of a head in between: 16 MFMAs
// per 32 points instead of 64 plus a VALU attention, and nothing but h in / h1 out touches HBM (the layer-by-layer path wrote
// xn2, q, P, att and read each back: 2.1 KB per point and block; this moves 1 KB).
//
// Backward: the gradients of A_s and M_s (per shape: sums over the shape's points) carry everything the weights Wq, Wo and the
// keys / values need:
//     dA_s[(h, j)][c] = sum_p dsim[(h, j)][p] xn2[c][p]         d Wq[d][c] = 1/4 sum_{s, j} k_{s, j}[d] dA_s[(h(d), j)][c]
//     dM_s[c][(h, j)] = sum_p dh1[c][p] P[(h, j)][p]            d k_{s, j}[d] = 1/4 sum_c dA_s[(h(d), j)][c] Wq[d][c]      (same for Wo, v)
// Sums over points are MFMA products with the points on the K axis, i.e. operands with the points along a lane's registers,
// while everything computed per point has the points along the lanes.  Two kernels therefore (the first one, and the forward, also exist
// inside the feed-forward kernels — train_ff_fused.h, FfArgs::at_frags — which is what the training path runs by default):
//   k_attn_bwd_dx     points on the lanes ("primary" layout, like the feed-forward kernel): dP = M_s^T dh1, softmax backward,
//                     dxn2 = A_s^T dsim, LayerNorm2 backward, dh out; the column sums for d gamma2, d beta2, d bo leave through a
//                     per-wave LDS tile that turns 32 points x 32 channels around
//   k_attn_bwd_param  computes sim / P / dP / dsim in the TRANSPOSED orientation — the MFMA operand layouts of A and B are symmetric,
//                     so swapping the operands of the same two register sets yields the transposed result (row index (h, j) on the
//                     lanes, 16 points in the registers, softmax over j = over the four lanes of a quad) — and multiplies them with
//                     transposed tiles of dh1 / xn2 (channels on the lanes).  Those come from the matrix unit as well: a product with
//                     a 0/1 selection matrix, xn2^T tile = I^T-blocks applied to the B-operand fragments (exact: every output is one
//                     bf16 input times 1.0), two MFMAs per 32 x 32 tile instead of a trip through LDS or strided reloads.
//                     dA_s, dM_s stay in accumulators over all tiles of a shape the workgroup owns.
// A small kernel folds (A_s, M_s) before the forward (k_attn_fold: bf16 MFMA fragments, 32 KiB per shape and block) and another
// unfolds the gradients after the backward (k_attn_unfold_*).
#pragma once
#include "train_ff_fused.h"

namespace dfx {
namespace afused {

using ffused::k_nat;
using ffused::k_reg;
using ffused::ln_rows;
using ffused::load_rows;
using ffused::rows_to_acc;
using ffused::xhalf;
using ffused::LN_EPS;
using ffused::mfma;
using ffused::pack8;
using ffused::rho;
using ffused::v16f;
using ffused::v4f;
using ffused::v8bf;
using ffused::v8f;

constexpr int C = 128, J = 4, HEADS = 8, HD = 16, HJ = HEADS * J;
using ffused::F_AS;
using ffused::F_AST;
using ffused::F_MS;
using ffused::F_MST;
using ffused::NSETS;
using ffused::SET_U4;
using ffused::SHAPE_U4;
using ffused::softmax_regs;
using ffused::softmax_bwd_regs;
using ffused::zero16;

// ---- fold: (k, v, Wq, Wo) -> fragments of A_s, M_s and their transposes; one workgroup per shape ----
// grid (B, depth): all blocks of the network in one launch (their keys / values sit side by side in one buffer, row stride ldkv)
constexpr int FOLD_MAX_DEPTH = 8;
struct FoldArgs {
  const float *kv;                      // (B J, ldkv): block i's keys at column 2 i C, values at (2 i + 1) C
  int ldkv;
  const float *wq[FOLD_MAX_DEPTH], *wo[FOLD_MAX_DEPTH];
  uint4 *frags[FOLD_MAX_DEPTH];
};
__global__ __launch_bounds__(256) void k_attn_fold(FoldArgs a) {
  __shared__ float As[HJ][C + 1], Ms[C][HJ + 1];
  const int s = blockIdx.x, t = threadIdx.x, blk = blockIdx.y, ldkv = a.ldkv;
  const float *__restrict__ k = a.kv + 2 * blk * C, *__restrict__ v = a.kv + (2 * blk + 1) * C;
  const float *__restrict__ wq = a.wq[blk], *__restrict__ wo = a.wo[blk];
  uint4 *__restrict__ frags = a.frags[blk];
  for (int idx = t; idx < HJ * C; idx += 256) {
    const int m = idx / C, c = idx % C, hd = m >> 2, j = m & 3;
    const float *kk = k + ((size_t)s * J + j) * ldkv + hd * HD;
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) a += kk[d] * wq[(size_t)(hd * HD + d) * C + c];
    As[m][c] = 0.25f * a;   // dim_head ** -0.5
  }
  for (int idx = t; idx < HJ * C; idx += 256) {
    const int c = idx / HJ, m = idx % HJ, hd = m >> 2, j = m & 3;
    const float *vv = v + ((size_t)s * J + j) * ldkv + hd * HD;
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) a += wo[(size_t)c * C + hd * HD + d] * vv[d];
    Ms[c][m] = a;
  }
  __syncthreads();
  for (int idx = t; idx < SHAPE_U4; idx += 256) {
    const int set = idx >> 9, tile = (idx >> 7) & 3, u = (idx >> 6) & 1, lane = idx & 63, i = lane & 31, hf = lane >> 5;
    __bf16 o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x;
      if (set == F_AS) x = As[i][32 * tile + k_nat(u, hf, e)];            // A of sim = A_s xn2 (K = channels)
      else if (set == F_MS) x = Ms[32 * tile + i][k_reg(u, hf, e)];       // A of M_s P (K = (h, j) in register order)
      else if (set == F_MST) x = Ms[32 * tile + k_nat(u, hf, e)][i];      // A of dP = M_s^T dh1 (K = channels)
      else x = As[k_reg(u, hf, e)][32 * tile + i];                        // A of dxn2 = A_s^T dsim (K = (h, j))
      o[e] = (__bf16)x;
    }
    frags[(size_t)s * SHAPE_U4 + idx] = *reinterpret_cast<const uint4 *>(o);
  }
}

struct AttnArgs {
  const uint4 *frags;     // [B][NSETS][4][2][64]
  const float *valid;     // (B, 4): 0 = masked key
  const float *g2, *b2;   // LayerNorm2 affine
  const float *bo;        // to_out bias
  const float *h;         // (R, 128) block input
  float *h1;              // forward out
  const float *dh1;       // backward in: gradient at h1
  float *dh;              // backward out: gradient at h
  float *part;            // param kernel: [B * split][2][32][128]  (dA_s, dM_s^T)
  float *cpart;           // dx kernel: [workgroups][3][128]        (d gamma2, d beta2, d bo)
  const uint4 *pk2;       // param kernel, optional: [R / 32][2][4][2][64] xn2 / dh1 of every tile as bf16 fragments (written by k_ff<true>);
                          // h and dh1 are then not read
  int N, split;           // points per shape (multiple of 32); workgroups per shape in the param kernel
  long long R;
};

__device__ __forceinline__ float quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }

// a row of fp32 values in the B-operand layout, rounded to bf16 fragments
__device__ __forceinline__ void row_frags(const float *__restrict__ row, int hf, uint4 (&f)[4][2]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float *p = row + 32 * c + k_nat(u, hf, 0);
      const v4f lo = *reinterpret_cast<const v4f *>(p), hi = *reinterpret_cast<const v4f *>(p + 4);
      const v8f t = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      f[c][u] = __builtin_bit_cast(uint4, __builtin_convertvector(t, v8bf));
    }
}

constexpr int NW = 4;   // wavefronts per workgroup: small workgroups, several per CU — these kernels are bound by memory latency

// ---- forward: one wavefront = 32 points ----
__global__ __launch_bounds__(NW * 64, 4) void k_attn_fwd_fused(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float gb[3 * C];   // gamma2 | beta2 | bo
  for (int i = threadIdx.x; i < 3 * C; i += NW * 64) gb[i] = i < C ? a.g2[i] : i < 2 * C ? a.b2[i - C] : a.bo[i - 2 * C];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, pj = lane & 31;
  const long long row0 = ((long long)blockIdx.x * NW + wave) * 32;
  if (row0 >= a.R) return;
  const int s = (int)(row0 / a.N);
  const long long row = row0 + pj;
  const uint4 *fr = a.frags + (size_t)s * SHAPE_U4 + lane;
  unsigned vmask = 0;
#pragma unroll
  for (int j = 0; j < J; ++j) vmask |= (a.valid[s * J + j] != 0.f ? 1u : 0u) << j;
  uint4 xn[4][2];
  float mu, rstd;
  v16f hacc[4];   // h in the accumulator layout, from the same read
  {
    v8f x[4][2];
    load_rows(a.h + row * C, hf, x);
    ln_rows(x, hf, gb, xn, mu, rstd);
    rows_to_acc(x, hacc);
  }
  v16f sim = zero16();
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) sim = mfma(fr[F_AS * SET_U4 + (c * 2 + u) * 64], xn[c][u], sim);
  softmax_regs(sim, vmask);
  const uint4 p0 = pack8(sim, 0), p1 = pack8(sim, 1);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    v16f acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f b = *reinterpret_cast<const v4f *>(gb + 2 * C + 32 * ct + 8 * q + 4 * hf);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[4 * q + m] = hacc[ct][4 * q + m] + b[m];
    }
    acc = mfma(fr[F_MS * SET_U4 + (ct * 2 + 0) * 64], p0, acc);
    acc = mfma(fr[F_MS * SET_U4 + (ct * 2 + 1) * 64], p1, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<v4f *>(a.h1 + row * C + 32 * ct + 8 * q + 4 * hf) = v4f{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  }
}

// ---- backward, gradient of the input: dh = dh1 + LN2'(A_s^T dsim); column sums of dxn2 xhat, dxn2, dh1 per workgroup ----
constexpr int TROW = 36;   // floats per row of the per-wave transposition tile: 16-byte aligned rows, conflict-free column reads
__global__ __launch_bounds__(NW * 64, 2) void k_attn_bwd_dx(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float gb[2 * C];
  __shared__ __attribute__((aligned(16))) float tiles[NW][32 * TROW];
  __shared__ float cred[NW][3][C];
  for (int i = threadIdx.x; i < 2 * C; i += NW * 64) gb[i] = i < C ? a.g2[i] : a.b2[i - C];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, pj = lane & 31;
  for (int i = threadIdx.x; i < NW * 3 * C; i += NW * 64) (&cred[0][0][0])[i] = 0.f;   // (each wave touches only its own rows afterwards)
  __syncthreads();
  // a workgroup walks tile groups blockIdx.x, + gridDim.x, ..: a bounded number of column-sum partials whatever R
  for (long long grp = blockIdx.x; grp * NW * 32 < a.R; grp += gridDim.x) {
  long long row0 = (grp * NW + wave) * 32;
  const bool live = row0 < a.R;
  if (!live) row0 = a.R - 32;   // trailing wavefronts recompute the last tile, store nothing and add zeros to the column sums
  const int s = (int)(row0 / a.N);
  const long long row = row0 + pj;
  const uint4 *fr = a.frags + (size_t)s * SHAPE_U4 + lane;
  unsigned vmask = 0;
#pragma unroll
  for (int j = 0; j < J; ++j) vmask |= (a.valid[s * J + j] != 0.f ? 1u : 0u) << j;
  float mu, rstd;
  v16f P = zero16();
  uint4 db[4][2];
  row_frags(a.dh1 + row * C, hf, db);   // (requested before the LayerNorm arithmetic needs its own loads)
  v16f dy[4], xh[4];
  {
    uint4 xn[4][2];
    v8f x[4][2];
    load_rows(a.h + row * C, hf, x);
    ln_rows(x, hf, gb, xn, mu, rstd);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        x[c][u] = (x[c][u] - mu) * rstd;
        P = mfma(fr[F_AS * SET_U4 + (c * 2 + u) * 64], xn[c][u], P);
      }
    rows_to_acc(x, xh);   // xhat in the accumulator layout for the LayerNorm backward, from the same read
  }
  softmax_regs(P, vmask);
  v16f ds = zero16();
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) ds = mfma(fr[F_MST * SET_U4 + (c * 2 + u) * 64], db[c][u], ds);
  softmax_bwd_regs(P, ds);
  const uint4 d0 = pack8(ds, 0), d1 = pack8(ds, 1);
  // dxn2 (accumulator layout: register r of tile ct = channel 32 ct + rho(r, hf)) and the LayerNorm backward
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    dy[ct] = mfma(fr[F_AST * SET_U4 + (ct * 2 + 0) * 64], d0, zero16());
    dy[ct] = mfma(fr[F_AST * SET_U4 + (ct * 2 + 1) * 64], d1, dy[ct]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * ct + 8 * q + 4 * hf;
      const v4f g = *reinterpret_cast<const v4f *>(gb + ch);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float dg = dy[ct][4 * q + m] * g[m];
        s1 += dg;
        s2 = fmaf(dg, xh[ct][4 * q + m], s2);
      }
    }
  }
  s1 += xhalf(s1), s2 += xhalf(s2);
  s1 *= (1.0f / C), s2 *= (1.0f / C);
  float *tt = tiles[wave];
  const float keep = live ? 1.f : 0.f;
  auto colsum = [&](const v16f &v, int which, int ct) {   // sum over the wave's 32 points of v, per channel of tile ct
#pragma unroll
    for (int r = 0; r < 16; ++r) tt[rho(r, hf) * TROW + pj] = v[r];
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const v4f x = *reinterpret_cast<const v4f *>(tt + pj * TROW + 16 * hf + 4 * k);
      t += (x[0] + x[1]) + (x[2] + x[3]);
    }
    t += xhalf(t);
    if (hf == 0) cred[wave][which][32 * ct + pj] += t * keep;
  };
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    v16f d1v, gx;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * ct + 8 * q + 4 * hf;
      const v4f d = *reinterpret_cast<const v4f *>(a.dh1 + row * C + ch), g = *reinterpret_cast<const v4f *>(gb + ch);
      v4f o;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int r = 4 * q + m;
        d1v[r] = d[m];
        gx[r] = dy[ct][r] * xh[ct][r];
        o[m] = d[m] + rstd * (dy[ct][r] * g[m] - s1 - xh[ct][r] * s2);
      }
      if (live) *reinterpret_cast<v4f *>(a.dh + row * C + ch) = o;
    }
    colsum(gx, 0, ct);
    colsum(dy[ct], 1, ct);
    colsum(d1v, 2, ct);
  }
  }
  __syncthreads();
  if (threadIdx.x < 3 * C / 2) {   // 192 threads x 2 columns
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int col = threadIdx.x * 2 + k;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += cred[w][col / C][col % C];
      a.cpart[(size_t)blockIdx.x * 3 * C + col] = t;
    }
  }
}

// a uint4 pointer that stays in the LDS address space through an opaque asm (a generic pointer would become flat loads)
typedef const __attribute__((address_space(3))) uint4 *LdsU4;
__device__ __forceinline__ uint4 lds_u4(LdsU4 p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *p;
#else
  (void)p;
  return uint4{};
#endif
}

// ---- backward, parameter side: dA_s, dM_s per shape ----
// grid = B * split workgroups; workgroup (s, k) walks the row tiles k * per .. (k + 1) * per of shape s, NW at a time
template <bool PACKED>
__global__ __launch_bounds__(NW * 64, 2) void k_attn_bwd_param(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float gb[2 * C];
  __shared__ __attribute__((aligned(16))) float red[2 * HJ * C];   // cross-wave sum of dA_s | dM_s^T (32 KiB)
  // the shape's [A_s | M_s^T] fragments (the B operands of sim^T and dP^T, the same for every tile): once into LDS — read from memory
  // they were 16 dependent L2 round trips per tile
  __shared__ __attribute__((aligned(16))) uint4 frs[2 * SET_U4];
  {
    const uint4 *src = a.frags + (size_t)(blockIdx.x / a.split) * SHAPE_U4;
    uint4 t[2 * SET_U4 / (NW * 64)];
#pragma unroll
    for (int k = 0; k < 2 * SET_U4 / (NW * 64); ++k) {
      const int i = k * NW * 64 + threadIdx.x;
      t[k] = src[(i < SET_U4 ? F_AS * SET_U4 : (F_MST - 1) * SET_U4) + i];
    }
#pragma unroll
    for (int k = 0; k < 2 * SET_U4 / (NW * 64); ++k) frs[k * NW * 64 + threadIdx.x] = t[k];
  }
  for (int i = threadIdx.x; i < 2 * C; i += NW * 64) gb[i] = i < C ? a.g2[i] : a.b2[i - C];
  for (int i = threadIdx.x; i < 2 * HJ * C; i += NW * 64) red[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, pj = lane & 31;
  const int s = blockIdx.x / a.split, part_k = blockIdx.x % a.split;
  const int tiles = a.N / 32, per = (tiles + a.split - 1) / a.split;
  const int t_begin = part_k * per, t_end = t_begin + per < tiles ? t_begin + per : tiles;
  const LdsU4 fr0 = (LdsU4)frs + lane;   // [0, SET_U4): A_s, [SET_U4, 2 SET_U4): M_s^T
  unsigned vmask = 0;
#pragma unroll
  for (int j = 0; j < J; ++j) vmask |= (a.valid[s * J + j] != 0.f ? 1u : 0u) << j;
  const bool keep_lane = (vmask >> (lane & 3)) & 1u;   // transposed layout: lane = (head, j), j = lane & 3
  // selection matrices: B[k][n] = (n == 16 u + k), this lane's eight k = 8 hf .. 8 hf + 7 of column n = lane & 31
  uint4 sel[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    __bf16 o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(pj == 16 * u + 8 * hf + e ? 1.0f : 0.0f);
    sel[u] = *reinterpret_cast<const uint4 *>(o);
  }
  // the transposed 32 x 32 tile of two B-operand fragments: lane = channel of the tile, registers = points rho(r, hf)
  auto transposed = [&](const uint4 &f0, const uint4 &f1, uint4 &t0, uint4 &t1) {
    v16f t = mfma(f0, sel[0], zero16());
    t = mfma(f1, sel[1], t);
    t0 = pack8(t, 0), t1 = pack8(t, 1);
  };
  v16f dAs[4], dMs[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) dAs[ct] = zero16(), dMs[ct] = zero16();
  for (int t = t_begin + wave; t < t_end; t += NW) {
    const long long row = (long long)s * a.N + (long long)t * 32 + pj;
    LdsU4 fr = fr0;
    asm volatile("" : "+v"(fr));   // the fragments are the same for every tile of the shape: hoisted out of the loop they would cost 64 registers (spilled)
    v16f PT = zero16(), dsT = zero16();
    uint4 xn[4][2];
    const uint4 *pk = PACKED ? a.pk2 + (size_t)((row - pj) / 32) * (2 * 8 * 64) + lane : nullptr;
    if (PACKED) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) xn[c][u] = pk[(c * 2 + u) * 64];
    } else {
      float mu, rstd;
      ln_rows(a.h + row * C, hf, gb, xn, mu, rstd);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) PT = mfma(xn[c][u], lds_u4(fr + (c * 2 + u) * 64), PT);   // [point][(h, j)]: lane = (h, j), registers = points
#pragma unroll
    for (int r = 0; r < 16; ++r) {   // softmax over j = over the quad
      const float x = keep_lane ? PT[r] : -3.402823466e38f;
      float m = fmaxf(x, quad_xor1(x));
      m = fmaxf(m, quad_xor2(m));
      const float e = __builtin_amdgcn_exp2f((x - m) * 1.44269504088896341f);   // the same hardware exp2 / rcp as ffused::softmax_regs
      float den = e + quad_xor1(e);
      den += quad_xor2(den);
      PT[r] = e * __builtin_amdgcn_rcpf(den);
    }
    {
      const uint4 pt0 = pack8(PT, 0), pt1 = pack8(PT, 1);        // A of dM_s^T: [(h, j)][points in register order]
      uint4 db[4][2];
      if (PACKED) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int u = 0; u < 2; ++u) db[c][u] = pk[(8 + c * 2 + u) * 64];
      } else {
        row_frags(a.dh1 + row * C, hf, db);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int u = 0; u < 2; ++u) dsT = mfma(db[c][u], lds_u4(fr + SET_U4 + (c * 2 + u) * 64), dsT);
        uint4 t0, t1;
        transposed(db[c][0], db[c][1], t0, t1);
        dMs[c] = mfma(pt0, t0, dMs[c]);
        dMs[c] = mfma(pt1, t1, dMs[c]);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {   // softmax backward
      const float pd = PT[r] * dsT[r];
      float dot = pd + quad_xor1(pd);
      dot += quad_xor2(dot);
      dsT[r] = PT[r] * (dsT[r] - dot);
    }
    const uint4 dt0 = pack8(dsT, 0), dt1 = pack8(dsT, 1);        // A of dA_s
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint4 t0, t1;
      transposed(xn[c][0], xn[c][1], t0, t1);
      dAs[c] = mfma(dt0, t0, dAs[c]);
      dAs[c] = mfma(dt1, t1, dAs[c]);
    }
  }
  // cross-wave sums in a fixed order (wave 0, 1, ..): accumulator register r of tile ct, lane (i, hf) = row rho(r, hf), column 32 ct + i
  for (int w = 0; w < NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = rho(r, hf) * C + 32 * ct + pj;
          red[o] += dAs[ct][r];
          red[HJ * C + o] += dMs[ct][r];
        }
    }
    __syncthreads();
  }
  float *out = a.part + (size_t)blockIdx.x * 2 * HJ * C;
  for (int i = threadIdx.x; i < 2 * HJ * C; i += NW * 64) out[i] = red[i];
}

// ---- unfold: per-shape dA_s, dM_s^T -> d k, d v (per shape) and d Wq, d Wo (sums over shapes and keys, fixed order) ----
// grid (B + 128): blocks < B do one shape's keys / values, the others one row d of the weights; 512 threads
struct UnfoldArgs {
  const float *part;        // [B * split][2][32][128]
  const float *k, *v;       // (B J, 128), row stride ldkv (the keys / values of all blocks sit side by side)
  const float *wq, *wo;     // (128, 128)
  float *dk, *dv;           // (B J, 128), row stride ldkv
  float *dwq, *dwo;         // (128, 128)
  float *sum;               // scratch [B][2][32][128]: partials of a shape summed (written by the shape blocks of launch 1)
  int B, split, ldkv;
};
// The unfold kernels run ONCE per backward, for all transformer blocks (blockIdx.z): the parameter kernel of block i leaves its partials in
// block i's own buffer, nothing on the gradient chain dh waits for them (round 4: ten launches of 128 - 512 workgroups -> two)
struct UnfoldBatch {
  UnfoldArgs blk[DFX_MAX_DEPTH];
};
// grid (J, B, depth): one block per key / value token of a shape; 256 threads = 128 channels d x 2 halves of the sum over c
__global__ __launch_bounds__(256) void k_attn_unfold_kv(UnfoldBatch batch) {
  const UnfoldArgs &a = batch.blk[blockIdx.z];
  __shared__ float dA[HEADS][C], dM[HEADS][C];   // rows (h, j) of this token: summed over the partials
  __shared__ float half_k[C], half_v[C];
  const int j = blockIdx.x, s = blockIdx.y, t = threadIdx.x;
  for (int idx = t; idx < 2 * HEADS * C; idx += 256) {
    const int which = idx >= HEADS * C, hd = (idx >> 7) & (HEADS - 1), c = idx & 127;
    const size_t o = (size_t)which * HJ * C + (size_t)(hd * J + j) * C + c;
    float x = 0.f;
    for (int q = 0; q < a.split; ++q) x += a.part[((size_t)s * a.split + q) * 2 * HJ * C + o];
    a.sum[(size_t)s * 2 * HJ * C + o] = x;
    (which ? dM : dA)[hd][c] = x;
  }
  __syncthreads();
  const int d = t & 127, hlf = t >> 7, hd = d >> 4, c0 = 64 * hlf;
  const float *wq = a.wq + (size_t)d * C + c0;   // this thread's own row: 16-byte loads
  float ak = 0.f, av = 0.f;
#pragma unroll 4
  for (int c4 = 0; c4 < 16; ++c4) {
    const v4f w = *reinterpret_cast<const v4f *>(wq + 4 * c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + 4 * c4 + e;
      ak = fmaf(dA[hd][c], w[e], ak);
      av = fmaf(dM[hd][c], a.wo[(size_t)c * C + d], av);
    }
  }
  if (hlf) half_k[d] = ak, half_v[d] = av;
  __syncthreads();
  if (!hlf) {
    a.dk[((size_t)s * J + j) * a.ldkv + d] = 0.25f * (ak + half_k[d]);
    a.dv[((size_t)s * J + j) * a.ldkv + d] = av + half_v[d];
  }
}
// one block per weight row d (of Wq) / column d (of Wo); 128 channels c x 8 groups of shapes (summed in group order)
constexpr int UW_D = 4;   // weight rows per block of k_attn_unfold_w: the 16 rows of a head read the same partial sums (one row per block: 330 MB of L2 traffic per backward)
__global__ __launch_bounds__(1024) void k_attn_unfold_w(UnfoldBatch batch) {   // grid (128 / UW_D, 1, depth)
  const UnfoldArgs &a = batch.blk[blockIdx.z];
  __shared__ float rq[8][UW_D][C], ro[8][UW_D][C];
  const int d0 = blockIdx.x * UW_D, c = threadIdx.x & 127, grp = threadIdx.x >> 7, hd = d0 >> 4;
  float aq[UW_D], ao[UW_D];
#pragma unroll
  for (int i = 0; i < UW_D; ++i) aq[i] = ao[i] = 0.f;
  for (int s = grp; s < a.B; s += 8)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float *sm = a.sum + (size_t)s * 2 * HJ * C + (size_t)(hd * J + j) * C + c;
      const float sq = sm[0], so = sm[HJ * C];
      const float *kr = a.k + ((size_t)s * J + j) * a.ldkv + d0, *vr = a.v + ((size_t)s * J + j) * a.ldkv + d0;
#pragma unroll
      for (int i = 0; i < UW_D; ++i) aq[i] = fmaf(kr[i], sq, aq[i]), ao[i] = fmaf(vr[i], so, ao[i]);   // (per output the same order as with one row per block)
    }
#pragma unroll
  for (int i = 0; i < UW_D; ++i) rq[grp][i][c] = aq[i], ro[grp][i][c] = ao[i];
  __syncthreads();
  if (grp < UW_D) {
    const int i = grp, d = d0 + i;
    float tq = 0.f, to = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) tq += rq[g][i][c], to += ro[g][i][c];
    a.dwq[(size_t)d * C + c] = 0.25f * tq;
    a.dwo[(size_t)c * C + d] = to;
  }
}

inline int dx_groups(long long R) {   // workgroups of k_attn_bwd_dx = rows of its column-sum partials
  const long long g = (R / 32 + NW - 1) / NW;
  return (int)(g < 1024 ? g : 1024);
}
inline int param_split(int B, int N) {   // workgroups per shape in the parameter kernel: ~4 workgroups per CU, at least two tiles per wavefront
  int split = 1;
  while (B * split < 512 && (N / 32) / (split * 2) >= 2 * NW) split *= 2;
  return split;
}

}  // namespace afused
}  // namespace dfx
