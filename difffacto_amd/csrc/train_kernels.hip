// The stage-1 training step: the cross-diffusion denoiser in training mode and its backward (below),
// PointNetV2 in train mode and the prior loss through the latent flows (further down), the loss gradient and the optimiser.
// Denoiser:
//   TransformerNet.forward / _forward_attn (attention.py:385-440, BasicTransformerBlock :296-306 with single_attn,
//   CrossAttention :179-204, FeedForward/GEGLU :50-57,77-94, timestep_embedding utils.py:7-24), optional dropout,
// in exact fp32 or with bf16 matrix products (gemm_bf16.h) (v_mfma_f32_32x32x2_f32 through the shared row-batched linear kernels), every intermediate that the
// backward needs kept in a caller-provided workspace (2.4 KB per point and block: 12.5 GB at B = 128, N = 2048 —
// sized for 288 GB of HBM, nothing is recomputed).  The persistent bf16 chain kernel is the sampling path; this file is
// the first correct training path: one launch per layer, parity with the reference's autograd is the bar
// (tests/test_gpu_train.py against tests/golden/train_grads_*.npz generated from the reference's own model).
//
// Gradient of a linear layer Y = X W^T + b over R rows:
//   dX = dY W          -> the shared forward kernel on a transposed copy of W (k_transpose, a few KB per call)
//   dW = dY^T X, db    -> k_wgrad: one wavefront owns a 64 x 64 tile of dW and a slab of rows, two rows per
//                         v_mfma_f32_32x32x2_f32 (lane (j, hf) feeds dY[r + hf][o0 + j] and X[r + hf][i0 + j]: 128-byte
//                         coalesced row segments), partial tiles per slab, deterministic second pass (no atomics)
#include <functional>
#include <mutex>
#include <unordered_map>

#include "dfx_common.h"
#include "dfx_dropout.h"
#include "gemm_bf16.h"
#include "train_attn_fused.h"
#include "mfma_linear.h"

namespace {

using dfx::lin::LinArgs;
using dfx::lin::v16f;
using dfx::lin::v4f;

constexpr int C = 128, J = 4, HEADS = 8, HD = 16, FH = 512, CTX = 522, CTXP = 528, TE = 256, TEH = 1024, XIN = 16;
constexpr float LN_EPS = 1e-5f;

// ---------------------------------------------------------------------------------------------------------------
// elementwise / row kernels
// ---------------------------------------------------------------------------------------------------------------

// Tensors that only matrix products consume (xn2, xn3, att, hid, d[a|g], dq) are STORED as bf16 when the products run in
// bf16 (template parameter BF of their producers): the product kernels would round them to bf16 anyway, so the values
// entering the MFMAs are the same and the HBM traffic of these tensors halves.
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));

// Dropout (nn.Dropout in train mode: attention.py:84 after the GEGLU, :177 after to_out): the contract of dfx_dropout.h — Philox4x32-7 keyed by the
// step's seed, counter = (group of eight consecutive elements, site), one 16-bit draw per element; factor = keep ? 1 / (1 - p) : 0.  The layer-by-layer
// kernels below work on four consecutive elements per thread = one half of a group.  The backward regenerates the factors from the same
// (seed, site, index): no mask is stored here (the fused kernels of train_ff_fused.h store one bit per element instead).
__device__ __forceinline__ v4f dropout4(unsigned long long seed, unsigned site, unsigned long long idx4, float p) {
  const dfx::DropKey k = dfx::drop_key(seed, p);
  const uint4 c = dfx::drop_group(k, site, idx4 >> 1);
  const unsigned w0 = (idx4 & 1) ? c.z : c.x, w1 = (idx4 & 1) ? c.w : c.y;
  return v4f{dfx::drop_keep_lo(w0, k.thr) ? k.keep : 0.f, dfx::drop_keep_hi(w0, k.thr) ? k.keep : 0.f,
             dfx::drop_keep_lo(w1, k.thr) ? k.keep : 0.f, dfx::drop_keep_hi(w1, k.thr) ? k.keep : 0.f};
}
struct Drop {
  float p;                  // 0 = off
  unsigned long long seed;
  unsigned site;
};
enum { SITE_TE = 1000 };    // sites: 2 i = attention output of block i, 2 i + 1 = feed-forward of block i, SITE_TE = time_embed
template <bool BF>
__device__ __forceinline__ void store4(float *base, size_t idx, v4f v) {   // idx in elements, multiple of 4
  if (BF) *reinterpret_cast<v4bf *>(reinterpret_cast<__bf16 *>(base) + idx) = __builtin_convertvector(v, v4bf);
  else *reinterpret_cast<v4f *>(base + idx) = v;
}
template <bool BF>
__device__ __forceinline__ void store1(float *base, size_t idx, float v) {
  if (BF) reinterpret_cast<__bf16 *>(base)[idx] = (__bf16)v;
  else base[idx] = v;
}

// xin (attention.py:398-405): [x | anchors | variances | one_hot(assignment)] per point, padded 13 -> 16
__global__ void k_build_xin(const float *__restrict__ x, const float *__restrict__ anc, const float *__restrict__ var,
                            const int32_t *__restrict__ asg, float *__restrict__ xin, int N, long long R) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const long long b = r / N;
  const int n = (int)(r % N);
  float v[XIN];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    v[c] = x[(b * 3 + c) * N + n];
    v[3 + c] = anc[r * 3 + c];
    v[6 + c] = var[r * 3 + c];
  }
  const int a = asg[r];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[9 + j] = a == j ? 1.f : 0.f;
  v[13] = v[14] = v[15] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) reinterpret_cast<v4f *>(xin + r * XIN)[q] = v4f{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}

// LayerNorm over 128 channels, two-pass like torch (mean, then biased variance of the centred values); 32 lanes per row
template <bool BF>
__global__ __launch_bounds__(256) void k_ln_fwd(const float *__restrict__ x, const float *__restrict__ g,
                                                 const float *__restrict__ be, float *__restrict__ y,
                                                 float *__restrict__ stats, long long R) {
  const int l = threadIdx.x & 31;
  const long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (r >= R) return;
  const v4f v = reinterpret_cast<const v4f *>(x + r * C)[l];
  float s = v[0] + v[1] + v[2] + v[3];
  for (int o = 16; o; o >>= 1) s += __shfl_xor(s, o, 32);
  const float mu = s * (1.0f / C);
  const v4f d = {v[0] - mu, v[1] - mu, v[2] - mu, v[3] - mu};
  float q = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  for (int o = 16; o; o >>= 1) q += __shfl_xor(q, o, 32);
  const float rstd = 1.0f / sqrtf(q * (1.0f / C) + LN_EPS);
  const v4f gv = reinterpret_cast<const v4f *>(g)[l], bv = reinterpret_cast<const v4f *>(be)[l];
  v4f o4;
#pragma unroll
  for (int e = 0; e < 4; ++e) o4[e] = d[e] * rstd * gv[e] + bv[e];
  store4<BF>(y, (size_t)r * C + 4 * l, o4);
  if (l == 0) stats[2 * r] = mu, stats[2 * r + 1] = rstd;
}

// LayerNorm backward: dx = rstd (dyg - mean(dyg) - xhat mean(dyg xhat)), dyg = dy gamma;  out = (resid ? resid : 0) + dx.
// Column sums of dy xhat (d gamma) and dy (d beta) per block of rows -> part[blockIdx][2][128]
constexpr int LNB_ROWS = 512;   // rows per block (8 row groups x 64 trips)
__global__ __launch_bounds__(256) void k_ln_bwd(const float *__restrict__ dy, const float *__restrict__ x,
                                                 const float *__restrict__ stats, const float *__restrict__ g,
                                                 const float *__restrict__ resid, float *__restrict__ out,
                                                 float *__restrict__ part, long long R) {
  __shared__ float red[8][2][C];
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const v4f gv = reinterpret_cast<const v4f *>(g)[l];
  v4f sg = {0, 0, 0, 0}, sb = {0, 0, 0, 0};
  for (int it = 0; it < LNB_ROWS / 8; ++it) {
    const long long r = (long long)blockIdx.x * LNB_ROWS + it * 8 + grp;
    if (r >= R) break;
    const v4f xv = reinterpret_cast<const v4f *>(x + r * C)[l], dv = reinterpret_cast<const v4f *>(dy + r * C)[l];
    const float mu = stats[2 * r], rstd = stats[2 * r + 1];
    v4f xh, dg;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[e] = (xv[e] - mu) * rstd;
      dg[e] = dv[e] * gv[e];
      s1 += dg[e];
      s2 += dg[e] * xh[e];
      sg[e] += dv[e] * xh[e];
      sb[e] += dv[e];
    }
    for (int o = 16; o; o >>= 1) s1 += __shfl_xor(s1, o, 32), s2 += __shfl_xor(s2, o, 32);
    s1 *= (1.0f / C), s2 *= (1.0f / C);
    v4f o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) o4[e] = rstd * (dg[e] - s1 - xh[e] * s2);
    if (resid) {
      const v4f rv = reinterpret_cast<const v4f *>(resid + r * C)[l];
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] += rv[e];
    }
    reinterpret_cast<v4f *>(out + r * C)[l] = o4;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[grp][0][4 * l + e] = sg[e], red[grp][1][4 * l + e] = sb[e];
  __syncthreads();
  const int t = threadIdx.x;   // 256 = 2 x 128
  float a = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) a += red[q][t >> 7][t & 127];
  part[(size_t)blockIdx.x * 2 * C + t] = a;
}

// out[c] = sum over nparts of part[p][c]  (second pass of every column reduction).  Block = 32 columns x 8 part groups;
// group q of G = blockDim.x / 32 adds parts q, q+G, ... in order, the G group sums are added in order: a fixed summation tree.
// (G = 32: the LayerNorm / bias column sums have 4 .. 32 blocks of 512 partial rows each — with 8 groups a launch was 16 us of
// serial adds, 35 launches per iteration.)
__global__ __launch_bounds__(1024) void k_sum_parts(const float *__restrict__ part, float *__restrict__ out, int nparts, int cols,
                                                     int ld_part) {
  __shared__ float red[32][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5, G = blockDim.x >> 5;
  float a = 0.f;
  if (c < cols)
    for (int p = q; p < nparts; p += G) a += part[(size_t)p * ld_part + c];
  red[q][threadIdx.x & 31] = a;
  __syncthreads();
  if (q == 0 && c < cols) {
    float t = red[0][threadIdx.x];
    for (int k = 1; k < G; ++k) t += red[k][threadIdx.x];
    out[c] = t;
  }
}

// The same for up to four outputs of `cols` columns each that sit side by side in the partial rows (part[p][which * cols + c]):
// one launch for e.g. (d gamma, d beta, d bias); four independent partial sums per thread keep the loads in flight.
struct SumOuts {
  float *o[4];
};
__global__ __launch_bounds__(1024) void k_sum_parts_multi(const float *__restrict__ part, SumOuts outs, int nparts, int cols, int ld_part) {
  __shared__ float red[32][32];
  const int per = cols / 32, which = blockIdx.x / per, c = (blockIdx.x % per) * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
  if (!outs.o[which]) return;   // (a row of the partials nobody wants: the whole workgroup leaves)
  const float *src = part + which * cols + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int p = q;
  for (; p + 96 < nparts; p += 128) {
    a0 += src[(size_t)p * ld_part], a1 += src[(size_t)(p + 32) * ld_part];
    a2 += src[(size_t)(p + 64) * ld_part], a3 += src[(size_t)(p + 96) * ld_part];
  }
  for (; p < nparts; p += 32) a0 += src[(size_t)p * ld_part];
  red[q][threadIdx.x & 31] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q == 0) {
    float t = red[0][threadIdx.x];
    for (int k = 1; k < 32; ++k) t += red[k][threadIdx.x];
    outs.o[which][c] = t;
  }
}

// Up to DFX_MAX_DEPTH x 6 outputs of `cols` columns in ONE launch: job (block b, which q) sums rows of part[b] (row length ld_part, the
// six vectors side by side).  grid = njobs x cols / 32.
struct SumJobs {
  const float *part[DFX_MAX_DEPTH];
  float *o[DFX_MAX_DEPTH][6];
};
__global__ __launch_bounds__(1024) void k_sum_parts_jobs(SumJobs jobs, int nparts, int cols, int ld_part) {
  __shared__ float red[32][32];
  const int per = cols / 32, job = blockIdx.x / per, b = job / 6, which = job % 6, c = (blockIdx.x % per) * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
  if (!jobs.o[b][which]) return;   // (a row of the partials nobody wants: the whole workgroup leaves)
  const float *src = jobs.part[b] + which * cols + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int p = q;
  for (; p + 96 < nparts; p += 128) {
    a0 += src[(size_t)p * ld_part], a1 += src[(size_t)(p + 32) * ld_part];
    a2 += src[(size_t)(p + 64) * ld_part], a3 += src[(size_t)(p + 96) * ld_part];
  }
  for (; p < nparts; p += 32) a0 += src[(size_t)p * ld_part];
  red[q][threadIdx.x & 31] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q == 0) {
    float t = red[0][threadIdx.x];
    for (int k = 1; k < 32; ++k) t += red[k][threadIdx.x];
    jobs.o[b][which][c] = t;
  }
}

__device__ __forceinline__ float gelu_erf(float g) { return 0.5f * g * (1.f + erff(g * 0.70710678118654752440f)); }

// [a | g] itself is stored as bf16 in bf16 mode (-DDFX_TRAIN_AG_F32 keeps it fp32): written once and read twice per block, it
// is the largest saved tensor (1 GB at B = 128 x 2048).  Unlike the product-only tensors this IS lossy for the backward:
// gelu'(g) and a are taken from the rounded values (the tolerance test covers it).
#ifdef DFX_TRAIN_AG_F32
constexpr bool AG_BF16 = false;
#else
constexpr bool AG_BF16 = true;
#endif
template <bool BF>
__device__ __forceinline__ v4f load4(const float *base, size_t idx) {
  if (BF) return __builtin_convertvector(*reinterpret_cast<const v4bf *>(reinterpret_cast<const __bf16 *>(base) + idx), v4f);
  return *reinterpret_cast<const v4f *>(base + idx);
}
// GEGLU (attention.py:55-57): hid = a * gelu(g), ag = [a | g] (R, 2 H); four consecutive units per thread (16-byte loads)
template <bool BF>
__global__ void k_geglu_fwd(const float *__restrict__ ag, float *__restrict__ hid, int H, long long total, Drop dr) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= total) return;
  const long long r = i / H;
  const int c = (int)(i % H);
  const v4f a = load4<BF && AG_BF16>(ag, (size_t)(r * 2 * H + c)), g = load4<BF && AG_BF16>(ag, (size_t)(r * 2 * H + H + c));
  v4f h = {a[0] * gelu_erf(g[0]), a[1] * gelu_erf(g[1]), a[2] * gelu_erf(g[2]), a[3] * gelu_erf(g[3])};
  if (dr.p > 0.f) h *= dropout4(dr.seed, dr.site, (unsigned long long)(i >> 2), dr.p);
  store4<BF>(hid, (size_t)i, h);
}
// d a = d hid gelu(g);  d g = d hid a (Phi(g) + g phi(g))
template <bool BF>
__global__ void k_geglu_bwd(const float *__restrict__ ag, const float *__restrict__ dhid, float *__restrict__ dag, int H,
                            long long total, Drop dr) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= total) return;
  const long long r = i / H;
  const int c = (int)(i % H);
  const v4f a = load4<BF && AG_BF16>(ag, (size_t)(r * 2 * H + c)), g = load4<BF && AG_BF16>(ag, (size_t)(r * 2 * H + H + c));
  v4f d = *reinterpret_cast<const v4f *>(dhid + i);
  if (dr.p > 0.f) d *= dropout4(dr.seed, dr.site, (unsigned long long)(i >> 2), dr.p);
  v4f da, dg;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float Phi = 0.5f * (1.f + erff(g[e] * 0.70710678118654752440f));
    const float phi = 0.39894228040143267794f * expf(-0.5f * g[e] * g[e]);
    da[e] = d[e] * g[e] * Phi;
    dg[e] = d[e] * a[e] * (Phi + g[e] * phi);
  }
  store4<BF>(dag, (size_t)(r * 2 * H + c), da);
  store4<BF>(dag, (size_t)(r * 2 * H + H + c), dg);
}

// out = x * factor (+ resid): the dropout behind to_out (forward: + block input; backward: the masked gradient)
__global__ void k_dropout(const float *__restrict__ x, const float *__restrict__ resid, float *__restrict__ out, long long total, Drop dr) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= total) return;
  v4f v = *reinterpret_cast<const v4f *>(x + i) * dropout4(dr.seed, dr.site, (unsigned long long)(i >> 2), dr.p);
  if (resid) v += *reinterpret_cast<const v4f *>(resid + i);
  *reinterpret_cast<v4f *>(out + i) = v;
}
__global__ void k_dropout_factors(float *__restrict__ out, long long total, Drop dr) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= total) return;
  *reinterpret_cast<v4f *>(out + i) = dropout4(dr.seed, dr.site, (unsigned long long)(i >> 2), dr.p);
}

// timestep_embedding (utils.py:7-24): [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(10000) k / 128)
__global__ void k_timestep_embedding(const int32_t *__restrict__ t, float *__restrict__ out, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * TE) return;
  const int b = i / TE, k = i % TE, half = TE / 2;
  const float f = expf(-9.21034037197618273607f * (float)(k % half) / (float)half);
  const float a = (float)t[b] * f;
  out[i] = k < half ? cosf(a) : sinf(a);
}

// ctx rows (attention.py:386-397): [(b, j)] = [part_code(b, :, j) | mean, var (b, :, j) | eye_j | t_emb(b) | 0 x 6]
__global__ void k_build_ctx(const float *__restrict__ code, const float *__restrict__ mv, const float *__restrict__ temb,
                            float *__restrict__ ctx, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * J * CTXP) return;
  const int c = i % CTXP, bj = i / CTXP, j = bj % J, b = bj / J;
  float v = 0.f;
  if (c < 256) v = code[((size_t)b * 256 + c) * J + j];
  else if (c < 262) v = mv[((size_t)b * 6 + (c - 256)) * J + j];
  else if (c < 266) v = (c - 262) == j ? 1.f : 0.f;
  else if (c < CTX) v = temb[(size_t)b * TE + (c - 266)];
  ctx[i] = v;
}
// backward of the above: d code, d mv, d t_emb[b] = sum_j d ctx[(b, j)][266:522]
__global__ void k_ctx_bwd(const float *__restrict__ dctx, float *__restrict__ dcode, float *__restrict__ dmv,
                          float *__restrict__ dtemb, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * CTX) return;
  const int c = i % CTX, b = i / CTX;
  const float *row = dctx + (size_t)b * J * CTXP + c;
  if (c < 256) {
    if (dcode)
      for (int j = 0; j < J; ++j) dcode[((size_t)b * 256 + c) * J + j] = row[(size_t)j * CTXP];
  } else if (c < 262) {
    if (dmv)
      for (int j = 0; j < J; ++j) dmv[((size_t)b * 6 + (c - 256)) * J + j] = row[(size_t)j * CTXP];
  } else if (c >= 266) {
    float a = 0.f;
    for (int j = 0; j < J; ++j) a += row[(size_t)j * CTXP];
    dtemb[(size_t)b * TE + (c - 266)] = a;
  }
}

// W (rows, cols) -> WT (cols_pad rows, ld = rows): the operand of the dX products; columns >= cols are not produced
__global__ void k_transpose(const float *__restrict__ W, float *__restrict__ WT, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads
  for (int k = ty; k < 32; k += 8)
    tile[k][tx] = (r0 + k < rows && c0 + tx < cols) ? W[(size_t)(r0 + k) * cols + c0 + tx] : 0.f;
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < cols && r0 + tx < rows) WT[(size_t)(c0 + k) * rows + r0 + tx] = tile[tx][k];
}
// W (rows, cols) -> Wp (rows, cols_pad) zero-padded (K of the shared linear kernel is consumed 8 at a time)
__global__ void k_pad_cols(const float *__restrict__ W, float *__restrict__ Wp, int rows, int cols, int cols_pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols_pad) return;
  const int c = i % cols_pad, r = i / cols_pad;
  Wp[i] = c < cols ? W[(size_t)r * cols + c] : 0.f;
}

// to_k / to_v of all blocks side by side: Wkv (2 depth x 128 rows, 522 -> 528 columns, zero padded) so that the keys and values of
// every block come from ONE product over the context tokens (the context is the same for all blocks), and their gradients back
struct KvPtrs {
  const float *p[2 * DFX_MAX_DEPTH];
};
struct KvMutPtrs {
  float *p[2 * DFX_MAX_DEPTH];
};
__global__ void k_pack_kv(KvPtrs src, float *__restrict__ dst, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C * CTXP) return;
  const int c = idx % CTXP, row = idx / CTXP;
  dst[idx] = c < CTX ? src.p[row / C][(size_t)(row % C) * CTX + c] : 0.f;
}
__global__ void k_unpack_kv(const float *__restrict__ src, KvMutPtrs dst, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C * CTX) return;
  const int c = idx % CTX, row = idx / CTX;
  dst.p[row / C][(size_t)(row % C) * CTX + c] = src[(size_t)row * CTXP + c];
}

// ---------------------------------------------------------------------------------------------------------------
// cross attention to the 4 part tokens (attention.py:179-204), forward with saved probabilities, and backward
// block = 32 points of one shape x 8 heads; k, v of the shape in LDS
// ---------------------------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(256) void k_attn_fwd(const float *__restrict__ q, const float *__restrict__ k,
                                                   const float *__restrict__ v, const float *__restrict__ valid,
                                                   float *__restrict__ p, float *__restrict__ att, int N) {
  __shared__ float ks[J][C], vs[J][C];
  const int b = blockIdx.y, h = threadIdx.x >> 5, i = threadIdx.x & 31;
  for (int e = threadIdx.x; e < J * C; e += 256) ks[e / C][e % C] = k[(size_t)b * J * C + e], vs[e / C][e % C] = v[(size_t)b * J * C + e];
  __syncthreads();
  const long long r = (long long)b * N + blockIdx.x * 32 + i;
  float qv[HD];
#pragma unroll
  for (int e = 0; e < HD / 4; ++e) {
    const v4f t = reinterpret_cast<const v4f *>(q + r * C + h * HD)[e];
    qv[4 * e] = t[0], qv[4 * e + 1] = t[1], qv[4 * e + 2] = t[2], qv[4 * e + 3] = t[3];
  }
  float sim[J], mx = -3.402823466e38f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < HD; ++e) s += qv[e] * ks[j][h * HD + e];
    s *= 0.25f;   // dim_head ** -0.5
    if (valid && valid[b * J + j] == 0.f) s = -3.402823466e38f;   // masked_fill_(~mask, -finfo.max)
    sim[j] = s;
    mx = fmaxf(mx, s);
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) sim[j] = expf(sim[j] - mx), den += sim[j];
  float o[HD];
#pragma unroll
  for (int e = 0; e < HD; ++e) o[e] = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    sim[j] /= den;
#pragma unroll
    for (int e = 0; e < HD; ++e) o[e] += sim[j] * vs[j][h * HD + e];
  }
  reinterpret_cast<v4f *>(p + r * (HEADS * J))[h] = v4f{sim[0], sim[1], sim[2], sim[3]};
#pragma unroll
  for (int e = 0; e < HD / 4; ++e) store4<BF>(att, (size_t)r * C + h * HD + 4 * e, v4f{o[4 * e], o[4 * e + 1], o[4 * e + 2], o[4 * e + 3]});
}

// d att -> d q (per point), d k / d v partial sums over the block's points -> part[(b, blockIdx.x)][2][J][C].
// A block covers ATT_BP = 128 points of one shape: every lane walks four points, keeping its 2 x 4 x 16 d k / d v
// contributions in registers, and the cross-lane reduction over the 32 lanes of a head runs once per block, not per point.
constexpr int ATT_BP = 128;
template <bool BF>
__global__ __launch_bounds__(256) void k_attn_bwd(const float *__restrict__ datt, const float *__restrict__ q,
                                                   const float *__restrict__ k, const float *__restrict__ v,
                                                   const float *__restrict__ p, float *__restrict__ dq,
                                                   float *__restrict__ part, int N) {
  __shared__ float ks[J][C], vs[J][C];
  const int b = blockIdx.y, h = threadIdx.x >> 5, i = threadIdx.x & 31;
  for (int e = threadIdx.x; e < J * C; e += 256) ks[e / C][e % C] = k[(size_t)b * J * C + e], vs[e / C][e % C] = v[(size_t)b * J * C + e];
  __syncthreads();
  float ak[J][HD], av[J][HD];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int e = 0; e < HD; ++e) ak[j][e] = 0.f, av[j][e] = 0.f;
  for (int pt = 0; pt < ATT_BP / 32; ++pt) {
    const int n = blockIdx.x * ATT_BP + pt * 32 + i;
    if (n >= N) break;   // N is a multiple of 32: whole groups of 32 lanes drop out together
    const long long r = (long long)b * N + n;
    float qv[HD], dv_[HD];
#pragma unroll
    for (int e = 0; e < HD / 4; ++e) {
      const v4f t = reinterpret_cast<const v4f *>(q + r * C + h * HD)[e], u = reinterpret_cast<const v4f *>(datt + r * C + h * HD)[e];
#pragma unroll
      for (int s = 0; s < 4; ++s) qv[4 * e + s] = t[s], dv_[4 * e + s] = u[s];
    }
    const v4f pv = reinterpret_cast<const v4f *>(p + r * (HEADS * J))[h];
    float dp[J], dot = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < HD; ++e) s += dv_[e] * vs[j][h * HD + e];
      dp[j] = s;
      dot += pv[j] * s;
    }
    float dqv[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) dqv[e] = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float ds = pv[j] * (dp[j] - dot) * 0.25f;   // d sim_j x scale (masked keys: p = 0 exactly)
#pragma unroll
      for (int e = 0; e < HD; ++e) {
        dqv[e] += ds * ks[j][h * HD + e];
        ak[j][e] += ds * qv[e];
        av[j][e] += pv[j] * dv_[e];
      }
    }
#pragma unroll
    for (int e = 0; e < HD / 4; ++e) store4<BF>(dq, (size_t)r * C + h * HD + 4 * e, v4f{dqv[4 * e], dqv[4 * e + 1], dqv[4 * e + 2], dqv[4 * e + 3]});
  }
  float *pk = part + ((size_t)(b * gridDim.x + blockIdx.x) * 2) * J * C;
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int e = 0; e < HD; ++e) {
      float a = ak[j][e], c = av[j][e];
      for (int o = 16; o; o >>= 1) a += __shfl_xor(a, o, 32), c += __shfl_xor(c, o, 32);
      if (i == 0) pk[j * C + h * HD + e] = a, pk[(J + j) * C + h * HD + e] = c;
    }
}
// d k, d v (B J, C) = sum over the nb blocks of a shape
__global__ void k_attn_bwd_finish(const float *__restrict__ part, float *__restrict__ dk, float *__restrict__ dv, int nb) {
  const int b = blockIdx.x, t = threadIdx.x;   // 512 threads = J * C
  float a = 0.f, c = 0.f;
  for (int q = 0; q < nb; ++q) {
    const float *pk = part + ((size_t)(b * nb + q) * 2) * J * C;
    a += pk[t];
    c += pk[J * C + t];
  }
  dk[(size_t)b * J * C + t] = a;
  dv[(size_t)b * J * C + t] = c;
}

// ---------------------------------------------------------------------------------------------------------------
// proj_out (3 x 128): eps (B, 3, N) = hn W^T + b; backward d hn, partial d W / d b
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_eps_fwd(const float *__restrict__ hn, const float *__restrict__ W,
                                                  const float *__restrict__ bias, float *__restrict__ eps, int N, long long R) {
  const int l = threadIdx.x & 31;
  const long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (r >= R) return;
  const v4f v = reinterpret_cast<const v4f *>(hn + r * C)[l];
  float s[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const v4f w = reinterpret_cast<const v4f *>(W + c * C)[l];
    float a = v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
    for (int o = 16; o; o >>= 1) a += __shfl_xor(a, o, 32);
    s[c] = a + bias[c];
  }
  if (l < 3) eps[((r / N) * 3 + l) * N + (r % N)] = s[l];
}
constexpr int EPSB_ROWS = 256;
__global__ __launch_bounds__(128) void k_eps_bwd(const float *__restrict__ deps, const float *__restrict__ hn,
                                                  const float *__restrict__ W, float *__restrict__ dhn,
                                                  float *__restrict__ part /*[blk][4][128]*/, int N, long long R) {
  const int kcol = threadIdx.x;
  const float w0 = W[kcol], w1 = W[C + kcol], w2 = W[2 * C + kcol];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
  for (int it = 0; it < EPSB_ROWS; ++it) {
    const long long r = (long long)blockIdx.x * EPSB_ROWS + it;
    if (r >= R) break;
    const long long b = r / N;
    const int n = (int)(r % N);
    const float d0 = deps[(b * 3 + 0) * N + n], d1 = deps[(b * 3 + 1) * N + n], d2 = deps[(b * 3 + 2) * N + n];
    const float h = hn[r * C + kcol];
    dhn[r * C + kcol] = d0 * w0 + d1 * w1 + d2 * w2;
    a0 += d0 * h, a1 += d1 * h, a2 += d2 * h;
    b0 += d0, b1 += d1, b2 += d2;
  }
  float *pp = part + (size_t)blockIdx.x * 4 * C;
  pp[kcol] = a0, pp[C + kcol] = a1, pp[2 * C + kcol] = a2;
  if (kcol < 3) pp[3 * C + kcol] = kcol == 0 ? b0 : kcol == 1 ? b1 : b2;
}

// d pred of the masked MSE (anchored_diffusion.py:840-847): gs 2 (pred - target) flag / (3 sum(flags)), or / (3 B N)
__global__ void k_mse_bwd(const float *__restrict__ target, const float *__restrict__ pred, const float *__restrict__ flags,
                          const double *__restrict__ acc, float gs, float *__restrict__ dpred, int N, long long total,
                          double count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long b = i / N;
  const int n = (int)(i % N);
  const float fl = flags ? flags[i] : 1.f;
  const float sc = flags ? (float)(2.0 * gs / (3.0 * acc[1])) : (float)(2.0 * gs / count);
  for (int c = 0; c < 3; ++c) {
    const long long o = (b * 3 + c) * N + n;
    dpred[o] = sc * (pred[o] - target[o]) * fl;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The two ends of the network as one pass each way (fused training path): nothing of width 128 is written that the next kernel
// only reads back.  32 lanes per point (lane l = channels 4 l .. 4 l + 3), 8 points per 256-thread step, like k_ln_fwd.
//   stem   hin = pre_norm(W_in xin + b_in)                          backward: recomputes h0 from xin; d W_in, d b_in, d gamma, d beta
//   head   eps = W_out post_norm(hfin) + b_out  -> (B, 3, N)        backward: recomputes the LayerNorm; dh, d W_out, d b_out, d gamma, d beta
// ---------------------------------------------------------------------------------------------------------------
constexpr int STEM_ROWS = 512, HEAD_ROWS = 128;   // points per block of the backward kernels (rows of their partial sums; the stem's partial is 8 KiB)
// Sum over the 32 lanes of a half-wave, in every lane.  Four steps inside the 16-lane rows are DPP modifiers of the adds (quad_perm [1,0,3,2] and
// [2,3,0,1], row_half_mirror, row_mirror: no LDS crossbar), one ds_swizzle exchanges the two rows — the five ds_bpermute round trips of a __shfl_xor
// butterfly were the latency chain of the stem / head kernels (four sums per pair of rows).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum32_xbar(float v) {   // the butterfly through the LDS crossbar: k_stem_bwd, whose VALU is the busier pipe (DPP form: 117 vs 107 us)
  for (int o = 16; o; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}
__device__ __forceinline__ float sum32(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
  return v;
}
struct StemW {   // this lane's four rows of W_in (13 columns) and bias
  float w[4][13], b[4];
};
__device__ __forceinline__ void stem_load(StemW &sw, const float *__restrict__ W, const float *__restrict__ b, int l) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int k = 0; k < 13; ++k) sw.w[e][k] = W[(4 * l + e) * 13 + k];
    sw.b[e] = b[4 * l + e];
  }
}
template <bool XBAR = false>
__device__ __forceinline__ void stem_row(const StemW &sw, const float *__restrict__ xrow, float (&x)[13], v4f &h0, float &mu, float &rstd) {
  const v4f a = *reinterpret_cast<const v4f *>(xrow), b = *reinterpret_cast<const v4f *>(xrow + 4), c = *reinterpret_cast<const v4f *>(xrow + 8);
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = a[k], x[4 + k] = b[k], x[8 + k] = c[k];
  x[12] = xrow[12];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = sw.b[e];
#pragma unroll
    for (int k = 0; k < 13; ++k) t = fmaf(sw.w[e][k], x[k], t);
    h0[e] = t;
  }
  const float s0 = h0[0] + h0[1] + h0[2] + h0[3];
  mu = (XBAR ? sum32_xbar(s0) : sum32(s0)) * (1.0f / C);
  const v4f d = {h0[0] - mu, h0[1] - mu, h0[2] - mu, h0[3] - mu};
  const float q0 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  rstd = 1.0f / sqrtf((XBAR ? sum32_xbar(q0) : sum32(q0)) * (1.0f / C) + LN_EPS);
}
__global__ __launch_bounds__(256) void k_stem_fwd(const float *__restrict__ xin, const float *__restrict__ W, const float *__restrict__ b,
                                                   const float *__restrict__ g, const float *__restrict__ be, float *__restrict__ hin, long long R) {
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  StemW sw;
  stem_load(sw, W, b, l);
  const v4f gv = reinterpret_cast<const v4f *>(g)[l], bv = reinterpret_cast<const v4f *>(be)[l];
  for (long long r = (long long)blockIdx.x * 8 + grp; r < R; r += (long long)gridDim.x * 8) {
    float x[13], mu, rstd;
    v4f h0;
    stem_row(sw, xin + r * XIN, x, h0, mu, rstd);
    v4f o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (h0[e] - mu) * rstd * gv[e] + bv[e];
    reinterpret_cast<v4f *>(hin + r * C)[l] = o;
  }
}
// Gradient at the denoiser's INPUT rows, for the two columns groups a caller may differentiate (stage 2 of the reference: `variance` reaches
// training_losses undetached, anchor_gen.py:1002-1020 — it enters through q_sample's sqrt(variance) * noise term of x_t and through the per-point
// feature columns): d x (B, 3, N) = columns 0..2, d variances (B, N, 3) = columns 6..8 of  d h0 W_in,  d h0 = LayerNorm'(dy) recomputed like k_stem_bwd.
__global__ __launch_bounds__(256) void k_stem_dx(const float *__restrict__ dy, const float *__restrict__ xin, const float *__restrict__ W,
                                                  const float *__restrict__ b, const float *__restrict__ g, float *__restrict__ d_x,
                                                  float *__restrict__ d_var, int N, long long R) {
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  StemW sw;
  stem_load(sw, W, b, l);
  const v4f gv = reinterpret_cast<const v4f *>(g)[l];
  for (long long r = (long long)blockIdx.x * 8 + grp; r < R; r += (long long)gridDim.x * 8) {
    float x[13], mu, rstd;
    v4f h0;
    stem_row(sw, xin + r * XIN, x, h0, mu, rstd);
    const v4f dv = reinterpret_cast<const v4f *>(dy + r * C)[l];
    v4f xh, dyg;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[e] = (h0[e] - mu) * rstd;
      dyg[e] = dv[e] * gv[e];
      s1 += dyg[e], s2 += dyg[e] * xh[e];
    }
    s1 = sum32(s1) * (1.0f / C), s2 = sum32(s2) * (1.0f / C);
    float px[3] = {0.f, 0.f, 0.f}, pv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = rstd * (dyg[e] - s1 - xh[e] * s2);   // gradient at h0
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] = fmaf(d0, sw.w[e][c], px[c]), pv[c] = fmaf(d0, sw.w[e][6 + c], pv[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) px[c] = sum32(px[c]), pv[c] = sum32(pv[c]);
    if (l == 0) {
      const long long bb = r / N;
      const int n = (int)(r % N);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (d_x) d_x[(bb * 3 + c) * N + n] = px[c];
        if (d_var) d_var[r * 3 + c] = pv[c];
      }
    }
  }
}
// part[block] = [d W_in (128 x 13) | d b_in (128) | d gamma (128) | d beta (128)] = 2048 floats
__global__ __launch_bounds__(256) void k_stem_bwd(const float *__restrict__ dy, const float *__restrict__ xin, const float *__restrict__ W,
                                                   const float *__restrict__ b, const float *__restrict__ g, float *__restrict__ part, long long R) {
  __shared__ float red[4][2048];
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  StemW sw;
  stem_load(sw, W, b, l);
  const v4f gv = reinterpret_cast<const v4f *>(g)[l];
  float dw[4][13], db[4], dg[4], dbe[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int k = 0; k < 13; ++k) dw[e][k] = 0.f;
    db[e] = dg[e] = dbe[e] = 0.f;
  }
#pragma unroll 2
  for (int it = 0; it < STEM_ROWS / 8; ++it) {
    const long long r = (long long)blockIdx.x * STEM_ROWS + it * 8 + grp;
    if (r >= R) break;
    float x[13], mu, rstd;
    v4f h0;
    stem_row<true>(sw, xin + r * XIN, x, h0, mu, rstd);
    const v4f dv = reinterpret_cast<const v4f *>(dy + r * C)[l];
    v4f xh, dyg;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[e] = (h0[e] - mu) * rstd;
      dyg[e] = dv[e] * gv[e];
      s1 += dyg[e], s2 += dyg[e] * xh[e];
      dg[e] += dv[e] * xh[e], dbe[e] += dv[e];
    }
    s1 = sum32_xbar(s1) * (1.0f / C), s2 = sum32_xbar(s2) * (1.0f / C);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = rstd * (dyg[e] - s1 - xh[e] * s2);   // gradient at h0
      db[e] += d0;
#pragma unroll
      for (int k = 0; k < 13; ++k) dw[e][k] = fmaf(d0, x[k], dw[e][k]);
    }
  }
  // eight row groups -> four LDS rows (groups 4..7 first, groups 0..3 add theirs), then a fixed-order sum
#pragma unroll
  for (int half = 1; half >= 0; --half) {
    if ((grp >> 2) == half) {
      float *rd = red[grp & 3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int k = 0; k < 13; ++k) rd[(4 * l + e) * 13 + k] = (half ? 0.f : rd[(4 * l + e) * 13 + k]) + dw[e][k];
        rd[1664 + 4 * l + e] = (half ? 0.f : rd[1664 + 4 * l + e]) + db[e];
        rd[1792 + 4 * l + e] = (half ? 0.f : rd[1792 + 4 * l + e]) + dg[e];
        rd[1920 + 4 * l + e] = (half ? 0.f : rd[1920 + 4 * l + e]) + dbe[e];
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 2048; i += 256) part[(size_t)blockIdx.x * 2048 + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
__device__ __forceinline__ void ln_row(const float *__restrict__ xrow, int l, v4f &xh, float &rstd) {
  const v4f v = reinterpret_cast<const v4f *>(xrow)[l];
  const float mu = sum32(v[0] + v[1] + v[2] + v[3]) * (1.0f / C);
  const v4f d = {v[0] - mu, v[1] - mu, v[2] - mu, v[3] - mu};
  rstd = 1.0f / sqrtf(sum32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / C) + LN_EPS);
  xh = d * rstd;
}
__global__ __launch_bounds__(256) void k_head_fwd(const float *__restrict__ hfin, const float *__restrict__ g, const float *__restrict__ be,
                                                   const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ eps, int N, long long R) {
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const v4f gv = reinterpret_cast<const v4f *>(g)[l], bv = reinterpret_cast<const v4f *>(be)[l];
  const v4f w0 = reinterpret_cast<const v4f *>(W)[l], w1 = reinterpret_cast<const v4f *>(W + C)[l], w2 = reinterpret_cast<const v4f *>(W + 2 * C)[l];
  for (long long r = (long long)blockIdx.x * 8 + grp; r < R; r += (long long)gridDim.x * 8) {
    v4f xh;
    float rstd;
    ln_row(hfin + r * C, l, xh, rstd);
    const v4f hn = xh * gv + bv;
    const float s0 = sum32(hn[0] * w0[0] + hn[1] * w0[1] + hn[2] * w0[2] + hn[3] * w0[3]);
    const float s1 = sum32(hn[0] * w1[0] + hn[1] * w1[1] + hn[2] * w1[2] + hn[3] * w1[3]);
    const float s2 = sum32(hn[0] * w2[0] + hn[1] * w2[1] + hn[2] * w2[2] + hn[3] * w2[3]);
    if (l < 3) eps[((r / N) * 3 + l) * N + (r % N)] = (l == 0 ? s0 : l == 1 ? s1 : s2) + bias[l];
  }
}
// part[block] = [d W_out (3 x 128) | d gamma (128) | d beta (128) | d b_out (3) | 0 x 29] = 672 floats
constexpr int HEAD_PART = 672;
__global__ __launch_bounds__(256) void k_head_bwd(const float *__restrict__ deps, const float *__restrict__ hfin, const float *__restrict__ g,
                                                   const float *__restrict__ be, const float *__restrict__ W, float *__restrict__ dh,
                                                   float *__restrict__ part, int N, long long R, bool hl) {
  // hl: dh leaves as the bf16-pair tiles the fused backward kernels pass between the blocks (train_ff_fused.h, TL_DH_HL) instead of fp32 rows
  __shared__ float red[8][HEAD_PART];
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const v4f gv = reinterpret_cast<const v4f *>(g)[l], bv = reinterpret_cast<const v4f *>(be)[l];
  const v4f w0 = reinterpret_cast<const v4f *>(W)[l], w1 = reinterpret_cast<const v4f *>(W + C)[l], w2 = reinterpret_cast<const v4f *>(W + 2 * C)[l];
  v4f a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, dg = a0, dbe = a0;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f;
  for (int it = 0; it < HEAD_ROWS / 8; ++it) {
    const long long r = (long long)blockIdx.x * HEAD_ROWS + it * 8 + grp;
    if (r >= R) break;
    const long long bb = r / N;
    const int n = (int)(r % N);
    const float d0 = deps[(bb * 3 + 0) * N + n], d1 = deps[(bb * 3 + 1) * N + n], d2 = deps[(bb * 3 + 2) * N + n];
    v4f xh;
    float rstd;
    if (!hl) {
      ln_row(hfin + r * C, l, xh, rstd);
    } else {   // the last block's forward kernel left the normalised row as bf16 fragments + 1 / std (train_ff_fused.h, TL_HEAD): this thread's four channels
      const char *tile = reinterpret_cast<const char *>(hfin + (r & ~31LL) * C);
      const int pj = (int)(r & 31), c = l >> 3;
      const uint2 hv = *reinterpret_cast<const uint2 *>(tile + ((c * 2 + ((l >> 2) & 1)) * 64 + pj + 32 * ((l >> 1) & 1)) * 16 + 8 * (l & 1));
      xh = v4f{__builtin_bit_cast(float, hv.x << 16), __builtin_bit_cast(float, hv.x & 0xffff0000u), __builtin_bit_cast(float, hv.y << 16),
               __builtin_bit_cast(float, hv.y & 0xffff0000u)};
      rstd = reinterpret_cast<const float *>(tile)[dfx::ffused::H1F_RSTD + pj];
    }
    const v4f hn = xh * gv + bv;
    const v4f dhn = d0 * w0 + d1 * w1 + d2 * w2;
    a0 += d0 * hn, a1 += d1 * hn, a2 += d2 * hn;
    b0 += d0, b1 += d1, b2 += d2;
    dg += dhn * xh, dbe += dhn;
    const v4f dyg = dhn * gv;
    const float s1 = sum32(dyg[0] + dyg[1] + dyg[2] + dyg[3]) * (1.0f / C);
    const float s2 = sum32(dyg[0] * xh[0] + dyg[1] * xh[1] + dyg[2] * xh[2] + dyg[3] * xh[3]) * (1.0f / C);
    const v4f o = rstd * (dyg - s1 - xh * s2);
    if (!hl) {
      reinterpret_cast<v4f *>(dh + r * C)[l] = o;
    } else {
      // this thread's channels 4 l .. 4 l + 3 of point pj = r % 32: hi = element 4 (l & 1) .. of block (c, u) = (l >> 3, (l >> 2) & 1), half-wave (l >> 1) & 1;
      // lo = registers 4 q .. 4 q + 3, q = (l >> 1) & 3, of accumulator tile c in half-wave l & 1 (block (c, q >> 1), second half of the tile)
      typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
      const v4bf hi = __builtin_convertvector(o, v4bf);
      const v4f rem = o - __builtin_convertvector(hi, v4f);
      const v4bf lo = __builtin_convertvector(rem, v4bf);
      char *tile = reinterpret_cast<char *>(dh + (r & ~31LL) * C);
      const int pj = (int)(r & 31), c = l >> 3, q = (l >> 1) & 3;
      *reinterpret_cast<uint2 *>(tile + ((c * 2 + ((l >> 2) & 1)) * 64 + pj + 32 * ((l >> 1) & 1)) * 16 + 8 * (l & 1)) = __builtin_bit_cast(uint2, hi);
      *reinterpret_cast<uint2 *>(tile + 8192 + ((c * 2 + (q >> 1)) * 64 + pj + 32 * (l & 1)) * 16 + 8 * (q & 1)) = __builtin_bit_cast(uint2, lo);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[grp][4 * l + e] = a0[e], red[grp][C + 4 * l + e] = a1[e], red[grp][2 * C + 4 * l + e] = a2[e];
    red[grp][3 * C + 4 * l + e] = dg[e], red[grp][4 * C + 4 * l + e] = dbe[e];
  }
  if (l == 0) red[grp][5 * C] = b0, red[grp][5 * C + 1] = b1, red[grp][5 * C + 2] = b2;
  if (l < 29) red[grp][5 * C + 3 + l] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < HEAD_PART; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][i];
    part[(size_t)blockIdx.x * HEAD_PART + i] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW (O, I) = dY^T X over R rows, db = column sums of dY
// grid (ceil(I/64), ceil(O/64), nslab); part[slab][O][I] (+ bpart[slab][O] from blockIdx.x == 0).  NW wavefronts per block share
// the tile: each takes a quarter of the slab's rows and the accumulators meet in LDS (fixed order: bit-reproducible) — the products
// over all 262 144 points had 2 wavefronts per SIMD, each waiting ~2 us for operands that come from HBM (40 % of the matrix peak);
// four times the wavefronts for the same number of partial tiles.
// ---------------------------------------------------------------------------------------------------------------
template <int NW, int NI = 2>   // NI = 32-column blocks of the tile along I: 2 (64 x 64) or 4 (64 x 128: every dY fragment feeds four MFMAs instead of two)
__device__ __forceinline__ void wgrad_tile(const float *__restrict__ dY, int ldy, const float *__restrict__ X, int ldx,
                                           float *__restrict__ pp, float *__restrict__ bp, int O, int I, long long r0, long long r1,
                                           int bx, int by) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, hf = lane >> 5;
  if (NW > 1) {   // this wavefront's share of the rows (an even number: rows go two per MFMA)
    const long long q = ((r1 - r0 + NW - 1) / NW + 1) & ~1ll;
    r0 = r0 + wave * q < r1 ? r0 + wave * q : r1;
    r1 = r0 + q < r1 ? r0 + q : r1;
  }
  const int i0 = bx * 32 * NI, o0 = by * 64;
  const bool oa = o0 + j < O, ob = o0 + 32 + j < O;
  const float *py = dY + o0 + j, *px = X + i0 + j;
  v16f acc[2][NI];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bs0 = 0.f, bs1 = 0.f;
  // eight rows (4 + 2 NI loads per lane, 8 NI MFMAs) per block; the loads of block q + 2 are issued before the MFMAs of block q (a ring of three register
  // sets; requests behind the last block re-read it, so that every path has the same number of loads in flight)
  struct Rows {
    float ya[4], yb[4], x[NI][4];
  };
  // (row base in scalar registers + a per-lane 32-bit offset that never changes: with the row index per lane the address arithmetic
  // was most of the 9 VALU instructions per MFMA of this kernel)
  // Columns past the edge are not predicated either: they are read from the last valid column instead, and what they contribute
  // lands only in rows / columns of the tile (and bias sums) that are never stored.
  const unsigned offy = (unsigned)(hf * ldy + min(o0 + j, O - 1)), offy2 = (unsigned)(hf * ldy + min(o0 + 32 + j, O - 1));
  unsigned offx[NI];
#pragma unroll
  for (int b = 0; b < NI; ++b) offx[b] = (unsigned)(hf * ldx + min(i0 + 32 * b + j, I - 1));
  auto load8 = [&](long long r, Rows &w) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float *ry = dY + (r + 2 * u) * ldy, *rx = X + (r + 2 * u) * ldx;
      w.ya[u] = ry[offy], w.yb[u] = ry[offy2];
#pragma unroll
      for (int b = 0; b < NI; ++b) w.x[b][u] = rx[offx[b]];
    }
  };
  auto mm8 = [&](const Rows &w) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int b = 0; b < NI; ++b) acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.ya[u], w.x[b][u], acc[0][b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NI; ++b) acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.yb[u], w.x[b][u], acc[1][b], 0, 0, 0);
      bs0 += w.ya[u], bs1 += w.yb[u];
    }
  };
  const long long nfull = (r1 - r0) / 8;
  long long r = r0 + 8 * nfull;
  if (nfull > 0) {
    // the row blocks come from HBM / the L2: requested RD - 1 blocks ahead through a ring of RD register sets
    constexpr int RD = 3;   // (same box, PointNetV2 training forward + backward: two sets 4.51 ms, three 4.48, four 4.48; the same sums in the same order)
    Rows ring[RD];
#pragma unroll
    for (int k = 0; k < RD - 1; ++k) load8(r0 + 8 * (k < nfull ? k : nfull - 1), ring[k]);
    long long q = 0;
    for (; q + RD <= nfull; q += RD) {
#pragma unroll
      for (int k = 0; k < RD; ++k) {
        const long long nx = q + k + RD - 1;
        load8(r0 + 8 * (nx < nfull ? nx : nfull - 1), ring[(k + RD - 1) % RD]);
        __builtin_amdgcn_sched_barrier(0);
        mm8(ring[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < RD - 1; ++k) {
      if (q + k < nfull) mm8(ring[k]);   // (the tail's blocks are in the ring already)
    }
  }
  for (; r < r1; r += 2) {
    const long long rr = r + hf;
    const bool in = rr < r1;
    const float ya = (in && oa) ? py[rr * ldy] : 0.f, yb = (in && ob) ? py[rr * ldy + 32] : 0.f;
#pragma unroll
    for (int b = 0; b < NI; ++b) {
      const float xb = (in && i0 + 32 * b + j < I) ? px[rr * ldx + 32 * b] : 0.f;
      acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya, xb, acc[0][b], 0, 0, 0);
      acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(yb, xb, acc[1][b], 0, 0, 0);
    }
    bs0 += ya, bs1 += yb;
  }
  if (NW > 1) {   // wavefronts 2, 3 -> 0, 1, then 1 -> 0 (two 17 KiB slots), two column blocks of the tile per pass
    __shared__ float red[2][4096 + 128];
#pragma unroll
    for (int b0 = 0; b0 < NI; b0 += 2) {
      auto put = [&](float *d) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) d[((a * 2 + b) * 16 + rg) * 64 + lane] = acc[a][b0 + b][rg];
        if (b0 == 0) d[4096 + lane] = bs0, d[4096 + 64 + lane] = bs1;
      };
      auto add = [&](const float *d) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) acc[a][b0 + b][rg] += d[((a * 2 + b) * 16 + rg) * 64 + lane];
        if (b0 == 0) bs0 += d[4096 + lane], bs1 += d[4096 + 64 + lane];
      };
      if (b0 > 0) __syncthreads();   // wavefront 0 is done with red[0] of the previous pass
      if (wave >= 2) put(red[wave - 2]);
      __syncthreads();
      if (wave < 2) add(red[wave]);
      __syncthreads();
      if (wave == 1) put(red[0]);
      __syncthreads();
      if (wave == 0) add(red[0]);
    }
    if (wave != 0) return;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NI; ++b) {
      const int i = i0 + 32 * b + j;   // C/D layout: column = lane % 32 (the B operand's index), row = (r&3) + 8(r>>2) + 4 hf
      if (i >= I) continue;
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int o = o0 + 32 * a + (rg & 3) + 8 * (rg >> 2) + 4 * hf;
        if (o < O) pp[(size_t)o * I + i] = acc[a][b][rg];
      }
    }
  if (bp && bx == 0) {
    bs0 += __shfl_xor(bs0, 32), bs1 += __shfl_xor(bs1, 32);
    if (hf == 0) {
      if (oa) bp[o0 + j] = bs0;
      if (ob) bp[o0 + 32 + j] = bs1;
    }
  }
}
template <int NI>
__global__ __launch_bounds__(256) void k_wgrad(const float *__restrict__ dY, int ldy, const float *__restrict__ X, int ldx,
                                               float *__restrict__ part, float *__restrict__ bpart, int O, int I,
                                               long long R, int rows_per_slab) {
  // Every 64 x 64 tile of a slab reads the slab's dY / X strips again (dY I / 64 times, X O / 64 times: 4.3 GB for the 512 x 256
  // product over 262 144 rows).  Workgroups go to the 8 XCDs round-robin by their linear id, each XCD with its own L2: a slab's tiles
  // are given ids of ONE residue mod 8, so that the re-reads hit that XCD's L2 instead of HBM.
  const int tiles = gridDim.x * gridDim.y;
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  int slab = lin / tiles, t = lin % tiles;
  if (gridDim.z % 8 == 0) {
    const int q = lin >> 3;
    slab = (q / tiles) * 8 + (lin & 7), t = q % tiles;
  }
  const long long r0 = (long long)slab * rows_per_slab;
  const long long r1 = r0 + rows_per_slab < R ? r0 + rows_per_slab : R;
  wgrad_tile<4, NI>(dY, ldy, X, ldx, part + (size_t)slab * O * I, bpart ? bpart + (size_t)slab * O : nullptr, O, I, r0, r1, t % gridDim.x, t / gridDim.x);
}
// Up to four independent few-row products in one launch (blockIdx.z = group): operands at uniform group strides, results
// through pointer tables (the per-part flows' parameters are separate tensors); all R rows in one slab, written directly
struct Ptr4 {
  float *p[4];
};
__global__ __launch_bounds__(256) void k_wgrad_g4(const float *__restrict__ dY, int ldy, long long dy_gs, const float *__restrict__ X,
                                                  int ldx, long long x_gs, Ptr4 dW, Ptr4 db, int O, int I, long long R) {
  const int g = blockIdx.z;
  wgrad_tile<4>(dY + g * dy_gs, ldy, X + g * x_gs, ldx, dW.p[g], db.p[g], O, I, 0, R, blockIdx.x, blockIdx.y);
}
struct CPtr4 {
  const float *p[4];
};
// W_g (rows, cols) -> WT + g * wt_gs (cols, rows) for up to four groups
__global__ void k_transpose_g4(CPtr4 W, float *__restrict__ WT, long long wt_gs, int rows, int cols) {
  __shared__ float tile[32][33];
  const float *Wg = W.p[blockIdx.z];
  float *T = WT + blockIdx.z * wt_gs;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) tile[k][tx] = (r0 + k < rows && c0 + tx < cols) ? Wg[(size_t)(r0 + k) * cols + c0 + tx] : 0.f;
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < cols && r0 + tx < rows) T[(size_t)(c0 + k) * rows + r0 + tx] = tile[tx][k];
}
// dW[o][i < I_valid] = sum over slabs; ld of dW = I_valid.  Eight groups of 32 lanes share the slabs of 32 outputs (group q adds
// slabs q, q+8, ... in order, the eight group sums are added in order: a fixed tree like k_sum_parts) — one thread per output
// walking all slabs took 290 us for proj_in's 1024 slabs.
__global__ __launch_bounds__(256) void k_wgrad_finish(const float *__restrict__ part, float *__restrict__ dW, int nslab, int O, int I,
                                                       int I_valid) {
  __shared__ float red[8][32];
  const int idx = blockIdx.x * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
  const bool ok = idx < O * I_valid;
  const int o = ok ? idx / I_valid : 0, i = ok ? idx % I_valid : 0;
  float a = 0.f;
  if (ok)
    for (int s = q; s < nslab; s += 8) a += part[((size_t)s * O + o) * I + i];
  red[q][threadIdx.x & 31] = a;
  __syncthreads();
  if (q == 0 && ok) {
    float t = red[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][threadIdx.x];
    dW[idx] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// optimizer (Runner.train: clip_grad_norm_(10) -> Adam.step, torch.optim.Adam defaults of optimizers.py:4-16)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_sumsq(const float *__restrict__ g, long long n, double *__restrict__ acc_part) {
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += (double)g[i] * g[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) acc_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// one wavefront: lane l adds parts l, l + 64, .. in order, then a fixed xor tree over the lanes
__global__ void k_sumsq_finish(const double *__restrict__ acc_part, int n, double *__restrict__ total /* += */) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += acc_part[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) *total += s;
}
// p, m, v updated in place; grads scaled by clip = min(1, max_norm / (norm + 1e-6)) read from the device
__global__ void k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                       long long n, const double *__restrict__ sumsq, float max_norm, float lr, float beta1, float beta2,
                       float eps, float weight_decay, float bc1, float bc2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float clip = 1.f;
  if (max_norm > 0.f) {
    const float norm = (float)sqrt(*sumsq);
    clip = fminf(1.f, max_norm / (norm + 1e-6f));
  }
  float gi = g[i] * clip;
  if (weight_decay != 0.f) gi += weight_decay * p[i];
  const float mi = beta1 * m[i] + (1.f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
  m[i] = mi, v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] = p[i] - (lr / bc1) * (mi / denom);
}

// ---------------------------------------------------------------------------------------------------------------
// PointNetV2 part encoder in train mode (pointnet.py:187-213): BatchNorm with batch statistics over the rows, masked max-pool
// ---------------------------------------------------------------------------------------------------------------
// Column sums over a slab of rows: MODE 0: sum x;  MODE 1: sum (x - mean)^2  (the second pass of the two-pass variance).
// grid (ceil(Cc / 64), nslab), 256 threads = 64 columns x 4 row groups; part[slab][Cc]
constexpr int BN_SLAB = 512;
template <int MODE>
__global__ __launch_bounds__(256) void k_col_stats(const float *__restrict__ x, const float *__restrict__ mean, float *__restrict__ part,
                                                    long long R, int Cc) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rgp = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float a = 0.f;
  if (c < Cc) {
    const float mu = MODE ? mean[c] : 0.f;
    const long long r1 = ((long long)blockIdx.y + 1) * BN_SLAB < R ? ((long long)blockIdx.y + 1) * BN_SLAB : R;
    for (long long r = (long long)blockIdx.y * BN_SLAB + rgp; r < r1; r += 4) {
      const float v = x[r * Cc + c] - mu;
      a += MODE ? v * v : v;
    }
  }
  red[rgp][cl] = a;
  __syncthreads();
  if (rgp == 0 && c < Cc) part[(size_t)blockIdx.y * Cc + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// sums -> mean;  centred sums of squares -> rstd (+ running statistics, nn.BatchNorm1d: unbiased variance, momentum)
__global__ void k_bn_mean(const float *__restrict__ sum, float *__restrict__ mean, float invR, int Cc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < Cc) mean[c] = sum[c] * invR;
}
__global__ void k_bn_rstd(const float *__restrict__ ssq, const float *__restrict__ mean, float *__restrict__ rstd, float *run_mean,
                          float *run_var, float invR, float unbias, float eps, float momentum, int Cc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cc) return;
  const float var = ssq[c] * invR;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (momentum >= 0.f && run_mean && run_var) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean[c];
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * unbias;
  }
}
// Batch statistics from the per-workgroup (count, mean, M2) partials the producing product left behind (k_lin_wide_lds<.., STATS>): merged per
// column in partial order (fixed: deterministic), then what k_bn_mean / k_bn_rstd do.  32 lanes share a column's partials and meet in a tree of
// stats_merge steps (one thread per column walking up to 512 partials was a 10 us chain of dependent loads).
__global__ __launch_bounds__(256) void k_bn_merge(const float *__restrict__ stats, int P, float *__restrict__ mean, float *__restrict__ rstd,
                                                  float *run_mean, float *run_var, float unbias, float eps, float momentum, int Cc) {
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (c >= Cc) return;
  float n = 0.f, mu = 0.f, m2 = 0.f;
  for (int p = l; p < P; p += 32) {
    const float *q = stats + ((size_t)p * Cc + c) * 3;
    dfx::lin::stats_merge(n, mu, m2, q[0], q[1], q[2]);
  }
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const float on = __shfl_xor(n, o), om = __shfl_xor(mu, o), oq = __shfl_xor(m2, o);
    // both partners must form the SAME merged triple: the lane with the lower index is the first operand for both
    float a_n = (l & o) ? on : n, a_m = (l & o) ? om : mu, a_q = (l & o) ? oq : m2;
    const float b_n = (l & o) ? n : on, b_m = (l & o) ? mu : om, b_q = (l & o) ? m2 : oq;
    dfx::lin::stats_merge(a_n, a_m, a_q, b_n, b_m, b_q);
    n = a_n, mu = a_m, m2 = a_q;
  }
  if (l != 0) return;
  const float var = m2 / n;
  mean[c] = mu;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (momentum >= 0.f && run_mean && run_var) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * unbias;
  }
}
// BatchNorm affine y = (z - mu) rs g + b: ONE definition with the contraction pinned off, shared by the forward (k_bn_apply), the
// pooling (k_pool_fwd) and the two backward passes that re-derive the ReLU mask from z instead of reading a stored y — the sign of a
// value within an ulp of 0 must not depend on how a particular kernel's multiply-adds were fused.
__device__ __forceinline__ float bn_affine(float z, float mu, float rs, float g, float b) {
#pragma clang fp contract(off)
  const float xh = (z - mu) * rs;
  const float sc = xh * g;
  return sc + b;
}
template <bool RELU>
__global__ void k_bn_apply(const float *__restrict__ z, const float *__restrict__ mean, const float *__restrict__ rstd,
                           const float *__restrict__ g, const float *__restrict__ b, float *__restrict__ y, long long total, int Cc) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= total) return;
  const int c = (int)(i % Cc);
  const v4f zv = *reinterpret_cast<const v4f *>(z + i), mu = *reinterpret_cast<const v4f *>(mean + c), rs = *reinterpret_cast<const v4f *>(rstd + c);
  const v4f gv = *reinterpret_cast<const v4f *>(g + c), bv = *reinterpret_cast<const v4f *>(b + c);
  v4f o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = bn_affine(zv[e], mu[e], rs[e], gv[e], bv[e]);
    if (RELU) o[e] = fmaxf(o[e], 0.f);
  }
  *reinterpret_cast<v4f *>(y + i) = o;
}
// backward, pass 1: gm = RELU ? dy (y > 0) : dy;  column sums of gm (d beta) and gm xhat (d gamma) -> part[slab][2][Cc].
// The ReLU mask is re-derived from z (k_bn_apply's arithmetic: y > 0) instead of reading y: one tensor less per pass.
template <bool RELU>
__global__ __launch_bounds__(256) void k_bn_bwd_part(const float *__restrict__ dy, const float *__restrict__ z,
                                                      const float *__restrict__ mean, const float *__restrict__ rstd,
                                                      const float *__restrict__ g, const float *__restrict__ be,
                                                      float *__restrict__ part, long long R, int Cc) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rgp = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float sb = 0.f, sg = 0.f;
  if (c < Cc) {
    const float mu = mean[c], rs = rstd[c], gv = g[c], bv = be[c];
    const long long r1 = ((long long)blockIdx.y + 1) * BN_SLAB < R ? ((long long)blockIdx.y + 1) * BN_SLAB : R;
    for (long long r = (long long)blockIdx.y * BN_SLAB + rgp; r < r1; r += 4) {
      float gm = dy[r * Cc + c];
      const float zv = z[r * Cc + c];
      if (RELU && !(bn_affine(zv, mu, rs, gv, bv) > 0.f)) gm = 0.f;
      sb += gm;
      sg += gm * (zv - mu) * rs;
    }
  }
  red[0][rgp][cl] = sb, red[1][rgp][cl] = sg;
  __syncthreads();
  if (rgp == 0 && c < Cc) {
    part[((size_t)blockIdx.y * 2) * Cc + c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    part[((size_t)blockIdx.y * 2 + 1) * Cc + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  }
}
// pass 2: dz = gamma rstd (gm - d beta / R - xhat d gamma / R)
template <bool RELU, bool ZERO_DY = false>
__global__ void k_bn_bwd_apply(const float *__restrict__ dy, const float *__restrict__ be, const float *__restrict__ z,
                               const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ g,
                               const float *__restrict__ dbeta, const float *__restrict__ dgamma, float *__restrict__ dz, float invR,
                               long long total, int Cc) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= total) return;
  const int c = (int)(i % Cc);
  const v4f zv = *reinterpret_cast<const v4f *>(z + i);
  v4f dv = {0.f, 0.f, 0.f, 0.f};
  if (!ZERO_DY) dv = *reinterpret_cast<const v4f *>(dy + i);
  v4f o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gm = (RELU && !(bn_affine(zv[e], mean[c + e], rstd[c + e], g[c + e], be[c + e]) > 0.f)) ? 0.f : dv[e];   // (the mask from z, like pass 1)
    const float xh = (zv[e] - mean[c + e]) * rstd[c + e];
    o[e] = g[c + e] * rstd[c + e] * (gm - dbeta[c + e] * invR - xh * dgamma[c + e] * invR);
  }
  *reinterpret_cast<v4f *>(dz + i) = o;
}
// BatchNorm over FEW rows (R <= BN_SLAB: the per-part heads normalise over the B shapes) in ONE launch per direction instead of seven / four:
// a workgroup owns 64 channels and all R rows.  The sums run in the order of the multi-launch path (k_col_stats' four row groups of one slab,
// k_sum_parts over that single partial): the same bits.
template <bool RELU>
__global__ __launch_bounds__(256) void k_bn_small_fwd(const float *__restrict__ z, const float *__restrict__ g, const float *__restrict__ b,
                                                       float *__restrict__ mean, float *__restrict__ rstd, float *run_mean, float *run_var,
                                                       float *__restrict__ y, int R, int Cc, float invR, float unbias, float eps, float momentum) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rgp = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const bool ok = c < Cc;
  // up to 128 rows: this thread's <= 32 values stay in registers for the three passes (ONE batch of loads instead of three chains of them)
  constexpr int KEEP = 32;
  const bool keep = R <= 4 * KEEP;
  float zk[KEEP];
  if (keep) {
#pragma unroll
    for (int q = 0; q < KEEP; ++q) zk[q] = (ok && rgp + 4 * q < R) ? z[(size_t)(rgp + 4 * q) * Cc + c] : 0.f;
  }
  float a = 0.f;
  if (ok) {
    if (keep) {
#pragma unroll
      for (int q = 0; q < KEEP; ++q)
        if (rgp + 4 * q < R) a += zk[q];
    } else {
      for (int r = rgp; r < R; r += 4) a += z[(size_t)r * Cc + c];
    }
  }
  red[rgp][cl] = a;
  __syncthreads();
  const float mu = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) * invR;
  __syncthreads();
  a = 0.f;
  if (ok) {
    if (keep) {
#pragma unroll
      for (int q = 0; q < KEEP; ++q)
        if (rgp + 4 * q < R) {
          const float v = zk[q] - mu;
          a += v * v;
        }
    } else {
      for (int r = rgp; r < R; r += 4) {
        const float v = z[(size_t)r * Cc + c] - mu;
        a += v * v;
      }
    }
  }
  red[rgp][cl] = a;
  __syncthreads();
  const float var = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) * invR;
  const float rs = 1.0f / sqrtf(var + eps);
  if (!ok) return;
  if (rgp == 0) {
    mean[c] = mu, rstd[c] = rs;
    if (momentum >= 0.f && run_mean && run_var) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * unbias;
    }
  }
  const float gv = g[c], bv = b[c];
  if (keep) {
#pragma unroll
    for (int q = 0; q < KEEP; ++q)
      if (rgp + 4 * q < R) {
        float o = bn_affine(zk[q], mu, rs, gv, bv);
        if (RELU) o = fmaxf(o, 0.f);
        y[(size_t)(rgp + 4 * q) * Cc + c] = o;
      }
    return;
  }
  for (int r = rgp; r < R; r += 4) {
    float o = bn_affine(z[(size_t)r * Cc + c], mu, rs, gv, bv);
    if (RELU) o = fmaxf(o, 0.f);
    y[(size_t)r * Cc + c] = o;
  }
}
template <bool RELU>
__global__ __launch_bounds__(256) void k_bn_small_bwd(const float *__restrict__ dy, const float *__restrict__ z, const float *__restrict__ mean,
                                                       const float *__restrict__ rstd, const float *__restrict__ g, const float *__restrict__ be,
                                                       float *__restrict__ dbeta, float *__restrict__ dgamma, float *__restrict__ dz, int R, int Cc,
                                                       float invR) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rgp = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const bool ok = c < Cc;
  float sb = 0.f, sg = 0.f, mu = 0.f, rs = 0.f, gv = 0.f, bv = 0.f;
  constexpr int KEEP = 32;   // (as in the forward: up to 128 rows stay in registers, dy already masked)
  const bool keep = R <= 4 * KEEP;
  float zk[KEEP], gk[KEEP];
  if (ok) {
    mu = mean[c], rs = rstd[c], gv = g[c], bv = be[c];
    if (keep) {
#pragma unroll
      for (int q = 0; q < KEEP; ++q) {
        const bool in = rgp + 4 * q < R;
        zk[q] = in ? z[(size_t)(rgp + 4 * q) * Cc + c] : 0.f;
        gk[q] = in ? dy[(size_t)(rgp + 4 * q) * Cc + c] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < KEEP; ++q)
        if (rgp + 4 * q < R) {
          if (RELU && !(bn_affine(zk[q], mu, rs, gv, bv) > 0.f)) gk[q] = 0.f;
          sb += gk[q];
          sg += gk[q] * (zk[q] - mu) * rs;
        }
    } else {
    for (int r = rgp; r < R; r += 4) {
      float gm = dy[(size_t)r * Cc + c];
      const float zv = z[(size_t)r * Cc + c];
      if (RELU && !(bn_affine(zv, mu, rs, gv, bv) > 0.f)) gm = 0.f;
      sb += gm;
      sg += gm * (zv - mu) * rs;
    }
    }
  }
  red[0][rgp][cl] = sb, red[1][rgp][cl] = sg;
  __syncthreads();
  if (!ok) return;
  const float db = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
  const float dg = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  if (rgp == 0) dbeta[c] = db, dgamma[c] = dg;
  if (keep) {
#pragma unroll
    for (int q = 0; q < KEEP; ++q)
      if (rgp + 4 * q < R) {
        const float xh = (zk[q] - mu) * rs;
        dz[(size_t)(rgp + 4 * q) * Cc + c] = gv * rs * (gk[q] - db * invR - xh * dg * invR);
      }
    return;
  }
  for (int r = rgp; r < R; r += 4) {
    const float zv = z[(size_t)r * Cc + c];
    const float gm = (RELU && !(bn_affine(zv, mu, rs, gv, bv) > 0.f)) ? 0.f : dy[(size_t)r * Cc + c];
    const float xh = (zv - mu) * rs;
    dz[(size_t)r * Cc + c] = gv * rs * (gm - db * invR - xh * dg * invR);
  }
}
// x (B,N,3) -> rows of 8 (zero padded)
__global__ void k_pn_rows(const float *__restrict__ x, float *__restrict__ X8, long long R) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  reinterpret_cast<v4f *>(X8 + r * 8)[0] = v4f{x[r * 3], x[r * 3 + 1], x[r * 3 + 2], 0.f};
  reinterpret_cast<v4f *>(X8 + r * 8)[1] = v4f{0.f, 0.f, 0.f, 0.f};
}
// pooled[b][a][c] = max_n y[b, n, c] attn[b, n, a] scale  (pointnet.py:194-198), with the arg max for the backward.  y = the last
// layer's BatchNorm output is formed HERE from z (k_bn_apply's arithmetic, no ReLU behind this layer): the 512-channel tensor is
// read once instead of read, written and read again.
// Block = 64 channels x PH point phases (thread (c, ph) walks n = ph, ph + PH, ...), grid (Cc / 64, B); the phases are merged
// through LDS keeping the FIRST maximum (lowest n) like the serial scan.
template <int A, int PH>
__global__ __launch_bounds__(64 * PH) void k_pool_fwd(const float *__restrict__ z, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                       const float *__restrict__ g, const float *__restrict__ be, const float *__restrict__ attn,
                                                       float *__restrict__ pooled, int32_t *__restrict__ arg, int N, int Cc, float scale) {
  __shared__ float sb[PH][A][64];
  __shared__ int si[PH][A][64];
  const int b = blockIdx.y, cl = threadIdx.x & 63, ph = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float best[A];
  int bi[A];
#pragma unroll
  for (int a = 0; a < A; ++a) best[a] = -3.402823466e38f, bi[a] = 0;
  if (c < Cc) {
    const float mu = mean[c], rs = rstd[c], gv = g[c], bv = be[c];
    // (eight points' loads in flight per thread; the A weights of a point as one wave-uniform 16-byte load when A = 4)
#pragma unroll 8
    for (int n = ph; n < N; n += PH) {
      const float v = bn_affine(z[((size_t)b * N + n) * Cc + c], mu, rs, gv, bv);
      float aw[A];
      if (A == 4) {
        const v4f t = *reinterpret_cast<const v4f *>(attn + ((size_t)b * N + n) * A);
#pragma unroll
        for (int a = 0; a < A; ++a) aw[a] = t[a];
      } else {
#pragma unroll
        for (int a = 0; a < A; ++a) aw[a] = attn[((size_t)b * N + n) * A + a];
      }
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const float w = v * aw[a] * scale;
        if (w > best[a]) best[a] = w, bi[a] = n;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < A; ++a) sb[ph][a][cl] = best[a], si[ph][a][cl] = bi[a];
  __syncthreads();
  if (ph == 0 && c < Cc) {
#pragma unroll
    for (int a = 0; a < A; ++a) {
      float m = sb[0][a][cl];
      int mi = si[0][a][cl];
#pragma unroll
      for (int q = 1; q < PH; ++q) {
        const float v = sb[q][a][cl];
        const int vi = si[q][a][cl];
        if (v > m || (v == m && vi < mi)) m = v, mi = vi;
      }
      pooled[((size_t)b * A + a) * Cc + c] = m, arg[((size_t)b * A + a) * Cc + c] = mi;
    }
  }
}
// The gradient behind the max-pool is zero except at the arg-max points (at most A per shape and channel), so the BatchNorm in front
// of the pool needs no pass over a dense dy: its column sums come from the B x A entries per channel (k_pool_bwd_stats), its dz is the
// dense "mean" part (k_bn_bwd_apply<false, true>: gm = 0) with the A entries per (shape, channel) recomputed in full afterwards
// (k_pool_bwd_fix: the dense formula with gm = the sum of the parts that share the point, in part order like k_pool_bwd's +=).
template <int A>
__device__ __forceinline__ void pool_entries(const float *__restrict__ dpooled, const int32_t *__restrict__ arg, const float *__restrict__ attn,
                                             int b, int c, int N, int Cc, float scale, int (&n)[A], float (&gm)[A]) {
  float val[A];
  int pt[A];
#pragma unroll
  for (int a = 0; a < A; ++a) {
    pt[a] = arg[((size_t)b * A + a) * Cc + c];
    val[a] = dpooled[((size_t)b * A + a) * Cc + c] * attn[((size_t)b * N + pt[a]) * A + a] * scale;
  }
#pragma unroll
  for (int a = 0; a < A; ++a) {   // n[a] = the point if part a is the first that selected it (else -1: skip), gm[a] = the point's total
    bool first = true;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < A; ++q) {
      if (q < a && pt[q] == pt[a]) first = false;
      if (q >= a && pt[q] == pt[a]) t += val[q];
    }
    n[a] = first ? pt[a] : -1;
    gm[a] = t;
  }
}
// d beta[c] = sum gm, d gamma[c] = sum gm xhat over the entries; block = 32 channels x 32 shape phases, summed in phase order
template <int A>
__global__ __launch_bounds__(1024) void k_pool_bwd_stats(const float *__restrict__ dpooled, const int32_t *__restrict__ arg,
                                                         const float *__restrict__ attn, const float *__restrict__ z,
                                                         const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         float *__restrict__ dbeta, float *__restrict__ dgamma, int B, int N, int Cc, float scale) {
  __shared__ float rb[32][32], rg[32][32];
  const int cl = threadIdx.x & 31, bq = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  float sb = 0.f, sg = 0.f;
  if (c < Cc) {
    const float mu = mean[c], rs = rstd[c];
    for (int b = bq; b < B; b += 32) {
      int n[A];
      float gm[A];
      pool_entries<A>(dpooled, arg, attn, b, c, N, Cc, scale, n, gm);
#pragma unroll
      for (int a = 0; a < A; ++a)
        if (n[a] >= 0) {
          const float xh = (z[((size_t)b * N + n[a]) * Cc + c] - mu) * rs;
          sb += gm[a], sg += gm[a] * xh;
        }
    }
  }
  rb[bq][cl] = sb, rg[bq][cl] = sg;
  __syncthreads();
  if (bq == 0 && c < Cc) {
    float tb = rb[0][cl], tg = rg[0][cl];
    for (int q = 1; q < 32; ++q) tb += rb[q][cl], tg += rg[q][cl];
    dbeta[c] = tb, dgamma[c] = tg;
  }
}
// the entries of dz: g rstd (gm - d beta / R - xhat d gamma / R), one thread per (shape, channel)
template <int A>
__global__ __launch_bounds__(256) void k_pool_bwd_fix(const float *__restrict__ dpooled, const int32_t *__restrict__ arg,
                                                       const float *__restrict__ attn, const float *__restrict__ z,
                                                       const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ g,
                                                       const float *__restrict__ dbeta, const float *__restrict__ dgamma, float *__restrict__ dz,
                                                       int N, int Cc, float scale, float invR) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= Cc) return;
  int n[A];
  float gm[A];
  pool_entries<A>(dpooled, arg, attn, b, c, N, Cc, scale, n, gm);
#pragma unroll
  for (int a = 0; a < A; ++a)
    if (n[a] >= 0) {
      const size_t o = ((size_t)b * N + n[a]) * Cc + c;
      const float xh = (z[o] - mean[c]) * rstd[c];
      dz[o] = g[c] * rstd[c] * (gm[a] - dbeta[c] * invR - xh * dgamma[c] * invR);
    }
}
// dy (zero-initialised) [b, arg, c] += d pooled[b][a][c] attn[b, arg, a] scale; one thread per (b, c): no atomics
template <int A>
__global__ __launch_bounds__(256) void k_pool_bwd(const float *__restrict__ dpooled, const int32_t *__restrict__ arg,
                                                   const float *__restrict__ attn, float *__restrict__ dy, int N, int Cc, float scale) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= Cc) return;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const int n = arg[((size_t)b * A + a) * Cc + c];
    dy[((size_t)b * N + n) * Cc + c] += dpooled[((size_t)b * A + a) * Cc + c] * attn[((size_t)b * N + n) * A + a] * scale;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// prior loss of the part encoder (part_encoders.py:1143-1182): coupling flows forward + log-det, log-likelihood, entropy
// rows are part-major: row = i * B + b, 256 latent channels, the conditioning half is [0,128) (swap: [128,256))
// ---------------------------------------------------------------------------------------------------------------
constexpr int ZD = 256, ZH = 128, NPART = 4;
__global__ void k_flow_gather(const float *__restrict__ z, float *__restrict__ x, int B) {   // z (B, 256, 4) -> x[i][b][c]
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NPART * B * ZD) return;
  const int c = idx % ZD, b = (idx / ZD) % B, i = idx / (ZD * B);
  x[idx] = z[((size_t)b * ZD + c) * NPART + i];
}
__global__ void k_flow_scatter(const float *__restrict__ x, float *__restrict__ z, int B) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NPART * B * ZD) return;
  const int c = idx % ZD, b = (idx / ZD) % B, i = idx / (ZD * B);
  z[((size_t)b * ZD + c) * NPART + i] = x[idx];
}
// flow.py:21-41 forward: y1 = x1 sigmoid(s + 2) + t, logdet += sum log sigmoid(s + 2); one block of 128 threads per row
__global__ __launch_bounds__(128) void k_coupling_fwd(const float *__restrict__ xin, const float *__restrict__ st, float *__restrict__ xout,
                                                       float *__restrict__ logdet, int swap) {
  __shared__ float red[2];
  const size_t row = blockIdx.x;
  const int c = threadIdx.x, xc = swap ? ZH : 0, xt = swap ? 0 : ZH;
  const float scale = 1.0f / (1.0f + expf(-(st[row * ZD + c] + 2.0f)));
  xout[row * ZD + xt + c] = xin[row * ZD + xt + c] * scale + st[row * ZD + ZH + c];
  xout[row * ZD + xc + c] = xin[row * ZD + xc + c];
  float l = logf(scale);
  for (int o = 32; o; o >>= 1) l += __shfl_xor(l, o);
  if ((c & 63) == 0) red[c >> 6] = l;
  __syncthreads();
  if (c == 0) logdet[row] += red[0] + red[1];
}
// backward: d st = [dy1 x1 s (1 - s) + dlogdet (1 - s) | dy1], dx1 = dy1 s, dx (conditioning half) = dy (the net's share is added later)
__global__ __launch_bounds__(128) void k_coupling_bwd(const float *__restrict__ dy, const float *__restrict__ xin, const float *__restrict__ st,
                                                       const float *__restrict__ dlogdet, float *__restrict__ dst, float *__restrict__ dx,
                                                       int swap) {
  const size_t row = blockIdx.x;
  const int c = threadIdx.x, xc = swap ? ZH : 0, xt = swap ? 0 : ZH;
  const float scale = 1.0f / (1.0f + expf(-(st[row * ZD + c] + 2.0f)));
  const float dy1 = dy[row * ZD + xt + c];
  dst[row * ZD + c] = dy1 * xin[row * ZD + xt + c] * scale * (1.0f - scale) + dlogdet[row] * (1.0f - scale);
  dst[row * ZD + ZH + c] = dy1;
  dx[row * ZD + xt + c] = dy1 * scale;
  dx[row * ZD + xc + c] = dy[row * ZD + xc + c];
}
__global__ void k_relu_mask(float *__restrict__ d, const float *__restrict__ h, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !(h[i] > 0.f)) d[i] = 0.f;
}
// per (part, shape): log p(w) + log det (0 for absent parts), posterior entropy, and the loss coefficient a = kl valid / (B n_valid)
__global__ __launch_bounds__(256) void k_prior_terms(const float *__restrict__ w, const float *__restrict__ logdet, const float *__restrict__ logvar,
                                                      const float *__restrict__ valid, float prior_var, float kl, int B, float *__restrict__ logp,
                                                      float *__restrict__ ent, float *__restrict__ acoef) {
  __shared__ float red[2][4];
  const int row = blockIdx.x, i = row / B, b = row % B, c = threadIdx.x;
  const float wv = w[(size_t)row * ZD + c];
  float s2 = wv * wv, sl = logvar[((size_t)b * NPART + i) * ZD + c];
  for (int o = 32; o; o >>= 1) s2 += __shfl_xor(s2, o), sl += __shfl_xor(sl, o);
  if ((c & 63) == 0) red[0][c >> 6] = s2, red[1][c >> 6] = sl;
  __syncthreads();
  if (c == 0) {
    const float sw = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), slv = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float v = valid[b * NPART + i];
    float nv = 0.f;
    for (int j = 0; j < NPART; ++j) nv += valid[b * NPART + j];
    // misc.py:301-317 with dim = 256, summed over the 256 elements: the constant -0.5 log(2 pi) * 256 enters 256 times
    const float lp = (float)ZD * (-logf(prior_var) - 0.91893853320467274178f * (float)ZD) - sw / (2.0f * prior_var) + logdet[row];
    logp[row] = v == 1.f ? lp : 0.f;
    ent[row] = 0.5f * slv + 0.5f * (float)ZD * 2.83787706640934548356f;   // 1 + log(2 pi)
    acoef[row] = kl * v / ((float)B * nv);
  }
}
__global__ __launch_bounds__(256) void k_prior_loss(const float *__restrict__ logp, const float *__restrict__ ent, const float *__restrict__ acoef,
                                                     int rows, float *__restrict__ loss) {
  __shared__ double red[4];
  double s = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) s += (double)acoef[r] * (double)(-logp[r] - ent[r]);
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
// d loss / d w = gs a w / prior_var (present parts), d loss / d logdet = -gs a, d loss / d logvar = -gs a / 2
__global__ void k_prior_bwd_init(const float *__restrict__ w, const float *__restrict__ acoef, const float *__restrict__ valid, float gs,
                                 float prior_var, int B, float *__restrict__ dy, float *__restrict__ dlogdet, float *__restrict__ dlogvar) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NPART * B * ZD) return;
  const int c = idx % ZD, row = idx / ZD, i = row / B, b = row % B;
  const float a = gs * acoef[row], present = valid[b * NPART + i] == 1.f ? 1.f : 0.f;
  dy[idx] = present * a * w[idx] / prior_var;
  if (c == 0) dlogdet[row] = -present * a;
  if (dlogvar) dlogvar[((size_t)b * NPART + i) * ZD + c] = -0.5f * a;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct Carver {
  char *base;
  size_t off = 0;
  template <class T>
  T *take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct BlockAct {
  float *hin, *st2, *xn2, *q, *k, *v, *p, *att, *h1, *st3, *xn3, *ag, *hid;
};
struct TrainWs {
  // saved by the forward
  float *xin, *h0, *st_pre, *hfin, *st_post, *hn, *ctx, *te_in, *te_ag, *te_hid, *te_out;
  BlockAct blk[DFX_MAX_DEPTH];
  // backward scratch
  float *dh, *dh2, *dwide, *dhid, *dq, *datt, *dk, *dv, *dctx, *dte_out, *dte_hid, *dte_ag, *wT, *wpad, *part, *bpart, *apart;
  float *valid;
  size_t part_floats, bpart_floats;
  // fused feed-forward (train_ff_fused.h): the block's W1 / W2 as bf16 MFMA fragments, re-packed by every forward
  uint4 *ff_frags[DFX_MAX_DEPTH];
  float *ff_b1p[DFX_MAX_DEPTH], *ff_b2p[DFX_MAX_DEPTH];
  // fused attention (train_attn_fused.h): per shape the folded (A_s, M_s) fragments of each block; gradient partials
  uint4 *at_frags[DFX_MAX_DEPTH];
  float *at_part[DFX_MAX_DEPTH], *at_sum[DFX_MAX_DEPTH], *cpart[DFX_MAX_DEPTH];   // per block: summed / unfolded behind the block loop, one launch each
  int at_split;
  // weight-stationary feed-forward gradients (k_ff_wgrad): per-slab partial tiles
  float *ffw_part[DFX_MAX_DEPTH], *ffw_bpart[DFX_MAX_DEPTH], *ffw_lnpart[DFX_MAX_DEPTH], *head_tab;
  int ffw_slabs;
  // keys / values of all blocks in one product: packed weights (and transposed), k | v of every block side by side, their gradients
  float *wkv, *wkvT, *kv, *dkv, *dwkv;
};

constexpr int WG_SLAB = 2048;   // rows per k_wgrad slab (64 for the few-row products over the context tokens / the batch)
inline int slab_rows(long long R) { return R <= 8192 ? 64 : WG_SLAB; }
inline int nslabs(long long R) { return (int)((R + slab_rows(R) - 1) / slab_rows(R)); }

size_t carve(TrainWs &w, void *base, int B, int N, int depth) {
  Carver c{static_cast<char *>(base)};
  const size_t R = (size_t)B * N, BJ = (size_t)B * J;
  w.xin = c.take<float>(R * XIN);
  w.h0 = c.take<float>(R * C);
  w.st_pre = c.take<float>(R * 2);
  w.hfin = nullptr;   // = output buffer of the last block (blk[depth].hin slot below)
  w.st_post = c.take<float>(R * 2);
  w.hn = c.take<float>(R * C);
  w.ctx = c.take<float>(BJ * CTXP);
  w.te_in = c.take<float>((size_t)B * TE);
  w.te_ag = c.take<float>((size_t)B * 2 * TEH);
  w.te_hid = c.take<float>((size_t)B * TEH);
  w.te_out = c.take<float>((size_t)B * TE);
  w.valid = c.take<float>(BJ);
  for (int i = 0; i < depth; ++i) {
    BlockAct &a = w.blk[i];
    a.hin = c.take<float>(R * C);
    a.st2 = c.take<float>(R * 2);
    a.xn2 = c.take<float>(R * C);
    a.q = c.take<float>(R * C);
    a.k = c.take<float>(BJ * C);
    a.v = c.take<float>(BJ * C);
    a.p = c.take<float>(R * HEADS * J);
    a.att = c.take<float>(R * C);
    a.h1 = c.take<float>(R * C);
    a.st3 = c.take<float>(R * 2);
    a.xn3 = c.take<float>(R * C);
    a.ag = c.take<float>(R * 2 * FH);
    a.hid = c.take<float>(R * FH);
  }
  w.hfin = c.take<float>(R * C);
  w.dh = c.take<float>(R * C);
  w.dh2 = c.take<float>(R * C);
  w.dwide = c.take<float>(R * 2 * FH);
  w.dhid = c.take<float>(R * FH);
  w.dq = c.take<float>(R * C);
  w.datt = c.take<float>(R * C);
  w.dk = c.take<float>(BJ * C);
  w.dv = c.take<float>(BJ * C);
  w.dctx = c.take<float>(BJ * CTXP);
  w.dte_out = c.take<float>((size_t)B * TE);
  w.dte_hid = c.take<float>((size_t)B * TEH);
  w.dte_ag = c.take<float>((size_t)B * 2 * TEH);
  w.wT = c.take<float>((size_t)2 * TEH * TE + 64);      // largest transposed weight: time_embed.net.0.proj (2048 x 256)
  w.wpad = c.take<float>((size_t)C * CTXP);
  // partial buffers of the two-pass reductions: the largest weight-gradient partial of each row count (points, context
  // tokens, batch rows), the LayerNorm / proj_out column sums
  const size_t ns_r = nslabs((long long)R), ns_bj = nslabs((long long)BJ), ns_b = nslabs(B);
  size_t pf = ns_r * (size_t)(2 * FH) * C;                                  // W1 (1024 x 128) over the points
  const size_t cands[] = {ns_bj * (size_t)C * CTXP, ns_b * (size_t)2 * TEH * TE, ((R + LNB_ROWS - 1) / LNB_ROWS) * 2 * C,
                          ((R + EPSB_ROWS - 1) / EPSB_ROWS) * 4 * C};
  for (size_t v : cands)
    if (v > pf) pf = v;
  w.part_floats = pf;
  w.part = c.take<float>(pf);
  size_t ns_max = ns_r > ns_bj ? ns_r : ns_bj;
  if (ns_b > ns_max) ns_max = ns_b;
  w.bpart_floats = (ns_max > 2048 ? ns_max : 2048) * (size_t)(2 * TEH);   // room for 2048 slabs of the widest bias
  w.bpart = c.take<float>(w.bpart_floats);
  w.apart = c.take<float>((size_t)B * (N / 32) * 2 * J * C);
  for (int i = 0; i < depth; ++i) {
    w.ff_frags[i] = c.take<uint4>(dfx::ffused::pack_bytes_frags() / sizeof(uint4));
    w.ff_b1p[i] = c.take<float>(3 * dfx::ffused::B1P_FLOATS);   // the folded bias three times: in the chunk loop's order, in natural order (PackArgs::b1f), and with the forward's fp16 scales (b1ps)
    w.ff_b2p[i] = c.take<float>(dfx::ffused::B2P_FLOATS);
  }
  for (int i = 0; i < depth; ++i) w.at_frags[i] = c.take<uint4>((size_t)B * dfx::afused::SHAPE_U4);
  w.at_split = dfx::afused::param_split(B, N);
  w.ffw_slabs = dfx::ffused::wgrad_slabs((long long)(R / 32));
  for (int i = 0; i < depth; ++i) {
    w.at_part[i] = c.take<float>((size_t)B * w.at_split * 2 * dfx::afused::HJ * C);
    w.at_sum[i] = c.take<float>((size_t)B * 2 * dfx::afused::HJ * C);
    const size_t groups = (size_t)dfx::ffused::ff_groups(B, N), a = groups * 6 * C, b = (size_t)dfx::afused::dx_groups((long long)R) * 3 * C;
    w.cpart[i] = c.take<float>(a > b ? a : b);
    w.ffw_part[i] = c.take<float>((size_t)w.ffw_slabs * dfx::ffused::NCHUNK * 12 * 1024);
    w.ffw_bpart[i] = c.take<float>((size_t)w.ffw_slabs * dfx::ffused::NCHUNK * 64);
    w.ffw_lnpart[i] = c.take<float>(dfx::ffused::LNPART_FLOATS);
    if (i == 0) w.head_tab = c.take<float>(512);   // (>= dfx::ffused::HEAD_TAB_FLOATS)
  }
  w.wkv = c.take<float>((size_t)2 * depth * C * CTXP);
  w.wkvT = c.take<float>((size_t)2 * depth * C * CTXP);
  w.dwkv = c.take<float>((size_t)2 * depth * C * CTXP);
  w.kv = c.take<float>(BJ * 2 * depth * C);
  w.dkv = c.take<float>(BJ * 2 * depth * C);
  return c.off;
}

// Fused feed-forward kernels (train_ff_fused.h) for bf16 products without dropout; dfx_debug_train_fused(0) restores the
// layer-by-layer path (A/B timing, and the reference for the fused path's own test).
bool g_ff_fused = true;
bool g_attn_in_ff = true;   // (debug: 2 in dfx_debug_train_fused keeps the attention forward as a kernel of its own)
// Fused path: the context branch of the forward (time-embedding MLP, keys / values, attention folds) and the parameter-gradient
// reductions of the backward run on the device's side stream (dfx::SideStream) beside the kernels over the points: they are chains of
// small-grid launches that leave most of the chip idle.  Same kernels, same operands, same results; dfx_debug_train_streams(0) = one stream.
int g_train_streams = 1;   // 0 = off, 1 = the default set, other values = explicit set of SS_* bits (A/B)
enum { SS_FWD_CTX = 2, SS_HEAD_EARLY = 4, SS_STEM_EARLY = 8, SS_LEAVES = 16, SS_STEM_END = 32, SS_DEFAULT = SS_HEAD_EARLY | SS_STEM_EARLY | SS_LEAVES };
inline int ss_mask() { return g_train_streams == 1 ? SS_DEFAULT : g_train_streams; }
// The switches are sampled ONCE per step, by the forward, and recorded per workspace on the host: the backward follows the record,
// not the switches, so toggling dfx_debug_train_fused between a forward and its backward (the A/B tests do) cannot pair a fused
// backward with the activations of a layer-by-layer forward.
struct PathRecord { bool fused, attn_in_ff; int prec; float dropout_p; int B, N; unsigned long long seq; };
unsigned long long g_path_seq = 0;   // forward counter: the record with the smallest seq is the one evicted when the table is full
std::mutex g_path_mu;
std::unordered_map<const void *, PathRecord> g_path;   // keyed by workspace pointer (a handful per process)
thread_local bool t_ff_fused = true, t_attn_in_ff = true;   // what the call in progress on this thread uses
thread_local const char *t_last_path = "none";                // dfx_debug_last_train_path(): the kernel family the last forward on this thread took
// (dropout > 0 takes the fused kernels when the attention sub-block sits inside them — the default — : k_ff<*, true> / k_ff_wgrad<true>)
inline bool ff_fused(bool bf, float dropout_p, long long R, int N) { return t_ff_fused && bf && (dropout_p == 0.f || t_attn_in_ff) && R % 32 == 0 && N % 32 == 0; }

// bf16 operands (fp32 accumulate, fp32 results) for the large products when the caller asked for DFX_PREC_BF16
thread_local int g_prec = DFX_PREC_F32;
// product-only tensors are stored as bf16 when every product over the points takes the bf16 kernel (see store4 above)
inline bool bf_store(long long R) { return g_prec == DFX_PREC_BF16 && R >= 256; }

// x_bf: X is one of the bf16-stored tensors (only when bf_store(R): the bf16 product kernel is then guaranteed to apply)
int lin(hipStream_t st, const float *X, int ldx, const float *W, const float *b, float *Y, int ldy, long long M, int N_,
        int K, const float *resid = nullptr, int ldr = 0, bool x_bf = false, bool y_bf = false) {
  if (g_prec == DFX_PREC_BF16) {
    dfx::gemm::GemmArgs g{};
    g.A = X, g.lda = ldx, g.B = W, g.ldb = K, g.bias = b, g.R = resid, g.ldr = ldr, g.C = Y, g.ldc = ldy, g.M = (int)M, g.N = N_, g.K = K;
    g.a_bf16 = x_bf, g.c_bf16 = y_bf;
    if (dfx::gemm::nt_ok(g)) {
      dfx::gemm::launch_nt(st, g);
      return dfx::check_launch("train: gemm_nt_bf16");
    }
  }
  if (x_bf || y_bf) return dfx::set_error(DFX_ERR_UNSUPPORTED, "train: bf16-stored operand without the bf16 product kernel (M=%lld N=%d K=%d)", M, N_, K);
  LinArgs a{};
  a.X = X, a.ldx = ldx, a.W = W, a.b = b, a.Y = Y, a.ldy = ldy, a.M = (int)M, a.N = N_, a.K = K;
  a.R = resid, a.ldr = ldr, a.r_mod = 0;
  if (M <= 4096) {   // few rows (context tokens, batch rows): 32 x 32 tiles, one wavefront each, fill more CUs than the wide kernel's 128 x 128
    const dim3 grid((N_ + 31) / 32, (unsigned)((M + 31) / 32), 1);
    if (resid) dfx::lin::k_lin<dfx::lin::EPI_RESID><<<grid, 64, 0, st>>>(a);
    else dfx::lin::k_lin<dfx::lin::EPI_NONE><<<grid, 64, 0, st>>>(a);
    return dfx::check_launch("train: linear");
  }
  if (resid) dfx::lin::launch<dfx::lin::EPI_RESID>(st, 1, a);
  else dfx::lin::launch<dfx::lin::EPI_NONE>(st, 1, a);
  return dfx::check_launch("train: linear");
}

// WT = W^T (cols rows of length rows) for dX = dY W
void transpose(hipStream_t st, const float *W, float *WT, int rows, int cols) {
  k_transpose<<<dim3((cols + 31) / 32, (rows + 31) / 32), 256, 0, st>>>(W, WT, rows, cols);
}

// dW (O x I_valid), db (O) from dY (R x O, ld ldy) and X (R x I, ld ldx)
// Slab size of a weight-gradient product: enough slabs to fill the chip (small outputs have few tiles per slab), few enough
// to keep the partial tiles within the workspace and the second pass short; a multiple of 64 rows.
inline int pick_slab(long long R, long long tiles, long long target_blocks, size_t part_floats, size_t out_floats) {
  long long ns = target_blocks / (tiles > 0 ? tiles : 1);
  if (ns < 1) ns = 1;
  const long long ns_cap = (long long)(part_floats / (out_floats ? out_floats : 1));
  if (ns > ns_cap) ns = ns_cap;
  if (ns < 1) ns = 1;
  long long slab = ((R + ns - 1) / ns + 63) / 64 * 64;
  if (slab < 64) slab = 64;
  return (int)slab;
}

struct PartBufs {   // scratch of the two-pass reductions
  float *part;
  size_t part_floats;
  float *bpart;
  size_t bpart_floats;
};
int wgrad(hipStream_t st, const PartBufs &w, const float *dY, int ldy, const float *X, int ldx, float *dW, float *db, int O, int I,
          int I_valid, long long R, bool dy_bf = false, bool x_bf = false) {
  bool done = false;
  int ns = 1;
  const size_t bcap = w.bpart_floats / (size_t)O;   // slabs the bias partials have room for
  if (g_prec == DFX_PREC_BF16) {
    dfx::gemm::GemmArgs g{};
    g.A = dY, g.lda = ldy, g.B = X, g.ldb = ldx, g.C = w.part, g.ldc = I, g.bpart = db ? w.bpart : nullptr, g.M = O, g.N = I, g.K = (int)R;
    g.rows_per_slab = pick_slab(R, (long long)(O / 128) * (I / 128), 512, w.part_floats, (size_t)O * I);
    ns = (int)((R + g.rows_per_slab - 1) / g.rows_per_slab);
    g.a_bf16 = dy_bf, g.b_bf16 = x_bf;
    if (R >= 256 && (size_t)ns <= bcap && dfx::gemm::tn_ok(g)) {
      dfx::gemm::launch_tn(st, g, ns);
      done = true;
    }
  }
  if (!done && (dy_bf || x_bf)) return dfx::set_error(DFX_ERR_UNSUPPORTED, "train: bf16-stored operand without the bf16 product kernel (O=%d I=%d)", O, I);
  if (!done) {
    int slab = pick_slab(R, (long long)((O + 63) / 64) * ((I + 63) / 64), 2048, w.part_floats, (size_t)O * I);
    if (R <= 512) slab = (int)((R + 63) / 64 * 64);   // a handful of rows: one slab, the tile is written straight into dW / db
    // ... unless the output has too few tiles to occupy the chip and goes through the finish kernel anyway (to_k / to_v over the
    // B x 4 context tokens: 18 workgroups walking 512 rows each took 71 us, ten times per iteration): slabs of 64 rows
    if (R <= 512 && R > 64 && I_valid != I && (long long)((O + 63) / 64) * ((I + 63) / 64) < 64) slab = 64;
    ns = (int)((R + slab - 1) / slab);
    if ((size_t)ns > bcap) {
      slab = (int)(((R + (long long)bcap - 1) / (long long)bcap + 63) / 64 * 64);
      ns = (int)((R + slab - 1) / slab);
    }
    if (ns == 1 && I_valid == I) {   // one slab: the "partial" tile is the result (few-row products: heads, flows, time embedding)
      k_wgrad<2><<<dim3((I + 63) / 64, (O + 63) / 64, 1), 256, 0, st>>>(dY, ldy, X, ldx, dW, db, O, I, R, slab);
      return dfx::check_launch("train: wgrad");
    }
    // many rows and I a multiple of 128: 64 x 128 tiles — a dY fragment feeds four MFMAs instead of two (6 loads per 8 MFMAs instead of 4 per 4) and the
    // slab's dY strip is read I / 128 times instead of I / 64; every output still sums its rows in the same order: the same bits.  Same box, PointNetV2
    // training forward + backward: 4.615 / 4.620 -> 4.342 / 4.337 ms
    if (I % 128 == 0 && R >= 8192) k_wgrad<4><<<dim3(I / 128, (O + 63) / 64, ns), 256, 0, st>>>(dY, ldy, X, ldx, w.part, db ? w.bpart : nullptr, O, I, R, slab);
    else k_wgrad<2><<<dim3((I + 63) / 64, (O + 63) / 64, ns), 256, 0, st>>>(dY, ldy, X, ldx, w.part, db ? w.bpart : nullptr, O, I, R, slab);
  }
  if (I_valid == I) k_sum_parts<<<(O * I + 31) / 32, 1024, 0, st>>>(w.part, dW, ns, O * I, O * I);   // parallel over slabs too
  else k_wgrad_finish<<<(O * I_valid + 31) / 32, 256, 0, st>>>(w.part, dW, ns, O, I, I_valid);
  if (db) k_sum_parts<<<(O + 31) / 32, 1024, 0, st>>>(w.bpart, db, ns, O, O);
  return dfx::check_launch("train: wgrad");
}

inline int wgrad(hipStream_t st, TrainWs &w, const float *dY, int ldy, const float *X, int ldx, float *dW, float *db, int O, int I,
                 int I_valid, long long R, bool dy_bf = false, bool x_bf = false) {
  return wgrad(st, PartBufs{w.part, w.part_floats, w.bpart, w.bpart_floats}, dY, ldy, X, ldx, dW, db, O, I, I_valid, R, dy_bf, x_bf);
}

int ln_bwd(hipStream_t st, TrainWs &w, const float *dy, const float *x, const float *stats, const float *g, const float *resid,
           float *out, float *dg, float *db, long long R) {
  const int nb = (int)((R + LNB_ROWS - 1) / LNB_ROWS);
  k_ln_bwd<<<nb, 256, 0, st>>>(dy, x, stats, g, resid, out, w.part, R);
  k_sum_parts<<<C / 32, 1024, 0, st>>>(w.part, dg, nb, C, 2 * C);
  k_sum_parts<<<C / 32, 1024, 0, st>>>(w.part + C, db, nb, C, 2 * C);
  return dfx::check_launch("train: ln_bwd");
}

inline float *mut(const float *p) { return const_cast<float *>(p); }

int check_args(const dfx_denoiser_weights *wt, const void *ws, size_t ws_bytes, int B, int N, const char *what) {
  DFX_REQUIRE(wt && ws, "%s: null argument", what);
  DFX_REQUIRE(wt->depth >= 1 && wt->depth <= DFX_MAX_DEPTH, "%s: depth %d", what, wt->depth);
  DFX_REQUIRE(B >= 1 && N >= 32 && N % 32 == 0, "%s: B >= 1 and N a multiple of 32 required (B=%d N=%d)", what, B, N);
  DFX_REQUIRE((long long)B * N < (1ll << 31), "%s: B*N too large", what);
  TrainWs t;
  const size_t need = carve(t, nullptr, B, N, wt->depth);
  DFX_REQUIRE(ws_bytes >= need, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
  DFX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "%s: workspace must be 256-byte aligned", what);
  return DFX_OK;
}

// ---- PointNetV2 train mode: workspace and BatchNorm helpers ----
struct PnWs {
  float *X8, *z[4], *y[4], *mean[4], *rstd[4];
  float *pooled;
  int32_t *arg;
  float *hz[2][2], *hy[2][2], *hmean[2][2], *hrstd[2][2];
  // backward scratch
  float *dA, *dB, *dpooled, *dh[2], *wpad, *wT, *sums;
  PartBufs pb;
};
constexpr int PN_C[5] = {8, 128, 128, 256, 512};   // trunk widths (input padded 3 -> 8)
size_t carve_pn(PnWs &w, void *base, int B, int N, int A, int /*zdim: the heads write straight into the caller's outputs*/) {
  Carver c{static_cast<char *>(base)};
  const size_t R = (size_t)B * N;
  w.X8 = c.take<float>(R * 8);
  for (int l = 0; l < 4; ++l) {
    w.z[l] = c.take<float>(R * PN_C[l + 1]);
    w.y[l] = c.take<float>(R * PN_C[l + 1]);
    w.mean[l] = c.take<float>(PN_C[l + 1]);
    w.rstd[l] = c.take<float>(PN_C[l + 1]);
  }
  w.pooled = c.take<float>((size_t)B * A * 512);
  w.arg = c.take<int32_t>((size_t)B * A * 512);
  const int hc[2] = {256, 128};
  for (int k = 0; k < 2; ++k)
    for (int l = 0; l < 2; ++l) {
      w.hz[k][l] = c.take<float>((size_t)B * A * hc[l]);
      w.hy[k][l] = c.take<float>((size_t)B * A * hc[l]);
      w.hmean[k][l] = c.take<float>((size_t)A * hc[l]);
      w.hrstd[k][l] = c.take<float>((size_t)A * hc[l]);
    }
  w.dA = c.take<float>(R * 512);
  w.dB = c.take<float>(R * 512);
  w.dpooled = c.take<float>((size_t)B * A * 512);
  w.dh[0] = c.take<float>((size_t)B * A * 256);
  w.dh[1] = c.take<float>((size_t)B * A * 256);
  w.wpad = c.take<float>(128 * 8);
  w.wT = c.take<float>((size_t)4 * 512 * 256);   // a transposed trunk weight (<= 512 x 256), or the four parts' transposed head weights side by side
  w.sums = c.take<float>(4 * 1024);
  const size_t nslab = (R + BN_SLAB - 1) / BN_SLAB;
  size_t pf = nslab * 2 * 1024;                            // BatchNorm column-sum partials (up to A * 256 channels)
  const size_t wg = (size_t)nslabs((long long)R) * 512 * 256;   // weight-gradient partials (conv4: 512 x 256)
  if (wg > pf) pf = wg;
  w.pb.part_floats = pf;
  w.pb.part = c.take<float>(pf);
  w.pb.bpart_floats = (size_t)2048 * 1024;
  w.pb.bpart = c.take<float>(w.pb.bpart_floats);
  return c.off;
}

// PointNetV2 training: batch statistics from the products' epilogues and one-launch BatchNorm for few rows (dfx_debug_bn_fused_stats(0): the
// multi-launch passes)
bool g_bn_fused_stats = true;
// y = [relu] BN_train(z) over R rows; leaves mean / rstd behind and updates the running statistics when momentum >= 0
// stat_parts > 0: the product that wrote z left that many rows of per-workgroup (count, mean, M2) partials in w.pb.part (lin_stats below): no pass over z
int bn_fwd(hipStream_t st, const PnWs &w, const float *z, long long R, int Cc, const float *g, const float *b, float *run_mean,
           float *run_var, float momentum, float eps, float *mean, float *rstd, float *y, bool relu, bool apply = true, int stat_parts = 0) {
  const int ns = (int)((R + BN_SLAB - 1) / BN_SLAB);
  const dim3 grid((Cc + 63) / 64, ns);
  if (stat_parts == 0 && apply && R <= BN_SLAB && g_bn_fused_stats) {   // few rows (the heads: R = B): statistics + apply in one launch, the same bits
    const float unbias = R > 1 ? (float)R / (float)(R - 1) : 1.0f;
    if (relu) k_bn_small_fwd<true><<<(Cc + 63) / 64, 256, 0, st>>>(z, g, b, mean, rstd, run_mean, run_var, y, (int)R, Cc, 1.0f / (float)R, unbias, eps, momentum);
    else k_bn_small_fwd<false><<<(Cc + 63) / 64, 256, 0, st>>>(z, g, b, mean, rstd, run_mean, run_var, y, (int)R, Cc, 1.0f / (float)R, unbias, eps, momentum);
    return dfx::check_launch("train: bn_fwd");
  }
  if (stat_parts > 0) {
    k_bn_merge<<<(Cc + 7) / 8, 256, 0, st>>>(w.pb.part, stat_parts, mean, rstd, run_mean, run_var, R > 1 ? (float)R / (float)(R - 1) : 1.0f, eps, momentum, Cc);
  } else {
  k_col_stats<0><<<grid, 256, 0, st>>>(z, nullptr, w.pb.part, R, Cc);
  k_sum_parts<<<(Cc + 31) / 32, 1024, 0, st>>>(w.pb.part, w.sums, ns, Cc, Cc);
  k_bn_mean<<<(Cc + 255) / 256, 256, 0, st>>>(w.sums, mean, 1.0f / (float)R, Cc);
  k_col_stats<1><<<grid, 256, 0, st>>>(z, mean, w.pb.part, R, Cc);
  k_sum_parts<<<(Cc + 31) / 32, 1024, 0, st>>>(w.pb.part, w.sums, ns, Cc, Cc);
  k_bn_rstd<<<(Cc + 255) / 256, 256, 0, st>>>(w.sums, mean, rstd, run_mean, run_var, 1.0f / (float)R,
                                              R > 1 ? (float)R / (float)(R - 1) : 1.0f, eps, momentum, Cc);
  }
  const long long total = R * Cc;
  if (!apply) return dfx::check_launch("train: bn_fwd");   // (the consumer forms y itself: k_pool_fwd)
  if (relu) k_bn_apply<true><<<(int)((total / 4 + 255) / 256), 256, 0, st>>>(z, mean, rstd, g, b, y, total, Cc);
  else k_bn_apply<false><<<(int)((total / 4 + 255) / 256), 256, 0, st>>>(z, mean, rstd, g, b, y, total, Cc);
  return dfx::check_launch("train: bn_fwd");
}
// dz from dy (gradient at the output of [relu] BN); d gamma, d beta written
// batch_stats = false: BatchNorm in eval() mode under autograd (mean / rstd are the running statistics, constants): dz = g rstd gm — the two
// correction terms of the batch-statistics form carry the factor 1 / R, which is passed as 0
int bn_bwd(hipStream_t st, const PnWs &w, const float *dy, const float *be, const float *z, long long R, int Cc, const float *g,
           const float *mean, const float *rstd, float *dz, float *dgamma, float *dbeta, bool relu, bool batch_stats = true) {
  const float invR = batch_stats ? 1.0f / (float)R : 0.0f;
  const int ns = (int)((R + BN_SLAB - 1) / BN_SLAB);
  const dim3 grid((Cc + 63) / 64, ns);
  if (R <= BN_SLAB && dz != dy && g_bn_fused_stats) {   // few rows: column sums + apply in one launch, the same bits (dz must not alias dy: the sums read all of dy first)
    if (relu) k_bn_small_bwd<true><<<(Cc + 63) / 64, 256, 0, st>>>(dy, z, mean, rstd, g, be, dbeta, dgamma, dz, (int)R, Cc, invR);
    else k_bn_small_bwd<false><<<(Cc + 63) / 64, 256, 0, st>>>(dy, z, mean, rstd, g, be, dbeta, dgamma, dz, (int)R, Cc, invR);
    return dfx::check_launch("train: bn_bwd");
  }
  if (relu) k_bn_bwd_part<true><<<grid, 256, 0, st>>>(dy, z, mean, rstd, g, be, w.pb.part, R, Cc);
  else k_bn_bwd_part<false><<<grid, 256, 0, st>>>(dy, z, mean, rstd, g, be, w.pb.part, R, Cc);
  k_sum_parts<<<(Cc + 31) / 32, 1024, 0, st>>>(w.pb.part, dbeta, ns, Cc, 2 * Cc);
  k_sum_parts<<<(Cc + 31) / 32, 1024, 0, st>>>(w.pb.part + Cc, dgamma, ns, Cc, 2 * Cc);
  const long long total = R * Cc;
  if (relu) k_bn_bwd_apply<true><<<(int)((total / 4 + 255) / 256), 256, 0, st>>>(dy, be, z, mean, rstd, g, dbeta, dgamma, dz, invR, total, Cc);
  else k_bn_bwd_apply<false><<<(int)((total / 4 + 255) / 256), 256, 0, st>>>(dy, be, z, mean, rstd, g, dbeta, dgamma, dz, invR, total, Cc);
  return dfx::check_launch("train: bn_bwd");
}
int check_pn(const dfx_pointnet_v2_weights *wt, const void *ws, size_t ws_bytes, int B, int N, const char *what) {
  DFX_REQUIRE(wt && ws, "%s: null argument", what);
  DFX_REQUIRE(wt->num_anchors == 4 && wt->zdim >= 4 && wt->zdim % 4 == 0, "%s: num_anchors 4 and zdim %% 4 == 0 required", what);
  DFX_REQUIRE(B >= 2 && N >= 1, "%s: B >= 2 (BatchNorm over the batch in the heads) and N >= 1 required", what);
  DFX_REQUIRE((long long)B * N < (1ll << 31), "%s: B*N too large", what);
  PnWs t;
  const size_t need = carve_pn(t, nullptr, B, N, wt->num_anchors, wt->zdim);
  DFX_REQUIRE(ws_bytes >= need, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
  DFX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "%s: workspace must be 256-byte aligned", what);
  return DFX_OK;
}
// grouped 1x1 convolution over B rows: group a reads columns a*cin.., writes a*cout..  (one launch, blockIdx.z = group)
int glin(hipStream_t st, const float *X, const float *W, const float *b, float *Y, int B, int A, int cin, int cout) {
  LinArgs a{};
  a.M = B, a.N = cout, a.K = cin;
  a.X = X, a.ldx = A * cin, a.x_gs = cin;
  a.W = W, a.w_gs = (long long)cout * cin;
  a.b = b, a.b_gs = cout;
  a.Y = Y, a.ldy = A * cout, a.y_gs = cout;
  dfx::lin::k_lin<dfx::lin::EPI_NONE><<<dim3((cout + 31) / 32, (B + 31) / 32, A), 64, 0, st>>>(a);
  return dfx::check_launch("train: grouped linear");
}

// ---- prior loss: workspace ----
struct FlowWs {
  float *xs[DFX_MAX_FLOW_DEPTH + 1], *h1[DFX_MAX_FLOW_DEPTH], *h2[DFX_MAX_FLOW_DEPTH], *st[DFX_MAX_FLOW_DEPTH];
  float *logdet, *acoef, *logp, *ent, *dlogdet;
  float *dy, *dy2, *dst, *dh1, *dh2, *wT;
  PartBufs pb;
};
size_t carve_flow(FlowWs &w, void *base, int B, int depth, int H) {
  Carver c{static_cast<char *>(base)};
  const size_t Rf = (size_t)NPART * B;
  for (int l = 0; l <= depth; ++l) w.xs[l] = c.take<float>(Rf * ZD);
  for (int l = 0; l < depth; ++l) {
    w.h1[l] = c.take<float>(Rf * H);
    w.h2[l] = c.take<float>(Rf * H);
    w.st[l] = c.take<float>(Rf * ZD);
  }
  w.logdet = c.take<float>(Rf);
  w.acoef = c.take<float>(Rf);
  w.logp = c.take<float>(Rf);
  w.ent = c.take<float>(Rf);
  w.dlogdet = c.take<float>(Rf);
  w.dy = c.take<float>(Rf * ZD);
  w.dy2 = c.take<float>(Rf * ZD);
  w.dst = c.take<float>(Rf * ZD);
  w.dh1 = c.take<float>(Rf * H);
  w.dh2 = c.take<float>(Rf * H);
  const size_t hm = (size_t)(H > ZD ? H : ZD);
  w.wT = c.take<float>(NPART * hm * hm);
  const size_t ns = (size_t)(B + 63) / 64 + 1;
  w.pb.part_floats = ns * hm * hm;
  w.pb.part = c.take<float>(w.pb.part_floats);
  w.pb.bpart_floats = ns * hm;
  w.pb.bpart = c.take<float>(w.pb.bpart_floats);
  return c.off;
}
int check_flow(const float *const *flow, int depth, int H, const void *ws, size_t ws_bytes, int B, const char *what) {
  DFX_REQUIRE(flow && ws, "%s: null argument", what);
  DFX_REQUIRE(depth >= 1 && depth <= DFX_MAX_FLOW_DEPTH && H >= 8 && H % 8 == 0 && B >= 1, "%s: depth %d hidden %d B %d", what, depth, H, B);
  FlowWs t;
  DFX_REQUIRE(ws_bytes >= carve_flow(t, nullptr, B, depth, H), "%s: workspace too small", what);
  DFX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "%s: workspace must be 256-byte aligned", what);
  for (int k = 0; k < NPART * depth * 6; ++k) DFX_REQUIRE(flow[k], "%s: null flow parameter %d", what, k);
  return DFX_OK;
}
// Y_g = [relu](X_g W_g^T + b_g) (+ R_g) for the four parts in one launch: X / Y / R at uniform group strides, W_g / b_g separate
// tensors (tables) or a contiguous array of transposed copies (W + g * w_gs)
template <int EPI>
int lin_g4(hipStream_t st, const float *X, int ldx, long long x_gs, const float *const *Wt, const float *const *bt, const float *Wc,
           long long w_gs, float *Y, int ldy, long long y_gs, int M, int N_, int K, const float *R = nullptr, int ldr = 0, long long r_gs = 0) {
  LinArgs a{};
  a.X = X, a.ldx = ldx, a.x_gs = x_gs, a.Y = Y, a.ldy = ldy, a.y_gs = y_gs, a.M = M, a.N = N_, a.K = K;
  a.R = R, a.ldr = ldr, a.r_gs = r_gs;
  if (Wt) {
    for (int g = 0; g < NPART; ++g) a.Wtab[g] = Wt[g], a.btab[g] = bt ? bt[g] : nullptr;
  } else {
    a.W = Wc, a.w_gs = w_gs;
  }
  dfx::lin::k_lin<EPI><<<dim3((N_ + 31) / 32, (M + 31) / 32, NPART), 64, 0, st>>>(a);
  return dfx::check_launch("train: grouped linear");
}


// ---- shared MLP of the PointNet++ layers in training mode (dfx_shared_mlp_train_*; pointnet2_modules.py:9-19, :62-70): the grouped tensor as rows,
// then PointNetV2's building blocks (lin, bn_fwd / bn_bwd with batch statistics, wgrad), a max over the neighbourhood with its arg-max ----
// x (B, Cc, L) -> rows (B L, ldr) [c < Cc; the padding columns Cc .. ldr - 1 are zeroed], one 32 x 32 tile per workgroup through LDS (both sides coalesced)
__global__ void k_smt_to_rows(const float *__restrict__ x, float *__restrict__ rows, int Cc, long long L, int ldr) {
  __shared__ float tile[32][33];
  const float *xb = x + (size_t)blockIdx.z * Cc * L;
  float *rb = rows + (size_t)blockIdx.z * L * ldr;
  const long long l0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) tile[k][tx] = (c0 + k < Cc && l0 + tx < L) ? xb[(size_t)(c0 + k) * L + l0 + tx] : 0.f;
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (l0 + k < L && c0 + tx < ldr) rb[(size_t)(l0 + k) * ldr + c0 + tx] = tile[tx][k];
}
// rows (B L, ldr) -> x (B, Cc, L)
__global__ void k_smt_from_rows(const float *__restrict__ rows, float *__restrict__ x, int Cc, long long L, int ldr) {
  __shared__ float tile[32][33];
  const float *rb = rows + (size_t)blockIdx.z * L * ldr;
  float *xb = x + (size_t)blockIdx.z * Cc * L;
  const long long l0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) tile[k][tx] = (l0 + k < L && c0 + tx < Cc) ? rb[(size_t)(l0 + k) * ldr + c0 + tx] : 0.f;
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < Cc && l0 + tx < L) xb[(size_t)(c0 + k) * L + l0 + tx] = tile[tx][k];
}
// max over the ns rows of a group (F.max_pool2d over nsample: the FIRST maximum wins, like torch's CPU / CUDA pooling kernels on ties): y (G ns, Cc) ->
// out (B, Cc, M) with G = B M, arg (G, Cc) = the winning row of the group
__global__ void k_smt_pool_fwd(const float *__restrict__ y, float *__restrict__ out, int32_t *__restrict__ arg, int M, int ns, int Cc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const long long g = blockIdx.y;
  if (c >= Cc) return;
  const float *p = y + (size_t)g * ns * Cc + c;
  float best = p[0];
  int at = 0;
  for (int s = 1; s < ns; ++s) {
    const float v = p[(size_t)s * Cc];
    if (v > best || (v != v && best == best)) best = v, at = s;   // (a NaN wins, as in torch)
  }
  const long long b = g / M, m = g % M;
  out[((size_t)b * Cc + c) * M + m] = best;
  arg[(size_t)g * Cc + c] = at;
}
// dy (G ns, Cc) = d_out (B, Cc, M) at the group's winning row, zero elsewhere
__global__ void k_smt_pool_bwd(const float *__restrict__ d_out, const int32_t *__restrict__ arg, float *__restrict__ dy, int M, int ns, int Cc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const long long g = blockIdx.y;
  if (c >= Cc) return;
  const long long b = g / M, m = g % M;
  const float v = d_out[((size_t)b * Cc + c) * M + m];
  const int at = arg[(size_t)g * Cc + c];
  float *p = dy + (size_t)g * ns * Cc + c;
  for (int s = 0; s < ns; ++s) p[(size_t)s * Cc] = s == at ? v : 0.f;
}
// a layer without BatchNorm: y = relu(z) (z already carries the convolution's bias), dz = dy (z > 0)
__global__ void k_smt_relu(const float *__restrict__ z, float *__restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fmaxf(z[i], 0.f);
}
__global__ void k_smt_relu_bwd(const float *__restrict__ dy, const float *__restrict__ z, float *__restrict__ dz, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dz[i] = z[i] > 0.f ? dy[i] : 0.f;
}
// eval() mode under autograd: the layer normalises with its running statistics
__global__ void k_smt_running_stats(const float *__restrict__ run_mean, const float *__restrict__ run_var, float *__restrict__ mean, float *__restrict__ rstd,
                                    float eps, int Cc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < Cc) mean[c] = run_mean[c], rstd[c] = 1.0f / sqrtf(run_var[c] + eps);
}
struct SmtWs {
  float *rows0, *z[DFX_MLP_MAX_LAYERS], *y[DFX_MLP_MAX_LAYERS], *mean[DFX_MLP_MAX_LAYERS], *rstd[DFX_MLP_MAX_LAYERS];
  int32_t *arg;
  float *dA, *dB, *wpad, *wT, *dwpad;
  PnWs pn;   // (bn_fwd / bn_bwd take their scratch — pb, sums — from a PnWs)
  int cp0;   // ch[0] rounded up to a multiple of 8 (K of the products is consumed 8 at a time)
};
size_t carve_smt(SmtWs &w, void *base, const dfx_shared_mlp_train *t, long long R, long long G) {
  Carver c{static_cast<char *>(base)};
  w.cp0 = (t->ch[0] + 7) / 8 * 8;
  w.rows0 = c.take<float>((size_t)R * w.cp0);
  int cmax = w.cp0;
  size_t wmax = 0;
  for (int l = 0; l < t->layers; ++l) {
    const int co = t->ch[l + 1], ci = l == 0 ? w.cp0 : t->ch[l];
    w.z[l] = c.take<float>((size_t)R * co);
    w.y[l] = c.take<float>((size_t)R * co);
    w.mean[l] = c.take<float>(co);
    w.rstd[l] = c.take<float>(co);
    cmax = std::max(cmax, co);
    wmax = std::max(wmax, (size_t)co * ci);
  }
  w.arg = c.take<int32_t>((size_t)G * t->ch[t->layers]);
  w.dA = c.take<float>((size_t)R * cmax);
  w.dB = c.take<float>((size_t)R * cmax);
  w.wpad = c.take<float>(wmax);
  w.wT = c.take<float>(wmax);
  w.dwpad = c.take<float>(wmax);
  w.pn.sums = c.take<float>(4 * 1024);
  const size_t nslab = (size_t)((R + BN_SLAB - 1) / BN_SLAB);
  size_t pf = nslab * 2 * (size_t)cmax;                                 // BatchNorm column-sum partials
  pf = std::max(pf, (size_t)nslabs(R) * wmax);                         // weight-gradient partials
  w.pn.pb.part_floats = pf;
  w.pn.pb.part = c.take<float>(pf);
  w.pn.pb.bpart_floats = (size_t)2048 * 1024;
  w.pn.pb.bpart = c.take<float>(w.pn.pb.bpart_floats);
  return c.off;
}
int check_smt(const dfx_shared_mlp_train *t, const void *ws, size_t ws_bytes, int B, int M, int ns, const char *what) {
  DFX_REQUIRE(t && ws, "%s: null argument", what);
  DFX_REQUIRE(t->layers >= 1 && t->layers <= DFX_MLP_MAX_LAYERS, "%s: 1..%d layers", what, DFX_MLP_MAX_LAYERS);
  DFX_REQUIRE(B >= 1 && M >= 1 && ns >= 1 && (long long)B * M * ns < (1ll << 31), "%s: bad sizes", what);
  DFX_REQUIRE(t->ch[0] >= 1, "%s: no input channels", what);
  for (int l = 0; l < t->layers; ++l) {
    DFX_REQUIRE(t->ch[l + 1] >= 4 && t->ch[l + 1] % 4 == 0 && t->ch[l + 1] <= 1024, "%s: layer %d: %d output channels (multiple of 4, <= 1024)", what, l, t->ch[l + 1]);
    DFX_REQUIRE(t->conv_w[l], "%s: layer %d: null weight", what, l);
    DFX_REQUIRE(t->bn_w[l] ? (t->bn_b[l] != nullptr) : true, "%s: layer %d: BatchNorm weight without bias", what, l);
  }
  SmtWs w;
  const size_t need = carve_smt(w, nullptr, t, (long long)B * M * ns, (long long)B * M);
  DFX_REQUIRE(ws_bytes >= need, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
  DFX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "%s: workspace must be 256-byte aligned", what);
  return DFX_OK;
}
}  // namespace

extern "C" {

size_t dfx_denoiser_train_workspace_bytes(int B, int N, int depth) {
  if (B < 1 || N < 32 || depth < 1 || depth > DFX_MAX_DEPTH) return 0;
  TrainWs t;
  return carve(t, nullptr, B, N, depth);
}

int dfx_denoiser_train_forward(const dfx_denoiser_weights *wt, void *workspace, size_t workspace_bytes, const float *x,
                               const int32_t *t, const float *ctx_code, const float *ctx_mv, const float *anchors,
                               const float *variances, const float *valid, const int32_t *assignment, float *eps, int B,
                               int N, int precision, float dropout_p, uint64_t dropout_seed, dfx_stream_t stream) {
  int rc = check_args(wt, workspace, workspace_bytes, B, N, "denoiser_train_forward");
  if (rc) return rc;
  DFX_REQUIRE(precision == DFX_PREC_F32 || precision == DFX_PREC_BF16, "denoiser_train_forward: precision %d", precision);
  DFX_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "denoiser_train_forward: dropout_p %g", (double)dropout_p);
  g_prec = precision;
  t_ff_fused = g_ff_fused, t_attn_in_ff = g_attn_in_ff;
  {
    std::lock_guard<std::mutex> lk(g_path_mu);
    // bounded table: a record is overwritten by the next forward into the same workspace (stale addresses cost nothing); when 4096
    // DIFFERENT workspaces have been seen, only the OLDEST record goes — never the record of a forward whose backward is pending
    // unless more than 4096 forwards are outstanding at once
    if (g_path.size() >= 4096 && g_path.find(workspace) == g_path.end()) {
      auto oldest = g_path.begin();
      for (auto it = g_path.begin(); it != g_path.end(); ++it)
        if (it->second.seq < oldest->second.seq) oldest = it;
      g_path.erase(oldest);
    }
    g_path[workspace] = PathRecord{t_ff_fused, t_attn_in_ff, precision, dropout_p, B, N, ++g_path_seq};
  }
  const bool bf = bf_store((long long)B * N);
  DFX_REQUIRE(x && t && ctx_code && ctx_mv && anchors && variances && assignment && eps, "denoiser_train_forward: null tensor");
  hipStream_t st = dfx::as_stream(stream);
  TrainWs w;
  carve(w, workspace, B, N, wt->depth);
  const long long R = (long long)B * N;
  const int BJ = B * J;
  const bool fused_ends = ff_fused(bf, dropout_p, R, N);
  t_last_path = fused_ends ? (dropout_p > 0.f ? "fused_bf16_dropout" : "fused_bf16")
                           : bf ? (dropout_p > 0.f ? "layer_bf16_dropout" : "layer_bf16") : (dropout_p > 0.f ? "layer_f32_dropout" : "layer_f32");
  // Two branches meet in front of the first block: the points (input rows, proj_in + pre_norm, the blocks' weight fragments) on the caller's
  // stream, the context (time embedding -> context rows -> keys / values of every block -> folded attention operands) on the side stream
  dfx::SideStream ss;
  if ((rc = ss.open(st, (ss_mask() & SS_FWD_CTX) && fused_ends))) return rc;
  hipStream_t sc = ss.side;
  if ((rc = ss.fork())) return rc;
  // time embedding -> context rows
  k_timestep_embedding<<<(B * TE + 255) / 256, 256, 0, sc>>>(t, w.te_in, B);
  if ((rc = lin(sc, w.te_in, TE, wt->te0_w, wt->te0_b, w.te_ag, 2 * TEH, B, 2 * TEH, TE))) return rc;
  k_geglu_fwd<false><<<(int)(((long long)B * TEH / 4 + 255) / 256), 256, 0, sc>>>(w.te_ag, w.te_hid, TEH, (long long)B * TEH, Drop{dropout_p, dropout_seed, SITE_TE});
  if ((rc = lin(sc, w.te_hid, TEH, wt->te2_w, wt->te2_b, w.te_out, TE, B, TE, TEH))) return rc;
  k_build_ctx<<<(BJ * CTXP + 255) / 256, 256, 0, sc>>>(ctx_code, ctx_mv, w.te_out, w.ctx, B);
  const int LDKV = 2 * wt->depth * C;
  if (fused_ends) {   // packed key / value weights of every block (the side stream's product below reads them: second fork)
    KvPtrs kp{};
    for (int i = 0; i < wt->depth; ++i) kp.p[2 * i] = wt->blk[i].to_k, kp.p[2 * i + 1] = wt->blk[i].to_v;
    k_pack_kv<<<(2 * wt->depth * C * CTXP + 255) / 256, 256, 0, st>>>(kp, w.wkv, 2 * wt->depth);
  }
  if (valid) DFX_HIP_TRY(hipMemcpyAsync(w.valid, valid, sizeof(float) * BJ, hipMemcpyDeviceToDevice, st));
  else DFX_HIP_TRY(hipMemsetAsync(w.valid, 0x3f, sizeof(float) * BJ, st));   // any non-zero value = keep
  if ((rc = ss.fork())) return rc;
  // proj_in + pre_norm
  k_build_xin<<<(int)((R + 255) / 256), 256, 0, st>>>(x, anchors, variances, assignment, w.xin, N, R);
  if (fused_ends) {
    k_stem_fwd<<<2048, 256, 0, st>>>(w.xin, wt->proj_in_w, wt->proj_in_b, wt->pre_norm_w, wt->pre_norm_b, w.blk[0].hin, R);
    // W1 / W2 of every block as bf16 MFMA fragments (train_ff_fused.h), one launch
    dfx::ffused::PackBatch pb{};
    for (int i = 0; i < wt->depth; ++i)
      pb.blk[i] = dfx::ffused::PackArgs{wt->blk[i].ff0_w, wt->blk[i].ff0_b, wt->blk[i].ff2_w, wt->blk[i].ff2_b, w.ff_frags[i], w.ff_b1p[i], w.ff_b2p[i],
                                        dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f, wt->blk[i].norm3_w, wt->blk[i].norm3_b,
                                        w.ff_b1p[i] + dfx::ffused::B1P_FLOATS, w.ff_b1p[i] + 2 * dfx::ffused::B1P_FLOATS};
    if (t_attn_in_ff)   // + the head's table: the last block's forward kernel computes eps itself
      pb.head_w = wt->proj_out_w, pb.head_b = wt->proj_out_b, pb.head_g = wt->post_norm_w, pb.head_be = wt->post_norm_b, pb.head_tab = w.head_tab;
    dfx::ffused::launch_pack(st, pb, wt->depth);
  } else {
    k_pad_cols<<<(C * XIN + 255) / 256, 256, 0, st>>>(wt->proj_in_w, w.wpad, C, 13, XIN);
    if ((rc = lin(st, w.xin, XIN, w.wpad, wt->proj_in_b, w.h0, C, R, C, XIN))) return rc;
    k_ln_fwd<false><<<(int)((R + 7) / 8), 256, 0, st>>>(w.h0, wt->pre_norm_w, wt->pre_norm_b, w.blk[0].hin, w.st_pre, R);
  }
  if (fused_ends) {   // keys and values of every block: one product over the B x 4 context tokens
    if ((rc = lin(sc, w.ctx, CTXP, w.wkv, nullptr, w.kv, LDKV, BJ, LDKV, CTXP))) return rc;
    dfx::afused::FoldArgs fo{};
    fo.kv = w.kv, fo.ldkv = LDKV;
    for (int i = 0; i < wt->depth; ++i) fo.wq[i] = wt->blk[i].to_q, fo.wo[i] = wt->blk[i].to_out_w, fo.frags[i] = w.at_frags[i];
    dfx::afused::k_attn_fold<<<dim3(B, wt->depth), 256, 0, sc>>>(fo);
  }
  if ((rc = ss.join())) return rc;
  dfx::ffused::FfChain chain{};
  for (int i = 0; i < wt->depth; ++i) {
    const dfx_block_weights &bw = wt->blk[i];
    BlockAct &a = w.blk[i];
    float *hout = i + 1 < wt->depth ? w.blk[i + 1].hin : w.hfin;
    const bool fused = ff_fused(bf, dropout_p, R, N);
    if (fused) {
      // the whole block in two launches (train_attn_fused.h, train_ff_fused.h): h1 = hin + attention(LN2(hin)), hout = h1 + FF(LN3(h1));
      // q, P, att, xn2, xn3, [a | g], hid never exist in memory
      dfx::ffused::FfArgs fa{};
      fa.frags = w.ff_frags[i], fa.b2p = w.ff_b2p[i], fa.g3 = bw.norm3_w, fa.b3 = bw.norm3_b;
      fa.b1p = w.ff_b1p[i] + (dfx::ffused::FWD_F16 && t_attn_in_ff ? 2 * dfx::ffused::B1P_FLOATS : 0);   // ff_fwd's table: scaled like its W1 tiles (PackArgs::b1ps)
      fa.h1 = a.h1, fa.h2 = hout, fa.R = R, fa.B = B, fa.N = N;
      // tile-major rows between the fused kernels (train_ff_fused.h): everything but the stem's output and the head's input
      // (+ h1 itself only as what the backward kernels want of it: the xhat3 fragments and 1 / std, TL_H1_FRAG)
      if (t_attn_in_ff) fa.tiled = dfx::ffused::TL_H1 | (i > 0 ? dfx::ffused::TL_HIN : 0) | (i + 1 < wt->depth ? dfx::ffused::TL_H2 : 0) | dfx::ffused::TL_H1_FRAG;
      if (t_attn_in_ff) {   // attention sub-block inside the feed-forward kernel's prologue: h1 computed from hin, written once
        fa.at_frags = w.at_frags[i], fa.valid = w.valid, fa.g2 = bw.norm2_w, fa.b2n = bw.norm2_b, fa.bo = bw.to_out_b;
        fa.hin = a.hin, fa.h1_out = a.h1;
        if (i + 1 == wt->depth) fa.tiled |= dfx::ffused::TL_HEAD, fa.head_tab = w.head_tab, fa.eps = eps;   // post_norm + proj_out in this kernel's epilogue
        if (dropout_p > 0.f) {   // the two sites of block i (the layer-by-layer path's numbering); bits -> a.p (R x 32 floats, unused by the fused path; 80 B per point needed)
          fa.dk = dfx::drop_key(dropout_seed, dropout_p), fa.site_att = (unsigned)(2 * i), fa.site_ff = (unsigned)(2 * i + 1);
          fa.dmask = reinterpret_cast<unsigned *>(a.p);
        }
      } else {
        dfx::afused::AttnArgs aa{};
        aa.frags = w.at_frags[i], aa.valid = w.valid, aa.g2 = bw.norm2_w, aa.b2 = bw.norm2_b, aa.bo = bw.to_out_b;
        aa.h = a.hin, aa.h1 = a.h1, aa.N = N, aa.R = R;
        dfx::afused::k_attn_fwd_fused<<<(int)((R / 32 + dfx::afused::NW - 1) / dfx::afused::NW), dfx::afused::NW * 64, 0, st>>>(aa);
      }
#if !defined(DFX_TRACE_FF) || defined(DFX_TRACE_FF_CHAIN)   // (the phase-trace build stamps single-block launches unless asked for the chain)
      if (t_attn_in_ff) {   // all blocks in ONE launch (k_ff_fwd_chain): collected here, launched behind the last one
        chain.blk[i] = fa;
        if (i + 1 == wt->depth) {
          chain.n = wt->depth;
          if (dfx::ffused::launch_ff_chain(st, chain)) return dfx::set_error(DFX_ERR_HIP, "train: fused forward chain launch");
        }
        continue;
      }
#endif
      if (dfx::ffused::launch_ff<false>(st, fa)) return dfx::set_error(DFX_ERR_HIP, "train: fused feed-forward launch");
      continue;
    }
    if (!fused) {
      if (bf) k_ln_fwd<true><<<(int)((R + 7) / 8), 256, 0, st>>>(a.hin, bw.norm2_w, bw.norm2_b, a.xn2, a.st2, R);
      else k_ln_fwd<false><<<(int)((R + 7) / 8), 256, 0, st>>>(a.hin, bw.norm2_w, bw.norm2_b, a.xn2, a.st2, R);
      if ((rc = lin(st, a.xn2, C, bw.to_q, nullptr, a.q, C, R, C, C, nullptr, 0, bf))) return rc;
    }
    k_pad_cols<<<(C * CTXP + 255) / 256, 256, 0, st>>>(bw.to_k, w.wpad, C, CTX, CTXP);
    if ((rc = lin(st, w.ctx, CTXP, w.wpad, nullptr, a.k, C, BJ, C, CTXP))) return rc;
    k_pad_cols<<<(C * CTXP + 255) / 256, 256, 0, st>>>(bw.to_v, w.wpad, C, CTX, CTXP);
    if ((rc = lin(st, w.ctx, CTXP, w.wpad, nullptr, a.v, C, BJ, C, CTXP))) return rc;
    if (bf) k_attn_fwd<true><<<dim3(N / 32, B), 256, 0, st>>>(a.q, a.k, a.v, w.valid, a.p, a.att, N);
    else k_attn_fwd<false><<<dim3(N / 32, B), 256, 0, st>>>(a.q, a.k, a.v, w.valid, a.p, a.att, N);
    if (dropout_p > 0.f) {   // h1 = dropout(att Wo^T + bo) + hin   (attention.py:177: to_out = Sequential(Linear, Dropout))
      if ((rc = lin(st, a.att, C, bw.to_out_w, bw.to_out_b, a.h1, C, R, C, C, nullptr, 0, bf))) return rc;
      k_dropout<<<(int)((R * C / 4 + 255) / 256), 256, 0, st>>>(a.h1, a.hin, a.h1, R * C, Drop{dropout_p, dropout_seed, (unsigned)(2 * i)});
    } else if ((rc = lin(st, a.att, C, bw.to_out_w, bw.to_out_b, a.h1, C, R, C, C, a.hin, C, bf))) return rc;
    if (bf) k_ln_fwd<true><<<(int)((R + 7) / 8), 256, 0, st>>>(a.h1, bw.norm3_w, bw.norm3_b, a.xn3, a.st3, R);
    else k_ln_fwd<false><<<(int)((R + 7) / 8), 256, 0, st>>>(a.h1, bw.norm3_w, bw.norm3_b, a.xn3, a.st3, R);
    if ((rc = lin(st, a.xn3, C, bw.ff0_w, bw.ff0_b, a.ag, 2 * FH, R, 2 * FH, C, nullptr, 0, bf, bf && AG_BF16))) return rc;
    const Drop dff{dropout_p, dropout_seed, (unsigned)(2 * i + 1)};
    if (bf) k_geglu_fwd<true><<<(int)((R * FH / 4 + 255) / 256), 256, 0, st>>>(a.ag, a.hid, FH, R * FH, dff);
    else k_geglu_fwd<false><<<(int)((R * FH / 4 + 255) / 256), 256, 0, st>>>(a.ag, a.hid, FH, R * FH, dff);
    if ((rc = lin(st, a.hid, FH, bw.ff2_w, bw.ff2_b, hout, C, R, C, FH, a.h1, C, bf))) return rc;
  }
  if (fused_ends) {
    if (!t_attn_in_ff) k_head_fwd<<<2048, 256, 0, st>>>(w.hfin, wt->post_norm_w, wt->post_norm_b, wt->proj_out_w, wt->proj_out_b, eps, N, R);
  } else {
    k_ln_fwd<false><<<(int)((R + 7) / 8), 256, 0, st>>>(w.hfin, wt->post_norm_w, wt->post_norm_b, w.hn, w.st_post, R);
    k_eps_fwd<<<(int)((R + 7) / 8), 256, 0, st>>>(w.hn, wt->proj_out_w, wt->proj_out_b, eps, N, R);
  }
  return dfx::check_launch("denoiser_train_forward");
}

int dfx_denoiser_train_backward(const dfx_denoiser_weights *wt, void *workspace, size_t workspace_bytes,
                                const float *d_eps, const dfx_denoiser_weights *grads, float *d_ctx_code, float *d_ctx_mv,
                                float *d_x, float *d_variances, int B, int N, int precision, float dropout_p, uint64_t dropout_seed,
                                dfx_stream_t stream) {
  int rc = check_args(wt, workspace, workspace_bytes, B, N, "denoiser_train_backward");
  if (rc) return rc;
  DFX_REQUIRE(precision == DFX_PREC_F32 || precision == DFX_PREC_BF16, "denoiser_train_backward: precision %d", precision);
  DFX_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "denoiser_train_backward: dropout_p %g", (double)dropout_p);
  g_prec = precision;
  {
    std::lock_guard<std::mutex> lk(g_path_mu);
    const auto it = g_path.find(workspace);
    DFX_REQUIRE(it != g_path.end(), "denoiser_train_backward: no forward has filled this workspace");
    const PathRecord &r = it->second;
    DFX_REQUIRE(r.prec == precision && r.dropout_p == dropout_p && r.B == B && r.N == N,
                "denoiser_train_backward: the forward of this workspace ran with precision %d, dropout %g, B %d, N %d", r.prec,
                (double)r.dropout_p, r.B, r.N);
    t_ff_fused = r.fused, t_attn_in_ff = r.attn_in_ff;
  }
  const bool bf = bf_store((long long)B * N);
  DFX_REQUIRE(d_eps && grads, "denoiser_train_backward: null tensor");
  hipStream_t st = dfx::as_stream(stream);
  TrainWs w;
  carve(w, workspace, B, N, wt->depth);
  const long long R = (long long)B * N;
  const int BJ = B * J;
  // proj_out, post_norm
  const bool fused_ends = ff_fused(bf, dropout_p, R, N);
  // Parameter-gradient reductions that nothing on the way back to the input waits for run on the side stream (joined before the return):
  // the head's and the stem's partial sums — in buffers of their own (w.hn, w.datt: free on the fused path), because the products behind
  // the block loop use w.part on the caller's stream meanwhile — and the leaves of the finishing kernels behind the block loop.
  dfx::SideStream ss;
  const int ssm = ss_mask();
  if ((rc = ss.open(st, (ssm & ~SS_FWD_CTX) && ssm != 0 && fused_ends))) return rc;
  hipStream_t sp = ss.side;
  std::function<void(hipStream_t)> head_sums;
  bool head_done = false;
  if (fused_ends) {
    const int nb = (int)((R + HEAD_ROWS - 1) / HEAD_ROWS);
    float *hp = w.hn;
    k_head_bwd<<<nb, 256, 0, st>>>(d_eps, w.hfin, wt->post_norm_w, wt->post_norm_b, wt->proj_out_w, w.dh, hp, N, R, t_attn_in_ff);
    head_sums = [&, nb, hp](hipStream_t s) {
      k_sum_parts<<<3 * C / 32, 1024, 0, s>>>(hp, mut(grads->proj_out_w), nb, 3 * C, HEAD_PART);
      k_sum_parts_multi<<<2 * C / 32, 1024, 0, s>>>(hp + 3 * C, SumOuts{{mut(grads->post_norm_w), mut(grads->post_norm_b), nullptr, nullptr}}, nb, C, HEAD_PART);
      k_sum_parts<<<1, 1024, 0, s>>>(hp + 5 * C, mut(grads->proj_out_b), nb, 3, HEAD_PART);
    };
    if (!ss.on) head_sums(st), head_done = true;
    else if (ssm & SS_HEAD_EARLY) {
      if ((rc = ss.fork())) return rc;
      head_sums(sp), head_done = true;
    }
  } else {
    const int nb = (int)((R + EPSB_ROWS - 1) / EPSB_ROWS);
    k_eps_bwd<<<nb, 128, 0, st>>>(d_eps, w.hn, wt->proj_out_w, w.dh2, w.part, N, R);
    k_sum_parts<<<3 * C / 32, 1024, 0, st>>>(w.part, mut(grads->proj_out_w), nb, 3 * C, 4 * C);
    k_sum_parts<<<1, 1024, 0, st>>>(w.part + 3 * C, mut(grads->proj_out_b), nb, 3, 4 * C);
  }
  if (!fused_ends && (rc = ln_bwd(st, w, w.dh2, w.hfin, w.st_post, wt->post_norm_w, nullptr, w.dh, mut(grads->post_norm_w), mut(grads->post_norm_b), R))) return rc;
  DFX_HIP_TRY(hipMemsetAsync(w.dctx, 0, sizeof(float) * (size_t)BJ * CTXP, st));
  bool stem_done = false;
  const bool dx_in_ff_all = t_attn_in_ff;
  float *dh_cur = w.dh;   // the gradient of the residual stream: where the next kernel on the way back finds it
  auto stem_backward = [&](hipStream_t s, float *part) {   // pre_norm, proj_in (fused ends): reads w.dh, the input rows and the weights
    const int nb = (int)((R + STEM_ROWS - 1) / STEM_ROWS);
    k_stem_bwd<<<nb, 256, 0, s>>>(dh_cur, w.xin, wt->proj_in_w, wt->proj_in_b, wt->pre_norm_w, part, R);
    k_sum_parts<<<C * 13 / 32, 1024, 0, s>>>(part, mut(grads->proj_in_w), nb, C * 13, 2048);
    k_sum_parts_multi<<<3 * C / 32, 1024, 0, s>>>(part + C * 13, SumOuts{{mut(grads->proj_in_b), mut(grads->pre_norm_w), mut(grads->pre_norm_b), nullptr}}, nb, C, 2048);
  };
  for (int i = wt->depth - 1; i >= 0; --i) {
    const dfx_block_weights &bw = wt->blk[i], &gw = grads->blk[i];
    BlockAct &a = w.blk[i];
    // feed-forward: h2 = h1 + W2 hid + b2, hid = a gelu(g), [a | g] = W1 xn3 + b1
    const bool fused = ff_fused(bf, dropout_p, R, N);
    if (fused) {
      // one pass over the points: xn3 = LN3(h1) and [a | g] recomputed, d hid = W2^T dh, GEGLU backward, dxn3 = W1^T d[a | g], LayerNorm3
      // backward -> dh1; hid, d[a | g] and xn3 leave the kernel once, as bf16, for the two weight-gradient products
      dfx::ffused::FfArgs fa{};
      fa.frags = w.ff_frags[i], fa.b1p = w.ff_b1p[i], fa.b2p = w.ff_b2p[i], fa.g3 = bw.norm3_w, fa.b3 = bw.norm3_b;
      fa.h1 = a.h1, fa.dh = dh_cur, fa.pk = reinterpret_cast<uint4 *>(w.dwide), fa.dh1 = w.dh2, fa.cpart = w.cpart[i], fa.R = R, fa.B = B, fa.N = N;
      const bool dx_in_ff = t_attn_in_ff;   // the attention's input gradient in the same kernel (dh1 leaves as fragments for the parameter kernel)
      // With the attention inside, the gradient travels from the head through the blocks as bf16-pair tiles (train_ff_fused.h, TL_DH_HL), alternating
      // between w.dh and w.dh2 (k_ff_wgrad reads block i's INCOMING gradient after k_ff<true> has written the outgoing one); block 0 writes the fp32
      // rows the stem reads (dh_cur behind the loop)
      const bool hl_in = dx_in_ff, hl_out = dx_in_ff && i > 0;   // (k_head_bwd writes the pair format too; block 0 hands fp32 rows to the stem)
      const float *dh_in_blk = dh_cur;
      if (t_attn_in_ff)   // + the layouts the forward left behind
        fa.tiled = dfx::ffused::TL_H1 | (i > 0 ? dfx::ffused::TL_HIN | dfx::ffused::TL_DHIN : 0) | (i + 1 < wt->depth ? dfx::ffused::TL_DH : 0) |
                   (hl_in ? dfx::ffused::TL_DH_HL : 0) | (hl_out ? dfx::ffused::TL_DHIN_HL : 0) | dfx::ffused::TL_H1_FRAG;
      if (dx_in_ff) {
        fa.at_frags = w.at_frags[i], fa.valid = w.valid, fa.g2 = bw.norm2_w, fa.b2n = bw.norm2_b, fa.bo = bw.to_out_b;
        fa.hin = a.hin, fa.dh_in = dh_cur = (dh_cur == w.dh ? w.dh2 : w.dh);
        fa.pk2 = reinterpret_cast<uint4 *>(w.dq);   // xn2 / dh1 as fragments for the parameter kernel (w.dq: free in this path)
      }
      if (dropout_p > 0.f) {   // the bits the forward left in a.p (ff_fused() admits dropout only with the attention inside these kernels)
        fa.dk = dfx::drop_key(dropout_seed, dropout_p), fa.site_att = (unsigned)(2 * i), fa.site_ff = (unsigned)(2 * i + 1);
        fa.dmask = reinterpret_cast<unsigned *>(a.p);
      }
      if (dfx::ffused::launch_ff<true>(st, fa)) return dfx::set_error(DFX_ERR_HIP, "train: fused feed-forward backward launch");
      if (i == 0 && dx_in_ff && ss.on && (ssm & SS_STEM_EARLY)) {   // w.dh is final: pre_norm / proj_in backward beside the rest of the block's parameter kernels
        if ((rc = ss.fork())) return rc;
        stem_backward(sp, w.datt);
        stem_done = true;
      }
      const int groups = (int)dfx::ffused::ff_groups(B, N);
      // (with the attention's input gradient in the same kernel, the six column sums of every block are one launch behind the loop)
      if (!dx_in_ff)
        k_sum_parts_multi<<<3 * C / 32, 1024, 0, st>>>(w.cpart[i], SumOuts{{nullptr, nullptr, mut(gw.ff2_b), nullptr}}, groups, C, 3 * C);
      // dW1, db1, dW2: weight-stationary, hid and d[a | g] recomputed from the tiles k_ff<true> left in w.dwide; the slab partials of every
      // block are summed in one launch behind the loop
      {
        dfx::ffused::FwArgs wa{w.ff_frags[i], w.ff_b1p[i] + dfx::ffused::B1P_FLOATS,
                               dx_in_ff ? reinterpret_cast<const uint4 *>(a.h1) : reinterpret_cast<const uint4 *>(w.dwide), w.ffw_part[i], w.ffw_bpart[i],
                               R / 32, w.ffw_slabs, dropout_p > 0.f ? reinterpret_cast<const unsigned *>(a.p) : nullptr,
                               hl_in ? reinterpret_cast<const uint4 *>(dh_in_blk) : reinterpret_cast<const uint4 *>(w.dwide) + dfx::ffused::PK_TILE_U4 / 2};
        // (without the attention inside — dfx_debug_train_fused(2) — k_ff<true> reads fp32 rows and leaves both fragment sets in w.dwide)
        if (dfx::ffused::launch_ff_wgrad(st, wa)) return dfx::set_error(DFX_ERR_HIP, "train: feed-forward weight-gradient launch");
      }
      // attention + LayerNorm2 (train_attn_fused.h): parameter side first (reads dh1 = w.dh2), then dh -> w.dh
      dfx::afused::AttnArgs aa{};
      aa.frags = w.at_frags[i], aa.valid = w.valid, aa.g2 = bw.norm2_w, aa.b2 = bw.norm2_b, aa.bo = bw.to_out_b;
      aa.h = a.hin, aa.dh1 = w.dh2, aa.dh = w.dh, aa.part = w.at_part[i], aa.cpart = w.cpart[i], aa.N = N, aa.split = w.at_split, aa.R = R;
      aa.pk2 = dx_in_ff ? reinterpret_cast<const uint4 *>(w.dq) : nullptr;
      if (dx_in_ff) dfx::afused::k_attn_bwd_param<true><<<B * w.at_split, dfx::afused::NW * 64, 0, st>>>(aa);
      else dfx::afused::k_attn_bwd_param<false><<<B * w.at_split, dfx::afused::NW * 64, 0, st>>>(aa);
      const int np = dfx::afused::dx_groups(R);
      if (!dx_in_ff) {
        dfx::afused::k_attn_bwd_dx<<<np, dfx::afused::NW * 64, 0, st>>>(aa);
        k_sum_parts_multi<<<3 * C / 32, 1024, 0, st>>>(w.cpart[i], SumOuts{{mut(gw.norm2_w), mut(gw.norm2_b), mut(gw.to_out_b), nullptr}}, np, C, 3 * C);
      }
    } else {
    if ((rc = wgrad(st, w, w.dh, C, a.hid, FH, mut(gw.ff2_w), mut(gw.ff2_b), C, FH, FH, R, false, bf))) return rc;
    transpose(st, bw.ff2_w, w.wT, C, FH);                                    // (512, 128)
    if ((rc = lin(st, w.dh, C, w.wT, nullptr, w.dhid, FH, R, FH, C))) return rc;
    const Drop dff{dropout_p, dropout_seed, (unsigned)(2 * i + 1)};
    if (bf) k_geglu_bwd<true><<<(int)((R * FH / 4 + 255) / 256), 256, 0, st>>>(a.ag, w.dhid, w.dwide, FH, R * FH, dff);
    else k_geglu_bwd<false><<<(int)((R * FH / 4 + 255) / 256), 256, 0, st>>>(a.ag, w.dhid, w.dwide, FH, R * FH, dff);
    if ((rc = wgrad(st, w, w.dwide, 2 * FH, a.xn3, C, mut(gw.ff0_w), mut(gw.ff0_b), 2 * FH, C, C, R, bf, bf))) return rc;
    transpose(st, bw.ff0_w, w.wT, 2 * FH, C);                                // (128, 1024)
    if ((rc = lin(st, w.dwide, 2 * FH, w.wT, nullptr, w.dh2, C, R, C, 2 * FH, nullptr, 0, bf))) return rc;
    if ((rc = ln_bwd(st, w, w.dh2, a.h1, a.st3, bw.norm3_w, w.dh, w.dh, mut(gw.norm3_w), mut(gw.norm3_b), R))) return rc;
    // attention: h1 = hin + Wo att + bo
    const float *dho = w.dh;   // gradient at the output of to_out: dh behind the dropout
    if (dropout_p > 0.f) {
      k_dropout<<<(int)((R * C / 4 + 255) / 256), 256, 0, st>>>(w.dh, nullptr, w.dh2, R * C, Drop{dropout_p, dropout_seed, (unsigned)(2 * i)});
      dho = w.dh2;
    }
    if ((rc = wgrad(st, w, dho, C, a.att, C, mut(gw.to_out_w), mut(gw.to_out_b), C, C, C, R, false, bf))) return rc;
    transpose(st, bw.to_out_w, w.wT, C, C);
    if ((rc = lin(st, dho, C, w.wT, nullptr, w.datt, C, R, C, C))) return rc;
    const int nab = (N + ATT_BP - 1) / ATT_BP;
    if (bf) k_attn_bwd<true><<<dim3(nab, B), 256, 0, st>>>(w.datt, a.q, a.k, a.v, a.p, w.dq, w.apart, N);
    else k_attn_bwd<false><<<dim3(nab, B), 256, 0, st>>>(w.datt, a.q, a.k, a.v, a.p, w.dq, w.apart, N);
    k_attn_bwd_finish<<<B, J * C, 0, st>>>(w.apart, w.dk, w.dv, nab);
    if ((rc = wgrad(st, w, w.dq, C, a.xn2, C, mut(gw.to_q), nullptr, C, C, C, R, bf, bf))) return rc;
    transpose(st, bw.to_q, w.wT, C, C);
    if ((rc = lin(st, w.dq, C, w.wT, nullptr, w.dh2, C, R, C, C, nullptr, 0, bf))) return rc;
    if ((rc = ln_bwd(st, w, w.dh2, a.hin, a.st2, bw.norm2_w, w.dh, w.dh, mut(gw.norm2_w), mut(gw.norm2_b), R))) return rc;
    }
    if (fused) continue;   // (keys / values of all blocks: one pair of products behind the loop)
    // keys / values of the 4 context tokens
    if ((rc = wgrad(st, w, w.dk, C, w.ctx, CTXP, mut(gw.to_k), nullptr, C, CTXP, CTX, BJ))) return rc;
    if ((rc = wgrad(st, w, w.dv, C, w.ctx, CTXP, mut(gw.to_v), nullptr, C, CTXP, CTX, BJ))) return rc;
    DFX_HIP_TRY(hipMemsetAsync(w.wT, 0, sizeof(float) * (size_t)CTXP * C, st));
    transpose(st, bw.to_k, w.wT, C, CTX);                                    // (522 -> 528 rows of 128)
    if ((rc = lin(st, w.dk, C, w.wT, nullptr, w.dctx, CTXP, BJ, CTXP, C, w.dctx, CTXP))) return rc;
    transpose(st, bw.to_v, w.wT, C, CTX);
    if ((rc = lin(st, w.dv, C, w.wT, nullptr, w.dctx, CTXP, BJ, CTXP, C, w.dctx, CTXP))) return rc;
  }
  if (ff_fused(bf, dropout_p, R, N)) {
    // parameter gradients of ALL blocks from the partials the block loop left behind: one launch per kind
    const int LDKV0 = 2 * wt->depth * C;
    dfx::ffused::FwFinishBatch fb{};
    dfx::afused::UnfoldBatch ub{};
    SumJobs sj{};
    for (int i = 0; i < wt->depth; ++i) {
      const dfx_block_weights &bw = wt->blk[i], &gw = grads->blk[i];
      fb.blk[i] = dfx::ffused::FwFinishArgs{w.ffw_part[i], w.ffw_bpart[i], mut(gw.ff0_w), mut(gw.ff0_b), mut(gw.ff2_w), w.ffw_slabs,
                                            dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f, bw.norm3_w, bw.norm3_b, bw.ff0_w,
                                            w.ffw_lnpart[i], mut(gw.norm3_w), mut(gw.norm3_b)};   // (d gamma3 / d beta3: k_ln3_param behind the slab sums)
      ub.blk[i] = dfx::afused::UnfoldArgs{w.at_part[i], w.kv + 2 * i * C, w.kv + (2 * i + 1) * C, bw.to_q, bw.to_out_w, w.dkv + 2 * i * C, w.dkv + (2 * i + 1) * C,
                                          mut(gw.to_q), mut(gw.to_out_w), w.at_sum[i], B, w.at_split, LDKV0};
      sj.part[i] = w.cpart[i];
      float *o6[6] = {nullptr, nullptr, mut(gw.ff2_b), mut(gw.norm2_w), mut(gw.norm2_b), mut(gw.to_out_b)};   // (rows 0, 1 of the partials are unused)
      for (int q = 0; q < 6; ++q) sj.o[i][q] = o6[q];
    }
    // the chain towards the context gradient (unfold_kv -> d Wk / d Wv / d ctx -> time-embedding MLP) stays on the caller's stream; the
    // leaves (slab sums of dW1 / dW2, dWq / dWo, the six column sums per block) go beside it
    if ((rc = ss.fork())) return rc;
    hipStream_t sl = (ssm & SS_LEAVES) ? sp : st;
    dfx::afused::k_attn_unfold_kv<<<dim3(dfx::afused::J, B, wt->depth), 256, 0, st>>>(ub);
    if (ss.on && !stem_done && dx_in_ff_all && (ssm & SS_STEM_END)) stem_backward(sp, w.datt), stem_done = true;
    dfx::ffused::launch_ff_wgrad_finish(sl, fb, wt->depth);
    if (t_attn_in_ff) k_sum_parts_jobs<<<wt->depth * 6 * (C / 32), 1024, 0, sl>>>(sj, (int)dfx::ffused::ff_groups(B, N), C, 6 * C);
    if (!head_done && head_sums) head_sums(sp), head_done = true;
    if ((rc = ss.fork())) return rc;   // (k_attn_unfold_w reads the per-shape sums k_attn_unfold_kv leaves in at_sum)
    dfx::afused::k_attn_unfold_w<<<dim3(C / dfx::afused::UW_D, 1, wt->depth), 1024, 0, sl>>>(ub);
    // d Wk, d Wv of every block and d ctx from the side-by-side key / value gradients
    const int n2 = 2 * wt->depth, LDKV = n2 * C;
    if ((rc = wgrad(st, w, w.dkv, LDKV, w.ctx, CTXP, w.dwkv, nullptr, LDKV, CTXP, CTXP, BJ))) return rc;
    KvMutPtrs gp{};
    for (int i = 0; i < wt->depth; ++i) gp.p[2 * i] = mut(grads->blk[i].to_k), gp.p[2 * i + 1] = mut(grads->blk[i].to_v);
    k_unpack_kv<<<(n2 * C * CTX + 255) / 256, 256, 0, st>>>(w.dwkv, gp, n2);
    transpose(st, w.wkv, w.wkvT, LDKV, CTXP);                                  // (528 rows of 2 depth x 128)
    if ((rc = lin(st, w.dkv, LDKV, w.wkvT, nullptr, w.dctx, CTXP, BJ, CTXP, LDKV))) return rc;
  }
  // pre_norm, proj_in
  if (fused_ends) {
    if (!stem_done) stem_backward(st, w.part);
  } else {
    if ((rc = ln_bwd(st, w, w.dh, w.h0, w.st_pre, wt->pre_norm_w, nullptr, w.dh2, mut(grads->pre_norm_w), mut(grads->pre_norm_b), R))) return rc;
    if ((rc = wgrad(st, w, w.dh2, C, w.xin, XIN, mut(grads->proj_in_w), mut(grads->proj_in_b), C, XIN, 13, R))) return rc;
  }
  // gradient at the input rows (optional; w.dh = the gradient behind pre_norm is final and row-major on either path)
  if (d_x || d_variances) k_stem_dx<<<2048, 256, 0, st>>>(dh_cur, w.xin, wt->proj_in_w, wt->proj_in_b, wt->pre_norm_w, d_x, d_variances, N, R);
  // context -> part codes, (mean, var), time embedding MLP
  k_ctx_bwd<<<(B * CTX + 255) / 256, 256, 0, st>>>(w.dctx, d_ctx_code, d_ctx_mv, w.dte_out, B);
  if ((rc = wgrad(st, w, w.dte_out, TE, w.te_hid, TEH, mut(grads->te2_w), mut(grads->te2_b), TE, TEH, TEH, B))) return rc;
  transpose(st, wt->te2_w, w.wT, TE, TEH);                                   // (1024, 256)
  if ((rc = lin(st, w.dte_out, TE, w.wT, nullptr, w.dte_hid, TEH, B, TEH, TE))) return rc;
  k_geglu_bwd<false><<<(int)(((long long)B * TEH / 4 + 255) / 256), 256, 0, st>>>(w.te_ag, w.dte_hid, w.dte_ag, TEH, (long long)B * TEH, Drop{dropout_p, dropout_seed, SITE_TE});
  if ((rc = wgrad(st, w, w.dte_ag, 2 * TEH, w.te_in, TE, mut(grads->te0_w), mut(grads->te0_b), 2 * TEH, TE, TE, B))) return rc;
  if ((rc = ss.join())) return rc;
  return dfx::check_launch("denoiser_train_backward");
}

int dfx_masked_mse_backward_f32(const float *target, const float *pred, const float *flags, const double *workspace2,
                                float grad_scale, float *d_pred, int B, int N, dfx_stream_t stream) {
  DFX_REQUIRE(target && pred && workspace2 && d_pred && B >= 1 && N >= 1, "masked_mse_backward: bad argument");
  const long long total = (long long)B * N;
  k_mse_bwd<<<(int)((total + 255) / 256), 256, 0, dfx::as_stream(stream)>>>(target, pred, flags, workspace2, grad_scale, d_pred, N,
                                                                               total, 3.0 * (double)B * N);
  return dfx::check_launch("masked_mse_backward");
}

// ---- PointNetV2 part encoder, train mode (pointnet.py:187-213 with nn.BatchNorm1d in training: batch statistics) ----
size_t dfx_pointnet_v2_train_workspace_bytes(int B, int N, int num_anchors, int zdim) {
  if (B < 2 || N < 1 || num_anchors != 4 || zdim < 4) return 0;
  PnWs t;
  return carve_pn(t, nullptr, B, N, num_anchors, zdim);
}

int dfx_pointnet_v2_train_forward(const dfx_pointnet_v2_weights *wt, void *workspace, size_t workspace_bytes, const float *x,
                                  const float *attn, float *m, float *v, float momentum, int B, int N, int precision,
                                  dfx_stream_t stream) {
  int rc = check_pn(wt, workspace, workspace_bytes, B, N, "pointnet_v2_train_forward");
  if (rc) return rc;
  DFX_REQUIRE(x && attn && m && v, "pointnet_v2_train_forward: null tensor");
  DFX_REQUIRE(precision == DFX_PREC_F32 || precision == DFX_PREC_BF16, "pointnet_v2_train_forward: precision %d", precision);
  g_prec = precision;
  hipStream_t st = dfx::as_stream(stream);
  const int A = wt->num_anchors, zd = wt->zdim;
  PnWs w;
  carve_pn(w, workspace, B, N, A, zd);
  const long long R = (long long)B * N;
  k_pn_rows<<<(int)((R + 255) / 256), 256, 0, st>>>(x, w.X8, R);
  k_pad_cols<<<(128 * 8 + 255) / 256, 256, 0, st>>>(wt->conv_w[0], w.wpad, 128, 3, 8);
  for (int l = 0; l < 4; ++l) {
    const float *in = l == 0 ? w.X8 : w.y[l - 1];
    const int K = PN_C[l], Co = PN_C[l + 1];
    // fp32 trunk: the product's epilogue leaves the BatchNorm batch statistics of its output (per-workgroup (count, mean, M2) partials): the two
    // statistics passes over z (0.64 ms of the 2.1 ms forward at 128 x 2048 points) are gone
    int parts = 0;
    if (g_prec == DFX_PREC_F32 && g_bn_fused_stats) {
      LinArgs la{};
      la.X = in, la.ldx = K, la.W = l == 0 ? w.wpad : wt->conv_w[l], la.b = wt->conv_b[l], la.Y = w.z[l], la.ldy = Co, la.M = (int)R, la.N = Co, la.K = K;
      la.stats = w.pb.part;
      if ((size_t)std::min<long long>((R + 255) / 256, 512) * Co * 3 <= w.pb.part_floats) parts = dfx::lin::launch_wide_lds_stats(st, la);
    }
    if (!parts && (rc = lin(st, in, K, l == 0 ? w.wpad : wt->conv_w[l], wt->conv_b[l], w.z[l], Co, R, Co, K))) return rc;
    if ((rc = bn_fwd(st, w, w.z[l], R, Co, wt->bn_w[l], wt->bn_b[l], mut(wt->bn_mean[l]), mut(wt->bn_var[l]), momentum, wt->bn_eps,
                     w.mean[l], w.rstd[l], w.y[l], l < 3, l < 3, parts))) return rc;
  }
  const float scale = wt->reweight_by_anchor ? (float)A : 1.0f;
  k_pool_fwd<4, 16><<<dim3(512 / 64, B), 1024, 0, st>>>(w.z[3], w.mean[3], w.rstd[3], wt->bn_w[3], wt->bn_b[3], attn, w.pooled, w.arg, N, 512, scale);
  const int hc[3] = {512, 256, 128};
  for (int k = 0; k < 2; ++k) {
    const float *in = w.pooled;
    for (int l = 0; l < 2; ++l) {
      if ((rc = glin(st, in, wt->head_w[k][l], wt->head_b[k][l], w.hz[k][l], B, A, hc[l], hc[l + 1]))) return rc;
      if ((rc = bn_fwd(st, w, w.hz[k][l], B, A * hc[l + 1], wt->head_bn_w[k][l], wt->head_bn_b[k][l], mut(wt->head_bn_mean[k][l]),
                       mut(wt->head_bn_var[k][l]), momentum, wt->bn_eps, w.hmean[k][l], w.hrstd[k][l], w.hy[k][l], true))) return rc;
      in = w.hy[k][l];
    }
    if ((rc = glin(st, in, wt->head_w[k][2], wt->head_b[k][2], k == 0 ? m : v, B, A, 128, zd))) return rc;
  }
  return dfx::check_launch("pointnet_v2_train_forward");
}

int dfx_pointnet_v2_train_backward(const dfx_pointnet_v2_weights *wt, void *workspace, size_t workspace_bytes, const float *attn,
                                   const float *dm, const float *dv, const dfx_pointnet_v2_weights *grads, int B, int N,
                                   int precision, dfx_stream_t stream) {
  int rc = check_pn(wt, workspace, workspace_bytes, B, N, "pointnet_v2_train_backward");
  if (rc) return rc;
  DFX_REQUIRE(attn && dm && dv && grads, "pointnet_v2_train_backward: null tensor");
  DFX_REQUIRE(precision == DFX_PREC_F32 || precision == DFX_PREC_BF16, "pointnet_v2_train_backward: precision %d", precision);
  g_prec = precision;
  hipStream_t st = dfx::as_stream(stream);
  const int A = wt->num_anchors, zd = wt->zdim;
  PnWs w;
  carve_pn(w, workspace, B, N, A, zd);
  const long long R = (long long)B * N;
  DFX_HIP_TRY(hipMemsetAsync(w.dpooled, 0, sizeof(float) * (size_t)B * A * 512, st));
  const int hc[4] = {512, 256, 128, zd};
  for (int k = 0; k < 2; ++k) {
    const float *dcur = k == 0 ? dm : dv;   // gradient at the output of head layer l (before its BatchNorm for l < 2)
    for (int l = 2; l >= 0; --l) {
      const int cin = hc[l], cout = hc[l + 1];
      const float *in = l == 0 ? w.pooled : w.hy[k][l - 1];
      const float *dz = dcur;
      if (l < 2) {   // through relu + BatchNorm over the B rows
        if ((rc = bn_bwd(st, w, dcur, wt->head_bn_b[k][l], w.hz[k][l], B, A * cout, wt->head_bn_w[k][l], w.hmean[k][l], w.hrstd[k][l], w.dh[1],
                         mut(grads->head_bn_w[k][l]), mut(grads->head_bn_b[k][l]), true))) return rc;
        dz = w.dh[1];
      }
      if (A == NPART) {   // the four parts' weight gradient, transpose and input gradient as ONE launch each (the flows' grouped kernels): 9 launches
                          // per head layer instead of ~14 x 4 — the heads' backward was ~100 launches of ~5 us
        Ptr4 gw{}, gb{};
        CPtr4 wp{};
        for (int a = 0; a < A; ++a)
          gw.p[a] = mut(grads->head_w[k][l]) + (size_t)a * cout * cin, gb.p[a] = mut(grads->head_b[k][l]) + a * cout, wp.p[a] = wt->head_w[k][l] + (size_t)a * cout * cin;
        const long long wt_gs = (long long)cout * cin;
        k_wgrad_g4<<<dim3((cin + 63) / 64, (cout + 63) / 64, NPART), 256, 0, st>>>(dz, A * cout, cout, in, A * cin, cin, gw, gb, cout, cin, B);
        k_transpose_g4<<<dim3((cin + 31) / 32, (cout + 31) / 32, NPART), 256, 0, st>>>(wp, w.wT, wt_gs, cout, cin);   // (cin, cout) per part
        if (l == 0) {   // d pooled accumulates over the two heads
          if ((rc = lin_g4<dfx::lin::EPI_RESID>(st, dz, A * cout, cout, nullptr, nullptr, w.wT, wt_gs, w.dpooled, A * cin, cin, B, cin, cout, w.dpooled, A * cin, cin))) return rc;
        } else if ((rc = lin_g4<dfx::lin::EPI_NONE>(st, dz, A * cout, cout, nullptr, nullptr, w.wT, wt_gs, w.dh[0], A * cin, cin, B, cin, cout))) return rc;
      } else
      for (int a = 0; a < A; ++a) {   // per-part weights: group a of the grouped 1x1 convolution
        if ((rc = wgrad(st, w.pb, dz + a * cout, A * cout, in + a * cin, A * cin, mut(grads->head_w[k][l]) + (size_t)a * cout * cin,
                        mut(grads->head_b[k][l]) + a * cout, cout, cin, cin, B))) return rc;
        transpose(st, wt->head_w[k][l] + (size_t)a * cout * cin, w.wT, cout, cin);   // (cin, cout)
        if (l == 0) {   // d pooled accumulates over the two heads
          if ((rc = lin(st, dz + a * cout, A * cout, w.wT, nullptr, w.dpooled + a * cin, A * cin, B, cin, cout, w.dpooled + a * cin, A * cin))) return rc;
        } else if ((rc = lin(st, dz + a * cout, A * cout, w.wT, nullptr, w.dh[0] + a * cin, A * cin, B, cin, cout))) return rc;
      }
      dcur = w.dh[0];
    }
  }
  // max-pool: the gradient goes to the arg-max point of every (shape, part, channel) — and the last BatchNorm's backward works from
  // those entries, no dense dy
  float *dy = w.dA, *dz = w.dB;
  {
    const float scale = wt->reweight_by_anchor ? (float)A : 1.0f;
    k_pool_bwd_stats<4><<<512 / 32, 1024, 0, st>>>(w.dpooled, w.arg, attn, w.z[3], w.mean[3], w.rstd[3], mut(grads->bn_b[3]), mut(grads->bn_w[3]), B, N, 512, scale);
    const long long total = R * 512;
    k_bn_bwd_apply<false, true><<<(int)((total / 4 + 255) / 256), 256, 0, st>>>(nullptr, nullptr, w.z[3], w.mean[3], w.rstd[3], wt->bn_w[3], grads->bn_b[3],
                                                                                 grads->bn_w[3], dz, 1.0f / (float)R, total, 512);
    k_pool_bwd_fix<4><<<dim3(2, B), 256, 0, st>>>(w.dpooled, w.arg, attn, w.z[3], w.mean[3], w.rstd[3], wt->bn_w[3], grads->bn_b[3], grads->bn_w[3], dz, N, 512,
                                                  scale, 1.0f / (float)R);
  }
  for (int l = 3; l >= 0; --l) {
    const int K = PN_C[l], Co = PN_C[l + 1];
    if (l < 3 && (rc = bn_bwd(st, w, dy, wt->bn_b[l], w.z[l], R, Co, wt->bn_w[l], w.mean[l], w.rstd[l], dz, mut(grads->bn_w[l]), mut(grads->bn_b[l]), true))) return rc;
    if ((rc = wgrad(st, w.pb, dz, Co, l == 0 ? w.X8 : w.y[l - 1], K, mut(grads->conv_w[l]), mut(grads->conv_b[l]), Co, K, l == 0 ? 3 : K, R))) return rc;
    if (l > 0) {
      transpose(st, wt->conv_w[l], w.wT, Co, K);   // (K, Co)
      if ((rc = lin(st, dz, Co, w.wT, nullptr, dy, K, R, K, Co))) return rc;
    }
  }
  return dfx::check_launch("pointnet_v2_train_backward");
}

// ---- prior loss (PartEncoder.get_prior_loss, part_encoders.py:1143-1182, use_flow = True) ----
size_t dfx_prior_loss_workspace_bytes(int B, int flow_depth, int flow_hidden) {
  if (B < 1 || flow_depth < 1 || flow_depth > DFX_MAX_FLOW_DEPTH || flow_hidden < 8) return 0;
  FlowWs t;
  return carve_flow(t, nullptr, B, flow_depth, flow_hidden);
}

int dfx_prior_loss_forward(const float *const *flow, int flow_depth, int flow_hidden, void *workspace, size_t workspace_bytes,
                           const float *part_code, const float *logvar, const float *valid, float prior_var, float kl_weight,
                           float *loss, float *log_p_part, float *entropy, int B, dfx_stream_t stream) {
  int rc = check_flow(flow, flow_depth, flow_hidden, workspace, workspace_bytes, B, "prior_loss_forward");
  if (rc) return rc;
  DFX_REQUIRE(part_code && logvar && valid && loss && prior_var > 0.f, "prior_loss_forward: bad argument");
  hipStream_t st = dfx::as_stream(stream);
  const int H = flow_hidden, Rf = NPART * B;
  const int prec_saved = g_prec;
  g_prec = DFX_PREC_F32;   // few-row products: exact fp32
  FlowWs w;
  carve_flow(w, workspace, B, flow_depth, H);
  k_flow_gather<<<(Rf * ZD + 255) / 256, 256, 0, st>>>(part_code, w.xs[0], B);
  DFX_HIP_TRY(hipMemsetAsync(w.logdet, 0, sizeof(float) * Rf, st));
  for (int l = 0; l < flow_depth; ++l) {
    const int swap = (l % 2 == 0), xc = swap ? ZH : 0;
    const float *wt[3][NPART], *bt[3][NPART];
    for (int i = 0; i < NPART; ++i)
      for (int k = 0; k < 3; ++k) wt[k][i] = flow[((size_t)i * flow_depth + l) * 6 + 2 * k], bt[k][i] = flow[((size_t)i * flow_depth + l) * 6 + 2 * k + 1];
    // net_s_t of the four parts in one launch each (rows part-major: group stride = B rows)
    if ((rc = lin_g4<dfx::lin::EPI_RELU>(st, w.xs[l] + xc, ZD, (long long)B * ZD, wt[0], bt[0], nullptr, 0, w.h1[l], H, (long long)B * H, B, H, ZH))) return rc;
    if ((rc = lin_g4<dfx::lin::EPI_RELU>(st, w.h1[l], H, (long long)B * H, wt[1], bt[1], nullptr, 0, w.h2[l], H, (long long)B * H, B, H, H))) return rc;
    if ((rc = lin_g4<dfx::lin::EPI_NONE>(st, w.h2[l], H, (long long)B * H, wt[2], bt[2], nullptr, 0, w.st[l], ZD, (long long)B * ZD, B, ZD, H))) return rc;
    k_coupling_fwd<<<Rf, 128, 0, st>>>(w.xs[l], w.st[l], w.xs[l + 1], w.logdet, swap);
  }
  k_prior_terms<<<Rf, 256, 0, st>>>(w.xs[flow_depth], w.logdet, logvar, valid, prior_var, kl_weight, B, w.logp, w.ent, w.acoef);
  k_prior_loss<<<1, 256, 0, st>>>(w.logp, w.ent, w.acoef, Rf, loss);
  if (log_p_part || entropy) {   // (B, 4) views of the part-major rows
    for (int i = 0; i < NPART; ++i) {
      if (log_p_part) DFX_HIP_TRY(hipMemcpy2DAsync(log_p_part + i, NPART * sizeof(float), w.logp + (size_t)i * B, sizeof(float), sizeof(float), B, hipMemcpyDeviceToDevice, st));
      if (entropy) DFX_HIP_TRY(hipMemcpy2DAsync(entropy + i, NPART * sizeof(float), w.ent + (size_t)i * B, sizeof(float), sizeof(float), B, hipMemcpyDeviceToDevice, st));
    }
  }
  g_prec = prec_saved;
  return dfx::check_launch("prior_loss_forward");
}

int dfx_prior_loss_backward(const float *const *flow, int flow_depth, int flow_hidden, void *workspace, size_t workspace_bytes,
                            const float *valid, float prior_var, float grad_scale, float *const *flow_grads, float *d_part_code,
                            float *d_logvar, int B, dfx_stream_t stream) {
  int rc = check_flow(flow, flow_depth, flow_hidden, workspace, workspace_bytes, B, "prior_loss_backward");
  if (rc) return rc;
  DFX_REQUIRE(valid && flow_grads && prior_var > 0.f, "prior_loss_backward: bad argument");
  for (int k = 0; k < NPART * flow_depth * 6; ++k) DFX_REQUIRE(flow_grads[k], "prior_loss_backward: null gradient buffer %d", k);
  hipStream_t st = dfx::as_stream(stream);
  const int H = flow_hidden, Rf = NPART * B;
  const int prec_saved = g_prec;
  g_prec = DFX_PREC_F32;
  FlowWs w;
  carve_flow(w, workspace, B, flow_depth, H);
  k_prior_bwd_init<<<(Rf * ZD + 255) / 256, 256, 0, st>>>(w.xs[flow_depth], w.acoef, valid, grad_scale, prior_var, B, w.dy, w.dlogdet, d_logvar);
  float *dy = w.dy, *dx = w.dy2;
  for (int l = flow_depth - 1; l >= 0; --l) {
    const int swap = (l % 2 == 0), xc = swap ? ZH : 0;
    k_coupling_bwd<<<Rf, 128, 0, st>>>(dy, w.xs[l], w.st[l], w.dlogdet, w.dst, dx, swap);
    CPtr4 wp[3];
    Ptr4 gw[3], gb[3];
    for (int i = 0; i < NPART; ++i)
      for (int k = 0; k < 3; ++k) {
        wp[k].p[i] = flow[((size_t)i * flow_depth + l) * 6 + 2 * k];
        gw[k].p[i] = flow_grads[((size_t)i * flow_depth + l) * 6 + 2 * k];
        gb[k].p[i] = flow_grads[((size_t)i * flow_depth + l) * 6 + 2 * k + 1];
      }
    const long long gsZ = (long long)B * ZD, gsH = (long long)B * H;
    const size_t hm = (size_t)(H > ZD ? H : ZD);
    const long long wt_gs = (long long)(hm * hm);
    // net_s_t.4: s_t = h2 W3^T + b3   (the four parts per launch)
    k_wgrad_g4<<<dim3((H + 63) / 64, (ZD + 63) / 64, NPART), 256, 0, st>>>(w.dst, ZD, gsZ, w.h2[l], H, gsH, gw[2], gb[2], ZD, H, B);
    k_transpose_g4<<<dim3((H + 31) / 32, (ZD + 31) / 32, NPART), 256, 0, st>>>(wp[2], w.wT, wt_gs, ZD, H);
    if ((rc = lin_g4<dfx::lin::EPI_NONE>(st, w.dst, ZD, gsZ, nullptr, nullptr, w.wT, wt_gs, w.dh2, H, gsH, B, H, ZD))) return rc;
    k_relu_mask<<<(int)(((long long)Rf * H + 255) / 256), 256, 0, st>>>(w.dh2, w.h2[l], (long long)Rf * H);
    // net_s_t.2
    k_wgrad_g4<<<dim3((H + 63) / 64, (H + 63) / 64, NPART), 256, 0, st>>>(w.dh2, H, gsH, w.h1[l], H, gsH, gw[1], gb[1], H, H, B);
    k_transpose_g4<<<dim3((H + 31) / 32, (H + 31) / 32, NPART), 256, 0, st>>>(wp[1], w.wT, wt_gs, H, H);
    if ((rc = lin_g4<dfx::lin::EPI_NONE>(st, w.dh2, H, gsH, nullptr, nullptr, w.wT, wt_gs, w.dh1, H, gsH, B, H, H))) return rc;
    k_relu_mask<<<(int)(((long long)Rf * H + 255) / 256), 256, 0, st>>>(w.dh1, w.h1[l], (long long)Rf * H);
    // net_s_t.0: input = the conditioning half of x; its gradient is added to the pass-through gradient already in dx
    k_wgrad_g4<<<dim3((ZH + 63) / 64, (H + 63) / 64, NPART), 256, 0, st>>>(w.dh1, H, gsH, w.xs[l] + xc, ZD, gsZ, gw[0], gb[0], H, ZH, B);
    k_transpose_g4<<<dim3((ZH + 31) / 32, (H + 31) / 32, NPART), 256, 0, st>>>(wp[0], w.wT, wt_gs, H, ZH);
    if ((rc = lin_g4<dfx::lin::EPI_RESID>(st, w.dh1, H, gsH, nullptr, nullptr, w.wT, wt_gs, dx + xc, ZD, gsZ, B, ZH, H, dx + xc, ZD, gsZ))) return rc;
    float *t = dy;
    dy = dx, dx = t;
  }
  if (d_part_code) k_flow_scatter<<<(Rf * ZD + 255) / 256, 256, 0, st>>>(dy, d_part_code, B);
  g_prec = prec_saved;
  return dfx::check_launch("prior_loss_backward");
}

// ---- PointNet++ shared MLP, training mode ----
size_t dfx_shared_mlp_train_workspace_bytes(const dfx_shared_mlp_train *t, int B, int M, int ns) {
  if (!t || t->layers < 1 || t->layers > DFX_MLP_MAX_LAYERS || B < 1 || M < 1 || ns < 1) return 0;
  SmtWs w;
  return carve_smt(w, nullptr, t, (long long)B * M * ns, (long long)B * M);
}

int dfx_shared_mlp_train_forward(const dfx_shared_mlp_train *t, void *workspace, size_t workspace_bytes, const float *x, float *out, int B, int M,
                                 int ns, int pool, int batch_stats, float momentum, dfx_stream_t stream) {
  int rc = check_smt(t, workspace, workspace_bytes, B, M, ns, "shared_mlp_train_forward");
  if (rc) return rc;
  DFX_REQUIRE(x && out, "shared_mlp_train_forward: null tensor");
  const int prec_saved = g_prec;
  g_prec = DFX_PREC_F32;
  hipStream_t st = dfx::as_stream(stream);
  const long long G = (long long)B * M, R = G * ns, L = (long long)M * ns;
  SmtWs w;
  carve_smt(w, workspace, t, R, G);
  k_smt_to_rows<<<dim3((unsigned)((L + 31) / 32), (w.cp0 + 31) / 32, B), 256, 0, st>>>(x, w.rows0, t->ch[0], L, w.cp0);
  const float *in = w.rows0;
  int K = w.cp0;
  for (int l = 0; l < t->layers; ++l) {
    const int co = t->ch[l + 1];
    const float *W = t->conv_w[l];
    if (l == 0 && w.cp0 != t->ch[0]) {
      k_pad_cols<<<(co * w.cp0 + 255) / 256, 256, 0, st>>>(t->conv_w[0], w.wpad, co, t->ch[0], w.cp0);
      W = w.wpad;
    }
    if ((rc = lin(st, in, K, W, t->conv_b[l], w.z[l], co, R, co, K))) break;
    const bool relu = (t->relu_mask >> l) & 1u;
    const float *act = w.y[l];
    if (t->bn_w[l] && batch_stats) {
      if ((rc = bn_fwd(st, w.pn, w.z[l], R, co, t->bn_w[l], t->bn_b[l], t->bn_mean[l], t->bn_var[l], momentum, t->bn_eps, w.mean[l], w.rstd[l], w.y[l], relu))) break;
    } else if (t->bn_w[l]) {   // eval(): running statistics, nothing updated
      if (!t->bn_mean[l] || !t->bn_var[l]) { rc = dfx::set_error(DFX_ERR_INVALID_ARG, "shared_mlp_train_forward: layer %d: running statistics required", l); break; }
      k_smt_running_stats<<<(co + 255) / 256, 256, 0, st>>>(t->bn_mean[l], t->bn_var[l], w.mean[l], w.rstd[l], t->bn_eps, co);
      if (relu) k_bn_apply<true><<<(int)((R * co / 4 + 255) / 256), 256, 0, st>>>(w.z[l], w.mean[l], w.rstd[l], t->bn_w[l], t->bn_b[l], w.y[l], R * co, co);
      else k_bn_apply<false><<<(int)((R * co / 4 + 255) / 256), 256, 0, st>>>(w.z[l], w.mean[l], w.rstd[l], t->bn_w[l], t->bn_b[l], w.y[l], R * co, co);
    } else if (relu) {
      k_smt_relu<<<(int)((R * co + 255) / 256), 256, 0, st>>>(w.z[l], w.y[l], R * co);
    } else {
      act = w.z[l];   // a plain linear layer: its output is what the next layer (or the caller) sees
    }
    in = act, K = co;
  }
  g_prec = prec_saved;
  if (rc) return rc;
  const int cl = t->ch[t->layers];
  if (pool) k_smt_pool_fwd<<<dim3((cl + 63) / 64, (unsigned)G), 64, 0, st>>>(in, out, w.arg, M, ns, cl);
  else k_smt_from_rows<<<dim3((unsigned)((L + 31) / 32), (cl + 31) / 32, B), 256, 0, st>>>(in, out, cl, L, cl);
  return dfx::check_launch("shared_mlp_train_forward");
}

int dfx_shared_mlp_train_backward(const dfx_shared_mlp_train *t, void *workspace, size_t workspace_bytes, const float *d_out,
                                  const dfx_shared_mlp_train *grads, float *d_x, int B, int M, int ns, int pool, int batch_stats, dfx_stream_t stream) {
  int rc = check_smt(t, workspace, workspace_bytes, B, M, ns, "shared_mlp_train_backward");
  if (rc) return rc;
  DFX_REQUIRE(d_out && grads, "shared_mlp_train_backward: null argument");
  for (int l = 0; l < t->layers; ++l)
    DFX_REQUIRE(grads->conv_w[l] && (t->conv_b[l] ? grads->conv_b[l] != nullptr : true) && (t->bn_w[l] ? grads->bn_w[l] && grads->bn_b[l] : true),
                "shared_mlp_train_backward: null gradient buffer in layer %d", l);
  const int prec_saved = g_prec;
  g_prec = DFX_PREC_F32;
  hipStream_t st = dfx::as_stream(stream);
  const long long G = (long long)B * M, R = G * ns, L = (long long)M * ns;
  SmtWs w;
  carve_smt(w, workspace, t, R, G);
  const int cl = t->ch[t->layers];
  float *dy = w.dA, *other = w.dB;
  if (pool) k_smt_pool_bwd<<<dim3((cl + 63) / 64, (unsigned)G), 64, 0, st>>>(d_out, w.arg, dy, M, ns, cl);
  else k_smt_to_rows<<<dim3((unsigned)((L + 31) / 32), (cl + 31) / 32, B), 256, 0, st>>>(d_out, dy, cl, L, cl);
  for (int l = t->layers - 1; l >= 0 && !rc; --l) {
    const int co = t->ch[l + 1], ci = l == 0 ? w.cp0 : t->ch[l], ci_valid = t->ch[l];
    const bool relu = (t->relu_mask >> l) & 1u;
    const bool plain_below = l > 0 && !t->bn_w[l - 1] && !((t->relu_mask >> (l - 1)) & 1u);   // (the layer below handed its z on: see the forward)
    const float *xin = l == 0 ? w.rows0 : plain_below ? w.z[l - 1] : w.y[l - 1];
    float *dz = other;
    if (t->bn_w[l]) {
      rc = bn_bwd(st, w.pn, dy, t->bn_b[l], w.z[l], R, co, t->bn_w[l], w.mean[l], w.rstd[l], dz, mut(grads->bn_w[l]), mut(grads->bn_b[l]), relu, batch_stats != 0);
    } else if (relu) {
      k_smt_relu_bwd<<<(int)((R * co + 255) / 256), 256, 0, st>>>(dy, w.z[l], dz, R * co);
    } else {
      dz = dy;   // plain linear layer
    }
    if (rc) break;
    if ((rc = wgrad(st, w.pn.pb, dz, co, xin, ci, mut(grads->conv_w[l]), t->conv_b[l] ? mut(grads->conv_b[l]) : nullptr, co, ci, ci_valid, R))) break;
    if (l > 0 || d_x) {   // d (input rows) = dz W
      const float *W = t->conv_w[l];
      if (ci != ci_valid) {
        k_pad_cols<<<(co * ci + 255) / 256, 256, 0, st>>>(t->conv_w[l], w.wpad, co, ci_valid, ci);
        W = w.wpad;
      }
      transpose(st, W, w.wT, co, ci);   // (ci, co)
      float *dprev = dz == dy ? other : dy;   // (dy's buffer is free once dz has been formed; a plain layer's dz IS dy: the other buffer then)
      if ((rc = lin(st, dz, co, w.wT, nullptr, dprev, ci, R, ci, co))) break;
      if (dprev != dy) std::swap(dy, other);
    }
    // next layer down: its dy is what was just written into `dy`; dz's buffer becomes the scratch
  }
  g_prec = prec_saved;
  if (rc) return rc;
  if (d_x) k_smt_from_rows<<<dim3((unsigned)((L + 31) / 32), (t->ch[0] + 31) / 32, B), 256, 0, st>>>(dy, d_x, t->ch[0], L, w.cp0);
  return dfx::check_launch("shared_mlp_train_backward");
}

// The dropout factors (0 or 1 / (1 - p)) of `n` consecutive elements of a site, as the training kernels apply them
// (site 2 i: behind to_out of block i, over (B N, 128); 2 i + 1: behind the GEGLU of block i, over (B N, 512); 1000: time_embed)
void dfx_debug_train_fused(int on) { g_ff_fused = on != 0, g_attn_in_ff = on != 2; }
const char *dfx_debug_last_train_path(void) { return t_last_path; }
void dfx_debug_bn_fused_stats(int on) { g_bn_fused_stats = on != 0; }
// Host-side run of the statistics arithmetic of k_lin_wide_lds<.., LM_STATS> / k_bn_merge (the same stats_merge, compiled for the host): `n` values cut
// into pieces of `chunk` (a lane's 16 rows of a tile), each piece as (count, mean, sum of squared deviations from its own mean), merged left to right.
void dfx_debug_stats_merge(const float *values, int n, int chunk, float *out3) {
  float cnt = 0.f, mean = 0.f, m2 = 0.f;
  for (int i0 = 0; i0 < n; i0 += chunk) {
    const int i1 = i0 + chunk < n ? i0 + chunk : n;
    float s = 0.f;
    for (int i = i0; i < i1; ++i) s += values[i];
    const float c = (float)(i1 - i0), mt = s / c;
    float q = 0.f;
    for (int i = i0; i < i1; ++i) {
      const float d = values[i] - mt;
      q = fmaf(d, d, q);
    }
    dfx::lin::stats_merge(cnt, mean, m2, c, mt, q);
  }
  out3[0] = cnt, out3[1] = mean, out3[2] = m2;
}
void dfx_debug_train_streams(int on) { g_train_streams = on < 0 ? 0 : on; }
// Host-side evaluation of the fused training kernels' row addressing (ffused::RowMap, the code the kernels compile): for a 32-point tile, the float
// offset of every (point, channel) as the B-operand-layout accessors and as the accumulator-layout accessors see it.  out_b, out_a: [32][128] int32.
void dfx_debug_rowmap(int tiled, int *out_b, int *out_a) {
  for (int lane = 0; lane < 64; ++lane) {
    const int pj = lane & 31, hf = lane >> 5;
    const dfx::ffused::RowMap m(tiled != 0, lane, pj, hf);
    for (int c = 0; c < 4; ++c) {
      for (int u = 0; u < 2; ++u)
        for (int e = 0; e < 8; ++e) out_b[pj * 128 + 32 * c + dfx::ffused::k_nat(u, hf, e)] = (int)m.b(c, u, e >> 2) + (e & 3);
      for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 4; ++k) out_a[pj * 128 + 32 * c + dfx::ffused::rho(4 * q + k, hf)] = (int)m.a(c, q) + k;
    }
  }
}

int dfx_debug_dropout_factors(uint64_t seed, int site, float p, float *out, long long n, dfx_stream_t stream) {
  DFX_REQUIRE(out && n >= 4 && n % 4 == 0 && p > 0.f && p < 1.f, "debug_dropout_factors: bad argument");
  k_dropout_factors<<<(int)((n / 4 + 255) / 256), 256, 0, dfx::as_stream(stream)>>>(out, n, Drop{p, seed, (unsigned)site});
  return dfx::check_launch("debug_dropout_factors");
}

// Test hook for the bf16 product kernels of gemm_bf16.h (tests/test_gpu_train.py):
//   tn = 0:  C (M, N) = A (M, K) B (N, K)^T + bias + resid                      (A fp32 or bf16-stored, B fp32)
//   tn = 1:  C (M, N) = A (K, M)^T B (K, N), db (M) = column sums of A          (A, B fp32 or bf16-stored); workspace for the
//            per-slab partials: (K / 64 + 1) * (M * N + M) floats
int dfx_debug_gemm_bf16(int tn, const void *A, int lda, int a_bf16, const void *B, int ldb, int b_bf16, const float *bias,
                        const float *resid, float *Cout, float *db, float *workspace, size_t workspace_floats, int M, int N, int K,
                        dfx_stream_t stream) {
  DFX_REQUIRE(A && B && Cout, "debug_gemm_bf16: null argument");
  hipStream_t st = dfx::as_stream(stream);
  dfx::gemm::GemmArgs g{};
  g.A = static_cast<const float *>(A), g.lda = lda, g.B = static_cast<const float *>(B), g.ldb = ldb, g.M = M, g.N = N, g.K = K;
  g.a_bf16 = a_bf16, g.b_bf16 = b_bf16;
  if (!tn) {
    g.bias = bias, g.R = resid, g.ldr = N, g.C = Cout, g.ldc = N;
    DFX_REQUIRE(dfx::gemm::nt_ok(g), "debug_gemm_bf16: shape not supported by the NT kernel");
    dfx::gemm::launch_nt(st, g);
    return dfx::check_launch("debug_gemm_bf16 nt");
  }
  const int slab = K <= 8192 ? 64 : 2048, ns = (K + slab - 1) / slab;
  DFX_REQUIRE(workspace && workspace_floats >= (size_t)ns * ((size_t)M * N + M), "debug_gemm_bf16: workspace too small");
  g.C = workspace, g.ldc = N, g.bpart = db ? workspace + (size_t)ns * M * N : nullptr, g.rows_per_slab = slab;
  DFX_REQUIRE(dfx::gemm::tn_ok(g), "debug_gemm_bf16: shape not supported by the TN kernel");
  dfx::gemm::launch_tn(st, g, ns);
  k_wgrad_finish<<<(M * N + 31) / 32, 256, 0, st>>>(workspace, Cout, ns, M, N, N);
  if (db) k_sum_parts<<<(M + 31) / 32, 1024, 0, st>>>(g.bpart, db, ns, M, M);
  return dfx::check_launch("debug_gemm_bf16 tn");
}

// Global gradient norm^2 accumulated over tensors (clip_grad_norm_): *sumsq += sum g^2.  workspace: 1024 doubles.
int dfx_grad_sumsq_accumulate(const float *g, long long n, double *workspace1024, double *sumsq, dfx_stream_t stream) {
  DFX_REQUIRE(g && workspace1024 && sumsq && n >= 1, "grad_sumsq: bad argument");
  long long nb = (n + 255) / 256;
  if (nb > 1024) nb = 1024;
  k_sumsq<<<(int)nb, 256, 0, dfx::as_stream(stream)>>>(g, n, workspace1024);
  k_sumsq_finish<<<1, 64, 0, dfx::as_stream(stream)>>>(workspace1024, (int)nb, sumsq);
  return dfx::check_launch("grad_sumsq");
}

// One Adam step on one tensor (torch.optim.Adam, amsgrad off), gradient clipped by min(1, max_norm / (sqrt(*sumsq) + 1e-6))
// when max_norm > 0 (torch.nn.utils.clip_grad_norm_); step = 1-based step count for the bias corrections.
int dfx_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, const double *sumsq,
                      float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      dfx_stream_t stream) {
  DFX_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 1 && step >= 1, "adam_step: bad argument");
  DFX_REQUIRE(max_norm <= 0.f || sumsq, "adam_step: clipping needs the gradient norm");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  k_adam<<<(int)((n + 255) / 256), 256, 0, dfx::as_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, n, sumsq, max_norm, lr, beta1,
                                                                    beta2, eps, weight_decay, bc1, bc2);
  return dfx::check_launch("adam_step");
}

}  // extern "C"
