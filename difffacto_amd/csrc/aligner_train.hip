// libdfx: training forward / backward of the part aligner (stage 2: configs/train_*_stage2.py, gen_*.py with train_aligner).
//
// Reference: PartAlignerTransformer.forward / _forward_attn (python/difffacto/models/encoders/part_encoders.py:88-143) with the shipped options
// (use_linear, single_attn, class_cond + add_class_cond, cimle with cond_noise_type 0, mask_out_unreferenced_code, dropout 0), its blocks
// BasicTransformerBlock._forward (python/difffacto/models/diffusions/nets/attention.py:296-306) = self-attention over the n_class part tokens
// (CrossAttention :179-204 with context = x, keys masked by valid_id) + GEGLU feed-forward (:50-57, 77-94), and torch autograd through them.
//
// One shape = n_class = 4 tokens; a training batch is B x K shapes = a few hundred tokens of width 256 (8 heads x 32), 5 blocks: ~0.2 GFLOP per
// direction — nothing for the matrix pipe to win, and the optimiser step of stage 2 only touches these 6.6 M parameters.  So this file is written for
// exactness (fp32 everywhere, erf GELU, fixed summation order: no atomics) and plainness: ONE tiled fp32 product kernel with general strides serves
// every Linear in all three orientations (Y = X W^T, dX = dY W, dW = dY^T X), LayerNorm / attention / GEGLU are small per-row kernels.
// The inference form of the same network (dfx_part_aligner, latents_kernels.hip) stays the MFMA path with fused epilogues.
#include <cstring>
#include <type_traits>

#include "dfx_common.h"

namespace {

constexpr int TS = 16;   // product tile
constexpr float LN_EPS = 1e-5f;

// C[m][n] (+)= sum_k A(m, k) B(k, n) + bias[n] + R[m][n],  A(m, k) = A[m sam + k sak],  B(k, n) = B[k sbk + n sbn]
__global__ __launch_bounds__(TS *TS) void k_mm(const float *__restrict__ A, long long sam, long long sak, const float *__restrict__ B, long long sbk,
                                                long long sbn, const float *__restrict__ bias, const float *R, int ldr, float *C, int ldc, int M, int N,
                                                int K) {   // (R may be C: every thread reads its own element before it writes it)
  __shared__ float sa[TS][TS + 1], sb[TS][TS + 1];
  const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
  const int m = blockIdx.y * TS + ty, n = blockIdx.x * TS + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += TS) {
    const int ka = k0 + tx, kb = k0 + ty;
    sa[ty][tx] = (m < M && ka < K) ? A[m * sam + ka * sak] : 0.f;
    sb[ty][tx] = (kb < K && n < N) ? B[kb * sbk + n * sbn] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TS; ++k) acc = fmaf(sa[ty][k], sb[k][tx], acc);
    __syncthreads();
  }
  if (m < M && n < N) {
    if (bias) acc += bias[n];
    if (R) acc += R[(size_t)m * ldr + n];
    C[(size_t)m * ldc + n] = acc;
  }
}
inline void mm(hipStream_t st, const float *A, long long sam, long long sak, const float *B, long long sbk, long long sbn, const float *bias,
               const float *R, int ldr, float *C, int ldc, int M, int N, int K) {
  k_mm<<<dim3((N + TS - 1) / TS, (M + TS - 1) / TS), TS * TS, 0, st>>>(A, sam, sak, B, sbk, sbn, bias, R, ldr, C, ldc, M, N, K);
}
// Y (M, N) = X (M, K) W^T (+ b) (+ R): nn.Linear
inline void linear(hipStream_t st, const float *X, int ldx, const float *W, const float *b, const float *R, float *Y, int ldy, int M, int N, int K) {
  mm(st, X, ldx, 1, W, 1, K, b, R, ldy, Y, ldy, M, N, K);
}
// dX (M, K) = dY (M, N) W (N, K) (+ R)
inline void linear_dx(hipStream_t st, const float *dY, int ldy, const float *W, const float *R, float *dX, int ldx, int M, int N, int K) {
  mm(st, dY, ldy, 1, W, K, 1, nullptr, R, ldx, dX, ldx, M, K, N);
}
// dW (N, K) = dY^T (N, M) X (M, K)
inline void linear_dw(hipStream_t st, const float *dY, int ldy, const float *X, int ldx, float *dW, int M, int N, int K) {
  mm(st, dY, 1, ldy, X, ldx, 1, nullptr, nullptr, 0, dW, K, N, K, M);
}
// out[n] = sum over the rows m = m0, m0 + step, .. < M of X[m][n]  (bias gradients; class_emb: step = J)
__global__ void k_colsum(const float *__restrict__ X, int ldx, float *__restrict__ out, int M, int N, int m0, int step) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int m = m0; m < M; m += step) s += X[(size_t)m * ldx + n];
  out[n] = s;
}

// tokens: X0[(r, j)][c] = c < Z ? code[r][c][j] : noise[r][c - Z] * noise_scale  (part_encoders.py:96-103 + rearrange 'b c n -> b n c')
__global__ void k_tokens(const float *__restrict__ code, const float *__restrict__ noise, float *__restrict__ X, int R, int Z, int J, int ND, float scale) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C = Z + ND;
  if (t >= (long long)R * J * C) return;
  const int c = t % C, j = (t / C) % J, r = t / ((long long)C * J);
  X[t] = c < Z ? code[((size_t)r * Z + c) * J + j] : noise[(size_t)r * ND + (c - Z)] * scale;
}
// x[(r, j)][c] += class_emb[j][c]  (part_encoders.py:116-118)
__global__ void k_add_class_emb(float *__restrict__ X, const float *__restrict__ emb, int M, int J, int C) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)M * C) return;
  const int c = t % C, j = (t / C) % J;
  X[t] += emb[(size_t)j * C + c];
}
// its transpose for the part codes: d code[r][c][j] = dX0[(r, j)][c]
__global__ void k_tokens_bwd(const float *__restrict__ dX, float *__restrict__ dcode, int R, int Z, int J, int IC) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)R * Z * J) return;
  const int j = t % J, c = (t / J) % Z, r = t / ((long long)J * Z);
  dcode[t] = dX[((size_t)r * J + j) * IC + c];
}

// LayerNorm over the last dimension (C a multiple of 64, <= 1024): one wavefront per row; stats = (mean, rstd) per row; two passes like torch
__global__ __launch_bounds__(256) void k_ln_fwd(const float *__restrict__ X, const float *__restrict__ g, const float *__restrict__ b, float *__restrict__ Y,
                                                 float *__restrict__ stats, int M, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float *x = X + (size_t)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[c];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mu = s / C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) q = fmaf(x[c] - mu, x[c] - mu, q);
  for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / C + LN_EPS);
  for (int c = lane; c < C; c += 64) Y[(size_t)row * C + c] = fmaf((x[c] - mu) * rstd, g[c], b[c]);
  if (lane == 0) stats[2 * row] = mu, stats[2 * row + 1] = rstd;
}
// dX = (R ? R : 0) + rstd (dy g - mean(dy g) - xhat mean(dy g xhat)); XH (M, C) receives xhat (for the gamma gradient), may alias nothing
__global__ __launch_bounds__(256) void k_ln_bwd(const float *__restrict__ dY, const float *__restrict__ X, const float *__restrict__ stats,
                                                 const float *__restrict__ g, const float *__restrict__ R, float *__restrict__ dX,
                                                 float *__restrict__ XH, int M, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float mu = stats[2 * row], rstd = stats[2 * row + 1];
  const float *x = X + (size_t)row * C, *dy = dY + (size_t)row * C;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float dg = dy[c] * g[c], xh = (x[c] - mu) * rstd;
    s1 += dg;
    s2 = fmaf(dg, xh, s2);
  }
  for (int o = 32; o; o >>= 1) s1 += __shfl_xor(s1, o, 64), s2 += __shfl_xor(s2, o, 64);
  s1 /= C, s2 /= C;
  for (int c = lane; c < C; c += 64) {
    const float xh = (x[c] - mu) * rstd;
    const float d = rstd * (dy[c] * g[c] - s1 - xh * s2);
    dX[(size_t)row * C + c] = R ? R[(size_t)row * C + c] + d : d;
    XH[(size_t)row * C + c] = xh;
  }
}
// d gamma[c] = sum_m dy xhat, d beta[c] = sum_m dy
__global__ void k_ln_param(const float *__restrict__ dY, const float *__restrict__ XH, float *__restrict__ dg, float *__restrict__ db, int M, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int m = 0; m < M; ++m) {
    const float d = dY[(size_t)m * C + c];
    a = fmaf(d, XH[(size_t)m * C + c], a);
    b += d;
  }
  dg[c] = a, db[c] = b;
}

// self-attention over the J <= 8 tokens of a shape (CrossAttention.forward, attention.py:179-204, context = x): one thread per (shape, head, query token).
// QKV (M, 3 C) = [q | k | v]; P (M, H, J) saved; O (M, C).  Keys of absent parts get -finfo.max before the softmax (:192-197).
template <int DH>
__global__ void k_attn_fwd(const float *__restrict__ QKV, const float *__restrict__ valid, float *__restrict__ P, float *__restrict__ O, int R, int J, int H, float scale) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * H * J) return;
  const int j = t % J, h = (t / J) % H, r = t / (J * H), C = H * DH;
  const float *q = QKV + ((size_t)r * J + j) * 3 * C + h * DH;
  float sim[8], mx = -3.402823466e38f;
  for (int jj = 0; jj < J; ++jj) {
    const float *k = QKV + ((size_t)r * J + jj) * 3 * C + C + h * DH;
    float s = 0.f;
    for (int d = 0; d < DH; ++d) s = fmaf(q[d], k[d], s);
    s *= scale;
    sim[jj] = valid[(size_t)r * J + jj] != 0.f ? s : -3.402823466e38f;
    mx = fmaxf(mx, sim[jj]);
  }
  float den = 0.f;
  for (int jj = 0; jj < J; ++jj) sim[jj] = expf(sim[jj] - mx), den += sim[jj];
  float *p = P + (((size_t)r * J + j) * H + h) * J;
  for (int jj = 0; jj < J; ++jj) sim[jj] /= den, p[jj] = sim[jj];
  float *o = O + ((size_t)r * J + j) * C + h * DH;
  for (int d = 0; d < DH; ++d) {
    float a = 0.f;
    for (int jj = 0; jj < J; ++jj) a = fmaf(sim[jj], QKV[((size_t)r * J + jj) * 3 * C + 2 * C + h * DH + d], a);
    o[d] = a;
  }
}
// backward: one thread per (shape, head) walks the J x J pairs in fixed order -> dQKV (no atomics)
template <int DH>
__global__ void k_attn_bwd(const float *__restrict__ QKV, const float *__restrict__ P, const float *__restrict__ dO, float *__restrict__ dQKV, int R, int J, int H,
                           float scale) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * H) return;
  const int h = t % H, r = t / H, C = H * DH;
  for (int j = 0; j < J; ++j)
    for (int d = 0; d < DH; ++d) {
      dQKV[((size_t)r * J + j) * 3 * C + h * DH + d] = 0.f;
      dQKV[((size_t)r * J + j) * 3 * C + C + h * DH + d] = 0.f;
      dQKV[((size_t)r * J + j) * 3 * C + 2 * C + h * DH + d] = 0.f;
    }
  for (int j = 0; j < J; ++j) {
    const float *p = P + (((size_t)r * J + j) * H + h) * J;
    const float *dout = dO + ((size_t)r * J + j) * C + h * DH;
    const float *q = QKV + ((size_t)r * J + j) * 3 * C + h * DH;
    float dP[8], dot = 0.f;
    for (int jj = 0; jj < J; ++jj) {
      const float *v = QKV + ((size_t)r * J + jj) * 3 * C + 2 * C + h * DH;
      float s = 0.f;
      for (int d = 0; d < DH; ++d) s = fmaf(dout[d], v[d], s);
      dP[jj] = s;
      dot = fmaf(p[jj], s, dot);
    }
    for (int jj = 0; jj < J; ++jj) {
      const float ds = p[jj] * (dP[jj] - dot) * scale;   // d sim (masked keys: p = 0 -> 0)
      const float *k = QKV + ((size_t)r * J + jj) * 3 * C + C + h * DH;
      float *dq = dQKV + ((size_t)r * J + j) * 3 * C + h * DH;
      float *dk = dQKV + ((size_t)r * J + jj) * 3 * C + C + h * DH;
      float *dv = dQKV + ((size_t)r * J + jj) * 3 * C + 2 * C + h * DH;
      for (int d = 0; d < DH; ++d) {
        dq[d] = fmaf(ds, k[d], dq[d]);
        dk[d] = fmaf(ds, q[d], dk[d]);
        dv[d] = fmaf(p[jj], dout[d], dv[d]);
      }
    }
  }
}

// GEGLU (attention.py:55-57): AG (M, 2 H) = [a | g]; hid = a gelu(g), erf form (F.gelu default)
__global__ void k_geglu_fwd(const float *__restrict__ AG, float *__restrict__ HID, long long M, int H) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * H) return;
  const long long m = t / H;
  const int c = (int)(t % H);
  const float a = AG[m * 2 * H + c], g = AG[m * 2 * H + H + c];
  HID[t] = a * 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
}
__global__ void k_geglu_bwd(const float *__restrict__ AG, const float *__restrict__ dHID, float *__restrict__ dAG, long long M, int H) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * H) return;
  const long long m = t / H;
  const int c = (int)(t % H);
  const float a = AG[m * 2 * H + c], g = AG[m * 2 * H + H + c], d = dHID[t];
  const float Phi = 0.5f * (1.0f + erff(g * 0.70710678118654752440f)), phi = 0.39894228040143267794f * expf(-0.5f * g * g);
  dAG[m * 2 * H + c] = d * g * Phi;
  dAG[m * 2 * H + H + c] = d * a * (Phi + g * phi);
}

// OUT (M, 6) <-> mean (R, 3, J), logvar (R, 3, J)   ('b n c -> b c n' + split, part_encoders.py:107-108,139)
__global__ void k_split(const float *__restrict__ OUT, float *__restrict__ mean, float *__restrict__ logvar, int R, int J) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * 6 * J) return;
  const int j = t % J, c = (t / J) % 6, r = t / (6 * J);
  const float v = OUT[((size_t)r * J + j) * 6 + c];
  if (c < 3) mean[((size_t)r * 3 + c) * J + j] = v;
  else logvar[((size_t)r * 3 + (c - 3)) * J + j] = v;
}
__global__ void k_split_bwd(const float *__restrict__ dmean, const float *__restrict__ dlogvar, float *__restrict__ dOUT, int R, int J) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * 6 * J) return;
  const int j = t % J, c = (t / J) % 6, r = t / (6 * J);
  dOUT[((size_t)r * J + j) * 6 + c] = c < 3 ? (dmean ? dmean[((size_t)r * 3 + c) * J + j] : 0.f) : (dlogvar ? dlogvar[((size_t)r * 3 + (c - 3)) * J + j] : 0.f);
}

inline int nblk(long long n, int bs = 256) { return (int)((n + bs - 1) / bs); }

struct Ws {
  // saved by the forward
  float *X0, *H[DFX_MAX_DEPTH + 1], *Xn2[DFX_MAX_DEPTH], *st2[DFX_MAX_DEPTH], *QKV[DFX_MAX_DEPTH], *P[DFX_MAX_DEPTH], *O[DFX_MAX_DEPTH], *H1[DFX_MAX_DEPTH],
      *Xn3[DFX_MAX_DEPTH], *st3[DFX_MAX_DEPTH], *AG[DFX_MAX_DEPTH], *HID[DFX_MAX_DEPTH], *Xnf, *stf, *OUT, *Wqkv;
  // backward scratch
  float *dH, *dH1, *dXn, *XH, *dQKV, *dO, *dAG, *dHID, *dOUT, *dX0;
};
size_t carve(Ws &w, char *base, int R, int J, int C, int IC, int H, int depth) {
  size_t off = 0;
  auto take = [&](size_t n) {
    off = (off + 255) & ~size_t(255);
    float *p = base ? reinterpret_cast<float *>(base + off) : nullptr;
    off += n * sizeof(float);
    return p;
  };
  const size_t M = (size_t)R * J;
  w.X0 = take(M * IC);
  for (int i = 0; i <= depth; ++i) w.H[i] = take(M * C);
  for (int i = 0; i < depth; ++i) {
    w.Xn2[i] = take(M * C), w.st2[i] = take(M * 2), w.QKV[i] = take(M * 3 * C), w.P[i] = take(M * H * J), w.O[i] = take(M * C), w.H1[i] = take(M * C);
    w.Xn3[i] = take(M * C), w.st3[i] = take(M * 2), w.AG[i] = take(M * 8 * C), w.HID[i] = take(M * 4 * C);
  }
  w.Xnf = take(M * C), w.stf = take(M * 2), w.OUT = take(M * 6), w.Wqkv = take((size_t)3 * C * C);
  w.dH = take(M * C), w.dH1 = take(M * C), w.dXn = take(M * C), w.XH = take(M * C), w.dQKV = take(M * 3 * C), w.dO = take(M * C);
  w.dAG = take(M * 8 * C), w.dHID = take(M * 4 * C), w.dOUT = take(M * 6), w.dX0 = take(M * IC);
  return off;
}
int check(const dfx_latent_weights *w, const char *who) {
  DFX_REQUIRE(w, "%s: null weights", who);
  DFX_REQUIRE(w->n_class >= 1 && w->n_class <= 8 && w->depth >= 1 && w->depth <= DFX_MAX_DEPTH, "%s: n_class %d / depth %d", who, w->n_class, w->depth);
  DFX_REQUIRE(w->d_head == 16 || w->d_head == 32 || w->d_head == 64, "%s: d_head %d not in {16, 32, 64}", who, w->d_head);
  DFX_REQUIRE((w->n_heads * w->d_head) % 64 == 0 && w->n_heads * w->d_head <= 1024, "%s: inner dim %d", who, w->n_heads * w->d_head);
  DFX_REQUIRE(w->cimle && w->noise_dim > 0, "%s: the training path is the cIMLE configuration (noise concatenated per token, no pre_norm)", who);
  return DFX_OK;
}
template <class F>
void attn_dispatch(int dh, F &&f) {
  if (dh == 16) f(std::integral_constant<int, 16>{});
  else if (dh == 32) f(std::integral_constant<int, 32>{});
  else f(std::integral_constant<int, 64>{});
}

}  // namespace

extern "C" {

size_t dfx_aligner_train_workspace_bytes(int B, int n_class, int zdim, int noise_dim, int n_heads, int d_head, int depth) {
  if (B <= 0 || n_class <= 0 || depth <= 0 || depth > DFX_MAX_DEPTH) return 0;
  Ws w;
  return carve(w, nullptr, B, n_class, n_heads * d_head, zdim + noise_dim, n_heads, depth);
}

int dfx_aligner_train_forward(const dfx_latent_weights *w, void *workspace, size_t workspace_bytes, const float *part_code, const float *valid,
                              const float *noise, float *mean, float *logvar, int B, dfx_stream_t stream) {
  int rc = check(w, "aligner_train_forward");
  if (rc) return rc;
  DFX_REQUIRE(B > 0 && part_code && valid && noise && mean && logvar && workspace, "aligner_train_forward: null argument");
  const int J = w->n_class, C = w->n_heads * w->d_head, IC = w->zdim + w->noise_dim, H = w->n_heads, M = B * J;
  Ws s;
  DFX_REQUIRE(carve(s, static_cast<char *>(workspace), B, J, C, IC, H, w->depth) <= workspace_bytes, "aligner_train_forward: workspace too small");
  hipStream_t st = dfx::as_stream(stream);
  k_tokens<<<nblk((long long)M * IC), 256, 0, st>>>(part_code, noise, s.X0, B, w->zdim, J, w->noise_dim, w->noise_scale);
  // x = proj_in(x) + class_emb[token]  (part_encoders.py:113-118; with cimle / cond_noise_type 0 no pre_norm, :119-131)
  linear(st, s.X0, IC, w->proj_in_w, w->proj_in_b, nullptr, s.H[0], C, M, C, IC);
  k_add_class_emb<<<nblk((long long)M * C), 256, 0, st>>>(s.H[0], w->class_emb, M, J, C);
  const float scale = 1.0f / sqrtf((float)w->d_head);
  for (int i = 0; i < w->depth; ++i) {
    const dfx_aligner_block_weights &k = w->blocks[i];
    k_ln_fwd<<<nblk(M, 4), 256, 0, st>>>(s.H[i], k.norm2_w, k.norm2_b, s.Xn2[i], s.st2[i], M, C);
    linear(st, s.Xn2[i], C, k.to_q, nullptr, nullptr, s.QKV[i], 3 * C, M, C, C);
    linear(st, s.Xn2[i], C, k.to_k, nullptr, nullptr, s.QKV[i] + C, 3 * C, M, C, C);
    linear(st, s.Xn2[i], C, k.to_v, nullptr, nullptr, s.QKV[i] + 2 * C, 3 * C, M, C, C);
    attn_dispatch(w->d_head, [&](auto dh) { k_attn_fwd<decltype(dh)::value><<<nblk((long long)B * H * J), 256, 0, st>>>(s.QKV[i], valid, s.P[i], s.O[i], B, J, H, scale); });
    linear(st, s.O[i], C, k.to_out_w, k.to_out_b, s.H[i], s.H1[i], C, M, C, C);                       // attn2(norm2(x)) + x   (attention.py:300)
    k_ln_fwd<<<nblk(M, 4), 256, 0, st>>>(s.H1[i], k.norm3_w, k.norm3_b, s.Xn3[i], s.st3[i], M, C);
    linear(st, s.Xn3[i], C, k.ff_proj_w, k.ff_proj_b, nullptr, s.AG[i], 8 * C, M, 8 * C, C);
    k_geglu_fwd<<<nblk((long long)M * 4 * C), 256, 0, st>>>(s.AG[i], s.HID[i], M, 4 * C);
    linear(st, s.HID[i], 4 * C, k.ff_out_w, k.ff_out_b, s.H1[i], s.H[i + 1], C, M, C, 4 * C);           // ff(norm3(x)) + x      (:305)
  }
  k_ln_fwd<<<nblk(M, 4), 256, 0, st>>>(s.H[w->depth], w->post_norm_w, w->post_norm_b, s.Xnf, s.stf, M, C);
  linear(st, s.Xnf, C, w->proj_out_w, w->proj_out_b, nullptr, s.OUT, 6, M, 6, C);
  k_split<<<nblk((long long)M * 6), 256, 0, st>>>(s.OUT, mean, logvar, B, J);
  return dfx::check_launch("aligner_train_forward");
}

/* grads: the same struct, pointing at the gradient buffers (overwritten; pre_norm_* are not touched: unused by this configuration).
 * d_part_code (B, zdim, n_class) or NULL. */
int dfx_aligner_train_backward(const dfx_latent_weights *w, void *workspace, size_t workspace_bytes, const float *valid, const float *d_mean,
                               const float *d_logvar, const dfx_latent_weights *grads, float *d_part_code, int B, dfx_stream_t stream) {
  int rc = check(w, "aligner_train_backward");
  if (rc) return rc;
  DFX_REQUIRE(B > 0 && valid && (d_mean || d_logvar) && grads && workspace, "aligner_train_backward: null argument");
  const int J = w->n_class, C = w->n_heads * w->d_head, IC = w->zdim + w->noise_dim, H = w->n_heads, M = B * J;
  Ws s;
  DFX_REQUIRE(carve(s, static_cast<char *>(workspace), B, J, C, IC, H, w->depth) <= workspace_bytes, "aligner_train_backward: workspace too small");
  hipStream_t st = dfx::as_stream(stream);
  auto mut = [](const float *p) { return const_cast<float *>(p); };
  const float scale = 1.0f / sqrtf((float)w->d_head);
  k_split_bwd<<<nblk((long long)M * 6), 256, 0, st>>>(d_mean, d_logvar, s.dOUT, B, J);
  linear_dw(st, s.dOUT, 6, s.Xnf, C, mut(grads->proj_out_w), M, 6, C);
  k_colsum<<<1, 64, 0, st>>>(s.dOUT, 6, mut(grads->proj_out_b), M, 6, 0, 1);
  linear_dx(st, s.dOUT, 6, w->proj_out_w, nullptr, s.dXn, C, M, 6, C);
  k_ln_bwd<<<nblk(M, 4), 256, 0, st>>>(s.dXn, s.H[w->depth], s.stf, w->post_norm_w, nullptr, s.dH, s.XH, M, C);
  k_ln_param<<<nblk(C), 256, 0, st>>>(s.dXn, s.XH, mut(grads->post_norm_w), mut(grads->post_norm_b), M, C);
  for (int i = w->depth - 1; i >= 0; --i) {
    const dfx_aligner_block_weights &k = w->blocks[i], &g = grads->blocks[i];
    // h_out = h1 + ff_out(hid) : dH is the gradient at h_out
    linear_dw(st, s.dH, C, s.HID[i], 4 * C, mut(g.ff_out_w), M, C, 4 * C);
    k_colsum<<<nblk(C), 256, 0, st>>>(s.dH, C, mut(g.ff_out_b), M, C, 0, 1);
    linear_dx(st, s.dH, C, k.ff_out_w, nullptr, s.dHID, 4 * C, M, C, 4 * C);
    k_geglu_bwd<<<nblk((long long)M * 4 * C), 256, 0, st>>>(s.AG[i], s.dHID, s.dAG, M, 4 * C);
    linear_dw(st, s.dAG, 8 * C, s.Xn3[i], C, mut(g.ff_proj_w), M, 8 * C, C);
    k_colsum<<<nblk(8 * C), 256, 0, st>>>(s.dAG, 8 * C, mut(g.ff_proj_b), M, 8 * C, 0, 1);
    linear_dx(st, s.dAG, 8 * C, k.ff_proj_w, nullptr, s.dXn, C, M, 8 * C, C);
    k_ln_bwd<<<nblk(M, 4), 256, 0, st>>>(s.dXn, s.H1[i], s.st3[i], k.norm3_w, s.dH, s.dH1, s.XH, M, C);   // dH1 = dH + LN3'(dxn3)
    k_ln_param<<<nblk(C), 256, 0, st>>>(s.dXn, s.XH, mut(g.norm3_w), mut(g.norm3_b), M, C);
    // h1 = h + to_out(att)
    linear_dw(st, s.dH1, C, s.O[i], C, mut(g.to_out_w), M, C, C);
    k_colsum<<<nblk(C), 256, 0, st>>>(s.dH1, C, mut(g.to_out_b), M, C, 0, 1);
    linear_dx(st, s.dH1, C, k.to_out_w, nullptr, s.dO, C, M, C, C);
    attn_dispatch(w->d_head, [&](auto dh) { k_attn_bwd<decltype(dh)::value><<<nblk((long long)B * H, 64), 64, 0, st>>>(s.QKV[i], s.P[i], s.dO, s.dQKV, B, J, H, scale); });
    linear_dw(st, s.dQKV, 3 * C, s.Xn2[i], C, mut(g.to_q), M, C, C);
    linear_dw(st, s.dQKV + C, 3 * C, s.Xn2[i], C, mut(g.to_k), M, C, C);
    linear_dw(st, s.dQKV + 2 * C, 3 * C, s.Xn2[i], C, mut(g.to_v), M, C, C);
    linear_dx(st, s.dQKV, 3 * C, k.to_q, nullptr, s.dXn, C, M, C, C);
    linear_dx(st, s.dQKV + C, 3 * C, k.to_k, s.dXn, s.dXn, C, M, C, C);
    linear_dx(st, s.dQKV + 2 * C, 3 * C, k.to_v, s.dXn, s.dXn, C, M, C, C);
    k_ln_bwd<<<nblk(M, 4), 256, 0, st>>>(s.dXn, s.H[i], s.st2[i], k.norm2_w, s.dH1, s.dH, s.XH, M, C);     // dH(in) = dH1 + LN2'(dxn2)
    k_ln_param<<<nblk(C), 256, 0, st>>>(s.dXn, s.XH, mut(g.norm2_w), mut(g.norm2_b), M, C);
  }
  // h0 = proj_in(x0) + class_emb[token]
  for (int j = 0; j < J; ++j) k_colsum<<<nblk(C), 256, 0, st>>>(s.dH, C, mut(grads->class_emb) + (size_t)j * C, M, C, j, J);
  linear_dw(st, s.dH, C, s.X0, IC, mut(grads->proj_in_w), M, C, IC);
  k_colsum<<<nblk(C), 256, 0, st>>>(s.dH, C, mut(grads->proj_in_b), M, C, 0, 1);
  if (d_part_code) {
    linear_dx(st, s.dH, C, w->proj_in_w, nullptr, s.dX0, IC, M, C, IC);
    k_tokens_bwd<<<nblk((long long)B * w->zdim * J), 256, 0, st>>>(s.dX0, d_part_code, B, w->zdim, J, IC);
  }
  return dfx::check_launch("aligner_train_backward");
}

}  // extern "C"
