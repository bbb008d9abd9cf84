// Latent sampler: per-part normalising flows in reverse + part-aligner transformer +
// the glue of PartEncoder.sample_latents, as small fp32 HIP kernels for gfx950.
//
//   reference                                                        here
//   flow.py:21-47  CouplingLayer.forward(reverse=True)               k_lin<RELU> x2 + k_lin<COUPLING>
//   flow.py:58-72  SequentialFlow (layers last -> first)             host loop in run_flows()
//   part_encoders.py:88-143 PartAlignerTransformer                   k_tokens, k_lin<*>, k_ln, k_attn
//   attention.py:179-204 CrossAttention (self-attention, key mask)   k_attn
//   attention.py:50-57,77-94 GEGLU feed-forward                      k_lin<GEGLU> + k_lin<RESID>
//   part_encoders.py:1072-1108 fixed_id mixing, K repeat, seg ids    k_mix_repeat, k_finish
//
// This runs once per batch next to a 1000-step chain (< 0.3 % of the wall time), so the design goal is
// exactness and few, simple kernels: every GEMM is one generic wave-per-32x32-tile kernel on the exact
// fp32 matrix pipe (v_mfma_f32_32x32x2_f32), evaluated transposed like the denoiser (output channels on
// the MFMA M axis = accumulator registers, rows/tokens on the N axis = lanes) so that the epilogues
// (bias, ReLU, residual, coupling update, GEGLU gate) are in-lane.  The K axis is consumed 8 at a time
// with one 16-byte load per operand and lane: lane (j, hf) holds k = k0 + 4 hf + s for the s-th MFMA.
#include <vector>

#include "dfx_common.h"
#include "mfma_linear.h"

namespace {

using namespace dfx::lin;

// LayerNorm over C channels (C % 64 == 0, C <= 1024), one wavefront per row, two-pass in registers.
__global__ __launch_bounds__(256) void k_ln(const float *__restrict__ X, const float *__restrict__ w,
                                            const float *__restrict__ b, float *__restrict__ Y, int M, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const int per = C >> 6;
  float v[16];
  float s = 0.f;
  for (int i = 0; i < per; ++i) v[i] = X[(size_t)row * C + lane + 64 * i], s += v[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / (float)C;
  float q = 0.f;
  for (int i = 0; i < per; ++i) v[i] -= mu, q += v[i] * v[i];
  for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.f / sqrtf(q / (float)C + 1e-5f);
  for (int i = 0; i < per; ++i) {
    const int c = lane + 64 * i;
    Y[(size_t)row * C + c] = v[i] * rstd * w[c] + b[c];
  }
}

// Self-attention over the J part tokens of one shape (attention.py:179-204 with context = x): one thread per
// (token, head).  QKV rows are [q | k | v] (3*inner).  Masked keys get -FLT_MAX before the softmax (:192-197).
template <int DH>
__global__ void k_attn(const float *__restrict__ QKV, const float *__restrict__ valid, float *__restrict__ O, int M,
                       int J, int heads, float scale) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * heads) return;
  const int m = t / heads, h = t % heads, bsh = m / J, inner = heads * DH;
  const float *q = QKV + (size_t)m * 3 * inner + h * DH;
  float qv[DH];
  for (int d = 0; d < DH; ++d) qv[d] = q[d];
  float sim[8], mx = -3.402823466e38f;
  for (int jj = 0; jj < J; ++jj) {
    const float *k = QKV + (size_t)(bsh * J + jj) * 3 * inner + inner + h * DH;
    float acc = 0.f;
    for (int d = 0; d < DH; ++d) acc += qv[d] * k[d];
    acc *= scale;
    if (valid && valid[bsh * J + jj] == 0.f) acc = -3.402823466e38f;
    sim[jj] = acc;
    mx = fmaxf(mx, acc);
  }
  float den = 0.f;
  for (int jj = 0; jj < J; ++jj) sim[jj] = expf(sim[jj] - mx), den += sim[jj];
  float out[DH];
  for (int d = 0; d < DH; ++d) out[d] = 0.f;
  for (int jj = 0; jj < J; ++jj) {
    const float p = sim[jj] / den;
    const float *v = QKV + (size_t)(bsh * J + jj) * 3 * inner + 2 * inner + h * DH;
    for (int d = 0; d < DH; ++d) out[d] += p * v[d];
  }
  float *o = O + (size_t)m * inner + h * DH;
  for (int d = 0; d < DH; ++d) o[d] = out[d];
}

// (S, Z, J) -> part-major (J, S, Z) working layout of the flows, scaled by sqrt(prior_var); and back.
__global__ void k_flow_in(const float *__restrict__ w, float *__restrict__ X, int S, int Z, int J, float scale) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)S * Z * J) return;
  const int j = t % J, c = (t / J) % Z, s = t / ((long long)J * Z);
  X[((size_t)j * S + s) * Z + c] = w[t] * scale;
}
__global__ void k_flow_out(const float *__restrict__ X, float *__restrict__ code, int S, int Z, int J) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)S * Z * J) return;
  const int j = t % J, c = (t / J) % Z, s = t / ((long long)J * Z);
  code[t] = X[((size_t)j * S + s) * Z + c];
}

// Token matrix of the aligner: row (b*J + j) = [part_code[b, :, j] | noise[b] * noise_scale]
// (part_encoders.py:91-103 with add_class_cond: no one-hot channels) then 'b c n -> b n c' (:113).
__global__ void k_tokens(const float *__restrict__ code, const float *__restrict__ noise, float *__restrict__ X,
                         int B, int Z, int J, int ND, float noise_scale) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C = Z + ND;
  if (t >= (long long)B * J * C) return;
  const int c = t % C, j = (t / C) % J, b = t / ((long long)C * J);
  X[t] = c < Z ? code[((size_t)b * Z + c) * J + j] : noise[(size_t)b * ND + (c - Z)] * noise_scale;
}

// h (B*J, 6) -> 'b n c -> b c n' and split (part_encoders.py:107-108,139): mean (B,3,J), logvar (B,3,J);
// params (B,6,J) = [mean | exp(logvar + lsv)] (part_encoders.py:1321).
__global__ void k_split(const float *__restrict__ H, float *__restrict__ mean, float *__restrict__ logvar,
                        float *__restrict__ params, int B, int J, float lsv) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * 6 * J) return;
  const int j = t % J, c = (t / J) % 6, b = t / (6 * J);
  const float v = H[((size_t)b * J + j) * 6 + c];
  if (c < 3) {
    if (mean) mean[((size_t)b * 3 + c) * J + j] = v;
  } else if (logvar) {
    logvar[((size_t)b * 3 + (c - 3)) * J + j] = v;
  }
  if (params) params[t] = c < 3 ? v : expf(v + lsv);
}

// part_encoders.py:1070-1086: fixed parts take shape 0's code / validity, every shape is repeated K times;
// with any fixed part the K aligner noises of shape 0 are shared by all shapes.
__global__ void k_mix_repeat(const float *__restrict__ code, const float *__restrict__ valid,
                             const float *__restrict__ noise, float *__restrict__ code_o, float *__restrict__ valid_o,
                             float *__restrict__ noise_o, int S, int K, int Z, int J, int ND, unsigned fixed_mask) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nc = (long long)S * K * Z * J, nv = (long long)S * K * J, nn = (long long)S * K * ND;
  if (t < nc) {
    const int j = t % J, c = (t / J) % Z;
    const int r = t / ((long long)J * Z), s = (fixed_mask >> j) & 1 ? 0 : r / K;
    code_o[t] = code[((size_t)s * Z + c) * J + j];
  } else if (t < nc + nv) {
    const long long u = t - nc;
    const int j = u % J, r = u / J;
    valid_o[u] = (fixed_mask >> j) & 1 ? fminf(fmaxf(valid[j] + 1.f, 0.f), 1.f) : valid[(size_t)(r / K) * J + j];
  } else if (t < nc + nv + nn) {
    const long long u = t - nc - nv;
    const int d = u % ND, r = u / ND;
    noise_o[u] = fixed_mask ? noise[(size_t)(r % K) * ND + d] : noise[u];
  }
}

// part_encoders.py:1105-1108: ids = arange * valid + argmax(valid) * (1 - valid); npoints // J consecutive
// points per part; per-point gathers of mean and logvar + lsv (gather_all :417-428).
__global__ void k_finish(const float *__restrict__ valid, const float *__restrict__ mean,
                         const float *__restrict__ logvar, int32_t *__restrict__ seg, float *__restrict__ mpp,
                         float *__restrict__ lpp, int R, int J, int npoints, float lsv) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = npoints / J, n_eff = per * J;
  if (t >= (long long)R * n_eff) return;
  const int p = t % n_eff, r = t / n_eff, j = p / per;
  int first = 0;
  float best = valid[(size_t)r * J];
  for (int jj = 1; jj < J; ++jj)
    if (valid[(size_t)r * J + jj] > best) best = valid[(size_t)r * J + jj], first = jj;   // torch.argmax: first maximum
  const float vj = valid[(size_t)r * J + j];
  const int id = (int)((float)j * vj + (float)first * (1.f - vj));
  if (seg) seg[(size_t)r * n_eff + p] = id;
  for (int c = 0; c < 3; ++c) {
    if (mpp) mpp[((size_t)r * 3 + c) * n_eff + p] = mean[((size_t)r * 3 + c) * J + id];
    if (lpp) lpp[((size_t)r * 3 + c) * n_eff + p] = logvar[((size_t)r * 3 + c) * J + id] + lsv;
  }
}

inline int nblk(long long n, int bs = 256) { return (int)((n + bs - 1) / bs); }

}  // namespace

struct dfx_latents {
  int J = 0, Z = 0, flow_depth = 0, flow_hidden = 0, depth = 0, heads = 0, dh = 0, cimle = 0, nd = 0, inner = 0,
      in_ch = 0;
  float noise_scale = 0, prior_std = 1, lsv = 0;
  float *wbuf = nullptr;
  struct FlowLayer { size_t w0, b0, w1, b1, w2, b2; };   // each [part][...] contiguous
  std::vector<FlowLayer> flow;
  struct Block { size_t n2w, n2b, wqkv, wo, bo, n3w, n3b, w1, b1, w2, b2; };
  std::vector<Block> blk;
  size_t proj_in_w = 0, proj_in_b = 0, class_emb = 0, pre_w = 0, pre_b = 0, post_w = 0, post_b = 0, out_w = 0,
         out_b = 0;
  float *ws = nullptr;
  size_t ws_floats = 0;
  int reserve(size_t floats) {
    if (floats <= ws_floats) return DFX_OK;
    if (ws) (void)hipFree(ws);   // hipFree synchronises: earlier launches that use the old block are done
    ws = nullptr, ws_floats = 0;
    if (hipMalloc(&ws, floats * sizeof(float)) != hipSuccess) return dfx::set_error(DFX_ERR_ALLOC, "latents: workspace");
    ws_floats = floats;
    return DFX_OK;
  }
};

namespace {

template <int EPI>
void lin(hipStream_t st, int groups, const LinArgs &a) { dfx::lin::launch<EPI>(st, groups, a); }

// SequentialFlow(reverse=True) for all parts at once on X (J, S, Z) (blockIdx.z = part).
void run_flows(const dfx_latents *h, float *X, float *H1, float *H2, int S, hipStream_t st) {
  const int Z = h->Z, Hd = h->flow_hidden, d = Z / 2, J = h->J;
  const float *W = h->wbuf;
  for (int l = h->flow_depth - 1; l >= 0; --l) {
    const bool swap = (l % 2 == 0);          // flow.py:78
    const int co = swap ? d : 0, to = swap ? 0 : d;   // conditioning half / transformed half (flow.py:23-24,44)
    const dfx_latents::FlowLayer &f = h->flow[l];
    LinArgs a{};
    a.M = S;
    a.X = X + co, a.x_gs = (long long)S * Z, a.ldx = Z, a.K = d;
    a.W = W + f.w0, a.w_gs = (long long)Hd * d, a.b = W + f.b0, a.b_gs = Hd;
    a.Y = H1, a.y_gs = (long long)S * Hd, a.ldy = Hd, a.N = Hd;
    lin<EPI_RELU>(st, J, a);
    a.X = H1, a.x_gs = (long long)S * Hd, a.ldx = Hd, a.K = Hd;
    a.W = W + f.w1, a.w_gs = (long long)Hd * Hd, a.b = W + f.b1;
    a.Y = H2;
    lin<EPI_RELU>(st, J, a);
    a.X = H2;
    a.W = W + f.w2, a.w_gs = (long long)2 * d * Hd, a.b = W + f.b2, a.b_gs = 2 * d;
    a.Y = X + to, a.y_gs = (long long)S * Z, a.ldy = Z, a.N = d;
    lin<EPI_COUPLING>(st, J, a);
  }
}

// PartAlignerTransformer on B shapes; code (B,Z,J), valid (B,J), noise (B,nd) device pointers.  ws layout is
// carved by the caller.  Writes mean / logvar / params (any may be null).
int run_aligner(dfx_latents *h, const float *code, const float *valid, const float *noise, float *mean,
                float *logvar, float *params, int B, float *ws, hipStream_t st) {
  const int J = h->J, M = B * J, C = h->inner, IC = h->in_ch;
  float *X0 = ws, *X = X0 + (size_t)M * IC, *Xn = X + (size_t)M * C, *QKV = Xn + (size_t)M * C,
        *O = QKV + (size_t)M * 3 * C, *G = O + (size_t)M * C, *Ho = G + (size_t)M * 4 * C;
  const float *W = h->wbuf;
  k_tokens<<<nblk((long long)M * IC), 256, 0, st>>>(code, noise, X0, B, h->Z, J, h->cimle ? h->nd : 0, h->noise_scale);
  LinArgs a{};
  a.M = M;
  a.X = X0, a.ldx = IC, a.K = IC, a.W = W + h->proj_in_w, a.b = W + h->proj_in_b, a.Y = X, a.ldy = C, a.N = C;
  a.R = W + h->class_emb, a.ldr = C, a.r_mod = J;    // x + class_emb[token] (part_encoders.py:116-118)
  lin<EPI_RESID>(st, 1, a);
  if (!h->cimle) k_ln<<<nblk(M, 4), 256, 0, st>>>(X, W + h->pre_w, W + h->pre_b, X, M, C);   // :130-131
  const float scale = 1.f / sqrtf((float)h->dh);
  for (int i = 0; i < h->depth; ++i) {
    const dfx_latents::Block &k = h->blk[i];
    k_ln<<<nblk(M, 4), 256, 0, st>>>(X, W + k.n2w, W + k.n2b, Xn, M, C);
    a = LinArgs{};
    a.M = M, a.X = Xn, a.ldx = C, a.K = C, a.W = W + k.wqkv, a.Y = QKV, a.ldy = 3 * C, a.N = 3 * C;
    lin<EPI_NONE>(st, 1, a);
    if (h->dh == 32) k_attn<32><<<nblk((long long)M * h->heads), 256, 0, st>>>(QKV, valid, O, M, J, h->heads, scale);
    else if (h->dh == 16) k_attn<16><<<nblk((long long)M * h->heads), 256, 0, st>>>(QKV, valid, O, M, J, h->heads, scale);
    else k_attn<64><<<nblk((long long)M * h->heads), 256, 0, st>>>(QKV, valid, O, M, J, h->heads, scale);
    a = LinArgs{};
    a.M = M, a.X = O, a.ldx = C, a.K = C, a.W = W + k.wo, a.b = W + k.bo, a.Y = X, a.ldy = C, a.N = C;
    a.R = X, a.ldr = C, a.r_mod = 0;                  // attn2(norm2(x)) + x (attention.py:300)
    lin<EPI_RESID>(st, 1, a);
    k_ln<<<nblk(M, 4), 256, 0, st>>>(X, W + k.n3w, W + k.n3b, Xn, M, C);
    a = LinArgs{};
    a.M = M, a.X = Xn, a.ldx = C, a.K = C, a.W = W + k.w1, a.b = W + k.b1, a.Y = G, a.ldy = 4 * C, a.N = 4 * C;
    lin<EPI_GEGLU>(st, 1, a);
    a = LinArgs{};
    a.M = M, a.X = G, a.ldx = 4 * C, a.K = 4 * C, a.W = W + k.w2, a.b = W + k.b2, a.Y = X, a.ldy = C, a.N = C;
    a.R = X, a.ldr = C, a.r_mod = 0;                  // ff(norm3(x)) + x (attention.py:305)
    lin<EPI_RESID>(st, 1, a);
  }
  k_ln<<<nblk(M, 4), 256, 0, st>>>(X, W + h->post_w, W + h->post_b, Xn, M, C);
  a = LinArgs{};
  a.M = M, a.X = Xn, a.ldx = C, a.K = C, a.W = W + h->out_w, a.b = W + h->out_b, a.Y = Ho, a.ldy = 6, a.N = 6;
  lin<EPI_NONE>(st, 1, a);
  k_split<<<nblk((long long)B * 6 * J), 256, 0, st>>>(Ho, mean, logvar, params, B, J, h->lsv);
  return dfx::check_launch("part_aligner");
}

size_t aligner_ws_floats(const dfx_latents *h, int B) {
  const size_t M = (size_t)B * h->J, C = h->inner;
  return M * (h->in_ch + C + C + 3 * C + C + 4 * C + 8);
}

}  // namespace

extern "C" {

int dfx_latents_create(dfx_latents **out, const dfx_latent_weights *w, dfx_stream_t stream) {
  DFX_REQUIRE(out && w, "latents_create: null argument");
  *out = nullptr;
  DFX_REQUIRE(w->n_class >= 1 && w->n_class <= 8, "latents_create: n_class %d outside [1,8]", w->n_class);
  DFX_REQUIRE(w->zdim > 0 && w->zdim % 16 == 0, "latents_create: zdim %d must be a multiple of 16", w->zdim);
  DFX_REQUIRE(w->flow_depth >= 0 && (w->flow_depth == 0 || (w->flow && w->flow_hidden > 0 && w->flow_hidden % 8 == 0)),
              "latents_create: bad flow description");
  DFX_REQUIRE(w->depth >= 1 && w->depth <= DFX_MAX_DEPTH, "latents_create: depth %d outside [1,%d]", w->depth, DFX_MAX_DEPTH);
  DFX_REQUIRE(w->d_head == 16 || w->d_head == 32 || w->d_head == 64, "latents_create: d_head %d not in {16,32,64}", w->d_head);
  const int inner = w->n_heads * w->d_head;
  DFX_REQUIRE(inner % 64 == 0 && inner <= 1024, "latents_create: inner dim %d must be a multiple of 64, <= 1024", inner);
  DFX_REQUIRE(!w->cimle || (w->noise_dim > 0 && w->noise_dim % 8 == 0), "latents_create: noise_dim %d must be a multiple of 8", w->noise_dim);
  DFX_REQUIRE(w->proj_in_w && w->proj_in_b && w->class_emb && w->post_norm_w && w->post_norm_b && w->proj_out_w && w->proj_out_b,
              "latents_create: null aligner weight");
  DFX_REQUIRE(w->cimle || (w->pre_norm_w && w->pre_norm_b), "latents_create: pre_norm weights required when !cimle");
  hipStream_t st = dfx::as_stream(stream);
  dfx_latents *h = new dfx_latents();
  h->J = w->n_class, h->Z = w->zdim, h->flow_depth = w->flow_depth, h->flow_hidden = w->flow_hidden;
  h->depth = w->depth, h->heads = w->n_heads, h->dh = w->d_head, h->cimle = w->cimle ? 1 : 0, h->nd = w->noise_dim;
  h->inner = inner, h->in_ch = w->zdim + (h->cimle ? w->noise_dim : 0);
  h->noise_scale = w->noise_scale, h->prior_std = sqrtf(w->prior_var), h->lsv = w->log_scale_var;

  // lay the weights out in one device buffer (16-byte aligned pieces); flows as [layer][tensor][part]
  struct Copy { size_t off; const float *src; size_t n; };
  std::vector<Copy> copies;
  size_t cur = 0;
  auto put = [&](const float *src, size_t n) {
    const size_t off = cur;
    copies.push_back({off, src, n});
    cur += (n + 3) & ~(size_t)3;
    return off;
  };
  const int J = h->J, Z = h->Z, d = Z / 2, Hd = h->flow_hidden;
  bool ok = true;
  for (int l = 0; l < h->flow_depth; ++l) {
    dfx_latents::FlowLayer f{};
    size_t *slots[6] = {&f.w0, &f.b0, &f.w1, &f.b1, &f.w2, &f.b2};
    const size_t sizes[6] = {(size_t)Hd * d, (size_t)Hd, (size_t)Hd * Hd, (size_t)Hd, (size_t)2 * d * Hd, (size_t)2 * d};
    for (int k = 0; k < 6; ++k)
      for (int p = 0; p < J; ++p) {
        const float *src = w->flow[((size_t)p * h->flow_depth + l) * 6 + k];
        ok = ok && src;
        const size_t off = put(src, sizes[k]);     // sizes are multiples of 4: parts are contiguous
        if (p == 0) *slots[k] = off;
      }
    h->flow.push_back(f);
  }
  const size_t C = inner;
  h->proj_in_w = put(w->proj_in_w, C * h->in_ch), h->proj_in_b = put(w->proj_in_b, C);
  h->class_emb = put(w->class_emb, (size_t)J * C);
  if (!h->cimle) h->pre_w = put(w->pre_norm_w, C), h->pre_b = put(w->pre_norm_b, C);
  h->post_w = put(w->post_norm_w, C), h->post_b = put(w->post_norm_b, C);
  h->out_w = put(w->proj_out_w, 6 * C), h->out_b = put(w->proj_out_b, 6);
  for (int i = 0; i < h->depth; ++i) {
    const dfx_aligner_block_weights &b = w->blocks[i];
    ok = ok && b.norm2_w && b.norm2_b && b.to_q && b.to_k && b.to_v && b.to_out_w && b.to_out_b && b.norm3_w &&
         b.norm3_b && b.ff_proj_w && b.ff_proj_b && b.ff_out_w && b.ff_out_b;
    dfx_latents::Block k{};
    k.n2w = put(b.norm2_w, C), k.n2b = put(b.norm2_b, C);
    k.wqkv = put(b.to_q, C * C), put(b.to_k, C * C), put(b.to_v, C * C);   // contiguous [q;k;v] rows
    k.wo = put(b.to_out_w, C * C), k.bo = put(b.to_out_b, C);
    k.n3w = put(b.norm3_w, C), k.n3b = put(b.norm3_b, C);
    k.w1 = put(b.ff_proj_w, 8 * C * C), k.b1 = put(b.ff_proj_b, 8 * C);
    k.w2 = put(b.ff_out_w, 4 * C * C), k.b2 = put(b.ff_out_b, C);
    h->blk.push_back(k);
  }
  if (!ok) {
    delete h;
    return dfx::set_error(DFX_ERR_INVALID_ARG, "latents_create: null weight pointer");
  }
  if (hipMalloc(&h->wbuf, cur * sizeof(float)) != hipSuccess) {
    delete h;
    return dfx::set_error(DFX_ERR_ALLOC, "latents_create: %zu bytes of weights", cur * sizeof(float));
  }
  for (const Copy &c : copies) {
    hipError_t e = hipMemcpyAsync(h->wbuf + c.off, c.src, c.n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
      (void)hipFree(h->wbuf);
      delete h;
      return dfx::set_error(DFX_ERR_HIP, "latents_create: weight copy: %s", hipGetErrorString(e));
    }
  }
  if (hipStreamSynchronize(st) != hipSuccess) {   // the caller's tensors may be released after create returns
    (void)hipFree(h->wbuf);
    delete h;
    return dfx::set_error(DFX_ERR_HIP, "latents_create: stream synchronise failed");
  }
  *out = h;
  return DFX_OK;
}

void dfx_latents_destroy(dfx_latents *h) {
  if (!h) return;
  if (h->wbuf) (void)hipFree(h->wbuf);
  if (h->ws) (void)hipFree(h->ws);
  delete h;
}

int dfx_flow_reverse(dfx_latents *h, const float *w, float *part_code, int S, dfx_stream_t stream) {
  DFX_REQUIRE(h && S >= 0, "flow_reverse: bad argument");
  if (S == 0) return DFX_OK;
  DFX_REQUIRE(w && part_code, "flow_reverse: null pointer");
  hipStream_t st = dfx::as_stream(stream);
  const size_t nX = (size_t)h->J * S * h->Z, nH = (size_t)h->J * S * (h->flow_hidden > 0 ? h->flow_hidden : 1);
  if (int e = h->reserve(nX + 2 * nH)) return e;
  float *X = h->ws, *H1 = X + nX, *H2 = H1 + nH;
  const long long n = (long long)S * h->Z * h->J;
  k_flow_in<<<nblk(n), 256, 0, st>>>(w, X, S, h->Z, h->J, h->prior_std);
  run_flows(h, X, H1, H2, S, st);
  k_flow_out<<<nblk(n), 256, 0, st>>>(X, part_code, S, h->Z, h->J);
  return dfx::check_launch("flow_reverse");
}

int dfx_part_aligner(dfx_latents *h, const float *part_code, const float *valid_id, const float *noise, float *mean,
                     float *logvar, int B, dfx_stream_t stream) {
  DFX_REQUIRE(h && B >= 0, "part_aligner: bad argument");
  if (B == 0) return DFX_OK;
  DFX_REQUIRE(part_code && valid_id && mean && logvar, "part_aligner: null pointer");
  DFX_REQUIRE(!h->cimle == !noise, "part_aligner: noise must be given iff the aligner was built with cimle");
  if (int e = h->reserve(aligner_ws_floats(h, B))) return e;
  return run_aligner(h, part_code, valid_id, noise, mean, logvar, nullptr, B, h->ws, dfx::as_stream(stream));
}

int dfx_sample_latents(dfx_latents *h, const float *w_noise, const float *part_code_in, const float *aligner_noise,
                       const float *valid_id, const int32_t *fixed_id, int S, int K, int npoints, float *part_code,
                       float *valid_out, float *noise_out, float *mean, float *logvar, float *params, int32_t *seg,
                       float *mean_per_point, float *logvar_per_point, dfx_stream_t stream) {
  DFX_REQUIRE(h && S >= 0 && K >= 1 && npoints >= 0, "sample_latents: bad sizes");
  if (S == 0) return DFX_OK;
  DFX_REQUIRE((w_noise != nullptr) != (part_code_in != nullptr), "sample_latents: give exactly one of w_noise / part_code_in");
  DFX_REQUIRE(valid_id && part_code && valid_out && mean && logvar, "sample_latents: null pointer");
  DFX_REQUIRE(!h->cimle == !aligner_noise, "sample_latents: aligner_noise must be given iff cimle");
  DFX_REQUIRE(h->cimle || K == 1, "sample_latents: K must be 1 without cimle (part_encoders.py:1068)");
  DFX_REQUIRE(!h->cimle || noise_out, "sample_latents: noise_out required with cimle");
  DFX_REQUIRE(npoints % h->J == 0, "sample_latents: npoints %d must be a multiple of n_class %d", npoints, h->J);
  hipStream_t st = dfx::as_stream(stream);
  unsigned fixed_mask = 0;
  if (fixed_id)
    for (int j = 0; j < h->J; ++j) fixed_mask |= (fixed_id[j] != 0 ? 1u : 0u) << j;
  const int J = h->J, Z = h->Z, R = S * K;
  const size_t nX = (size_t)J * S * Z, nH = (size_t)J * S * (h->flow_hidden > 0 ? h->flow_hidden : 1);
  const size_t n_code = (size_t)S * Z * J;
  const size_t flow_ws = nX + 2 * nH, al_ws = aligner_ws_floats(h, R);
  if (int e = h->reserve(n_code + (flow_ws > al_ws ? flow_ws : al_ws))) return e;
  float *code0 = h->ws, *scratch = code0 + n_code;
  const float *code_src = part_code_in;
  if (w_noise) {                                    // part_encoders.py:1054-1060
    float *X = scratch, *H1 = X + nX, *H2 = H1 + nH;
    if (h->flow_depth > 0) {
      k_flow_in<<<nblk((long long)n_code), 256, 0, st>>>(w_noise, X, S, Z, J, h->prior_std);
      run_flows(h, X, H1, H2, S, st);
      k_flow_out<<<nblk((long long)n_code), 256, 0, st>>>(X, code0, S, Z, J);
    } else {
      k_flow_in<<<nblk((long long)n_code), 256, 0, st>>>(w_noise, X, S, Z, J, h->prior_std);
      k_flow_out<<<nblk((long long)n_code), 256, 0, st>>>(X, code0, S, Z, J);
    }
    code_src = code0;
  }
  const int ND = h->cimle ? h->nd : 0;
  const long long nmix = (long long)R * Z * J + (long long)R * J + (long long)R * ND;
  k_mix_repeat<<<nblk(nmix), 256, 0, st>>>(code_src, valid_id, aligner_noise, part_code, valid_out, noise_out, S, K, Z, J,
                                           ND, fixed_mask);
  if (int e = run_aligner(h, part_code, valid_out, h->cimle ? noise_out : nullptr, mean, logvar, params, R, scratch, st))
    return e;
  if (npoints > 0 && (seg || mean_per_point || logvar_per_point))
    k_finish<<<nblk((long long)R * npoints), 256, 0, st>>>(valid_out, mean, logvar, seg, mean_per_point, logvar_per_point, R,
                                                          J, npoints, h->lsv);
  return dfx::check_launch("sample_latents");
}

}  // extern "C"
