/* libdfx — MI355X-native (gfx950) kernels for DiffFacto's reverse-diffusion sampling hot path.
 *
 * C-ABI drop-in boundary.  Plain pointers and sizes only; every pointer is a
 * DEVICE pointer unless stated otherwise; every launch goes on the caller's HIP stream
 * (`stream` is a hipStream_t passed as void*; NULL = the default stream), no hidden syncs.
 * Inputs are borrowed, outputs are caller-allocated and fully written by the call (no
 * dependence on pre-zeroed outputs unless stated).  Return value: 0 = DFX_OK, negative =
 * error (never exit(), unlike the reference's CUDA_CHECK_ERRORS,
 * pointnet2_ops_lib/pointnet2_ops/_ext-src/include/cuda_utils.h:30-39); dfx_last_error() gives the
 * message for the calling thread.  Single caller thread per device.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repo root; SRC = pointnet2_ops_lib/pointnet2_ops/_ext-src/src).
 */
#ifndef DFX_H
#define DFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFX_OK 0
#define DFX_ERR_INVALID_ARG (-1)
#define DFX_ERR_HIP (-2)
#define DFX_ERR_UNSUPPORTED (-3)
#define DFX_ERR_ALLOC (-4)

#define DFX_MAX_DEPTH 8

/* arithmetic of the dense contraction (LayerNorm, softmax, GELU, posterior are fp32 in both) */
#define DFX_PREC_F32 0  /* v_mfma_f32_32x32x2_f32: exact fp32, the parity gate            */
#define DFX_PREC_BF16 1 /* v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate       */

typedef void *dfx_stream_t;
#define DFX_MAX_SIDE_STREAMS 64   /* see dfx_denoiser_train_forward */

int dfx_version(void);
/* Bumped whenever an existing entry point changes its argument list (round 2 inserted `shape_offset` into the four sampling
 * calls: 2; round 4 added dfx_last_kernel_variant: 3 — additive, bumped so that a binding written against 3 does not load a library without it;
 * round 5: d_x / d_variances of the denoiser backward: 4; round 6 added dfx_shared_mlp_train_*: 5).
 * Bindings compare dfx_abi_version() with the DFX_ABI_VERSION they were written against at load time (_ffi.py does)
 * instead of shifting arguments silently. */
#define DFX_ABI_VERSION 5
int dfx_abi_version(void);
const char *dfx_last_error(void);

/* ------------------------------------------------------------------------------------------
 * pointnet2_ops primitives — replace the pybind module `pointnet2_ops._ext`
 * (SRC/bindings.cpp:6-19); fp32 + int32, contiguous.
 * ------------------------------------------------------------------------------------------ */

/* _ext.gather_points  (SRC/sampling.cpp:15-38 -> SRC/sampling_gpu.cu:8-30)
 * out[b,c,j] = points[b,c,idx[b,j]];  points (B,C,N), idx (B,M) -> out (B,C,M) */
int dfx_gather_points_f32(const float *points, const int32_t *idx, float *out, int B, int C, int N, int M,
                          dfx_stream_t stream);

/* _ext.gather_points_grad  (SRC/sampling.cpp:40-64 -> SRC/sampling_gpu.cu:34-57)
 * grad_points (B,C,N) is ZEROED by the call, then scatter-added from grad_out (B,C,M). */
int dfx_gather_points_grad_f32(const float *grad_out, const int32_t *idx, float *grad_points, int B, int C, int N,
                               int M, dfx_stream_t stream);

/* _ext.furthest_point_sampling  (SRC/sampling.cpp:66-87 -> SRC/sampling_gpu.cu:69-229)
 * xyz (B,N,3) -> idx (B,M).  Starts at index 0, skips points with |p|^2 <= 1e-3, reproduces the
 * reference block-reduction tie rule for opt_n_threads(N) threads.  `tmp` (B,N) floats is scratch
 * (the reference's `temp`, filled with 1e10 by the call); may be NULL when N <= dfx_fps_max_resident(). */
int dfx_furthest_point_sampling_f32(const float *xyz, float *tmp, int32_t *idx, int B, int N, int M,
                                    dfx_stream_t stream);
int dfx_fps_max_resident(void);

/* _ext.ball_query  (SRC/ball_query.cpp:8-32 -> SRC/ball_query_gpu.cu:9-54)
 * new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,nsample): first `nsample` indices k (ascending) with
 * d2 < radius^2; first hit pre-fills all slots; no hit -> zeros.  Argument order is the C++ op's. */
int dfx_ball_query_f32(const float *new_xyz, const float *xyz, int32_t *idx, int B, int N, int M, float radius,
                       int nsample, dfx_stream_t stream);

/* _ext.group_points  (SRC/group_points.cpp:12-36 -> SRC/group_points_gpu.cu:8-39)
 * out[b,c,j,k] = points[b,c,idx[b,j,k]];  points (B,C,N), idx (B,npoints,nsample). */
int dfx_group_points_f32(const float *points, const int32_t *idx, float *out, int B, int C, int N, int npoints,
                         int nsample, dfx_stream_t stream);

/* _ext.group_points_grad  (SRC/group_points.cpp:38-62 -> SRC/group_points_gpu.cu:43-75); zeroes grad_points. */
int dfx_group_points_grad_f32(const float *grad_out, const int32_t *idx, float *grad_points, int B, int C, int N,
                              int npoints, int nsample, dfx_stream_t stream);

/* _ext.three_nn  (SRC/interpolate.cpp:14-40 -> SRC/interpolate_gpu.cu:9-68)
 * unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) SQUARED distances, idx (B,n,3). */
int dfx_three_nn_f32(const float *unknown, const float *known, float *dist2, int32_t *idx, int B, int n, int m,
                     dfx_stream_t stream);

/* _ext.three_interpolate (SRC/interpolate.cpp:42-70 -> SRC/interpolate_gpu.cu:72-111)
 * points (B,c,m), idx (B,n,3), weight (B,n,3) -> out (B,c,n). */
int dfx_three_interpolate_f32(const float *points, const int32_t *idx, const float *weight, float *out, int B,
                              int c, int m, int n, dfx_stream_t stream);

/* _ext.three_interpolate_grad (SRC/interpolate.cpp:72-99 -> SRC/interpolate_gpu.cu:116-154); zeroes grad_points (B,c,m). */
int dfx_three_interpolate_grad_f32(const float *grad_out, const int32_t *idx, const float *weight,
                                   float *grad_points, int B, int c, int n, int m, dfx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Cross-diffusion denoiser + anchored DDPM reverse process.
 * Replaces, for the shipped gen_* configs (inner 128 = 8 heads x 16, n_class 4, ctx 522,
 * GEGLU FF 128 -> 2*512 -> 128, single cross-attention per block):
 *   TransformerNet.forward        python/difffacto/models/diffusions/nets/attention.py:385-440
 *   AnchoredDiffusion.p_mean_variance / p_sample / p_sample_loop_progressive
 *                                 python/difffacto/models/diffusions/anchored_diffusion.py:227-395,450-484,528-588
 *   gather_operation of the per-part params (part_encoders.py:417-428) — folded in: the kernels
 *   index the (B,3,4) part params by `seg` instead of materialising (B,3,N) tensors.
 * ------------------------------------------------------------------------------------------ */

/* fp32 device pointers to the reference parameters (torch Linear layout: weight (out,in)),
 * named by their state_dict keys relative to `diffusion.model.`. */
typedef struct dfx_block_weights {
  const float *norm2_w, *norm2_b;   /* transformer_blocks.i.norm2.{weight,bias}      (128)     */
  const float *to_q;                /* ...attn2.to_q.weight                          (128,128) */
  const float *to_k, *to_v;         /* ...attn2.to_{k,v}.weight                      (128,522) */
  const float *to_out_w, *to_out_b; /* ...attn2.to_out.0.{weight,bias}               (128,128),(128) */
  const float *norm3_w, *norm3_b;   /* ...norm3.{weight,bias}                        (128)     */
  const float *ff0_w, *ff0_b;       /* ...ff.net.0.proj.{weight,bias}                (1024,128),(1024) */
  const float *ff2_w, *ff2_b;       /* ...ff.net.2.{weight,bias}                     (128,512),(128) */
} dfx_block_weights;

typedef struct dfx_denoiser_weights {
  int depth;                          /* number of transformer blocks (<= DFX_MAX_DEPTH)  */
  const float *proj_in_w, *proj_in_b; /* proj_in.{weight,bias}            (128,13),(128)  */
  const float *pre_norm_w, *pre_norm_b;
  const float *post_norm_w, *post_norm_b;
  const float *proj_out_w, *proj_out_b; /* proj_out.{weight,bias}         (3,128),(3)     */
  const float *te0_w, *te0_b;           /* time_embed.net.0.proj.{weight,bias} (2048,256),(2048) */
  const float *te2_w, *te2_b;           /* time_embed.net.2.{weight,bias}      (256,1024),(256)  */
  dfx_block_weights blk[DFX_MAX_DEPTH];
} dfx_denoiser_weights;

typedef struct dfx_denoiser dfx_denoiser; /* opaque; owns repacked weights + per-t tables on the device */

/* Build the frozen denoiser for one diffusion schedule (AnchoredDiffusion.__init__,
 * anchored_diffusion.py:62-112, mode='linear'): repacks weights into MFMA fragment order for
 * `precision`, tabulates the time-embedding MLP and the per-t attention/posterior constants for
 * t = 0..num_timesteps-1.  The source parameters are only read during this call (stream-ordered:
 * keep them alive until the stream has passed it). */
int dfx_denoiser_create(dfx_denoiser **out, const dfx_denoiser_weights *w, int num_timesteps, double beta_1,
                        double beta_T, int precision, dfx_stream_t stream);
void dfx_denoiser_destroy(dfx_denoiser *d);
int dfx_denoiser_num_timesteps(const dfx_denoiser *d);
int dfx_denoiser_precision(const dfx_denoiser *d);
/* DFX_PREC_BF16 engines: 1 if b1' = b1 + W1 beta3 rides in one hidden channel's K slot of the packed W1 (no accumulator initialisers in the chain
 * kernels; exact in real arithmetic — a LayerNorm output sums to zero, so one channel is redundant — and every weight of a row then rounds with a
 * step that follows that channel's column of W1 diag(gamma3)).  Create picks the channel from the weights: 127, or — when 127's column is an
 * outlier (ratio > 4 of its row's mean magnitude) — the channel with the smallest column over all blocks, exchanged with 127 in a private copy
 * of the parameters (a relabelling of the hidden channels: the same function).  0 only if every channel is an outlier of some block (ratio > 8):
 * the engine then has the plain pack and runs the direct kernel (~3x slower).  *ratio (may be NULL) receives the chosen channel's ratio.
 * fp32 engines: 0. */
int dfx_denoiser_w1_fold(const dfx_denoiser *d, float *ratio);

/* Copies the 8 fp32 schedule tables the kernels use to HOST memory, each `num_timesteps` long, in the
 * order: sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1, _coef2, _coef3,
 * posterior_variance, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod  (out: host float[8*T]). */
int dfx_denoiser_get_tables(const dfx_denoiser *d, float *host_out);

/* Per-batch static context (once per batch of shapes, before the T-loop): the 4 part tokens'
 * K/V projections folded with to_q / to_out and the per-part proj_in constants.
 *   part_code (B,256,4)  mean (B,3,4)  var (B,3,4) = exp(logvar)   [ctx list of prepare_ctx,
 *   part_encoders.py:1317-1326]   valid (B,4) floats (mask of attention.py:192-197)
 * `ctx_out` is a caller-allocated device buffer of dfx_shape_ctx_bytes(d, B) bytes. */
size_t dfx_shape_ctx_bytes(const dfx_denoiser *d, int B);
int dfx_shape_ctx_prepare(const dfx_denoiser *d, const float *part_code, const float *mean, const float *var,
                          const float *valid, void *ctx_out, int B, dfx_stream_t stream);

/* One TransformerNet.forward: eps (B,3,N) = eps_theta(x (B,3,N), t) with anchors/variances/one-hot
 * taken from the shape context through seg (B,N) int32 in [0,4).  N % 32 == 0.  `t` is the same for
 * the whole batch (anchored_diffusion.py:576). */
int dfx_denoise_eps(const dfx_denoiser *d, const void *shape_ctx, const float *x, const int32_t *seg, int t,
                    float *eps, int B, int N, dfx_stream_t stream);

/* One p_sample (anchored_diffusion.py:450-484): x_prev = mean + 1[t!=0] sqrt(var) z.
 * noise (B,3,N) = the z of :476, or NULL -> in-kernel Philox4x32-10 normals keyed by
 * (seed, global point id, t), global point id = (shape_offset + b) * N + n: `shape_offset` is the global index of this
 * call's shape 0, so a batch that is split over several calls or GPUs draws exactly the normals of
 * the unsplit call, whatever the split.  x_prev may alias x.  pred_xstart (B,3,N) is optional (NULL to skip):
 * the x_0 prediction p_sample returns next to the sample (:484). */
int dfx_p_sample(const dfx_denoiser *d, const void *shape_ctx, const float *x, const int32_t *seg, int t,
                 const float *noise, uint64_t seed, uint64_t shape_offset, float *x_prev, float *pred_xstart, int B, int N,
                 dfx_stream_t stream);

/* The whole reverse chain in ONE persistent launch (p_sample_loop_progressive :528-588 driven by
 * AnchorDiffAE.decode, python/difffacto/models/networks/anchor_gen.py:145-169): x_t stays in registers
 * for all T steps.
 *   x_T_noise  (B,3,N) the randn of :564, or NULL -> Philox
 *   step_noise (T,B,3,N), step_noise[i] = z of the i-th executed step (t = T-1-i), or NULL -> Philox
 *   seed, shape_offset  Philox key and global index of shape 0 (see dfx_p_sample): rank r of a sharded run passes the
 *              index of its first shape, and the clouds do not depend on the number of GPUs
 *   pred       (B,N,3)  final cloud, already transposed like decode's 'pred'
 *   traj       NULL, or (n_keep,B,N,3) snapshots for t = T, then every t>0 with t % ret_interval == 0 in
 *              descending order (decode's ret_traj entries); n_keep = dfx_chain_num_snapshots(T, ret_interval). */
int dfx_chain_num_snapshots(int num_timesteps, int ret_interval);
int dfx_sample_chain(const dfx_denoiser *d, const void *shape_ctx, const int32_t *seg, const float *x_T_noise,
                     const float *step_noise, uint64_t seed, uint64_t shape_offset, int ret_interval, float *traj, float *pred,
                     int B, int N, dfx_stream_t stream);

/* Training-style forward evaluation (forward only; anchored_diffusion.py:760-853 in eval mode):
 *   dfx_q_sample_f32    x_t = sqrt_acp[t_b] (x0 - a) + a + sqrt(1 - acp[t_b]) L noise   (:148-173), t (B,) int32 on the device
 *   dfx_denoise_eps_t   dfx_denoise_eps with one timestep per shape
 *   dfx_masked_mse_f32  ((target - pred)^2 * flags).mean(1).sum() / flags.sum()  (:840-847; flags (B,N) or NULL = plain mean);
 *                       workspace2 = 2 doubles on the device, loss = 1 float on the device
 * Dropout and gradients are not part of this path. */
int dfx_q_sample_f32(const dfx_denoiser *d, const void *shape_ctx, const int32_t *seg, const int32_t *t, const float *x_start,
                     const float *noise, float *x_t, int B, int N, dfx_stream_t stream);
int dfx_denoise_eps_t(const dfx_denoiser *d, const void *shape_ctx, const float *x, const int32_t *seg, const int32_t *t,
                      float *eps, int B, int N, dfx_stream_t stream);
int dfx_masked_mse_f32(const float *target, const float *pred, const float *flags, double *workspace2, float *loss, int B,
                       int N, dfx_stream_t stream);

/* DDIM branch (anchored_diffusion.py:114-124 step lists, :368-377 xt_dir, :480-481 update):
 *   x_prev = (x0 - a) sqrt(acp_prev[t]) + a + L xt_dir_coeff[t] eps + eta 1[t != 0] sqrt(variance) z,
 *   xt_dir_coeff = sqrt(1 - acp - eta^2 posterior_variance) (float64 on the host).
 * `steps` is a HOST array, ascending like the reference's self.steps (e.g. 'quad': [0,0,2,5,10,16,23,32]); it is
 * executed in reverse; n_steps <= 128; steps[0] must be 0 for the chain.  step_noise is (n_steps,B,3,N) or NULL.
 * traj slot k still means t = (T / ret_interval - k) * ret_interval; slots of timesteps that are not visited are
 * left untouched. */
int dfx_p_sample_ddim(const dfx_denoiser *d, const void *shape_ctx, const float *x, const int32_t *seg, int t, float eta,
                      const float *noise, uint64_t seed, uint64_t shape_offset, float *x_prev, float *pred_xstart, int B, int N,
                      dfx_stream_t stream);
int dfx_sample_chain_ddim(const dfx_denoiser *d, const void *shape_ctx, const int32_t *seg, const int32_t *steps,
                          int n_steps, float eta, const float *x_T_noise, const float *step_noise, uint64_t seed,
                          uint64_t shape_offset, int ret_interval, float *traj, float *pred, int B, int N, dfx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Chamfer-L2 — replaces the `chamfer` extension
 * (python/difffacto/metrics/chamfer_dist/chamfer.cu: forward :15-170, backward :173-230; bound in
 * python/difffacto/metrics/chamfer_dist/__init__.py:14-27).
 * ------------------------------------------------------------------------------------------ */

/* chamfer.forward(xyz1 (B,N,3), xyz2 (B,M,3)) -> dist1 (B,N), dist2 (B,M) squared NN distances, idx1, idx2 int32 */
int dfx_chamfer_forward_f32(const float *xyz1, const float *xyz2, float *dist1, float *dist2, int32_t *idx1,
                            int32_t *idx2, int B, int N, int M, dfx_stream_t stream);

/* chamfer.backward(...) -> grad_xyz1 (B,N,3), grad_xyz2 (B,M,3); both zeroed by the call, atomics like the reference */
int dfx_chamfer_backward_f32(const float *xyz1, const float *xyz2, const int32_t *idx1, const int32_t *idx2,
                             const float *grad_dist1, const float *grad_dist2, float *grad_xyz1, float *grad_xyz2,
                             int B, int N, int M, dfx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Latent sampler — the once-per-batch producer of decode's inputs:
 * PartEncoder.sample_latents (python/difffacto/models/encoders/part_encoders.py:1052-1110) =
 * per-part normalising flows run in reverse (python/difffacto/models/encoders/flow.py:21-47,58-72)
 * + PartAlignerTransformer (part_encoders.py:20-143) + seg-mask ids / per-point gathers
 * (part_encoders.py:1105-1108, :417-428) + PartEncoderForTransformerDecoder.prepare_ctx (:1317-1326).
 * fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32); supported configuration = the shipped gen configs
 * (class_cond + add_class_cond, use_linear, single_attn, cimle with cond_noise_type 0, or no cimle).
 * ------------------------------------------------------------------------------------------ */
typedef struct dfx_aligner_block_weights { /* BasicTransformerBlock(single_attn) of the aligner, device pointers */
  const float *norm2_w, *norm2_b;   /* (inner) */
  const float *to_q, *to_k, *to_v;  /* (inner, inner), bias-free */
  const float *to_out_w, *to_out_b; /* (inner, inner), (inner) */
  const float *norm3_w, *norm3_b;
  const float *ff_proj_w, *ff_proj_b; /* (8*inner, inner), (8*inner): GEGLU */
  const float *ff_out_w, *ff_out_b;   /* (inner, 4*inner), (inner) */
} dfx_aligner_block_weights;

typedef struct dfx_latent_weights {
  int32_t n_class, zdim;                /* 4, 256 */
  int32_t flow_depth, flow_hidden;      /* 14, 256; flow_depth 0 = use_flow False */
  /* host array of n_class*flow_depth*6 device pointers, [part][layer][w0,b0,w1,b1,w2,b2] = net_s_t.{0,2,4} */
  const float *const *flow;
  int32_t depth, n_heads, d_head;       /* 5, 8, 32 */
  int32_t cimle, noise_dim;             /* 1, 32 (cond_noise_type 0: noise*noise_scale concatenated per token) */
  float noise_scale, prior_var, log_scale_var;
  const float *proj_in_w, *proj_in_b;   /* (inner, zdim + cimle*noise_dim) */
  const float *class_emb;               /* (n_class, inner) */
  const float *pre_norm_w, *pre_norm_b; /* applied only when !cimle (part_encoders.py:115-131) */
  const float *post_norm_w, *post_norm_b;
  const float *proj_out_w, *proj_out_b; /* (6, inner) */
  dfx_aligner_block_weights blocks[DFX_MAX_DEPTH];
} dfx_latent_weights;

typedef struct dfx_latents dfx_latents; /* opaque; owns a copy of the weights + a grow-only workspace */

int dfx_latents_create(dfx_latents **out, const dfx_latent_weights *w, dfx_stream_t stream);
void dfx_latents_destroy(dfx_latents *h);

/* flow[i](w[..., i], reverse=True) for every part (part_encoders.py:1054-1060):
 * w (S,zdim,n_class) standard normal -> part_code (S,zdim,n_class); w is scaled by sqrt(prior_var) first. */
int dfx_flow_reverse(dfx_latents *h, const float *w, float *part_code, int S, dfx_stream_t stream);

/* PartAlignerTransformer.forward (part_encoders.py:88-109): part_code (B,zdim,n_class), valid_id (B,n_class)
 * {0,1} floats = the key mask, noise (B,noise_dim) (NULL iff !cimle) -> mean (B,3,n_class), logvar (B,3,n_class). */
int dfx_part_aligner(dfx_latents *h, const float *part_code, const float *valid_id, const float *noise, float *mean,
                     float *logvar, int B, dfx_stream_t stream);

/* The whole sample_latents after the random draws (part_encoders.py:1052-1110; no selective sampling):
 *   w_noise (S,zdim,n_class) randn of :1054, or part_code_in (S,zdim,n_class) given codes (then w_noise NULL)
 *   aligner_noise (S*K,noise_dim) randn of :1065 (NULL iff !cimle; then K must be 1)
 *   valid_id (S,n_class); fixed_id: HOST array of n_class 0/1 flags (:1072-1082)
 * outputs, R = S*K rows (any may be NULL to skip, except those the chain needs):
 *   part_code (R,zdim,n_class), valid_out (R,n_class), noise_out (R,noise_dim), mean/logvar (R,3,n_class),
 *   params (R,6,n_class) = [mean | exp(logvar + log_scale_var)] (= ctx[1]), seg (R,npoints) int32,
 *   mean_per_point / logvar_per_point (R,3,npoints) (logvar_per_point includes log_scale_var). */
int dfx_sample_latents(dfx_latents *h, const float *w_noise, const float *part_code_in, const float *aligner_noise,
                       const float *valid_id, const int32_t *fixed_id, int S, int K, int npoints, float *part_code,
                       float *valid_out, float *noise_out, float *mean, float *logvar, float *params, int32_t *seg,
                       float *mean_per_point, float *logvar_per_point, dfx_stream_t stream);

/* Training forward / backward of the part aligner (stage 2: configs/train_*_stage2.py and gen_*.py with train_aligner; replaces torch autograd through
 * PartAlignerTransformer, part_encoders.py:88-143, for the shipped options: cimle with cond_noise_type 0, class_cond + add_class_cond, single_attn,
 * mask_out_unreferenced_code, dropout 0).  Exact fp32, no atomics (aligner_train.hip).  `w` holds the parameter pointers (flow fields unused), `grads` the
 * gradient buffers in the same struct (overwritten; pre_norm_* untouched: unused by this configuration).
 *   part_code (B,zdim,n_class), valid (B,n_class) 0/1 = the key mask, noise (B,noise_dim) -> mean / logvar (B,3,n_class)
 *   backward: d_mean / d_logvar (B,3,n_class) (either may be NULL = zero) -> grads, d_part_code (B,zdim,n_class) or NULL
 *   workspace: dfx_aligner_train_workspace_bytes(...) device bytes, shared by the forward and the backward of a step. */
size_t dfx_aligner_train_workspace_bytes(int B, int n_class, int zdim, int noise_dim, int n_heads, int d_head, int depth);
int dfx_aligner_train_forward(const dfx_latent_weights *w, void *workspace, size_t workspace_bytes, const float *part_code, const float *valid,
                              const float *noise, float *mean, float *logvar, int B, dfx_stream_t stream);
int dfx_aligner_train_backward(const dfx_latent_weights *w, void *workspace, size_t workspace_bytes, const float *valid, const float *d_mean,
                               const float *d_logvar, const dfx_latent_weights *grads, float *d_part_code, int B, dfx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PointNet++ set-abstraction / feature-propagation layers, eval mode — the part of
 * pointnet2_ops the reference leaves to PyTorch: QueryAndGroup / GroupAll
 * (pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py:296-333, :349-381) + build_shared_mlp's Conv2d 1x1 +
 * BatchNorm2d + ReLU stack (pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:9-19) + max_pool2d over nsample
 * (:62-70), and PointnetFPModule.forward (:170-209).  BatchNorm uses its running statistics (folded into the
 * convolution at create time); training mode: dfx_shared_mlp_train_* below.
 * ------------------------------------------------------------------------------------------ */
typedef struct dfx_shared_mlp dfx_shared_mlp; /* opaque; owns folded/packed weights + a grow-only workspace */

/* n_layers <= 4; channels[n_layers+1]; per layer device pointers: conv_w (Cout,Cin) [Conv2d weight (Cout,Cin,1,1)],
 * conv_b (Cout) or NULL, and either all of bn_w/bn_b/bn_mean/bn_var (Cout) or bn_w[l] == NULL (no BatchNorm). */
int dfx_shared_mlp_create(dfx_shared_mlp **out, int n_layers, const int32_t *channels, const float *const *conv_w,
                          const float *const *conv_b, const float *const *bn_w, const float *const *bn_b,
                          const float *const *bn_mean, const float *const *bn_var, float eps, uint32_t relu_mask,
                          dfx_stream_t stream); /* relu_mask bit l: ReLU after layer l (build_shared_mlp: all ones) */
void dfx_shared_mlp_destroy(dfx_shared_mlp *h);
int dfx_shared_mlp_is_fused(const dfx_shared_mlp *h); /* 1: the gather+MLP+max-pool kernel serves this MLP in one launch */

/* _PointnetSAModuleBase.forward after the centres are known: grouper -> mlp -> max over nsample.
 *   xyz (B,N,3); new_xyz (B,M,3) centres (NULL: no centring, GroupAll); features (B,C,N) or NULL (C = 0);
 *   idx (B,M,ns) int32 from ball_query, or NULL = GroupAll (M = 1, ns = N, every point);
 *   out (B, C_out, M).  force_general != 0 selects the layer-by-layer path (A/B and parity of the two paths). */
int dfx_sa_forward_f32(dfx_shared_mlp *h, const float *xyz, const float *new_xyz, const float *features,
                       const int32_t *idx, int use_xyz, float *out, int B, int N, int M, int ns, int C, int force_general,
                       dfx_stream_t stream);

/* PointnetFPModule.forward: unknown (B,n,3), known (B,m,3) or NULL, unknow_feats (B,C1,n) or NULL (C1 = 0),
 * known_feats (B,C2,m) [(B,C2,1) when known is NULL] -> out (B, C_out, n). */
int dfx_fp_forward_f32(dfx_shared_mlp *h, const float *unknown, const float *known, const float *unknow_feats,
                       const float *known_feats, float *out, int B, int n, int m, int C1, int C2, dfx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same shared MLP in TRAINING mode (round 6): build_shared_mlp's Conv2d 1x1 [+ BatchNorm2d with BATCH statistics and the running-
 * statistics update of nn.BatchNorm2d: momentum, unbiased variance] + ReLU stack over a grouped tensor, + max over nsample
 * (pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:9-19, :62-70; PointnetFPModule's mlp :170-209 with pool = 0), forward with saved
 * activations and backward — what autograd through nn.Conv2d / nn.BatchNorm2d / F.max_pool2d computes in the reference.  Exact fp32.
 *   x (B, ch[0], M, ns) = the grouper's output (QueryAndGroup / GroupAll through dfx_group_points, or the interpolated + skip features with
 *   ns = 1; a Linear + BatchNorm1d head over a batch of rows, e.g. PointNet2SSG.fc_layer (models/encoders/pointnet2.py:47-56): B = 1, M = rows, ns = 1); out (B, ch[layers], M) when pool != 0, else (B, ch[layers], M, ns).  ch[l] % 4 == 0 for l >= 1; layers <= DFX_MLP_MAX_LAYERS.
 *   conv_b[l] == NULL when the layer has BatchNorm (bn_w[l] != NULL); bn_mean / bn_var: running statistics, updated IN PLACE by the forward
 *   when momentum >= 0.  grads: the same struct whose conv_w / conv_b / bn_w / bn_b name WRITABLE buffers of the parameters' shapes.
 *   batch_stats = 1: BatchNorm of train() mode; 0: eval() mode under autograd (the running statistics normalise, nothing is updated); the same value
 *   for the forward and its backward.  d_x (B, ch[0], M, ns) or NULL.  workspace: dfx_shared_mlp_train_workspace_bytes, 256-byte aligned; the backward reads what the forward of
 *   the SAME (B, M, ns) call left in it.
 * ------------------------------------------------------------------------------------------ */
#define DFX_MLP_MAX_LAYERS 4
typedef struct dfx_shared_mlp_train {
  int layers;
  int ch[DFX_MLP_MAX_LAYERS + 1];
  const float *conv_w[DFX_MLP_MAX_LAYERS], *conv_b[DFX_MLP_MAX_LAYERS];
  const float *bn_w[DFX_MLP_MAX_LAYERS], *bn_b[DFX_MLP_MAX_LAYERS];
  float *bn_mean[DFX_MLP_MAX_LAYERS], *bn_var[DFX_MLP_MAX_LAYERS];
  float bn_eps;
  uint32_t relu_mask; /* bit l: ReLU behind layer l (build_shared_mlp: all ones; a Linear head's last layer: 0) */
} dfx_shared_mlp_train;
size_t dfx_shared_mlp_train_workspace_bytes(const dfx_shared_mlp_train *w, int B, int M, int ns);
int dfx_shared_mlp_train_forward(const dfx_shared_mlp_train *w, void *workspace, size_t workspace_bytes, const float *x, float *out, int B, int M,
                                 int ns, int pool, int batch_stats, float momentum, dfx_stream_t stream);
int dfx_shared_mlp_train_backward(const dfx_shared_mlp_train *w, void *workspace, size_t workspace_bytes, const float *d_out,
                                  const dfx_shared_mlp_train *grads, float *d_x, int B, int M, int ns, int pool, int batch_stats, dfx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PointNetV2 masked max-pool part encoder, eval mode — python/difffacto/models/encoders/pointnet.py:124-213
 * with per_part_mlp=True: conv1..4 + BN (ReLU after the first three) over the points, x * attn_weight (* num_anchors)
 * max-pooled over N per part (:194-198), per-part grouped-Conv1d heads 512 -> 256 -> 128 -> zdim for m and v (:200-203).
 * ------------------------------------------------------------------------------------------ */
typedef struct dfx_pointnet_v2_weights { /* fp32 device pointers, reference state_dict tensors */
  int32_t num_anchors, zdim, reweight_by_anchor;
  float bn_eps;
  const float *conv_w[4], *conv_b[4];                           /* conv{1..4}.weight (Cout,Cin,1), .bias */
  const float *bn_w[4], *bn_b[4], *bn_mean[4], *bn_var[4];      /* bn{1..4} */
  const float *head_w[2][3], *head_b[2][3];                     /* [0] = mlp_m, [1] = mlp_v: .0 / .3 / .6 weight (A*Cout,Cin,1), bias */
  const float *head_bn_w[2][2], *head_bn_b[2][2], *head_bn_mean[2][2], *head_bn_var[2][2]; /* .1 / .4 */
} dfx_pointnet_v2_weights;
typedef struct dfx_pointnet_v2 dfx_pointnet_v2;
int dfx_pointnet_v2_create(dfx_pointnet_v2 **out, const dfx_pointnet_v2_weights *w, dfx_stream_t stream);
void dfx_pointnet_v2_destroy(dfx_pointnet_v2 *h);
/* x (B,N,3), attn_weight (B,N,num_anchors) -> m, v (B,num_anchors,zdim) */
int dfx_pointnet_v2_forward_f32(dfx_pointnet_v2 *h, const float *x, const float *attn, float *m, float *v, int B, int N,
                                dfx_stream_t stream);

/* The same encoder in TRAIN mode (nn.BatchNorm1d with batch statistics over the B N points / over the B shapes in the heads) and
 * its backward (encoder side): m, v as above; the backward takes d m, d v (B,num_anchors,zdim) and writes the
 * gradients of conv / BatchNorm / head weights and biases into `grads` (same struct, writable pointers; its running-statistics
 * pointers are ignored).  momentum >= 0 updates the running statistics in place like nn.BatchNorm1d (unbiased variance),
 * momentum < 0 leaves them alone.  workspace: dfx_pointnet_v2_train_workspace_bytes(B, N, 4, zdim), 256-byte aligned, shared
 * by the forward and the backward of one step.  num_anchors = 4, B >= 2.  precision as for the denoiser. */
size_t dfx_pointnet_v2_train_workspace_bytes(int B, int N, int num_anchors, int zdim);
int dfx_pointnet_v2_train_forward(const dfx_pointnet_v2_weights *w, void *workspace, size_t workspace_bytes, const float *x,
                                  const float *attn, float *m, float *v, float momentum, int B, int N, int precision,
                                  dfx_stream_t stream);
int dfx_pointnet_v2_train_backward(const dfx_pointnet_v2_weights *w, void *workspace, size_t workspace_bytes, const float *attn,
                                   const float *dm, const float *dv, const dfx_pointnet_v2_weights *grads, int B, int N,
                                   int precision, dfx_stream_t stream);

/* Prior loss of the part encoder with use_flow (PartEncoder.get_prior_loss, part_encoders.py:1143-1182) and its backward
 * (encoder side): per part, the latents go FORWARD through that part's coupling layers (flow.py:21-41,
 * logpx - log det), log N(w; 0, prior_var) with the reference's normalisation (misc.py:301-317 called with dim = zdim and summed
 * over the zdim elements), minus the posterior's Gaussian entropy (misc.py:292-295), averaged over the valid parts of a shape
 * and over the batch, times kl_weight.  n_class = 4, zdim = 256.
 *   flow / flow_grads: HOST arrays of 4 * flow_depth * 6 device pointers, [part][layer][w0,b0,w1,b1,w2,b2] = net_s_t.{0,2,4}
 *   (as in dfx_latent_weights); flow_grads buffers are overwritten.  part_code (B,256,4), logvar (B,4,256), valid (B,4) 0/1.
 *   loss: one float on the device; log_p_part / entropy (B,4) or NULL.  d_part_code (B,256,4) / d_logvar (B,4,256) or NULL.
 *   workspace: dfx_prior_loss_workspace_bytes(B, flow_depth, flow_hidden), shared by the forward and the backward of a step. */
#define DFX_MAX_FLOW_DEPTH 16
size_t dfx_prior_loss_workspace_bytes(int B, int flow_depth, int flow_hidden);
int dfx_prior_loss_forward(const float *const *flow, int flow_depth, int flow_hidden, void *workspace, size_t workspace_bytes,
                           const float *part_code, const float *logvar, const float *valid, float prior_var, float kl_weight,
                           float *loss, float *log_p_part, float *entropy, int B, dfx_stream_t stream);
int dfx_prior_loss_backward(const float *const *flow, int flow_depth, int flow_hidden, void *workspace, size_t workspace_bytes,
                            const float *valid, float prior_var, float grad_scale, float *const *flow_grads, float *d_part_code,
                            float *d_logvar, int B, dfx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Approximate EMD by auction — replaces the `emd` extension
 * (python/difffacto/metrics/emd/emd_cuda.cu: forward :236-284, backward :286-316; bound in emd_module.py:17-51).
 * xyz1, xyz2 (B,n,3) in [0,1]^3 -> dist (B,n) squared distance of point j to its matched target, assignment (B,n) int32.
 * One persistent workgroup per cloud pair; n <= 8192 (any n, not only multiples of 1024; B unbounded).
 * workspace: dfx_emd_workspace_bytes(B, n) device bytes.  emd_backward: grad_xyz1 (B,n,3) (grad_xyz2 is zero in the reference).
 * ------------------------------------------------------------------------------------------ */
size_t dfx_emd_workspace_bytes(int B, int n);
int dfx_emd_forward_f32(const float *xyz1, const float *xyz2, float *dist, int32_t *assignment, void *workspace, int B, int n,
                        float eps, int iters, dfx_stream_t stream);
int dfx_emd_backward_f32(const float *xyz1, const float *xyz2, const float *grad_dist, const int32_t *assignment,
                         float *grad_xyz1, int B, int n, dfx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode denoiser: forward with saved activations, backward, loss gradient, optimizer.
 * Replaces autograd through TransformerNet.forward / _forward_attn (python/difffacto/models/diffusions/nets/attention.py:385-440;
 * BasicTransformerBlock :296-306, CrossAttention :179-204, FeedForward/GEGLU :50-57,77-94; nn.Dropout optional, see below), the
 * mse_loss of AnchoredDiffusion.training_losses (anchored_diffusion.py:840-847), and Runner.train's
 * clip_grad_norm_ + Adam.step (runner.py:312-316, optimizers.py:4-16).
 *   precision DFX_PREC_F32: exact fp32 throughout (the parity gate).  DFX_PREC_BF16: the large matrix products (every
 *   linear layer over the B*N points: forward, dX, dW) round their operands to bf16 on the way into LDS and accumulate
 *   in fp32 (v_mfma_f32_32x32x16_bf16); activations, gradients, LayerNorm / softmax / GELU and the optimiser stay fp32.
 *   The same value must be passed to the forward and the backward of one step.
 *   dropout_p in [0, 1): nn.Dropout of train() mode behind every to_out (attention.py:177) and GEGLU (:84, time_embed
 *   included); factor(seed, site, element i) = 0 when a 16-bit draw is below round(p 2^16), else 1 / (1 - p); the draw is 16 bits of
 *   Philox4x32-7 keyed by the 64-bit dropout_seed, counter = (i >> 3 = group of EIGHT consecutive elements of the site's row-major
 *   tensor, site, 0xD20F0), element i & 7 = half (i & 1) of word (i & 7) >> 1 (csrc/dfx_dropout.h is the one definition); sites: 2 b =
 *   behind to_out of block b over (B N, 128), 2 b + 1 = behind the GEGLU of block b over (B N, 512), 1000 = time_embed over (B, 1024).
 *   The backward regenerates / re-reads them (pass the same p and seed); torch's own CUDA dropout stream is launch-geometry dependent
 *   and not reproducible, so this is libdfx's contract, checked against torch autograd by replaying the factors
 *   (dfx_debug_dropout_factors, include/dfx_debug.h).
 *   x (B,3,N); t (B,) int32; ctx_code (B,256,4) and ctx_mv (B,6,4) = the two tensors of the reference's ctx list;
 *   anchors, variances (B,N,3) per point (the caller's gather, as at anchored_diffusion.py:261); valid (B,4) 0/1 or
 *   NULL; assignment (B,N) int32; eps (B,3,N).
 *   workspace: dfx_denoiser_train_workspace_bytes(B, N, depth) device bytes, 256-byte aligned; the backward reads what
 *   the forward of the SAME (B, N) call left in it.
 *   Streams: the bf16 path forks small-grid work (context branch, parameter-gradient reductions) onto an internal side stream per
 *   (device, caller stream) pair and joins it before returning — stream-ordered, capturable.  A process gets at most
 *   DFX_MAX_SIDE_STREAMS such pairs (never evicted); callers beyond that run everything on their own stream (same results).
 *   grads: a dfx_denoiser_weights whose pointers name WRITABLE device buffers of the parameters' shapes; every one of
 *   them is overwritten (not accumulated).  d_ctx_code / d_ctx_mv: (B,256,4) / (B,6,4) or NULL.
 *   d_x (B,3,N) / d_variances (B,N,3) or NULL (ABI 4): the gradient at the input x and at the per-point variance feature columns — stage 2 of the
 *   reference differentiates `variance` (it reaches training_losses undetached, anchor_gen.py:1002-1020), through q_sample and through these columns.
 * ------------------------------------------------------------------------------------------ */
size_t dfx_denoiser_train_workspace_bytes(int B, int N, int depth);
int dfx_denoiser_train_forward(const dfx_denoiser_weights *w, void *workspace, size_t workspace_bytes, const float *x,
                               const int32_t *t, const float *ctx_code, const float *ctx_mv, const float *anchors,
                               const float *variances, const float *valid, const int32_t *assignment, float *eps, int B,
                               int N, int precision, float dropout_p, uint64_t dropout_seed, dfx_stream_t stream);
int dfx_denoiser_train_backward(const dfx_denoiser_weights *w, void *workspace, size_t workspace_bytes,
                                const float *d_eps, const dfx_denoiser_weights *grads, float *d_ctx_code, float *d_ctx_mv,
                                float *d_x, float *d_variances, int B, int N, int precision, float dropout_p, uint64_t dropout_seed,
                                dfx_stream_t stream);
/* d loss / d pred of dfx_masked_mse_f32, times grad_scale; workspace2 = the two doubles its forward left behind */
int dfx_masked_mse_backward_f32(const float *target, const float *pred, const float *flags, const double *workspace2,
                                float grad_scale, float *d_pred, int B, int N, dfx_stream_t stream);
/* *sumsq += sum(g^2) (zero it before the first tensor); workspace1024 = 1024 doubles on the device */
int dfx_grad_sumsq_accumulate(const float *g, long long n, double *workspace1024, double *sumsq, dfx_stream_t stream);
/* torch.optim.Adam step on one tensor, gradient scaled by min(1, max_norm / (sqrt(*sumsq) + 1e-6)) if max_norm > 0 */
int dfx_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, const double *sumsq,
                      float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      dfx_stream_t stream);

/* Name + average duration bookkeeping for bench.py: duration in ms of the last dfx_sample_chain /
 * dfx_p_sample / dfx_denoise_eps launch measured with HIP events on `stream` when profiling is enabled. */
void dfx_set_event_timing(int enable);
float dfx_last_kernel_ms(void);
/* Which kernel the last dfx_sample_chain / dfx_p_sample / dfx_denoise_eps (and their _ddim / _t forms) launched on this host
 * thread's process: "k_denoise_pipe<8>", "k_denoise_pipe<4>", "k_denoise_pipe<2>", "k_denoise_coop", "k_denoise_pipe2",
 * "k_denoise_pipe_f32<8|4|2>", "k_denoise<bf16>", "k_denoise<f32>" ("" before the first launch).  The choice is made from the
 * batch shape; every variant of one precision produces the same bits.  Static storage, never NULL. */
const char *dfx_last_kernel_variant(void);

#ifdef __cplusplus
}
#endif
#endif /* DFX_H */
