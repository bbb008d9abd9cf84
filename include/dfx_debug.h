/* libdfx test and tuning hooks — NOT part of the drop-in ABI of include/dfx.h.
 *
 * Nothing here replaces a reference interface: these are the switches tests/ and tools/ use to run two implementations of the
 * same entry point against each other (A/B), to sweep a launch shape, or to replay random factors into the oracle.  They are
 * process-global and not thread-safe; a product binding never includes this header.  Results are valid under
 * every setting (all variants are parity-tested against each other); only the choice of kernel changes.
 */
#ifndef DFX_DEBUG_H
#define DFX_DEBUG_H
#include "dfx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Dropout factors (0 or 1/(1-p)) of n consecutive elements of a site: 2 i = behind to_out of block i over (B N, 128),
 * 2 i + 1 = behind the GEGLU of block i over (B N, 512), 1000 = time_embed over (B, 1024).  n % 4 == 0. */
int dfx_debug_dropout_factors(uint64_t seed, int site, float p, float *out, long long n, dfx_stream_t stream);
/* Debug / A-B switch: 0 routes the bf16 training path through the layer-by-layer kernels instead of the fused ones (default 1;
 * the fused path applies to DFX_PREC_BF16 with dropout_p == 0); 2 = fused, but the attention forward and its input gradient run as
 * kernels of their own instead of inside the feed-forward kernels. */
void dfx_debug_train_fused(int on);
/* The kernel family the calling thread's most recent dfx_denoiser_train_forward took (its backward follows the same record):
 * "fused_bf16", "fused_bf16_dropout" (k_ff_fwd_chain / k_ff<true> / k_ff_wgrad), "layer_bf16(_dropout)" (layer-by-layer kernels, bf16
 * products), "layer_f32(_dropout)" (exact fp32; also what DFX_PREC_BF16 takes below 256 rows), "none" before the first call. */
const char *dfx_debug_last_train_path(void);
/* Debug / A-B switch: 0 keeps every launch of the fused training path on the caller's stream (default 1: the context branch of the forward
 * and the parameter-gradient reductions of the backward run on a per-device side stream, forked from and joined into the caller's stream
 * inside the call).  Same kernels and operands either way: bit-identical results. */
void dfx_debug_train_streams(int on);
/* Debug / A-B switch: 0 = the PointNetV2 training forward computes its BatchNorm batch statistics with two passes over each layer's output
 * (mean, then centred sum of squares); default 1 = from (count, mean, M2) partials the fp32 product kernels leave in their epilogues
 * (>= 8192 rows), equal up to fp32 rounding of the statistics; and BatchNorm over at most 512 rows (the per-part heads: over the B shapes) as one
 * launch per direction instead of seven / four, bit-identical to the multi-launch passes. */
void dfx_debug_bn_fused_stats(int on);
/* Host-side run (no GPU) of that statistics arithmetic: n values in pieces of `chunk`, each piece as (count, mean, M2), merged left to right with
 * the kernels' own merge function; out3 = (count, mean, sum of squared deviations). */
void dfx_debug_stats_merge(const float *values, int n, int chunk, float *out3);
/* Host-side table of the fused training kernels' row addressing inside a 32-point tile (tiled = 1: the tile-major layout between the fused
 * kernels; 0: row-major): float offset of (point, channel) through the B-operand-layout and the accumulator-layout accessors; [32][128] int32 each. */
void dfx_debug_rowmap(int tiled, int *out_b, int *out_a);
/* The few-row fp32 products (latent front end, time-embedding MLP; mfma_linear.h) split their sum over K across four wavefronts when the call has at
 * most 256 output tiles — a different grouping of the same fp32 terms, chosen by the call's row count.  mode 1: split whenever K >= 128 (one grouping for
 * every batch size: sharded runs reproduce a single-process run's latents bit for bit whatever the shard size); 0: never; -1: automatic (default). */
void dfx_debug_lin_split_k(int mode);
/* Debug / A-B switch: 1 keeps the EMD auction's state in global memory for every n (default 0: in LDS when n <= 2688). */
void dfx_debug_emd_state_global(int on);
/* Debug / sweep: workgroup shape of the register-resident FPS kernel (threads in {256, 512, 1024} x points per thread in {2..32},
 * used when threads * points >= N; 0, 0 = automatic). */
void dfx_debug_fps_shape(int threads, int points_per_thread);
/* Test hook for the bf16 product kernels of the training path (csrc/gemm_bf16.h): tn = 0: C (M,N) = A (M,K) B (N,K)^T + bias +
 * resid; tn = 1: C (M,N) = A (K,M)^T B (K,N) and db (M) = column sums of A, workspace >= (K/64 + 1) (M N + M) floats.
 * a_bf16 / b_bf16: the operand is stored as bf16 (lda / ldb in elements).  N % 128 == 0 (and M % 128 == 0 for tn = 1). */
int dfx_debug_gemm_bf16(int tn, const void *A, int lda, int a_bf16, const void *B, int ldb, int b_bf16, const float *bias,
                        const float *resid, float *C, float *db, float *workspace, size_t workspace_floats, int M, int N, int K,
                        dfx_stream_t stream);
/* Debug/A-B switch: force the direct (non LDS-pipelined) kernel for every launch. */
/* Box calibration for bench.py: a bare v_mfma_f32_32x32x16_bf16 stream (one wavefront per SIMD on every CU, pseudo-random operands, nothing
 * else) of `iters` x 16 MFMAs per wavefront; *ms_out = HIP-event duration, *tflops_out = executed MFMA TFLOP/s — what this chip sustains on the
 * matrix pipe at its power cap and clocks (iters = 150000 runs ~50 ms).  Synchronises `stream`. */
int dfx_debug_bare_mfma(int iters, float *ms_out, double *tflops_out, dfx_stream_t stream);
void dfx_debug_force_direct(int on);
/* dfx_denoiser_create's decision about the W1 bias fold of bf16 engines (dfx_denoiser_w1_fold): -1 = from the weights (default: channel 127
 * while its column of W1 diag(gamma3) is ordinary, else the hidden channel with the smallest column over all blocks, exchanged with 127 at
 * pack time), 0 = never (plain pack, direct kernel), 1 = always on channel 127 (round 5's form).  Applies to engines created afterwards. */
void dfx_debug_w1_fold(int mode);
/* The hidden channel whose K slot of the packed W1 carries b1' (127 unless create relabelled the channels; -1 = engine without the fold). */
int dfx_debug_w1_fold_channel(const dfx_denoiser *d);
/* Debug: wavefronts per workgroup of the pipelined chain kernel (8, 4 or 2), or 1 = the co-operative latency kernel (one
 * 32-point tile per workgroup, eight wavefronts on it; for an fp32 denoiser: the direct kernel), or 64 = k_denoise_pipe2 (bf16 only:
 * four wavefronts of two 32-point tiles each; when its 256-point workgroup tiles would pad a shape by more than 3x —
 * ceil(N / 256) * 256 > 3 N — the request falls back SILENTLY to the 8-wavefront kernel, and an fp32 denoiser ignores it), or 16 =
 * k_denoise_coop2 (bf16 only: the co-operative kernel with two 32-point tiles per workgroup; N % 64 != 0 falls back to the one-tile kernel);
 * 0 = chosen from the batch size; any other value is treated as 0.  All variants of one precision are bit-identical.
 * dfx_last_kernel_variant() (dfx.h) names the kernel a launch actually took. */
void dfx_debug_pipe_waves(int nw);
/* Slot-boundary clock stamps of two wavefronts of workgroup 0 (device buffer of 2*capacity uint64; NULL = off).
 * Only effective in a library built with -DDFX_TRACE (tools/experiments/trace_slots.py builds one). */
void dfx_debug_trace(void *device_buf, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* DFX_DEBUG_H */
