// libdfx hot path: the cross-diffusion denoiser eps_theta(x_t, t) fused with the anchored-DDPM
// posterior update, as ONE kernel that can run a single evaluation, a single p_sample step, or the
// whole T-step reverse chain with x_t resident in registers (no HBM round trip per step).
//
// Replaces TransformerNet.forward (python/difffacto/models/diffusions/nets/attention.py:385-440) and
// AnchoredDiffusion.p_mean_variance / p_sample / p_sample_loop_progressive
// (python/difffacto/models/diffusions/anchored_diffusion.py:227-395, 450-484, 528-588).
//
// Mapping (see denoiser_internal.h): one wavefront = 32 points; the residual stream h^T
// (128 channels x 32 points) stays in 64 accumulator VGPRs for the whole chain; every dense
// contraction is a v_mfma_f32_32x32x16_bf16 (or the exact v_mfma_f32_32x32x2_f32) with the WEIGHT
// tile as the A operand and the register-resident activation as the B operand, so consecutive GEMMs
// chain register-to-register.  LayerNorm / softmax / GELU / posterior are fp32 VALU on the same
// registers; the 4-key attention needs no q/k/v at run time (folded into A_s / M_s at prepare time).
#include <cmath>
#include <type_traits>

#include "denoiser_internal.h"

#pragma clang fp contract(fast)

using namespace dfx;

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

enum { MODE_EPS = 0, MODE_PSAMPLE = 1, MODE_CHAIN = 2 };
constexpr int DDIM_MAX_STEPS = 128;

struct KParams {
  DenoiserDev d;
  const float *part;     // shape ctx regions
  const float *cpart;
  const uint4 *as_ms;
  const float *x_in;     // (B,3,N)        eps / p_sample
  const int32_t *seg;    // (B,N)
  const float *noise;    // p_sample: (B,3,N) | chain: (nsteps,B,3,N) | null -> Philox
  const float *xT_noise; // chain: (B,3,N) | null -> Philox
  float *out;            // eps: (B,3,N) | p_sample: (B,3,N) | chain: pred (B,N,3)
  float *traj;           // chain snapshots (n_keep,B,N,3) or null
  float *xstart;         // p_sample: optional pred_xstart (B,3,N)
  unsigned long long seed;
  unsigned long long shape0;   // global index of this launch's shape 0 (Philox counters are keyed by the GLOBAL point id)
  int B, N, t0, nsteps, ret_interval, mode;
  // DDIM branch (anchored_diffusion.py:114-124, :368-377, :480-481): ddim_n > 0 = the executed timestep list
  // (descending, e.g. 'quad' [32,23,16,10,5,2,0,0]) and xt_dir_coeff[t] = sqrt(1 - acp[t] - eta^2 posterior_variance[t])
  const int32_t *t_shape;  // MODE_EPS only: per-shape timestep (B,), training-style evaluation (anchored_diffusion.py:760-853)
  int ddim_n;
  float ddim_eta;
  int ddim_t[DDIM_MAX_STEPS];
  float ddim_xdc[DDIM_MAX_STEPS];
  unsigned long long *trace;  // debug: s_memtime stamps of waves 0 and 4 of workgroup 0 at every slot boundary
  int trace_cap;
};

// ----------------------------------------------------------------------------------------------
// Activation fragments: the B operand view of one 32-channel accumulator tile.
template <int PREC>
struct Act;
template <>
struct Act<DFX_PREC_BF16> {
  v8bf f[2];
  __device__ __forceinline__ void set(const v16f &x) {
    v8f lo = __builtin_shufflevector(x, x, 0, 1, 2, 3, 4, 5, 6, 7);
    v8f hi = __builtin_shufflevector(x, x, 8, 9, 10, 11, 12, 13, 14, 15);
    f[0] = __builtin_convertvector(lo, v8bf);
    f[1] = __builtin_convertvector(hi, v8bf);
  }
};
template <>
struct Act<DFX_PREC_F32> {
  v16f x;
  __device__ __forceinline__ void set(const v16f &v) { x = v; }
};

// acc += Wtile (32 rows x 32 k)  *  act (32 k x 32 points).   `w` already includes the lane offset.
template <int PREC>
__device__ __forceinline__ void mma_tile(v16f &acc, const uint4 *__restrict__ w, const Act<PREC> &a);

template <>
__device__ __forceinline__ void mma_tile<DFX_PREC_BF16>(v16f &acc, const uint4 *__restrict__ w,
                                                        const Act<DFX_PREC_BF16> &a) {
  const uint4 u0 = w[0], u1 = w[64];
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, u0), a.f[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, u1), a.f[1], acc, 0, 0, 0);
}

template <>
__device__ __forceinline__ void mma_tile<DFX_PREC_F32>(v16f &acc, const uint4 *__restrict__ w,
                                                       const Act<DFX_PREC_F32> &a) {
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const v4f u = __builtin_bit_cast(v4f, w[r4 * 64]);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u[e], a.x[4 * r4 + e], acc, 0, 0, 0);
  }
}

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// LayerNorm statistics over the 128 channels of each point (64 in-lane + the partner half-wave).
__device__ __forceinline__ void ln_stats(const v16f (&h)[4], float &mean, float &rstd) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += h[c][r];
  s += xhalf(s);
  mean = s * (1.0f / 128.0f);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dlt = h[c][r] - mean;
      q = fmaf(dlt, dlt, q);
    }
  q += xhalf(q);
  rstd = 1.0f / sqrtf(q * (1.0f / 128.0f) + 1e-5f);  // nn.LayerNorm eps (attention.py:348-349, 286-287)
}

// bf16 path: single pass over the registers (sum and sum of squares on four independent chains each), variance as
// E[x^2] - mean^2 in fp32 — the normalised value is rounded to bf16 right after, so the cancellation error
// (<= 2^-24 E[x^2] / var relative) is far below the operand rounding — and one FMA per element for the normalisation.
__device__ __forceinline__ void ln_stats_fast(const v16f (&h)[4], float &mean, float &rstd) {
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r & 3] += h[c][r];
      q[r & 3] = fmaf(h[c][r], h[c][r], q[r & 3]);
    }
  float st = (s[0] + s[1]) + (s[2] + s[3]), qt = (q[0] + q[1]) + (q[2] + q[3]);
  st += xhalf(st);
  qt += xhalf(qt);
  mean = st * (1.0f / 128.0f);
  const float var = fmaxf(fmaf(-mean, mean, qt * (1.0f / 128.0f)), 0.f);
  rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
}

template <int PREC>
__device__ __forceinline__ void ln_to_act(const v16f (&h)[4], Act<PREC> (&xn)[4]) {
  float mean, rstd;
  if (PREC == DFX_PREC_BF16) {
    ln_stats_fast(h, mean, rstd);
    const float nmr = -mean * rstd;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v16f t;
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = fmaf(h[c][r], rstd, nmr);
      xn[c].set(t);
    }
    return;
  }
  ln_stats(h, mean, rstd);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v16f t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = (h[c][r] - mean) * rstd;
    xn[c].set(t);
  }
}

__device__ __forceinline__ v16f zero16() {
  v16f z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// bf16 path, GEMM1 of the feed-forward: LayerNorm3's output xhat sums to zero over its 128 channels, so channel 127 is redundant
// (xhat_127 = - sum of the others).  denoiser_setup.hip packs W1'' = W1'[., k] - W1'[., 127] for k < 127 and puts b1' into column 127;
// here channel 127's K slot — register 15 of tile 3 in the upper half-wave (rho(15, 1) = 31) — carries the constant 1, so
// [a | g] = W1'' xhat'' comes out of the MFMAs WITH its bias, started from C = 0: the 8 ds_read_b128 of accumulator initialisers per
// record are gone (2.1 % of a record's energy, profiles/r02_energy_budget.txt; measured +2.5 %, profiles/r04_headline_experiments.txt).
// Exact in real arithmetic; in bf16 the weight differences round ~1.4x coarser and the implicit channel carries the rounding noise of
// the other 127: one evaluation deviates from fp32 by rms 8.1e-4 instead of 6.6e-4 (same file).  fp32 kernels keep the plain form.
__device__ __forceinline__ void bias_slot_one(Act<DFX_PREC_BF16> (&xn)[4], int hf) {
  if (hf) xn[3].f[1][7] = (__bf16)1.0f;
}
__device__ __forceinline__ void bias_slot_one(Act<DFX_PREC_F32> (&)[4], int) {}

__device__ __forceinline__ float gelu_erf(float x) {  // F.gelu default (attention.py:57)
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ void add_cvec(v16f (&h)[4], const float *__restrict__ v /* + hf*64 applied */) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float4 t = *reinterpret_cast<const float4 *>(v + c * 16 + r4 * 4);
      h[c][r4 * 4 + 0] += t.x;
      h[c][r4 * 4 + 1] += t.y;
      h[c][r4 * 4 + 2] += t.z;
      h[c][r4 * 4 + 3] += t.w;
    }
}

// ----------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG + Box-Muller (perf runs; parity runs feed explicit noise).
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

__device__ __forceinline__ void philox_normal3(unsigned long long seed, unsigned long long gid, unsigned t,
                                               unsigned stream, float (&z)[3]) {
  const uint4 r = philox4x32_10(make_uint4((unsigned)gid, (unsigned)(gid >> 32), t, stream),
                                make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
  const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u1 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u3 = ((float)(r.w >> 8) + 0.5f) * (1.0f / 16777216.0f);
  // hardware transcendentals (v_log_f32; v_sin_f32 / v_cos_f32 take revolutions): plenty for a noise source and
  // far fewer instructions than the libm versions in the persistent kernel's step epilogue
  const float ra = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));  // -2 ln(u) = -2 ln2 log2(u)
  const float rb = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
  z[0] = ra * __builtin_amdgcn_cosf(u1);
  z[1] = ra * __builtin_amdgcn_sinf(u1);
  z[2] = rb * __builtin_amdgcn_cosf(u3);
}

// ----------------------------------------------------------------------------------------------
// Building blocks shared by the direct (v0) and the LDS-pipelined kernels.  Pointers may be global
// or LDS: after inlining the compiler infers the address space from the call site.

// proj_in (13 -> 128) + pre_norm.  x-columns on the VALU; the 10 per-part-constant inputs
// (anchors | variances | onehot, attention.py:398-408) are pre-folded into cpart[seg].
template <bool FAST = false>
__device__ __forceinline__ void proj_in_prenorm(v16f (&h)[4], const float (&x)[3], const float *cpart,
                                                const float4 *winx, const float2 *pregb) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float4 cp = *reinterpret_cast<const float4 *>(cpart + c * 16 + r4 * 4);
      const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 w = winx[c * 16 + r4 * 4 + e];
        h[c][r4 * 4 + e] = fmaf(w.z, x[2], fmaf(w.y, x[1], fmaf(w.x, x[0], cpv[e])));
      }
    }
  float mean, rstd;
  if (FAST) ln_stats_fast(h, mean, rstd);
  else ln_stats(h, mean, rstd);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 gb = pregb[c * 16 + r];
      h[c][r] = fmaf((h[c][r] - mean) * rstd, gb.x, gb.y);  // affine kept: h is the residual stream
    }
}

__device__ __forceinline__ void load16(v16f &v, const float *src) {
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const float4 t4 = *reinterpret_cast<const float4 *>(src + r4 * 4);
    v[r4 * 4 + 0] = t4.x; v[r4 * 4 + 1] = t4.y; v[r4 * 4 + 2] = t4.z; v[r4 * 4 + 3] = t4.w;
  }
}

// Masked softmax over the 4 keys of each head, in place (registers 4g..4g+3 = keys 0..3 of head 2g+hf; attention.py:195-198).
template <bool FAST_RCP = false>   // bf16 kernels: the hardware reciprocal (1 ulp) instead of the correctly rounded division (~10 VALU instructions, four per block)
__device__ __forceinline__ void softmax4(v16f &sim, unsigned vmask) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float sj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sj[j] = (vmask >> j) & 1u ? sim[4 * g + j] : -3.402823466e38f;  // :195-197
    const float m = fmaxf(fmaxf(sj[0], sj[1]), fmaxf(sj[2], sj[3]));
    float e[4], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      e[j] = __expf(sj[j] - m);
      sum += e[j];
    }
    const float inv = FAST_RCP ? __builtin_amdgcn_rcpf(sum) : 1.0f / sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) sim[4 * g + j] = e[j] * inv;
  }
}

// Cross attention to the 4 part tokens (attention.py:179-204 with q/k/v folded away, see denoiser_setup.hip):
//   sim = A_s LN2(h) + sbias;  P = softmax over the 4 keys of each head (masked);  h += M_s P + c_t
// `rec` = this lane's view of the attention record (tiles 0..3 = A_s, 4..7 = M_s).
template <int PREC>
__device__ __forceinline__ void attention(v16f (&h)[4], const uint4 *rec, const float *sbias, const float *ct,
                                          unsigned vmask) {
  constexpr int TSTRIDE = tile_units(PREC) * 64;
  Act<PREC> xn[4];
  ln_to_act<PREC>(h, xn);
  v16f sim;
  load16(sim, sbias);
#pragma unroll
  for (int c = 0; c < 4; ++c) mma_tile<PREC>(sim, rec + c * TSTRIDE, xn[c]);
  softmax4<PREC == DFX_PREC_BF16>(sim, vmask);
  Act<PREC> pa;
  pa.set(sim);
#pragma unroll
  for (int t = 0; t < 4; ++t) mma_tile<PREC>(h[t], rec + (4 + t) * TSTRIDE, pa);
  add_cvec(h, ct);
}

// GELU for the bf16 path: x * sigmoid(x (c1 + c3 x^2)) with (c1, c3) = (1.60031416, 0.06940179) fitted to the
// exact erf form (max abs error 2.7e-4 over all x, below the bf16 rounding of the hidden activation it
// feeds; the standard tanh constants give 4.7e-4).  The argument is monotone in x, so no clamp is needed:
// +-inf / huge inputs give sigmoid = 1 / 0 exactly.  6 plain VALU + v_exp_f32 + v_rcp_f32 per element.
__device__ __forceinline__ float gelu_sigmoid_arg(float x) {
  const float x2 = x * x;
  return x * fmaf(-0.100125614f, x2, -2.30876530f);  // -log2(e) * x (c1 + c3 x^2)
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(gelu_sigmoid_arg(x));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int PREC>
__device__ __forceinline__ float gelu_for(float x) {
  return PREC == DFX_PREC_BF16 ? gelu_fast(x) : gelu_erf(x);
}

// ---- packed-fp16 GELU (bf16 path; see denoiser_internal.h) ----------------------------------------------------------
// B operand of GEMM2: 16 hidden values of this lane as fp16, same element order as Act.
struct HidAct {
  uint4 f[2];
};
__device__ __forceinline__ h2 pk_f16(float lo, float hi) { return __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
// gelu(g) = g Phi(g),  Phi(g) ~ 1/2 + u R(z),  u = g / 2,  z = min(u^2 - m, L^2 - m),  R of degree 5 (minimax fit of g Phi(g) on
// [-6, 6]: 3.9e-4 in exact arithmetic; evaluated in fp16 with the centred argument z the error of a * gelu(g) is rms 8.5e-4 for
// a, g ~ N(0, 1.5), tools/experiments/fit_gelu_poly.py).  Ten packed instructions per PAIR of values and no transcendental (an exp / rcp
// pair is quarter rate and not packed: the sigmoid form cost 6 packed + 4 transcendental instructions per pair = 2.2x the
// VALU cycles).  g arrives as g / 2 (FF_G_SCALE), a as a / 16 (FF_A_SCALE); fp16 overflow is benign: z saturates at the
// min, the clamp modifier saturates Phi to [0, 1].
__device__ __forceinline__ h2 h2c(float v) { return h2{(_Float16)v, (_Float16)v}; }
// (a, g) -> packed fp16 first: after these sixteen conversions the accumulators are dead and their next initialisers
// (b1 of the next chunk) can be fetched from LDS underneath the GELU arithmetic instead of after it.
__device__ __forceinline__ void gelu16_f16_cvt(const v16f &a, const v16f &g, h2 (&aa)[8], h2 (&gg)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) gg[i] = pk_f16(g[2 * i], g[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 8; ++i) aa[i] = pk_f16(a[2 * i], a[2 * i + 1]);
}
__device__ __forceinline__ void gelu16_f16_math(const h2 (&aa)[8], const h2 (&gg)[8], HidAct &hid) {
  // Stage by stage over the eight packed pairs: eight independent instructions between any two dependent ones (a
  // dependent v_pk_* costs a wait state on gfx950; written pair by pair hipcc serialises the whole chain through three
  // registers: 88 dependent instructions + 53 s_nop, 1176 cycles per slot against 984 like this).
  h2 y[8], z[8], r[8];
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
  hid.f[0] = make_uint4(__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1]), __builtin_bit_cast(unsigned, y[2]),
                        __builtin_bit_cast(unsigned, y[3]));
  hid.f[1] = make_uint4(__builtin_bit_cast(unsigned, y[4]), __builtin_bit_cast(unsigned, y[5]), __builtin_bit_cast(unsigned, y[6]),
                        __builtin_bit_cast(unsigned, y[7]));
}
// h_tile += W2 fragment (A, from LDS / L2) x hid fragment (B)
__device__ __forceinline__ v16f mma_hid(const uint4 &w, const uint4 &hid, v16f acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, w), __builtin_bit_cast(v8h, hid), acc, 0, 0, 0);
}

// One hidden chunk (32 units) of the GEGLU feed-forward (attention.py:50-57,77-94):
//   a = W1a xn + b1a; g = W1g xn + b1g; hid = a * gelu(g); h += W2[:, chunk] hid.  Hidden stays in registers.
template <int PREC>
__device__ __forceinline__ void ff_chunk(v16f (&h)[4], const Act<PREC> (&xn)[4], const uint4 *ck, const uint4 *ck2,
                                         const float *b1, bool fold) {
  constexpr int TSTRIDE = tile_units(PREC) * 64;
  v16f a, g;
  if (PREC == DFX_PREC_BF16 && fold) {
    a = zero16(), g = zero16();   // b1' rides on the constant-one K slot (bias_slot_one)
  } else {                        // fp32, and bf16 engines whose weights ruled the fold out (DenoiserDev::w1_fold = 0): plain W1', b1' initialisers
    load16(a, b1);
    load16(g, b1 + 32);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    mma_tile<PREC>(a, ck + (0 + c) * TSTRIDE, xn[c]);
    mma_tile<PREC>(g, ck + (4 + c) * TSTRIDE, xn[c]);
  }
  v16f hid;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (PREC == DFX_PREC_BF16)   // g arrives pre-scaled, hid carries the same factor (denoiser_internal.h)
      hid[r] = a[r] * gelu_for<PREC>(g[r] * (1.0f / FF_G_SCALE)) * FF_G_SCALE;
    else
      hid[r] = a[r] * gelu_for<PREC>(g[r]);
  }
  if (PREC == DFX_PREC_BF16) {   // `a` arrives pre-scaled, W2 is fp16 (denoiser_internal.h)
    HidAct hf16;
    h2 y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = pk_f16(hid[2 * i], hid[2 * i + 1]);
    hf16.f[0] = make_uint4(__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1]), __builtin_bit_cast(unsigned, y[2]),
                           __builtin_bit_cast(unsigned, y[3]));
    hf16.f[1] = make_uint4(__builtin_bit_cast(unsigned, y[4]), __builtin_bit_cast(unsigned, y[5]), __builtin_bit_cast(unsigned, y[6]),
                           __builtin_bit_cast(unsigned, y[7]));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint4 *w = ck2 + (8 + t) * TSTRIDE;
      h[t] = mma_hid(w[0], hf16.f[0], h[t]);
      h[t] = mma_hid(w[64], hf16.f[1], h[t]);
    }
    return;
  }
  Act<PREC> ha;
  ha.set(hid);
#pragma unroll
  for (int t = 0; t < 4; ++t) mma_tile<PREC>(h[t], ck2 + (8 + t) * TSTRIDE, ha);
}

// ---- building blocks of the LDS-pipelined bf16 kernel ---------------------------------------------------
// Every wavefront's stream is cut into pure MFMA bursts ("M slots") and pure VALU bursts ("V slots"), and the two
// wavefronts of a SIMD run them in anti-phase (k_denoise_pipe), so that the matrix pipe of a SIMD always has one
// wavefront feeding it while the other one does the fp32 work.  Measured rules behind the structure
// (tools/ubench/agpr_overlap.hip, stage_mix.hip, lds_bw.hip):
//   * a VALU instruction next to MFMAs costs ~3.1 cycles of SIMD issue (transcendentals ~10), every MFMA takes
//     ~18 cycles of VALU issue away: T(SIMD) ~ max(32.6 nMFMA, 3.1 nVALU + 10 nTRANS + 18 nMFMA);
//   * packed fp32 VALU (v_pk_mul/fma/add_f32) serialises against the matrix pipe: never next to MFMAs
//     (the library is built with -fno-slp-vectorize, +10 %);
//   * LDS delivers ~240 B/clk/CU with eight wavefronts reading; one wavefront sustains ~32 B/clk (latency bound).
__device__ __forceinline__ v8bf as_bf(const uint4 &u) { return __builtin_bit_cast(v8bf, u); }

constexpr int MFMA_PRIO = 0, VALU_PRIO = 3;   // s_setprio inside M / V slots

// M slot of FF record j: h += W2[:, chunk j-1] hid (S3: 8 MFMAs) then a,g = b1[j] + W1[chunk j] xn (S1: 16 MFMAs).
// The 24 A-fragment units are fetched in batches of eight ds_read_b128 running one batch ahead of the MFMAs.
// Slot-boundary clock stamps for tools/experiments/trace_slots.py; compiled in only with -DDFX_TRACE (the bookkeeping costs
// ~10 scalar instructions per stamp site even when switched off at run time).
struct Tracer {
#ifdef DFX_TRACE
  unsigned long long *buf;  // nullptr = off
  int cap, count;
  __device__ __forceinline__ void stamp(int tag) {
    if (buf && count < cap) {
      if ((threadIdx.x & 63) == 0) buf[count] = ((unsigned long long)tag << 56) | (__builtin_readcyclecounter() & 0xffffffffffffffull);
      ++count;
    }
  }
#else
  __device__ __forceinline__ void stamp(int) {}
#endif
};

// Fragment indices (uint4 units relative to this lane's record pointer).  FF record: W2 tile ct = i&3, unit q = i>>2 (a tile's
// two MFMAs are four MFMAs apart: a dependent accumulate right behind its predecessor waits for it when the wave is alone on its SIMD);
// W1 unit order (c, q, part) -> tile part*4 + c, unit q (half 0: c = 0,1; half 1: c = 2,3).  Attention record: A_s tiles
// 0..3, M_s tiles 4..7.
__device__ __forceinline__ constexpr int w2_frag(int i) { return (8 + (i & 3)) * 128 + (i >> 2) * 64; }   // tile i & 3, unit i >> 2: four accumulators in rotation
__device__ __forceinline__ constexpr int w1_frag(int i, int half) { return ((i & 1) * 4 + 2 * half + (i >> 2)) * 128 + ((i >> 1) & 1) * 64; }
__device__ __forceinline__ constexpr int as_frag(int i) { return (i >> 1) * 128 + (i & 1) * 64; }
__device__ __forceinline__ constexpr int ms_frag(int i) { return (4 + (i & 3)) * 128 + (i >> 2) * 64; }

// Tail prefetch: every M slot finds the A fragments of its first eight MFMAs in registers (P): they are read at the
// TAIL of the previous M slot of the same wavefront — after its last MFMA has been issued, while the matrix pipe drains
// and the wave has nothing else to issue — so the burst starts without the LDS round trip (~300 cycles per slot) and the
// VALU-bound V slot in between carries no extra LDS instructions.  Needs the next record complete in LDS one management
// barrier earlier: the ring runs three records ahead in five slots.
// Interleaved M slots: every MFMA is followed by ONE LDS read that refills the fragment register it has just consumed with
// the fragment needed eight MFMAs later (in place: the MFMA reads its A operand at issue), so the wavefront's read
// instructions issue while the matrix pipe works instead of in bursts during which it drains.  The last eight refills are
// the tail prefetch for the next M slot.  `issue.at(i, n)` issues the wave's ring-DMA pieces between the MFMAs.

// MFMA order of a full FF record (GEMM2 of chunk j-1 and GEMM1 of chunk j are independent): eight groups of
//     GEMM2 MFMA k (accumulator h[k & 3])  |  GEMM1 MFMA 2k (a)  |  GEMM1 MFMA 2k+1 (g)
// so that an accumulator is touched every third MFMA at the earliest (the r01 order — 8 x GEMM2, then 16 x GEMM1 with its two
// accumulators alternating — made every GEMM1 MFMA depend on the one two before it).  Measured: no difference (B = 1: 87.7 vs
// 87.0 ms per chain, B = 128: within box noise) — a lone wavefront's M slot takes 41 cycles per MFMA in either order (slot trace:
// 984 cycles), so the accumulate latency is not what stretches it; the order stays because it costs nothing and one ring loop
// is simpler than three.
// m-th MFMA of the record -> its A fragment (uint4 index relative to the lane's record pointer)
__device__ __forceinline__ constexpr int ff_frag(int m) {
  const int k = m / 3, r = m % 3;
  if (r == 0) return w2_frag(k);
  const int e = 2 * k + r - 1;             // GEMM1 MFMA 0..15: half e >> 3, index e & 7 within the half
  return w1_frag(e & 7, e >> 3);
}
// what the next M slot of this wavefront starts with (tail prefetch): a full record (mixed order), the block's last record
// (GEMM2 only) or the next block's attention record (A_s)
enum { NEXT_FULL = 0, NEXT_LAST = 1, NEXT_AS = 2 };
template <int NEXT>
__device__ __forceinline__ constexpr int next_frag(int i) { return NEXT == NEXT_FULL ? ff_frag(i) : NEXT == NEXT_LAST ? w2_frag(i) : as_frag(i); }

template <bool S3, bool S1, int NEXT, class Issue>
__device__ __forceinline__ void ff_m(v16f (&h)[4], const Act<DFX_PREC_BF16> (&xn)[4], v16f &a, v16f &g,
                                     const HidAct &hid, const uint4 *ck, uint4 (&P)[8], const uint4 *ck_next, Tracer &tr,
                                     Issue &issue) {
  __builtin_amdgcn_sched_barrier(0);
  tr.stamp(4);
  if (MFMA_PRIO) __builtin_amdgcn_s_setprio(MFMA_PRIO);
  auto gemm1 = [&](int e, const uint4 &w) {   // GEMM1 MFMA e = 0..15
    const int i = e & 7, half = e >> 3;
    v16f &acc = (i & 1) ? g : a;
    if (e < 2) {   // first MFMA of a / g: C = 0 (inline constant) — b1' rides on the constant-one K slot (bias_slot_one), no initialiser reads
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w), xn[2 * half + (i >> 2)].f[(i >> 1) & 1], zero16(), 0, 0, 0);
      return;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w), xn[2 * half + (i >> 2)].f[(i >> 1) & 1], acc, 0, 0, 0);
  };
  if (S3 && S1) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {   // P is a ring of eight fragment registers: MFMA m takes P[m & 7], refilled in place with fragment m + 8
      const int k = m / 3, r = m % 3;
      if (r == 0) h[k & 3] = mma_hid(P[m & 7], hid.f[k >> 2], h[k & 3]);
      else gemm1(2 * k + r - 1, P[m & 7]);
      issue.at(m, 24);
      P[m & 7] = m + 8 < 24 ? ck[ff_frag(m + 8)] : ck_next[next_frag<NEXT>(m + 8 - 24)];
    }
  } else if (S1) {                  // first FF record of a block: GEMM1 only; P holds its first eight fragments
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      gemm1(m, P[m & 7]);
      issue.at(m, 16);
      P[m & 7] = m + 8 < 16 ? ck[w1_frag((m + 8) & 7, (m + 8) >> 3)] : ck_next[next_frag<NEXT>(m + 8 - 16)];
    }
  } else {                          // last FF record of a block: GEMM2 only
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      h[i & 3] = mma_hid(P[i], hid.f[i >> 2], h[i & 3]);
      issue.at(i, 8);
      P[i] = ck_next[next_frag<NEXT>(i)];
    }
  }
  // keep the program order MFMA, read, MFMA, read, ...
#pragma unroll
  for (int i = 0; i < (S3 && S1 ? 24 : S1 ? 16 : 8); ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
}

// V slot of the feed-forward: hid = a * gelu(g) as the fp16 B operand of GEMM2.
__device__ __forceinline__ void ff_v(v16f &a, v16f &g, HidAct &hid, Tracer &tr) {
  if (VALU_PRIO) __builtin_amdgcn_s_setprio(VALU_PRIO);
  h2 aa[8], gg[8];
  gelu16_f16_cvt(a, g, aa, gg);
  gelu16_f16_math(aa, gg, hid);
  // Pin the result here: without a use in this slot LLVM sinks the whole GELU past the record barrier into the
  // consumer's M slot, and the two groups' VALU bursts collide instead of running in anti-phase.
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u w = __builtin_bit_cast(v4u, hid.f[q]);
    asm volatile("" : "+v"(w));
    hid.f[q] = __builtin_bit_cast(uint4, w);
  }
  tr.stamp(8);
  if (VALU_PRIO) __builtin_amdgcn_s_setprio(0);
}

// attention M slots: sim = sbias + A_s xn (8 MFMAs);  h += M_s P (8 MFMAs)
template <class Issue>
__device__ __forceinline__ void attn_m0(v16f &sim, const Act<DFX_PREC_BF16> (&xn)[4], const uint4 *rec, const float *sbias,
                                        uint4 (&P)[8], bool have_p, Issue &issue) {
  if (!have_p) {
#pragma unroll
    for (int i = 0; i < 8; ++i) P[i] = rec[as_frag(i)];
  }
  load16(sim, sbias);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sim = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(P[i]), xn[i >> 1].f[i & 1], sim, 0, 0, 0);
    issue.at(i, 8);
    P[i] = rec[ms_frag(i)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void attn_m1(v16f (&h)[4], const Act<DFX_PREC_BF16> &pa, const uint4 *rec, uint4 (&P)[8],
                                        const uint4 *ck_next) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(P[i]), pa.f[i >> 2], h[i & 3], 0, 0, 0);
    P[i] = ck_next[w1_frag(i, 0)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// attention V slot: masked softmax over the 4 keys of each head (registers 4g..4g+3 = keys of head 2g+hf)
__device__ __forceinline__ void attn_softmax(v16f &sim, Act<DFX_PREC_BF16> &pa, unsigned vmask) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float sj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sj[j] = (vmask >> j) & 1u ? sim[4 * g + j] : -3.402823466e38f;  // attention.py:195-197
    const float m = fmaxf(fmaxf(sj[0], sj[1]), fmaxf(sj[2], sj[3]));
    float e[4], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      e[j] = __expf(sj[j] - m);
      sum += e[j];
    }
    const float inv = __builtin_amdgcn_rcpf(sum);   // (as softmax4<true>: every bf16 kernel takes the same reciprocal; +0.25 % in a same-box A/B)
#pragma unroll
    for (int j = 0; j < 4; ++j) sim[4 * g + j] = e[j] * inv;
  }
  pa.set(sim);
}

// post_norm (affine folded into W_out) + proj_out (128 -> 3) on the VALU.
template <bool FAST = false>
__device__ __forceinline__ void post_eps(const v16f (&h)[4], const float4 *wout, const float (&bout)[4],
                                         float (&eps)[3]) {
  float mean, rstd;
  if (FAST) ln_stats_fast(h, mean, rstd);
  else ln_stats(h, mean, rstd);
  const float nmr = -mean * rstd;
  float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 w = wout[c * 16 + r];
      const float v = FAST ? fmaf(h[c][r], rstd, nmr) : (h[c][r] - mean) * rstd;
      e0 = fmaf(w.x, v, e0);
      e1 = fmaf(w.y, v, e1);
      e2 = fmaf(w.z, v, e2);
    }
  eps[0] = e0 + xhalf(e0) + bout[0];
  eps[1] = e1 + xhalf(e1) + bout[1];
  eps[2] = e2 + xhalf(e2) + bout[2];
}

// Timestep of the step-th executed step (wave-uniform): T-1, T-2, ... for DDPM, the list for DDIM.
__device__ __forceinline__ int step_t(const KParams &p, int step, int s) {
  if (p.t_shape) return __builtin_amdgcn_readfirstlane(p.t_shape[s]);   // wave-uniform (one shape per wave)
  if (p.ddim_n > 0) return p.ddim_t[step < p.ddim_n ? step : p.ddim_n - 1];
  return p.t0 - step;
}

// Per-point state that lives in registers for the whole chain.
struct PointState {
  float x[3], anc[3], var[3], L[3];
  int s, n, sg;
  unsigned long long gid;
  bool live = true;   // false: a wavefront past the end of its shape's last (partial) 256-point tile recomputes that shape's last 32 points and stores nothing
};

__device__ __forceinline__ void point_init(const KParams &p, PointState &ps, int s, int n, unsigned long long gid,
                                           unsigned &vmask) {
  ps.s = s; ps.n = n; ps.gid = gid;
  const float *part = p.part + (size_t)s * 32;
  ps.sg = p.seg[(size_t)s * p.N + n];  // replaces gather_operation (part_encoders.py:417-428)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ps.anc[i] = part[i * 4 + ps.sg];
    ps.var[i] = part[12 + i * 4 + ps.sg];
    ps.L[i] = sqrtf(ps.var[i]);  // anchored_diffusion.py:306 / :562
  }
  vmask = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) vmask |= (part[24 + j] != 0.f ? 1u : 0u) << j;  // mask.to(bool), attention.py:196
  vmask = __builtin_amdgcn_readfirstlane(vmask);
  const int hf = (threadIdx.x >> 5) & 1;
  if (p.mode == MODE_CHAIN) {
    float z[3];
    if (p.xT_noise) {
#pragma unroll
      for (int i = 0; i < 3; ++i) z[i] = p.xT_noise[((size_t)s * 3 + i) * p.N + n];
    } else {
      philox_normal3(p.seed, gid, (unsigned)p.d.T, 1u, z);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) ps.x[i] = ps.L[i] * z[i] + ps.anc[i];  // anchored_diffusion.py:563-564
    if (p.traj && p.d.T % p.ret_interval == 0 && hf == 0 && ps.live) {
      float *o = p.traj + ((size_t)s * p.N + n) * 3;  // snapshot 0 <-> t = T
      o[0] = ps.x[0]; o[1] = ps.x[1]; o[2] = ps.x[2];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) ps.x[i] = p.x_in[((size_t)s * 3 + i) * p.N + n];
  }
}

// What happens to eps after the network: store it (MODE_EPS), or the anchored posterior update
// (anchored_diffusion.py:306-319,365-367,378-380,401-409,175-213,476-483; reference op order, no contraction)
// plus the bookkeeping of AnchorDiffAE.decode (anchor_gen.py:160-167).  Returns true when the kernel is done.
__device__ __forceinline__ bool step_epilogue(const KParams &p, PointState &ps, const float (&eps)[3], int step, int t,
                                              const float *z_ready = nullptr, const float *tab_row = nullptr) {
  const int hf = ps.live ? (threadIdx.x >> 5) & 1 : 1;   // stores are done by half-wave 0 of live wavefronts
  const int s = ps.s, n = ps.n;
  if (p.mode == MODE_EPS) {
    if (hf == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) p.out[((size_t)s * 3 + i) * p.N + n] = eps[i];
    }
    return true;
  }
  float z[3];
  if (z_ready) {   // (co-operative kernel: an idle wavefront has drawn the noise and fetched the table row already)
#pragma unroll
    for (int i = 0; i < 3; ++i) z[i] = z_ready[i];
  } else if (p.noise) {
#pragma unroll
    for (int i = 0; i < 3; ++i) z[i] = p.noise[(((size_t)step * p.B + s) * 3 + i) * p.N + n];
  } else {
    philox_normal3(p.seed, ps.gid, (unsigned)t, 0u, z);
  }
  {
#pragma clang fp contract(off)
    const float *tb = tab_row ? tab_row : p.d.tab + (size_t)t * 8;
    const float sra = tb[0], srm1 = tb[1], c1 = tb[2], c2 = tb[3], c3 = tb[4], pv = tb[5];
    const float nz = t != 0 ? 1.0f : 0.0f;
    const bool ddim = p.ddim_n > 0;
    const float sap = sqrtf(tb[6]);   // torch.sqrt of the fp32 alphas_cumprod_prev[t] (:481)
    const float xdc = ddim ? p.ddim_xdc[step < p.ddim_n ? step : p.ddim_n - 1] : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float x0 = sra * (ps.x[i] - ps.anc[i]) + ps.anc[i] - srm1 * ps.L[i] * eps[i];
      if (p.xstart && hf == 0) p.xstart[((size_t)s * 3 + i) * p.N + n] = x0;
      const float mv = pv * ps.var[i];
      if (ddim) {   // (x0 - a) sqrt(acp_prev) + a + L xt_dir_coeff eps + eta 1[t != 0] sqrt(var) z
        const float xt_dir = ps.L[i] * xdc * eps[i];
        ps.x[i] = (x0 - ps.anc[i]) * sap + ps.anc[i] + xt_dir + p.ddim_eta * nz * sqrtf(mv) * z[i];
      } else {
        const float mu = c1 * x0 + c2 * ps.x[i] + c3 * ps.anc[i];
        ps.x[i] = mu + nz * sqrtf(mv) * z[i];
      }
    }
  }
  if (p.mode == MODE_PSAMPLE) {
    if (hf == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) p.out[((size_t)s * 3 + i) * p.N + n] = ps.x[i];
    }
    return true;
  }
  if (hf == 0) {
    if (t == 0) {
      float *o = p.out + ((size_t)s * p.N + n) * 3;
      o[0] = ps.x[0]; o[1] = ps.x[1]; o[2] = ps.x[2];
    } else if (p.traj && t % p.ret_interval == 0) {
      const int k = p.d.T / p.ret_interval - t / p.ret_interval;
      float *o = p.traj + (((size_t)k * p.B + s) * p.N + n) * 3;
      o[0] = ps.x[0]; o[1] = ps.x[1]; o[2] = ps.x[2];
    }
  }
  return false;
}

// ----------------------------------------------------------------------------------------------
// v0 "direct" kernel: every operand comes straight from global memory (L2-resident).  Any N % 32 == 0,
// both precisions.  It is the general fallback and the exact-fp32 parity path.
template <int PREC, int NW>
__global__ void __launch_bounds__(NW * 64) k_denoise(const KParams p) {
  constexpr int TSTRIDE = tile_units(PREC) * 64;  // uint4 elements per 32x32 weight tile
  constexpr int AREC = asms_bytes(PREC) / 16;     // uint4 per attention record
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  const long long g0 = ((long long)blockIdx.x * NW + wave) * 32;
  if (g0 >= (long long)p.B * p.N) return;
  const int s = __builtin_amdgcn_readfirstlane((int)(g0 / p.N));
  const int n = (int)(g0 - (long long)s * p.N) + pj;
  const int depth = p.d.depth;
  PointState ps;
  unsigned vmask;
  point_init(p, ps, s, n, ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)n, vmask);
  const float *cpart = p.cpart + ((size_t)s * NCLS + ps.sg) * INNER + hf * 64;
  const uint4 *asms_s = p.as_ms + (size_t)s * depth * AREC + lane;

  for (int step = 0; step < p.nsteps; ++step) {
    const int t = step_t(p, step, s);
    v16f h[4];
    proj_in_prenorm(h, ps.x, cpart, p.d.win_x + hf * 64, p.d.pre_gb + hf * 64);
    for (int b = 0; b < depth; ++b) {
      const BlockPack &bp = p.d.blk[b];
      const uint4 *rec = asms_s + (size_t)b * AREC;
      attention<PREC>(h, rec, reinterpret_cast<const float *>(rec - lane + 8 * TSTRIDE) + hf * 16,
                      bp.ct + (size_t)t * CT_ROW + hf * 64, vmask);
      Act<PREC> xn[4];
      ln_to_act<PREC>(h, xn);
      const bool fold = p.d.w1_fold != 0;   // (wave-uniform)
      if (fold) bias_slot_one(xn, hf);
#pragma unroll 1
      for (int u = 0; u < FF_CHUNKS; ++u)
        ff_chunk<PREC>(h, xn, bp.chunks + (size_t)u * CHUNK_TILES * TSTRIDE + lane,
                       bp.chunks + (size_t)(u + FF_SKEW) * CHUNK_TILES * TSTRIDE + lane, bp.bconst + u * 64 + hf * 16, fold);
      add_cvec(h, bp.bconst + BCONST_B2_OFF + hf * 64);
    }
    float eps[3];
    post_eps(h, p.d.wout + hf * 64, p.d.bout, eps);
    if (step_epilogue(p, ps, eps, step, t)) return;
  }
}

// ----------------------------------------------------------------------------------------------
// LDS-pipelined kernel (bf16): one workgroup = 8 wavefronts = 256 points of ONE shape (a partial last tile idles wavefronts).
//
// Weight streaming.  All weights stream L2 -> LDS through a 5-slot ring of 24 KiB records filled by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wavefront-instruction, 3 per wave per record), three records ahead of the
// compute, with a counted s_waitcnt vmcnt (never 0 in steady state).  Record sequence per transformer block:
// [attention record] then 17 x [FF record j = W1 of chunk j | W2 of chunk j-1].  The DMA is issued from inline asm so
// that hipcc does not serialise the ring behind vmcnt(0) waits (it cannot prove that a ds_read does not alias an
// in-flight LDS-DMA); the data hazards are handled by the slot protocol below.
//
// Slots.  Each wavefront alternates pure-VALU slots (V) and pure-MFMA slots (M).  Group B (waves 4-7) runs one
// slot behind group A (waves 0-3): each record's management barrier (below) is in front of A's M slot but in
// front of B's preceding V slot, and waves w / w+4 share a SIMD, so after every barrier one wavefront of each
// SIMD starts an MFMA burst while its partner starts a VALU burst:
//     block:  V0 (b2 of the previous block | step boundary | LN2)  M0 (A_s)  V1 (softmax)  M1 (M_s)  V2 (+c_t, LN3)
//             M(F0: GEMM1 0)  V (GELU 0)  M(F1: GEMM2 0 + GEMM1 1)  V (GELU 1) ... M(F16: GEMM2 15)
// One barrier per record per wavefront; both groups execute the same number of barriers.  Every M slot ends by reading
// the first eight A fragments of the wave's NEXT M slot (tail prefetch, ff_m / attn_m0 / attn_m1).
//
// Ring protocol.  The barrier in front of the M slot that first reads record r (for group A; for B it is the
// barrier in front of the preceding V slot — the SAME global barrier) is r's management barrier:
//     before it  every wave waits for its own DMA pieces of r+1 (vmcnt(its pieces per record): r+2 may still be in flight)
//     after it   every wave issues its pieces of record r+3 into slot (r+3)%5
//   RAW: the issuing waves' vmcnt + the barrier precede every read of r and every tail-prefetch read of r+1.
//   WAR: slot (r+3)%5 held record r-2, whose last reader (B; for an attention record B's V2, which ends before the
//        management barrier of r) finished at least one barrier earlier.
// Wavefronts per workgroup: 8 (256 points) when that fills the chip; 4 or 2 for small batches, so that a single shape still
// spreads over 16 / 32 CUs (same per-wave instruction stream, bit-identical results; the ring then takes 6 / 12 pieces per wave).
constexpr int PIPE_NW = 8;
constexpr int SLOT_BYTES = 24 * 1024;
constexpr int NSLOT = 5;   // records in flight ahead of the compute: NSLOT - 2

// Ring DMA: 24 pieces of 1 KiB per record, 24 / NW per wavefront.
constexpr int RING_PIECES = SLOT_BYTES / 1024;
constexpr int RECORDS_PER_BLOCK = 1 + FF_STAGES;
// LDS map (bytes)
constexpr int L_RING = 0;
constexpr int L_BCONST = L_RING + NSLOT * SLOT_BYTES;  // 2 x block-constant record (b1', b2)
constexpr int L_WINX = L_BCONST + 2 * BCONST_BYTES;    // float4[128]
constexpr int L_PREGB = L_WINX + 2048;                 // float2[128]
constexpr int L_WOUT = L_PREGB + 1024;                 // float4[128]
constexpr int L_CPART = L_WOUT + 2048;                 // float[4][128]
constexpr int L_DUMMY = L_CPART + 2048;                // sink for padding DMAs (one wave's pieces)
constexpr int PSTATE_FIELDS = 13;                      // x[3] anc[3] var[3] L[3] seg
template <int NW>
struct PipeCfg {
  static_assert(NW == 8 || NW == 4 || NW == 2, "wavefronts per workgroup");
  static constexpr int CALLS = RING_PIECES / NW;                       // ring pieces per wave and record
  static constexpr int PTS = NW * 32;                                  // points per workgroup
  static constexpr int L_PSTATE = L_DUMMY + CALLS * 1024;              // per-point chain state parked between steps: float[13][PTS]
  static constexpr int L_TOTAL = L_PSTATE + PSTATE_FIELDS * PTS * 4;
  static_assert(L_TOTAL <= 160 * 1024, "LDS budget");
};
static_assert(asms_bytes(DFX_PREC_BF16) + 1024 <= SLOT_BYTES && chunk_bytes(DFX_PREC_BF16) == SLOT_BYTES, "slot layout");

extern __shared__ __attribute__((aligned(1024))) unsigned char pipe_smem[];

// one 1 KiB LDS-DMA: lane l copies 16 B from gbase + voff(l) to LDS byte address lds_addr + 16 l
__device__ __forceinline__ void dma1k(const void *gbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gbase), "s"(lds_addr)
               : "memory");  // m0 is reserved: hipcc re-materialises it before each of its own uses
}
__device__ __forceinline__ const char *pin_ptr(const char *q) {   // wave-uniform pointer -> scalar registers
  const unsigned long long g = (unsigned long long)(uintptr_t)q;
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(g >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)g);
  return (const char *)(uintptr_t)(((unsigned long long)hi << 32) | lo);   // (the builtin returns int: no sign extension of `lo`)
}
// the same with the (wave-uniform by construction) operands pinned to scalar registers: behind a chain of uniform selects hipcc
// may hold them in VGPRs, which the "s" constraints reject
__device__ __forceinline__ void dma1k_pinned(const void *gbase, unsigned voff, unsigned lds_addr) {
  dma1k(pin_ptr(reinterpret_cast<const char *>(gbase)), voff, (unsigned)__builtin_amdgcn_readfirstlane(lds_addr));
}

// N consecutive 1 KiB pieces: the instruction's immediate offset applies to BOTH the global and the LDS address
// (checked on gfx950: tools/ubench/dma_offset.hip), so one M0 / one SGPR base serve all of them
#define DFX_DMA_HEAD "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
#define DFX_DMA_MORE(off) "\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #off
#define DFX_DMA_ASM(str) asm volatile(str ::"v"(voff), "s"(gbase), "s"(lds_addr) : "memory")
template <int N>
__device__ __forceinline__ void dma_nk(const void *gbase, unsigned voff, unsigned lds_addr) {   // ONE asm block per four pieces: M0 stays ours
  static_assert(N >= 0 && N <= 12, "pieces per wave");
  if constexpr (N > 4) {   // the immediate offset has 12 bits
    dma_nk<4>(gbase, voff, lds_addr);
    dma_nk<N - 4>(reinterpret_cast<const char *>(gbase) + 4096, voff, lds_addr + 4096);
  } else {
    if (N == 1) DFX_DMA_ASM(DFX_DMA_HEAD);
    if (N == 2) DFX_DMA_ASM(DFX_DMA_HEAD DFX_DMA_MORE(1024));
    if (N == 3) DFX_DMA_ASM(DFX_DMA_HEAD DFX_DMA_MORE(1024) DFX_DMA_MORE(2048));
    if (N == 4) DFX_DMA_ASM(DFX_DMA_HEAD DFX_DMA_MORE(1024) DFX_DMA_MORE(2048) DFX_DMA_MORE(3072));
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Block b's pack = block 0's + b * stride: plain scalar arithmetic instead of a look-up in the kernel-argument table
// (an s_load whose s_waitcnt lgkmcnt(0) also drains every LDS read in flight, once per record).
__device__ __forceinline__ BlockPack block_pack(const KParams &p, int b) {
  const long long off = (long long)b * p.d.blk_stride;
  return BlockPack{reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(p.d.blk[0].chunks) + off),
                   reinterpret_cast<const float *>(reinterpret_cast<const char *>(p.d.blk[0].bconst) + off),
                   reinterpret_cast<const float *>(reinterpret_cast<const char *>(p.d.blk[0].ct) + off)};
}

// Next record to fetch (all wave-uniform).
struct DmaState {
  int step, b, k, seq, slot;  // k: 0 = attention record, 1..17 = FF record k-1; seq = running block number
  const char *ff_src;         // this wave's window of the next FF record
};

// Issue this wave's pieces (CALLS_A / CALLS_B of them, starting at piece q0) of the next record and advance the state.
template <int NC>
__device__ __forceinline__ void issue_pieces(const KParams &p, DmaState &st, int q0, unsigned voff, unsigned lds0, int s) {
  const unsigned ring = lds0 + L_RING + st.slot * SLOT_BYTES;
  if (NC == 0) {
  } else if (st.step >= p.nsteps) {  // past the end: padding pieces keep the vmcnt bookkeeping uniform
    dma_nk<NC>(p.d.blk[0].chunks, voff, lds0 + L_DUMMY);
  } else if (st.k > 0) {  // FF record: 24 contiguous KiB (the common case: keep it lean)
    dma_nk<NC>(st.ff_src, voff, ring + q0 * 1024);
    st.ff_src += SLOT_BYTES;
  } else {  // attention record: 17 KiB shape record | 5 KiB block constants | 1 KiB c_t row | 1 padding piece
    const BlockPack bp = block_pack(p, st.b);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int q = q0 + j;
      const char *src = reinterpret_cast<const char *>(bp.bconst);
      unsigned dst = lds0 + L_DUMMY;
      if (q < 17) {
        src = reinterpret_cast<const char *>(p.as_ms) + ((size_t)s * p.d.depth + st.b) * asms_bytes(DFX_PREC_BF16) + q * 1024;
        dst = ring + q * 1024;
      } else if (q < 22) {
        src = reinterpret_cast<const char *>(bp.bconst) + (q - 17) * 1024;
        dst = lds0 + L_BCONST + (st.seq & 1) * BCONST_BYTES + (q - 17) * 1024;
      } else if (q == 22) {
        src = reinterpret_cast<const char *>(bp.ct + (size_t)step_t(p, st.step, s) * CT_ROW);
        dst = ring + 17 * 1024;
      }
      dma1k_pinned(src, voff, dst);
    }
    st.ff_src = reinterpret_cast<const char *>(bp.chunks) + q0 * 1024;
  }
}

// The same bookkeeping, but only the (wave-uniform) source / destination of this wave's pieces: the loads themselves are
// issued one at a time from inside the wave's next M slot (Issuer).
template <int NC>
struct Pieces {
  const char *src[NC];
  unsigned dst[NC];
};
template <int NC>
__device__ __forceinline__ void prepare_pieces(const KParams &p, DmaState &st, int q0, unsigned lds0, int s, Pieces<NC> &pc) {
  const unsigned ring = lds0 + L_RING + st.slot * SLOT_BYTES;
  if (st.step >= p.nsteps) {
#pragma unroll
    for (int j = 0; j < NC; ++j) pc.src[j] = reinterpret_cast<const char *>(p.d.blk[0].chunks), pc.dst[j] = lds0 + L_DUMMY;
  } else if (st.k > 0) {
#pragma unroll
    for (int j = 0; j < NC; ++j) pc.src[j] = st.ff_src + j * 1024, pc.dst[j] = ring + (q0 + j) * 1024;
    st.ff_src += SLOT_BYTES;
  } else {
    const BlockPack bp = block_pack(p, st.b);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int q = q0 + j;
      const char *src = reinterpret_cast<const char *>(bp.bconst);
      unsigned dst = lds0 + L_DUMMY;
      if (q < 17) {
        src = reinterpret_cast<const char *>(p.as_ms) + ((size_t)s * p.d.depth + st.b) * asms_bytes(DFX_PREC_BF16) + q * 1024;
        dst = ring + q * 1024;
      } else if (q < 22) {
        src = reinterpret_cast<const char *>(bp.bconst) + (q - 17) * 1024;
        dst = lds0 + L_BCONST + (st.seq & 1) * BCONST_BYTES + (q - 17) * 1024;
      } else if (q == 22) {
        src = reinterpret_cast<const char *>(bp.ct + (size_t)step_t(p, st.step, s) * CT_ROW);
        dst = ring + 17 * 1024;
      }
      // (the 64-bit address arithmetic above may run on the VALU: back to scalar registers for the "s" operands of the DMA)
      pc.src[j] = pin_ptr(src), pc.dst[j] = (unsigned)__builtin_amdgcn_readfirstlane(dst);
    }
    st.ff_src = reinterpret_cast<const char *>(bp.chunks) + q0 * 1024;
  }
}

__device__ __forceinline__ void advance_record(const KParams &p, DmaState &st) {
  st.slot = st.slot + 1 == NSLOT ? 0 : st.slot + 1;
  if (++st.k == RECORDS_PER_BLOCK) {
    st.k = 0;
    ++st.seq;
    if (++st.b == p.d.depth) {
      st.b = 0;
      ++st.step;
    }
  }
}

template <int NW>
__device__ __forceinline__ void issue_record(const KParams &p, DmaState &st, int wave, unsigned voff, unsigned lds0, int s) {
  issue_pieces<PipeCfg<NW>::CALLS>(p, st, wave * PipeCfg<NW>::CALLS, voff, lds0, s);
  advance_record(p, st);
}

// The DMA of the record three ahead is issued one piece at a time from inside the wave's own M slot, between MFMAs (the two
// groups' M slots are in anti-phase, so at most four waves issue at a time, one KiB per ~250 cycles).  Issued all at once
// behind the record's management barrier, the eight waves' 24 KiB hit the texture-address path (64 B/clk) together and
// every wave sits ~400 cycles in the issue stall on the critical path of the record (slot trace, tools/experiments/run_trace.sh).
template <int NW>
struct Issuer {
  static constexpr int CALLS = PipeCfg<NW>::CALLS;
  // 8-wave workgroups fill the chip: their pieces are spread over the M slot (comment above).  The 4- and 2-wave workgroups of
  // small batches (<= 32 workgroups in flight, no contention on the texture-address path) issue theirs at once behind the
  // record's management barrier: 6 / 12 (source, destination) pairs would not fit the scalar registers of the M slot.
  static constexpr bool SPREAD = NW == 8;
  const KParams &p;
  DmaState &st;
  int wave;
  unsigned voff, lds0;
  int s;
  Pieces<SPREAD ? CALLS : 1> pc;
  // (scalar) source / destination bookkeeping of the pieces of the next M slot; runs in the V slot before it
  __device__ __forceinline__ void m_begin() {
    if constexpr (SPREAD) {
      prepare_pieces<CALLS>(p, st, wave * CALLS, lds0, s, pc);
      advance_record(p, st);
    }
  }
  // right behind a record's management barrier
  __device__ __forceinline__ void after_barrier() {
    if constexpr (!SPREAD) issue_record<NW>(p, st, wave, voff, lds0, s);
  }
  // after MFMA i of the n of an M slot
  __device__ __forceinline__ void at(int i, int n) {
    if constexpr (SPREAD) {
#pragma unroll
      for (int k = 0; k < CALLS; ++k)
        if (i == (k * n) / CALLS) dma1k(pc.src[k], voff, pc.dst[k]);
    }
  }
};

// The per-point state (x_t, anchor, variance, sqrt(variance), part id) is touched once per diffusion step; parked in LDS
// in between (13 KiB per workgroup) it costs no VGPRs during the 90 record slots of a step — hipcc would otherwise keep
// it in scratch memory (80 B per lane: ~8 MB of write-back traffic per step and launch).
__device__ __forceinline__ void pstate_store(float *ps_lds, int pt, int PTS, const PointState &ps, bool all) {
#pragma unroll
  for (int i = 0; i < 3; ++i) ps_lds[i * PTS + pt] = ps.x[i];
  if (all) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      ps_lds[(3 + i) * PTS + pt] = ps.anc[i];
      ps_lds[(6 + i) * PTS + pt] = ps.var[i];
      ps_lds[(9 + i) * PTS + pt] = ps.L[i];
    }
    ps_lds[12 * PTS + pt] = __int_as_float(ps.sg);
  }
}
__device__ __forceinline__ void pstate_load(const float *ps_lds, int pt, int PTS, PointState &ps, bool all) {
#pragma unroll
  for (int i = 0; i < 3; ++i) ps.x[i] = ps_lds[i * PTS + pt];
  if (all) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      ps.anc[i] = ps_lds[(3 + i) * PTS + pt];
      ps.var[i] = ps_lds[(6 + i) * PTS + pt];
      ps.L[i] = ps_lds[(9 + i) * PTS + pt];
    }
  }
  ps.sg = __float_as_int(ps_lds[12 * PTS + pt]);
}

template <int NW>
__global__ void __launch_bounds__(NW * 64, 2) k_denoise_pipe(const KParams p) {
  constexpr int PREC = DFX_PREC_BF16;
  constexpr int PIPE_NW = NW, PTS = PipeCfg<NW>::PTS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  // XCD-aware placement: workgroups are handed to the 8 XCDs round-robin by blockIdx, and every XCD has its own 4 MiB
  // L2.  All workgroups stream the same block weights (2 MiB per step), but the attention records are per shape
  // (85 KiB each): with the natural order the wpg workgroups of one shape land on wpg different XCDs and every L2 sees
  // every shape.  Remap so that consecutive workgroups of ONE XCD walk through the workgroups of one shape.
  const int wpg = (p.N + PIPE_NW * 32 - 1) / (PIPE_NW * 32);   // workgroups per shape (the last one may be partial)
  int bid = blockIdx.x;
  {
    const int per = 8 * wpg;                        // workgroups of 8 shapes = one remap period
    const int full = ((int)gridDim.x / per) * per;  // the tail (B % 8 shapes) keeps the natural order
    if (bid < full) {
      const int x = bid & 7, q = (bid % per) >> 3;  // XCD, position on that XCD within the period (0 .. wpg-1)
      bid = (bid / per) * per + x * wpg + q;        // shape (period*8 + x), workgroup q of it
    }
  }
  // one shape per workgroup; N % 32 == 0, so whole wavefronts fall off the end of a shape's last tile: those recompute the
  // shape's last 32 points (same barriers, same DMA duties) and store nothing
  const int s = __builtin_amdgcn_readfirstlane(bid / wpg);
  int n0 = __builtin_amdgcn_readfirstlane((bid - s * wpg) * (PIPE_NW * 32) + wave * 32);
  const bool live = n0 < p.N;
  if (!live) n0 = p.N - 32;
  const int n = n0 + pj;
  const unsigned long long gid = ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)n;   // Philox key: GLOBAL point id
  const int depth = p.d.depth;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)pipe_smem);
  const unsigned voff = lane * 16;
  const bool grpA = wave < PIPE_NW / 2;

  // ---- prologue DMA: records 0, 1, 2 in flight while the per-point state is set up ----
  DmaState dma{0, 0, 0, 0, 0, nullptr};
  issue_record<NW>(p, dma, wave, voff, lds0, s);
  issue_record<NW>(p, dma, wave, voff, lds0, s);
  issue_record<NW>(p, dma, wave, voff, lds0, s);

  // ---- chain-invariant small operands -> LDS (plain loads; not part of the ring) ----
  {
    float4 *winx = reinterpret_cast<float4 *>(pipe_smem + L_WINX);
    float2 *pregb = reinterpret_cast<float2 *>(pipe_smem + L_PREGB);
    float4 *wout = reinterpret_cast<float4 *>(pipe_smem + L_WOUT);
    float *cp = reinterpret_cast<float *>(pipe_smem + L_CPART);
    const int tid = threadIdx.x;
    if (tid < 128) {
      winx[tid] = p.d.win_x[tid];
      pregb[tid] = p.d.pre_gb[tid];
      wout[tid] = p.d.wout[tid];
    }
    for (int i = tid; i < NCLS * INNER; i += NW * 64) cp[i] = p.cpart[(size_t)s * NCLS * INNER + i];
  }
  unsigned vmask;
  float *ps_lds = reinterpret_cast<float *>(pipe_smem + PipeCfg<NW>::L_PSTATE);
  const int pt = wave * 32 + pj;   // point slot inside the workgroup (both half-waves hold the same point)
  {
    PointState ps0;
    ps0.live = live;
    point_init(p, ps0, s, n, gid, vmask);
    pstate_store(ps_lds, pt, PTS, ps0, true);   // both half-waves hold the same point: identical values
  }
  __syncthreads();

  const float4 *winx = reinterpret_cast<const float4 *>(pipe_smem + L_WINX) + hf * 64;
  const float2 *pregb = reinterpret_cast<const float2 *>(pipe_smem + L_PREGB) + hf * 64;
  const float4 *wout = reinterpret_cast<const float4 *>(pipe_smem + L_WOUT) + hf * 64;

  Issuer<NW> issue_in_m{p, dma, wave, voff, lds0, s, {}};
  // slot boundary; `mgmt` = this barrier is a record's management barrier for this wave's group
#define DFX_SLOT(mgmt)                   \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    if (mgmt) {                          \
      tr.stamp(1);                       \
      wait_vmcnt<PipeCfg<NW>::CALLS>();  \
      __builtin_amdgcn_s_barrier();      \
      tr.stamp(2);                       \
      issue_in_m.after_barrier();        \
    } else {                             \
      tr.stamp(3);                       \
    }                                    \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
  // pointer to this lane's view of the record in ring slot `cur`, then advance
#define DFX_NEXT_RECORD()                                                               \
  do {                                                                                  \
    ck = reinterpret_cast<const uint4 *>(pipe_smem + L_RING + cur * SLOT_BYTES) + lane; \
    cur = cur + 1 == NSLOT ? 0 : cur + 1;                                               \
  } while (0)
  // the record that the next DFX_NEXT_RECORD() will hand out (complete in LDS since the previous management barrier)
#define DFX_PEEK_RECORD() (reinterpret_cast<const uint4 *>(pipe_smem + L_RING + cur * SLOT_BYTES) + lane)

#ifdef DFX_TRACE
  Tracer tr{(p.trace != nullptr && bid == 0 && (wave & 3) == 0) ? p.trace + (size_t)(wave >> 2) * p.trace_cap : nullptr,
            p.trace_cap, 0};
#else
  Tracer tr;
#endif

  uint4 P[8];   // first MFMA batch of the next M slot (tail prefetch)
  int cur = 0;  // ring slot of the next record to be consumed
  int seq = 0;  // running block number (parity selects the block-constant buffer)
  v16f h[4];
  bool done = false;
  for (int step = 0; step <= p.nsteps && !done; ++step) {
    for (int b = 0; b < depth; ++b, ++seq) {
      const uint4 *ck;
      Act<PREC> xn[4];
      // ---- V0: finish the previous block / step, start this one ----
      DFX_SLOT(!grpA && step < p.nsteps);
      issue_in_m.m_begin();
      if (seq > 0)  // b2 of the previous block (other block-constant buffer)
        add_cvec(h, reinterpret_cast<const float *>(pipe_smem + L_BCONST + ((seq - 1) & 1) * BCONST_BYTES) +
                        BCONST_B2_OFF + hf * 64);
      if (b == 0) {
        PointState ps;
        ps.s = s, ps.n = n, ps.gid = gid, ps.live = live;
        pstate_load(ps_lds, pt, PTS, ps, step > 0);
        if (step > 0) {
          float eps[3];
          post_eps<true>(h, wout, p.d.bout, eps);
          done = step_epilogue(p, ps, eps, step - 1, step_t(p, step - 1, s));
          if (!done) pstate_store(ps_lds, pt, PTS, ps, false);   // both half-waves hold the same point and write the same x
        }
        if (step == p.nsteps) done = true;
        if (!done) {
          const float *cpart = reinterpret_cast<const float *>(pipe_smem + L_CPART) + ps.sg * INNER + hf * 64;
          proj_in_prenorm<true>(h, ps.x, cpart, winx, pregb);
        }
      }
      if (done) break;
      ln_to_act<PREC>(h, xn);
      // ---- M0: sim = sbias + A_s xn ----
      DFX_SLOT(grpA);
      DFX_NEXT_RECORD();
      const uint4 *rec = ck;
      v16f sim;
      attn_m0(sim, xn, rec, reinterpret_cast<const float *>(rec - lane + 1024) + hf * 16, P, seq > 0, issue_in_m);
      // ---- V1: softmax ----
      DFX_SLOT(false);
      Act<PREC> pa;
      attn_softmax(sim, pa, vmask);
      // ---- M1: h += M_s P ----
      DFX_SLOT(false);
      attn_m1(h, pa, rec, P, DFX_PEEK_RECORD());
      // ---- V2: + c_t, LN3 ----
      DFX_SLOT(!grpA);
      issue_in_m.m_begin();
      add_cvec(h, reinterpret_cast<const float *>(rec - lane + 1088) + hf * 64);
      ln_to_act<PREC>(h, xn);
      // ---- feed-forward: M(F0) V M(F1) V ... M(F16) ----
      v16f a, g;
      HidAct hid;
      bias_slot_one(xn, hf);   // b1' enters GEMM1 through channel 127's K slot: no accumulator initialisers
      DFX_SLOT(grpA);
      DFX_NEXT_RECORD();
      ff_m<false, true, NEXT_FULL>(h, xn, a, g, hid, ck, P, DFX_PEEK_RECORD(), tr, issue_in_m);
#pragma unroll 1
      for (int j = 1; j < FF_CHUNKS - 1; ++j) {
        DFX_SLOT(!grpA);
        issue_in_m.m_begin();
        ff_v(a, g, hid, tr);
        DFX_SLOT(grpA);
        DFX_NEXT_RECORD();
        ff_m<true, true, NEXT_FULL>(h, xn, a, g, hid, ck, P, DFX_PEEK_RECORD(), tr, issue_in_m);
      }
      DFX_SLOT(!grpA);   // record F15: the next one is the block's last (GEMM2 only): its tail prefetch differs
      issue_in_m.m_begin();
      ff_v(a, g, hid, tr);
      DFX_SLOT(grpA);
      DFX_NEXT_RECORD();
      ff_m<true, true, NEXT_LAST>(h, xn, a, g, hid, ck, P, DFX_PEEK_RECORD(), tr, issue_in_m);
      DFX_SLOT(!grpA);
      issue_in_m.m_begin();
      ff_v(a, g, hid, tr);
      DFX_SLOT(grpA);
      DFX_NEXT_RECORD();
      ff_m<true, false, NEXT_AS>(h, xn, a, g, hid, ck, P, DFX_PEEK_RECORD(), tr, issue_in_m);
    }
  }
#undef DFX_SLOT
#undef DFX_NEXT_RECORD
#undef DFX_PEEK_RECORD
  wait_vmcnt<0>();  // drain padding DMAs before the LDS allocation is released
}

// ----------------------------------------------------------------------------------------------
// k_denoise_pipe2: the bf16 chain with TWO point tiles (64 points) per wavefront — VERDICT r2 item 1(a).
//
// The pipelined kernel above reads every A (weight) fragment from LDS once per 32 points: eight wavefronts x 24 KiB per record,
// ~10 % of the energy of a record on a chip that sits on its power cap (profiles/r02_energy_budget.txt).  Here a workgroup is four
// wavefronts of 64 points (the same 256-point tile, LDS map and ring), ONE wavefront per SIMD with the whole 512-entry register
// file: every fragment read feeds two MFMAs (tile 0, tile 1), and the b1 accumulator initialisers are read once and enter both
// tiles' first MFMA as its C operand (dst != src C), so the LDS -> register traffic per point halves.  There is no partner
// wavefront to run the VALU work beside the MFMAs, so the feed-forward is software-pipelined inside the wave (the r02 DFX_SWP
// experiment, now with two tiles): for FF record j
//     stage A   GEMM1 of chunk j   (16 fragments x 2 tiles = 32 MFMAs)  with the packed-fp16 GELU of chunk j-1 behind them (5 v_pk per MFMA)
//     stage B   GEMM2 of chunk j-1 (8 x 2 = 16 MFMAs)                   with (a, g) of chunk j -> packed fp16 and b1 of chunk j+1 -> registers
// Same device functions and the same MFMA order per accumulator as k_denoise_pipe: bit-identical results (tested).
constexpr int P2_NW = 4;
constexpr int P2_LDS = PipeCfg<P2_NW>::L_PSTATE + PSTATE_FIELDS * 256 * 4;   // the 4-wave ring map (6 DMA pieces per wave) with 256 point slots
struct GeluRegs {
  h2 y[8], z[8], r[8];
};
// operation K = stage * 8 + pair of gelu16_f16_math (same instructions in the same order per pair)
template <int K>
__device__ __forceinline__ void gelu_op(GeluRegs &t, const h2 (&aa)[8], const h2 (&gg)[8]) {
  constexpr int st = K / 8, i = K % 8;
  if (st == 0) t.z[i] = __builtin_elementwise_fma(gg[i], gg[i], h2c(-1.62f));
  if (st == 1) t.y[i] = aa[i] * gg[i];
  if (st == 2) t.z[i] = __builtin_elementwise_min(t.z[i], h2c(1.62f));
  if (st == 3) t.r[i] = __builtin_elementwise_fma(t.z[i], h2c(-0.0011402554f), h2c(0.0057853916f));
  if (st == 4) t.r[i] = __builtin_elementwise_fma(t.r[i], t.z[i], h2c(-0.0158536041f));
  if (st == 5) t.r[i] = __builtin_elementwise_fma(t.r[i], t.z[i], h2c(0.0409006897f));
  if (st == 6) t.r[i] = __builtin_elementwise_fma(t.r[i], t.z[i], h2c(-0.1098130657f));
  if (st == 7) t.r[i] = __builtin_elementwise_fma(t.r[i], t.z[i], h2c(0.3885767652f));
  if (st == 8) asm("v_pk_fma_f16 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(t.z[i]) : "v"(gg[i]), "v"(t.r[i]));   // Phi
  if (st == 9) t.y[i] = t.y[i] * t.z[i];
}
template <int K0, int N>
__device__ __forceinline__ void gelu_ops(GeluRegs &t, const h2 (&aa)[8], const h2 (&gg)[8]) {
  if constexpr (N > 0) {
    gelu_op<K0>(t, aa, gg);
    gelu_ops<K0 + 1, N - 1>(t, aa, gg);
  }
}
__device__ __forceinline__ void hid_from(const GeluRegs &t, HidAct &hid) {
  hid.f[0] = make_uint4(__builtin_bit_cast(unsigned, t.y[0]), __builtin_bit_cast(unsigned, t.y[1]), __builtin_bit_cast(unsigned, t.y[2]),
                        __builtin_bit_cast(unsigned, t.y[3]));
  hid.f[1] = make_uint4(__builtin_bit_cast(unsigned, t.y[4]), __builtin_bit_cast(unsigned, t.y[5]), __builtin_bit_cast(unsigned, t.y[6]),
                        __builtin_bit_cast(unsigned, t.y[7]));
}
// fragment order of this kernel's records: the 16 W1 fragments (GEMM1 MFMA e: half e >> 3, index e & 7), then the 8 W2 fragments
enum { P2_NEXT_W1 = 0, P2_NEXT_W2 = 1, P2_NEXT_AS = 2 };
template <int NEXT>
__device__ __forceinline__ constexpr int p2_next_frag(int i) { return NEXT == P2_NEXT_W1 ? w1_frag(i, 0) : NEXT == P2_NEXT_W2 ? w2_frag(i) : as_frag(i); }

// b1 of one chunk -> registers (the C operand of both tiles' first GEMM1 MFMAs)
__device__ __forceinline__ void load_b1(v16f &b1a, v16f &b1g, const float *b1) {
  typedef __attribute__((address_space(3))) const float lds_cf;
  unsigned addr = (unsigned)(uintptr_t)(lds_cf *)b1;
  asm volatile("" : "+v"(addr));   // one base register, immediate offsets
  const float *src = (const float *)(lds_cf *)(uintptr_t)addr;
  load16(b1a, src);
  load16(b1g, src + 32);
}

// GEMM1 accumulators (a, g) in ARCHITECTURAL registers.  With a 512-register budget hipcc selects the accumulator-file form for
// every MFMA builtin, and the GELU's conversions would then pay one v_accvgpr_read per value (64 per record: a third more VALU
// instructions).  The C/D file of an MFMA is chosen per instruction (A and B independently), so these — and only these — MFMAs are
// inline asm: D / C = VGPRs, A = the fragment from LDS (VGPR), B = the LayerNorm output (accumulator file: nothing else reads it).
// Hazards (guide §5.7 item 2): the inputs are written long before (ds_read + the compiler's lgkmcnt wait; xn one record earlier); D is
// read only by the NEXT MFMA on the same accumulator as its whole C (0 states) and by the conversions of stage B, at least three
// MFMA issues (> 96 cycles) after the accumulator's last MFMA (8-pass XDL: 12 states).
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_v_init(v16f &acc, const uint4 &A, const v8bf &B, const v16f &C) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(__builtin_bit_cast(v4u, A)), "a"(B), "v"(C));
}
__device__ __forceinline__ void mfma_v_acc(v16f &acc, const uint4 &A, const v8bf &B) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(__builtin_bit_cast(v4u, A)), "a"(B));
}
template <int E>   // the ten operations behind the E-th fragment of stage A: tile 0's GELU rides on fragments 0..7, tile 1's on 8..15
__device__ __forceinline__ void gelu_slot(GeluRegs &t, const h2 (&aa)[2][8], const h2 (&gg)[2][8], int half_slot) {
  constexpr int tl = E >> 3, base = (E & 7) * 10;
  if (half_slot == 0) gelu_ops<base, 5>(t, aa[tl], gg[tl]);
  else gelu_ops<base + 5, 5>(t, aa[tl], gg[tl]);
}

// The wave's DMA duty for the record three ahead: FF records (24 contiguous KiB) are issued one piece at a time from inside the
// MFMA stream (two scalar bases, immediate offsets); the irregular attention record at once behind its barrier.
struct Issuer2 {
  const char *src;   // nullptr: nothing deferred
  unsigned dst;
  unsigned voff;
  template <int Q>
  __device__ __forceinline__ void piece() const {
    if (src) {
      if (Q < 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(src), "s"(dst), "i"(Q * 1024) : "memory");
      else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(src + 4096), "s"(dst + 4096), "i"((Q - 4) * 1024) : "memory");
    }
  }
};

// One FF record for both tiles.  FIRST: record F0 (GEMM1 of chunk 0 only); LAST: record F16 (GELU + GEMM2 of chunk 15 only).
// P enters with the record's first eight fragments and leaves with the next record's (tail prefetch, as in k_denoise_pipe).
template <bool FIRST, bool LAST, int NEXT>
__device__ __forceinline__ void ff2(v16f (&h)[2][4], const Act<DFX_PREC_BF16> (&xn)[2][4], v16f (&a)[2], v16f (&g)[2], h2 (&aa)[2][8],
                                    h2 (&gg)[2][8], v16f &b1a, v16f &b1g, const uint4 *ck, uint4 (&P)[8], const uint4 *ck_next,
                                    const float *b1_next, const Issuer2 &dma, Tracer &tr) {
  GeluRegs t;
  HidAct hid[2];
  __builtin_amdgcn_sched_barrier(0);
  tr.stamp(30);
  if constexpr (!LAST) {
    auto stage_a = [&](auto ec) {
      constexpr int e = decltype(ec)::value;
      constexpr int k = e & 7, half = e >> 3;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        v16f &acc = (k & 1) ? g[tl] : a[tl];
        if (e < 2) mfma_v_init(acc, P[k], xn[tl][2 * half + (k >> 2)].f[(k >> 1) & 1], (k & 1) ? b1g : b1a);
        else mfma_v_acc(acc, P[k], xn[tl][2 * half + (k >> 2)].f[(k >> 1) & 1]);
        if (tl == 1) P[k] = e + 8 < 16 ? ck[w1_frag((e + 8) & 7, (e + 8) >> 3)] : FIRST ? ck_next[p2_next_frag<NEXT>(e + 8 - 16)] : ck[w2_frag(e + 8 - 16)];
        if constexpr (!FIRST) gelu_slot<e>(t, aa, gg, tl);
        if constexpr (!FIRST && e == 7) {
          if (tl == 1) hid_from(t, hid[0]);
        }
        if (tl == 0 && e % 3 == 0 && e / 3 < 6) {   // DMA pieces behind fragments 0, 3, 6, 9, 12, 15
          if (e == 0) dma.piece<0>();
          if (e == 3) dma.piece<1>();
          if (e == 6) dma.piece<2>();
          if (e == 9) dma.piece<3>();
          if (e == 12) dma.piece<4>();
          if (e == 15) dma.piece<5>();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    stage_a(std::integral_constant<int, 0>{});  stage_a(std::integral_constant<int, 1>{});  stage_a(std::integral_constant<int, 2>{});
    stage_a(std::integral_constant<int, 3>{});  stage_a(std::integral_constant<int, 4>{});  stage_a(std::integral_constant<int, 5>{});
    stage_a(std::integral_constant<int, 6>{});  stage_a(std::integral_constant<int, 7>{});  stage_a(std::integral_constant<int, 8>{});
    stage_a(std::integral_constant<int, 9>{});  stage_a(std::integral_constant<int, 10>{}); stage_a(std::integral_constant<int, 11>{});
    stage_a(std::integral_constant<int, 12>{}); stage_a(std::integral_constant<int, 13>{}); stage_a(std::integral_constant<int, 14>{});
    stage_a(std::integral_constant<int, 15>{});
    if constexpr (!FIRST) hid_from(t, hid[1]);
    tr.stamp(31);
  } else {
    gelu_ops<0, 80>(t, aa[0], gg[0]);
    hid_from(t, hid[0]);
    gelu_ops<0, 80>(t, aa[1], gg[1]);
    hid_from(t, hid[1]);
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (!FIRST) {
    const float *b1q = nullptr;
    if (!LAST && b1_next) {   // one opaque base register, immediate offsets
      typedef __attribute__((address_space(3))) const float lds_cf;
      unsigned addr = (unsigned)(uintptr_t)(lds_cf *)b1_next;
      asm volatile("" : "+v"(addr));
      b1q = (const float *)(lds_cf *)(uintptr_t)addr;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        h[tl][i & 3] = mma_hid(P[i], hid[tl].f[i >> 2], h[tl][i & 3]);
        if (tl == 1) P[i] = ck_next[p2_next_frag<NEXT>(i)];
        if constexpr (!LAST) {   // (a, g) of this chunk -> packed fp16: a got its last MFMA before g did
          if (i < 4) {
#pragma unroll
            for (int q = 2 * i; q < 2 * i + 2; ++q) aa[tl][q] = pk_f16(a[tl][2 * q], a[tl][2 * q + 1]);
          } else {
#pragma unroll
            for (int q = 2 * (i - 4); q < 2 * (i - 4) + 2; ++q) gg[tl][q] = pk_f16(g[tl][2 * q], g[tl][2 * q + 1]);
          }
        }
        if constexpr (!LAST) {   // b1 of the next chunk -> registers, a quarter per fragment (landed long before the next record's first MFMA)
          if (tl == 1 && b1q) {
            const v4f q4 = *reinterpret_cast<const v4f *>(b1q + (i < 4 ? 4 * i : 32 + 4 * (i - 4)));
            v16f &dst = i < 4 ? b1a : b1g;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) dst[4 * (i & 3) + e4] = q4[e4];
          }
        }
        if constexpr (LAST) {
          if (tl == 0 && i < 6) {
            if (i == 0) dma.piece<0>();
            if (i == 1) dma.piece<1>();
            if (i == 2) dma.piece<2>();
            if (i == 3) dma.piece<3>();
            if (i == 4) dma.piece<4>();
            if (i == 5) dma.piece<5>();
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    // Stage A's MFMAs are inline asm writing VGPRs: the hazard recogniser does not see them, and on this path no stage B sits
    // between the last MFMA (e = 15, tile 1 -> g[1]) and the conversions that read a / g.  An 8-pass XDL write needs 11 wait states
    // before a VALU read (18 for 16 passes): 24 explicit ones, once per block, whatever order the scheduler gives the conversions.
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) gelu16_f16_cvt(a[tl], g[tl], aa[tl], gg[tl]);
    __builtin_amdgcn_sched_barrier(0);
    if (b1_next) load_b1(b1a, b1g, b1_next);
  }
  __builtin_amdgcn_sched_barrier(0);
  tr.stamp(32);
}

__global__ void __launch_bounds__(P2_NW * 64, 1) k_denoise_pipe2(const KParams p) {
  constexpr int PREC = DFX_PREC_BF16;
  constexpr int NW = P2_NW, PTS = PipeCfg<NW>::PTS * 2;
  static_assert(PTS == 256 && P2_LDS <= 160 * 1024, "LDS budget");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  const int wpg = (p.N + PTS - 1) / PTS;
  int bid = blockIdx.x;
  {
    const int per = 8 * wpg;
    const int full = ((int)gridDim.x / per) * per;
    if (bid < full) {
      const int x = bid & 7, q = (bid % per) >> 3;
      bid = (bid / per) * per + x * wpg + q;
    }
  }
  const int s = __builtin_amdgcn_readfirstlane(bid / wpg);
  const int depth = p.d.depth;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)pipe_smem);
  const unsigned voff = lane * 16;
  // the wave's two tiles are tiles 2 wave and 2 wave + 1 of the workgroup's eight (the point slots of k_denoise_pipe<8>'s waves)
  int nn[2], ptv[2];
  bool lv[2];
  unsigned long long gidv[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    int n0 = __builtin_amdgcn_readfirstlane((bid - s * wpg) * PTS + (2 * wave + tl) * 32);
    lv[tl] = n0 < p.N;
    if (!lv[tl]) n0 = p.N - 32;
    nn[tl] = n0 + pj;
    ptv[tl] = (2 * wave + tl) * 32 + pj;
    gidv[tl] = ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)nn[tl];
  }

  DmaState dma{0, 0, 0, 0, 0, nullptr};
  issue_record<NW>(p, dma, wave, voff, lds0, s);
  issue_record<NW>(p, dma, wave, voff, lds0, s);
  issue_record<NW>(p, dma, wave, voff, lds0, s);
  {
    float4 *winx = reinterpret_cast<float4 *>(pipe_smem + L_WINX);
    float2 *pregb = reinterpret_cast<float2 *>(pipe_smem + L_PREGB);
    float4 *wout = reinterpret_cast<float4 *>(pipe_smem + L_WOUT);
    float *cp = reinterpret_cast<float *>(pipe_smem + L_CPART);
    const int tid = threadIdx.x;
    if (tid < 128) {
      winx[tid] = p.d.win_x[tid];
      pregb[tid] = p.d.pre_gb[tid];
      wout[tid] = p.d.wout[tid];
    }
    for (int i = tid; i < NCLS * INNER; i += NW * 64) cp[i] = p.cpart[(size_t)s * NCLS * INNER + i];
  }
  unsigned vmask = 0;
  float *ps_lds = reinterpret_cast<float *>(pipe_smem + PipeCfg<NW>::L_PSTATE);
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    PointState ps0;
    ps0.live = lv[tl];
    point_init(p, ps0, s, nn[tl], gidv[tl], vmask);
    pstate_store(ps_lds, ptv[tl], PTS, ps0, true);
  }
  __syncthreads();
  const float4 *winx = reinterpret_cast<const float4 *>(pipe_smem + L_WINX) + hf * 64;
  const float2 *pregb = reinterpret_cast<const float2 *>(pipe_smem + L_PREGB) + hf * 64;
  const float4 *wout = reinterpret_cast<const float4 *>(pipe_smem + L_WOUT) + hf * 64;

  // `defer`: the record that follows is an FF record with an MFMA stream to hide the DMA pieces in
  Issuer2 isr{nullptr, 0u, voff};
#ifdef DFX_TRACE
  Tracer tr{(p.trace != nullptr && bid == 0 && wave == 0) ? p.trace : nullptr, p.trace_cap, 0};
#else
  Tracer tr;
#endif
#define DFX_RECORD2(defer)                                                              \
  do {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    /* the (scalar) bookkeeping of a deferred FF record runs AHEAD of the barrier: with one wavefront per SIMD nobody */ \
    /* covers what sits between the barrier and the first MFMA */                      \
    isr.src = nullptr;                                                                  \
    const bool deferred_ = (defer) && dma.k > 0 && dma.step < p.nsteps;                 \
    if (deferred_) {                                                                    \
      isr.src = pin_ptr(dma.ff_src);                                                    \
      isr.dst = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + L_RING + dma.slot * SLOT_BYTES + wave * PipeCfg<NW>::CALLS * 1024); \
      dma.ff_src += SLOT_BYTES;                                                         \
      advance_record(p, dma);                                                           \
    }                                                                                   \
    ck = reinterpret_cast<const uint4 *>(pipe_smem + L_RING + cur * SLOT_BYTES) + lane; \
    cur = cur + 1 == NSLOT ? 0 : cur + 1;                                               \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    tr.stamp(1);                                                                        \
    wait_vmcnt<PipeCfg<NW>::CALLS>();                                                   \
    __builtin_amdgcn_s_barrier();                                                       \
    tr.stamp(2);                                                                        \
    if (!deferred_) issue_record<NW>(p, dma, wave, voff, lds0, s);                      \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
#define DFX_PEEK2() (reinterpret_cast<const uint4 *>(pipe_smem + L_RING + cur * SLOT_BYTES) + lane)

  uint4 P[8];
  int cur = 0, seq = 0;
  v16f h[2][4];
  bool done = false;
  for (int step = 0; step <= p.nsteps && !done; ++step) {
    for (int b = 0; b < depth; ++b, ++seq) {
      const uint4 *ck;
      Act<PREC> xn[2][4];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        if (seq > 0)
          add_cvec(h[tl], reinterpret_cast<const float *>(pipe_smem + L_BCONST + ((seq - 1) & 1) * BCONST_BYTES) + BCONST_B2_OFF + hf * 64);
        if (b == 0) {
          PointState ps;
          ps.s = s, ps.n = nn[tl], ps.gid = gidv[tl], ps.live = lv[tl];
          pstate_load(ps_lds, ptv[tl], PTS, ps, step > 0);
          bool dn = false;
          if (step > 0) {
            float eps[3];
            post_eps<true>(h[tl], wout, p.d.bout, eps);
            dn = step_epilogue(p, ps, eps, step - 1, step_t(p, step - 1, s));
            if (!dn) pstate_store(ps_lds, ptv[tl], PTS, ps, false);
          }
          if (step == p.nsteps) dn = true;
          if (!dn) proj_in_prenorm<true>(h[tl], ps.x, reinterpret_cast<const float *>(pipe_smem + L_CPART) + ps.sg * INNER + hf * 64, winx, pregb);
          done = dn;   // (wave-uniform and the same for both tiles: mode / step count)
        }
        if (!done) ln_to_act<PREC>(h[tl], xn[tl]);
      }
      if (done) break;
      // ---- attention record ----
      DFX_RECORD2(false);
      const uint4 *rec = ck;
      {
        if (seq == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) P[i] = rec[as_frag(i)];
        }
        v16f sb, sim[2];
        load16(sb, reinterpret_cast<const float *>(rec - lane + 1024) + hf * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int tl = 0; tl < 2; ++tl)
            sim[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(P[i]), xn[tl][i >> 1].f[i & 1], i == 0 ? sb : sim[tl], 0, 0, 0);
          P[i] = rec[ms_frag(i)];
          __builtin_amdgcn_sched_barrier(0);
        }
        tr.stamp(21);
        Act<PREC> pa[2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) attn_softmax(sim[tl], pa[tl], vmask);
        __builtin_amdgcn_sched_barrier(0);
        tr.stamp(22);
        const uint4 *nx = DFX_PEEK2();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int tl = 0; tl < 2; ++tl)
            h[tl][i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(P[i]), pa[tl].f[i >> 2], h[tl][i & 3], 0, 0, 0);
          P[i] = nx[w1_frag(i, 0)];
          __builtin_amdgcn_sched_barrier(0);
        }
        tr.stamp(23);
      }
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        add_cvec(h[tl], reinterpret_cast<const float *>(rec - lane + 1088) + hf * 64);
        ln_to_act<PREC>(h[tl], xn[tl]);
        bias_slot_one(xn[tl], hf);
        // LN3's output is read by GEMM1's MFMAs only: into the accumulator file HERE, once per block (an "a" operand whose value
        // lives in VGPRs is otherwise copied in front of every asm MFMA: 64 v_accvgpr_write per record)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          asm volatile("" : "+a"(xn[tl][c].f[0]));
          asm volatile("" : "+a"(xn[tl][c].f[1]));
        }
      }
      // ---- feed-forward ----
      v16f a[2], g[2], b1a, b1g;
      h2 aa[2][8], gg[2][8];
      // b1' rides on the constant-one K slot: the first MFMA's C operand is zero for every chunk.  The zeros are made opaque so that they
      // stay in registers: a rematerialised v_mov right in front of an asm MFMA is a VALU-write -> SrcC hazard the compiler does not see
      b1a = zero16(), b1g = zero16();
      asm volatile("" : "+v"(b1a), "+v"(b1g));
      DFX_RECORD2(true);
      ff2<true, false, P2_NEXT_W1>(h, xn, a, g, aa, gg, b1a, b1g, ck, P, DFX_PEEK2(), nullptr, isr, tr);
#pragma unroll 1
      for (int j = 1; j < FF_CHUNKS - 1; ++j) {
        DFX_RECORD2(true);
        ff2<false, false, P2_NEXT_W1>(h, xn, a, g, aa, gg, b1a, b1g, ck, P, DFX_PEEK2(), nullptr, isr, tr);
      }
      DFX_RECORD2(true);
      ff2<false, false, P2_NEXT_W2>(h, xn, a, g, aa, gg, b1a, b1g, ck, P, DFX_PEEK2(), nullptr, isr, tr);
      DFX_RECORD2(true);
      ff2<false, true, P2_NEXT_AS>(h, xn, a, g, aa, gg, b1a, b1g, ck, P, DFX_PEEK2(), nullptr, isr, tr);
    }
  }
#undef DFX_RECORD2
#undef DFX_PEEK2
  wait_vmcnt<0>();
}

// ----------------------------------------------------------------------------------------------
// LDS-pipelined kernel, exact fp32 (v_mfma_f32_32x32x2_f32): the reference-precision sampler.  The reference computes in
// fp32 end to end (attention.py:296-306, anchored_diffusion.py:227-395); this is that arithmetic at the structure of the bf16
// chain kernel — one workgroup = NW wavefronts x 32 points of one shape, the residual stream in registers for the whole chain,
// the posterior fused, and every weight streamed L2 -> LDS through the same 5-slot ring of 24 KiB records by LDS-DMA three
// records ahead, one workgroup barrier per record.  Same device functions and the same MFMA order per accumulator as the direct
// kernel k_denoise<DFX_PREC_F32>: the results are bit-identical (tested), so every golden of the fp32 path holds for it.
//
// What differs from the bf16 kernel is the balance: an fp32 MFMA is 64 cycles of matrix pipe for 4 bytes of A operand per lane
// (the bf16 one: 32 cycles for 16 bytes), so operand delivery is a twentieth of the bf16 kernel's — one ds_read_b128 feeds four
// MFMAs — and the fp32 VALU work (LayerNorm, softmax, erf-GELU, posterior) is ~15 % of the matrix time of one wavefront, hidden by
// the other wavefront of the SIMD without any slot choreography.  The direct kernel takes every fragment from L2 (~600 cycles
// away, 4 KiB tiles, two or three requests in flight per wavefront) and idles the matrix pipe 70 % of the time; here the pipe waits only
// at the record barriers.
//
// Ring records (24 KiB = six fp32 tiles = 24 DMA pieces of 1 KiB; RECORDS_F32 = 2 + 2 x 17 per transformer block):
//     k = 0      A_s tiles 0..3 | sbias piece (+16 KiB) | c_t row (+17 KiB); block constants -> their own double buffer
//     k = 1      M_s tiles 0..3
//     k = 2 + 2j FF record j, first half:  W1 g-rows x k-tiles 0..3 (tiles 4..7 of the packed record) | W1 a-rows x k-tiles 0, 1
//     k = 3 + 2j FF record j, second half: W1 a-rows x k-tiles 2, 3 | W2 row tiles 0..3 x hidden chunk j-1 (tiles 8..11)
// so g is complete after the first half and the second half ends with the MFMAs that do not feed the GELU (GEMM2 of the chunk before).
// Second half of FF record j of the fp32 chain: `a` x k-tiles 2, 3 (tiles 0, 1 of the half-record) and GEMM2 of chunk j - 1 (tiles 2..5),
// with the erf of chunk j's g BEHIND the MFMAs: g is complete since the first half, an fp32 MFMA keeps the matrix pipe busy for 64
// cycles after its issue, and the wave spends them on the VALU — gelu_erf(g[r]) overwrites g[r], one element per NM / 16 MFMAs
// (erff branches per lane; that only cuts basic blocks, the MFMAs in flight do not care).  Before this the two wavefronts of a SIMD
// reached the GELU together and the pipe idled for its whole length (80 % busy).  Same MFMA order per accumulator, same GELU
// function on the same values: bit-identical to the direct kernel.
template <bool FIRST, bool LAST>
__device__ __forceinline__ void ff_f32_second_half(v16f (&h)[4], const Act<DFX_PREC_F32> (&xn)[4], v16f &a, v16f &g, Act<DFX_PREC_F32> &hid,
                                                   const uint4 *rl) {
  constexpr int TS = tile_units(DFX_PREC_F32) * 64;
  constexpr int T0 = LAST ? 2 : 0, T1 = FIRST ? 2 : 6;   // tiles of the half-record that carry MFMAs here
  constexpr int NM = (T1 - T0) * 16, PER = NM / 16;
  v4f w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int t = T0 + m / 16, r4 = (m % 16) / 4, e = m % 4;
    if (e == 0) w = __builtin_bit_cast(v4f, rl[t * TS + r4 * 64]);
    if (t < 2) a = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], xn[2 + t].x[4 * r4 + e], a, 0, 0, 0);
    else h[t - 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], hid.x[4 * r4 + e], h[t - 2], 0, 0, 0);
    if constexpr (!LAST) {
      if ((m + 1) % PER == 0) {
        const int r = (m + 1) / PER - 1;
        g[r] = gelu_erf(g[r]);
      }
    }
  }
  if constexpr (!LAST) {
#pragma unroll
    for (int r = 0; r < 16; ++r) hid.x[r] = a[r] * g[r];
  }
}

constexpr int RECORDS_F32 = 2 + 2 * FF_STAGES;
static_assert(tile_bytes(DFX_PREC_F32) * 6 == SLOT_BYTES && asms_bytes(DFX_PREC_F32) == 33 * 1024, "fp32 ring records");

struct DmaStateF {
  int step, b, k, seq, slot;
};

template <int NC>
__device__ __forceinline__ void issue_pieces_f32(const KParams &p, const DmaStateF &st, int q0, unsigned voff, unsigned lds0, int s) {
  const unsigned ring = lds0 + L_RING + st.slot * SLOT_BYTES;
  if (st.step >= p.nsteps) {  // past the end: padding pieces keep the vmcnt bookkeeping uniform
    dma_nk<NC>(p.d.blk[0].chunks, voff, lds0 + L_DUMMY);
    return;
  }
  const BlockPack bp = block_pack(p, st.b);
  const char *asms = reinterpret_cast<const char *>(p.as_ms) + ((size_t)s * p.d.depth + st.b) * asms_bytes(DFX_PREC_F32);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int q = q0 + j;
    const char *src = reinterpret_cast<const char *>(bp.bconst);   // padding piece
    unsigned dst = lds0 + L_DUMMY;
    if (st.k == 0) {
      if (q < 16) src = asms + q * 1024, dst = ring + q * 1024;
      else if (q == 16) src = asms + 32 * 1024, dst = ring + 16 * 1024;
      else if (q < 22) src = reinterpret_cast<const char *>(bp.bconst) + (q - 17) * 1024, dst = lds0 + L_BCONST + (st.seq & 1) * BCONST_BYTES + (q - 17) * 1024;
      else if (q == 22) src = reinterpret_cast<const char *>(bp.ct + (size_t)step_t(p, st.step, s) * CT_ROW), dst = ring + 17 * 1024;
    } else if (st.k == 1) {
      if (q < 16) src = asms + (16 + q) * 1024, dst = ring + q * 1024;
    } else {
      const char *rec = reinterpret_cast<const char *>(bp.chunks) + (size_t)((st.k - 2) >> 1) * chunk_bytes(DFX_PREC_F32);
      if (((st.k - 2) & 1) == 0) src = rec + (q < 16 ? (16 + q) * 1024 : (q - 16) * 1024);
      else src = rec + (q < 8 ? (8 + q) * 1024 : (32 + q - 8) * 1024);
      dst = ring + q * 1024;
    }
    dma1k_pinned(src, voff, dst);
  }
}

template <int NW>
__device__ __forceinline__ void issue_record_f32(const KParams &p, DmaStateF &st, int wave, unsigned voff, unsigned lds0, int s) {
  issue_pieces_f32<PipeCfg<NW>::CALLS>(p, st, wave * PipeCfg<NW>::CALLS, voff, lds0, s);
  st.slot = st.slot + 1 == NSLOT ? 0 : st.slot + 1;
  if (++st.k == RECORDS_F32) {
    st.k = 0;
    ++st.seq;
    if (++st.b == p.d.depth) {
      st.b = 0;
      ++st.step;
    }
  }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64, 2) k_denoise_pipe_f32(const KParams p) {
  constexpr int PREC = DFX_PREC_F32;
  constexpr int PTS = PipeCfg<NW>::PTS, TS = tile_units(PREC) * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  // XCD-aware placement and tile bookkeeping: as in k_denoise_pipe
  const int wpg = (p.N + NW * 32 - 1) / (NW * 32);
  int bid = blockIdx.x;
  {
    const int per = 8 * wpg;
    const int full = ((int)gridDim.x / per) * per;
    if (bid < full) {
      const int x = bid & 7, q = (bid % per) >> 3;
      bid = (bid / per) * per + x * wpg + q;
    }
  }
  const int s = __builtin_amdgcn_readfirstlane(bid / wpg);
  int n0 = __builtin_amdgcn_readfirstlane((bid - s * wpg) * (NW * 32) + wave * 32);
  const bool live = n0 < p.N;
  if (!live) n0 = p.N - 32;
  const int n = n0 + pj;
  const unsigned long long gid = ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)n;
  const int depth = p.d.depth;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)pipe_smem);
  const unsigned voff = lane * 16;

  DmaStateF dma{0, 0, 0, 0, 0};
  issue_record_f32<NW>(p, dma, wave, voff, lds0, s);
  issue_record_f32<NW>(p, dma, wave, voff, lds0, s);
  issue_record_f32<NW>(p, dma, wave, voff, lds0, s);
  {
    float4 *winx = reinterpret_cast<float4 *>(pipe_smem + L_WINX);
    float2 *pregb = reinterpret_cast<float2 *>(pipe_smem + L_PREGB);
    float4 *wout = reinterpret_cast<float4 *>(pipe_smem + L_WOUT);
    float *cp = reinterpret_cast<float *>(pipe_smem + L_CPART);
    const int tid = threadIdx.x;
    if (tid < 128) {
      winx[tid] = p.d.win_x[tid];
      pregb[tid] = p.d.pre_gb[tid];
      wout[tid] = p.d.wout[tid];
    }
    for (int i = tid; i < NCLS * INNER; i += NW * 64) cp[i] = p.cpart[(size_t)s * NCLS * INNER + i];
  }
  unsigned vmask;
  float *ps_lds = reinterpret_cast<float *>(pipe_smem + PipeCfg<NW>::L_PSTATE);
  const int pt = wave * 32 + pj;
  {
    PointState ps0;
    ps0.live = live;
    point_init(p, ps0, s, n, gid, vmask);
    pstate_store(ps_lds, pt, PTS, ps0, true);
  }
  __syncthreads();
  const float4 *winx = reinterpret_cast<const float4 *>(pipe_smem + L_WINX) + hf * 64;
  const float2 *pregb = reinterpret_cast<const float2 *>(pipe_smem + L_PREGB) + hf * 64;
  const float4 *wout = reinterpret_cast<const float4 *>(pipe_smem + L_WOUT) + hf * 64;

  // record boundary: this wave's pieces of the record after next have landed, everybody is done with the slot that is refilled next
  // (ring protocol of k_denoise_pipe), then the pieces of the record three ahead are issued
#ifdef DFX_TRACE   // phase stamps of waves 0 and 4 (one SIMD) of workgroup 0: tools/experiments/trace_f32.py
  Tracer tr{(p.trace != nullptr && bid == 0 && (wave & 3) == 0) ? p.trace + (size_t)(wave >> 2) * p.trace_cap : nullptr, p.trace_cap, 0};
#else
  Tracer tr;
#endif
#define DFX_RECORD(ptr)                                                                      \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    tr.stamp(1);                                                                             \
    wait_vmcnt<PipeCfg<NW>::CALLS>();                                                        \
    __builtin_amdgcn_s_barrier();                                                            \
    tr.stamp(2);                                                                             \
    issue_record_f32<NW>(p, dma, wave, voff, lds0, s);                                       \
    tr.stamp(3);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    ptr = reinterpret_cast<const uint4 *>(pipe_smem + L_RING + cur * SLOT_BYTES);            \
    cur = cur + 1 == NSLOT ? 0 : cur + 1;                                                    \
  } while (0)

  int cur = 0, seq = 0;
  v16f h[4];
  bool done = false;
  for (int step = 0; step <= p.nsteps && !done; ++step) {
    for (int b = 0; b < depth; ++b, ++seq) {
      const float *bc = reinterpret_cast<const float *>(pipe_smem + L_BCONST + (seq & 1) * BCONST_BYTES);
      if (seq > 0)  // b2 of the previous block (other block-constant buffer)
        add_cvec(h, reinterpret_cast<const float *>(pipe_smem + L_BCONST + ((seq - 1) & 1) * BCONST_BYTES) + BCONST_B2_OFF + hf * 64);
      if (b == 0) {
        PointState ps;
        ps.s = s, ps.n = n, ps.gid = gid, ps.live = live;
        pstate_load(ps_lds, pt, PTS, ps, step > 0);
        if (step > 0) {
          float eps[3];
          post_eps(h, wout, p.d.bout, eps);
          done = step_epilogue(p, ps, eps, step - 1, step_t(p, step - 1, s));
          if (!done) pstate_store(ps_lds, pt, PTS, ps, false);
        }
        if (step == p.nsteps) done = true;
        if (!done) proj_in_prenorm(h, ps.x, reinterpret_cast<const float *>(pipe_smem + L_CPART) + ps.sg * INNER + hf * 64, winx, pregb);
      }
      if (done) break;
      const uint4 *rec;
      Act<PREC> xn[4];
      ln_to_act<PREC>(h, xn);
      // ---- attention: sim = sbias + A_s LN2(h);  P = softmax4;  h += M_s P + c_t ----
      DFX_RECORD(rec);
      const float *ct = reinterpret_cast<const float *>(rec) + 17 * 256 + hf * 64;   // (read during the NEXT record: its slot is refilled one barrier later)
      v16f sim;
      load16(sim, reinterpret_cast<const float *>(rec) + 16 * 256 + hf * 16);
#pragma unroll
      for (int c = 0; c < 4; ++c) mma_tile<PREC>(sim, rec + lane + c * TS, xn[c]);
      softmax4(sim, vmask);
      Act<PREC> pa;
      pa.set(sim);
      DFX_RECORD(rec);
#pragma unroll
      for (int t = 0; t < 4; ++t) mma_tile<PREC>(h[t], rec + lane + t * TS, pa);
      add_cvec(h, ct);
      ln_to_act<PREC>(h, xn);
      // ---- feed-forward: 17 records of two halves; GEMM2 of chunk j-1 rides in the second half of record j ----
      Act<PREC> hid;
      v16f a, g;
      auto first_half = [&](int j) {   // g x k-tiles 0..3, a x k-tiles 0, 1
        load16(a, bc + j * 64 + hf * 16);
        load16(g, bc + j * 64 + hf * 16 + 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) mma_tile<PREC>(g, rec + lane + c * TS, xn[c]);
#pragma unroll
        for (int c = 0; c < 2; ++c) mma_tile<PREC>(a, rec + lane + (4 + c) * TS, xn[c]);
      };
      DFX_RECORD(rec);
      first_half(0);
      DFX_RECORD(rec);
      ff_f32_second_half<true, false>(h, xn, a, g, hid, rec + lane);
#pragma unroll 1
      for (int j = 1; j < FF_CHUNKS; ++j) {
        DFX_RECORD(rec);
        first_half(j);
        DFX_RECORD(rec);
        ff_f32_second_half<false, false>(h, xn, a, g, hid, rec + lane);
      }
      DFX_RECORD(rec);   // (record 16, first half: zero tiles of the packed layout — consumed to keep the ring uniform)
      DFX_RECORD(rec);
      ff_f32_second_half<false, true>(h, xn, a, g, hid, rec + lane);
    }
  }
#undef DFX_RECORD
  wait_vmcnt<0>();  // drain padding DMAs before the LDS allocation is released
}

// The step boundary runs on one wavefront with nobody to hide its LDS latency behind: the same arithmetic as post_eps /
// proj_in_prenorm (operation for operation — the results are bit-identical), with the operand reads of a whole 16-channel tile
// issued ahead of the tile before's arithmetic (hipcc, short of registers over the whole kernel, otherwise reads one operand at a time).
// (fences: an empty asm that consumes the batch's results and clobbers memory — reads cannot cross it, the arithmetic is tied to it by its
// data; a sched_barrier alone orders only what instruction selection has already laid out, and that is every read first.)
#define DFX_TIE8(a) asm volatile("" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]), "+v"((a)[4]), "+v"((a)[5]), "+v"((a)[6]), "+v"((a)[7]) :: "memory")
__device__ __forceinline__ void post_eps_tiles(const v16f (&h)[4], const float4 *wout, const float (&bout)[4], float (&eps)[3]) {
  float mean, rstd;
  ln_stats_fast(h, mean, rstd);
  const float nmr = -mean * rstd;
  float e0 = 0.f, e1 = 0.f, e2 = 0.f;
  float4 w[2][8];   // half a tile per batch
#pragma unroll
  for (int r = 0; r < 8; ++r) w[0][r] = wout[r];
  asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2) :: "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < 7) {
#pragma unroll
      for (int r = 0; r < 8; ++r) w[(k + 1) & 1][r] = wout[(k + 1) * 8 + r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float v = fmaf(h[k >> 1][(k & 1) * 8 + r], rstd, nmr);
      e0 = fmaf(w[k & 1][r].x, v, e0);
      e1 = fmaf(w[k & 1][r].y, v, e1);
      e2 = fmaf(w[k & 1][r].z, v, e2);
    }
    asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2) :: "memory");
  }
  eps[0] = e0 + xhalf(e0) + bout[0];
  eps[1] = e1 + xhalf(e1) + bout[1];
  eps[2] = e2 + xhalf(e2) + bout[2];
}

__device__ __forceinline__ void proj_in_prenorm_tiles(v16f (&h)[4], const float (&x)[3], const float *cpart, const float4 *winx, const float2 *pregb) {
  float4 w[2][8], cp[2][2];
  auto fetch = [&](int k) {
#pragma unroll
    for (int r4 = 0; r4 < 2; ++r4) cp[k & 1][r4] = *reinterpret_cast<const float4 *>(cpart + k * 8 + r4 * 4);
#pragma unroll
    for (int r = 0; r < 8; ++r) w[k & 1][r] = winx[k * 8 + r];
  };
  fetch(0);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < 7) fetch(k + 1);
    float o[8];
#pragma unroll
    for (int r4 = 0; r4 < 2; ++r4) {
      const float cpv[4] = {cp[k & 1][r4].x, cp[k & 1][r4].y, cp[k & 1][r4].z, cp[k & 1][r4].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 ww = w[k & 1][r4 * 4 + e];
        o[r4 * 4 + e] = fmaf(ww.z, x[2], fmaf(ww.y, x[1], fmaf(ww.x, x[0], cpv[e])));
      }
    }
    DFX_TIE8(o);
#pragma unroll
    for (int e = 0; e < 8; ++e) h[k >> 1][(k & 1) * 8 + e] = o[e];
  }
  float2 gb[2][8];
#pragma unroll
  for (int r = 0; r < 8; ++r) gb[0][r] = pregb[r];
  float mean, rstd;
  ln_stats_fast(h, mean, rstd);
  asm volatile("" : "+v"(mean), "+v"(rstd) :: "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < 7) {
#pragma unroll
      for (int r = 0; r < 8; ++r) gb[(k + 1) & 1][r] = pregb[(k + 1) * 8 + r];
    }
    float o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = fmaf((h[k >> 1][(k & 1) * 8 + r] - mean) * rstd, gb[k & 1][r].x, gb[k & 1][r].y);
    DFX_TIE8(o);
#pragma unroll
    for (int e = 0; e < 8; ++e) h[k >> 1][(k & 1) * 8 + e] = o[e];
  }
}

constexpr int COOP_NW = 8, COOP_TILES = 4;
static_assert(FF_CHUNKS == 2 * COOP_NW, "two rounds of one chunk per wavefront");
constexpr int COOP_W2_FIRST = 8;                                  // first W2 chunk staged in LDS (the LDS budget holds eight)
constexpr int CL_XN = 0;                                          // 8 x 64 uint4: LN3 output as the B operand of GEMM1
constexpr int CL_HID = CL_XN + 8 * 1024;                          // [16][2][64] uint4: GELU output of every chunk
constexpr int CL_BC = CL_HID + FF_CHUNKS * 2048;                  // 2 x block constants (b1', b2), by block parity
constexpr int CL_AT = CL_BC + 2 * BCONST_BYTES;                   // attention record (17 KiB) + c_t row (1 KiB)
constexpr int CL_HS = CL_AT + asms_bytes(DFX_PREC_BF16) + 1024;   // home of the residual stream h: 4 tiles x 4 KiB, [tile][q][lane] float4
constexpr int CL_CONST = CL_HS + 16 * 1024;                       // chain-invariant operands: W_in x-columns (2 KiB) | pre_norm (1 KiB) | W_out (2 KiB) | cpart (2 KiB)
constexpr int CL_W2 = CL_CONST + 7 * 1024;                        // W2 tiles of chunks 8..15 (8 KiB each)
constexpr int CL_Z = CL_W2 + (FF_CHUNKS - COOP_W2_FIRST) * 8192;   // this step's noise z[3][32] (384 B) | posterior table row (32 B at +512)
constexpr int CL_PS = CL_Z + 1024;                                 // per-point chain state (x, anchor, variance, L, part id: 13 x 32 floats), home between step boundaries
constexpr int CL_TOTAL = CL_PS + 2048;
static_assert(CL_TOTAL <= 160 * 1024, "LDS budget of the co-operative kernel");

__global__ void __launch_bounds__(COOP_NW * 64, 2) k_denoise_coop(const KParams p) {
  constexpr int PREC = DFX_PREC_BF16;
  constexpr int TSTRIDE = tile_units(PREC) * 64, AREC = asms_bytes(PREC) / 16;
  uint4 (*s_xn)[64] = reinterpret_cast<uint4 (*)[64]>(pipe_smem + CL_XN);
  uint4 (*s_hid)[2][64] = reinterpret_cast<uint4 (*)[2][64]>(pipe_smem + CL_HID);
  uint4 *s_at = reinterpret_cast<uint4 *>(pipe_smem + CL_AT);
  float *s_hs = reinterpret_cast<float *>(pipe_smem + CL_HS);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  const long long g0 = (long long)blockIdx.x * 32;
  const int s = __builtin_amdgcn_readfirstlane((int)(g0 / p.N));
  const int n = (int)(g0 - (long long)s * p.N) + pj;
  const int depth = p.d.depth;
  const bool w0 = wave == 0, tile_owner = wave < COOP_TILES;
  float *ps_lds = reinterpret_cast<float *>(pipe_smem + CL_PS);
  unsigned vmask = 0;
  if (w0) {   // (the state would otherwise sit in scratch memory for the whole chain: a global-memory round trip per field at every step boundary)
    PointState ps0;
    point_init(p, ps0, s, n, ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)n, vmask);
    pstate_store(ps_lds, pj, 32, ps0, true);
  }
  const uint4 *asms_s = p.as_ms + (size_t)s * depth * AREC;
  {   // chain-invariant small operands -> LDS (wave 0 reads them at every step boundary)
    float4 *c_winx = reinterpret_cast<float4 *>(pipe_smem + CL_CONST);
    float2 *c_pregb = reinterpret_cast<float2 *>(pipe_smem + CL_CONST + 2048);
    float4 *c_wout = reinterpret_cast<float4 *>(pipe_smem + CL_CONST + 3072);
    float *c_cp = reinterpret_cast<float *>(pipe_smem + CL_CONST + 5120);
    const int tid = threadIdx.x;
    if (tid < 128) c_winx[tid] = p.d.win_x[tid], c_pregb[tid] = p.d.pre_gb[tid], c_wout[tid] = p.d.wout[tid];
    c_cp[tid] = p.cpart[(size_t)s * NCLS * INNER + tid];
    __syncthreads();
  }
  const float4 *winx = reinterpret_cast<const float4 *>(pipe_smem + CL_CONST) + hf * 64;
  const float2 *pregb = reinterpret_cast<const float2 *>(pipe_smem + CL_CONST + 2048) + hf * 64;
  const float4 *wout = reinterpret_cast<const float4 *>(pipe_smem + CL_CONST + 3072) + hf * 64;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)pipe_smem);
  const unsigned voff = lane * 16;

  // One register file for every role (one kernel = one register allocation):
  //   R[32]   phase H: GEMM1 fragments of round 0 (R[0..15]) / round 1 (R[16..31]), in the accumulation order a(c,0) a(c,1) g(c,0)
  //           g(c,1), c = 0..3;   phase G (tile owners): W2 fragments of chunks 0..7 of the own tile, R[2 u + q]
  //   h[4]    wave 0, phase A only;   ht   tile owners, phase G only — in between h lives in LDS
  uint4 R[32];
  // (scalar base + lane offset: with the base pinned to SGPRs the sixteen loads share one address VGPR; left to itself hipcc keeps
  // 64-bit per-lane addresses, spills them, and every scratch reload — a vector-memory operation like the prefetches in flight —
  // drains the whole prefetch with s_waitcnt vmcnt(0))
  auto load_w1 = [&](int base, const uint4 *chunks, int u) {
    const uint4 *ck = reinterpret_cast<const uint4 *>(pin_ptr(reinterpret_cast<const char *>(chunks + (size_t)u * CHUNK_TILES * TSTRIDE))) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      R[base + 4 * c + 0] = ck[(0 + c) * TSTRIDE], R[base + 4 * c + 1] = ck[(0 + c) * TSTRIDE + 64];
      R[base + 4 * c + 2] = ck[(4 + c) * TSTRIDE], R[base + 4 * c + 3] = ck[(4 + c) * TSTRIDE + 64];
    }
  };
  auto xn_frag = [&](int k) -> v8bf { return __builtin_bit_cast(v8bf, s_xn[k][lane]); };   // k-th uint4 of xn3 (tile k >> 1, unit k & 1)
  // h's LDS home: tile c, quad q of lane l at float4 index (4 c + q) * 64 + l
  auto hs_ptr = [&](int c, int q) -> v4f * { return reinterpret_cast<v4f *>(s_hs + ((c * 4 + q) * 64 + lane) * 4); };

#ifdef DFX_TRACE   // phase stamps of wave 0 (row 0 of the trace buffer) and of wave 5 (row 1) of workgroup 0
  Tracer tr{(p.trace != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 5)) ? p.trace + (size_t)(wave ? 1 : 0) * p.trace_cap : nullptr, p.trace_cap, 0};
#else
  Tracer tr;
#endif
  // proj_in + pre_norm of the chain state -> h's LDS home (wave 0: before the first step and at every step boundary)
  auto enter_step = [&](const PointState &ps) {
    v16f h[4];
    proj_in_prenorm_tiles(h, ps.x, reinterpret_cast<const float *>(pipe_smem + CL_CONST + 5120) + ps.sg * INNER + hf * 64, winx, pregb);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) *hs_ptr(c, q) = v4f{h[c][4 * q], h[c][4 * q + 1], h[c][4 * q + 2], h[c][4 * q + 3]};
  };
  if (w0) {
    PointState ps;
    pstate_load(ps_lds, pj, 32, ps, false);
    enter_step(ps);
  }
  int seq = 0;
  for (int step = 0; step < p.nsteps; ++step) {
    const int t = step_t(p, step, s);
    for (int b = 0; b < depth; ++b, ++seq) {
      const BlockPack bp = block_pack(p, b);
      float *s_bc = reinterpret_cast<float *>(pipe_smem + CL_BC + (seq & 1) * BCONST_BYTES);
      // ---- top of the block: operands of phase A -> LDS (waves 4..7, which left phase G's barrier first), round-0 fragments requested
      if (!tile_owner) {
        constexpr int NA = asms_bytes(PREC) / 1024, NB = BCONST_BYTES / 1024, NH = COOP_NW - COOP_TILES;   // 17 + 1 + 5 pieces over 4 waves
#pragma unroll
        for (int i = 0; i < (NA + 1 + NB + NH - 1) / NH; ++i) {
          const int k = i * NH + wave - COOP_TILES;
          if (k < NA) dma1k_pinned(reinterpret_cast<const char *>(asms_s + (size_t)b * AREC) + k * 1024, voff, lds0 + CL_AT + k * 1024);
          else if (k == NA) dma1k_pinned(reinterpret_cast<const char *>(bp.ct + (size_t)t * CT_ROW), voff, lds0 + CL_AT + NA * 1024);
          else if (k < NA + 1 + NB) dma1k_pinned(reinterpret_cast<const char *>(bp.bconst) + (k - NA - 1) * 1024, voff, lds0 + CL_BC + (seq & 1) * BCONST_BYTES + (k - NA - 1) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      load_w1(0, bp.chunks, wave);   // round 0: in flight through phase A
      tr.stamp(10);
      __syncthreads();   // 0: attention record, c_t, block constants of this block are in LDS; h's home holds the previous block's result
      tr.stamp(11);
      if (!w0) {         // ---- phase A for the idle waves: W2 tiles of chunks 8..15 -> LDS
        constexpr int NP = (FF_CHUNKS - COOP_W2_FIRST) * 8, NH = COOP_NW - 1;   // 64 pieces over 7 waves
#pragma unroll
        for (int i = 0; i < (NP + NH - 1) / NH; ++i) {
          const int k = i * NH + wave - 1;
          if (k < NP)
            dma1k_pinned(reinterpret_cast<const char *>(bp.chunks + (size_t)(COOP_W2_FIRST + (k >> 3) + FF_SKEW) * CHUNK_TILES * TSTRIDE + 8 * TSTRIDE) + (k & 7) * 1024,
                         voff, lds0 + CL_W2 + k * 1024);
        }
        if (wave == COOP_NW - 1 && b == depth - 1 && p.mode != MODE_EPS) {   // the step's noise and posterior coefficients, ready for wave 0's epilogue
          float z[3];
          if (p.noise) {
#pragma unroll
            for (int i = 0; i < 3; ++i) z[i] = p.noise[(((size_t)step * p.B + s) * 3 + i) * p.N + n];
          } else {
            philox_normal3(p.seed, ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)n, (unsigned)t, 0u, z);
          }
          float *s_z = reinterpret_cast<float *>(pipe_smem + CL_Z);
          if (hf == 0) s_z[pj] = z[0], s_z[32 + pj] = z[1], s_z[64 + pj] = z[2];
          if (lane < 8) s_z[128 + lane] = p.d.tab[(size_t)t * 8 + lane];
        }
      } else {           // ---- phase A: wave 0
        v16f h[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4f x = *hs_ptr(c, q);
            h[c][4 * q] = x[0], h[c][4 * q + 1] = x[1], h[c][4 * q + 2] = x[2], h[c][4 * q + 3] = x[3];
          }
        const uint4 *rec = s_at + lane;
        attention<PREC>(h, rec, reinterpret_cast<const float *>(s_at + 8 * TSTRIDE) + hf * 16, reinterpret_cast<const float *>(s_at + AREC) + hf * 64, vmask);
        Act<PREC> xo[4];
        ln_to_act<PREC>(h, xo);
        bias_slot_one(xo, hf);
#pragma unroll
        for (int c = 0; c < 4; ++c) s_xn[2 * c][lane] = __builtin_bit_cast(uint4, xo[c].f[0]), s_xn[2 * c + 1][lane] = __builtin_bit_cast(uint4, xo[c].f[1]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) *hs_ptr(c, q) = v4f{h[c][4 * q], h[c][4 * q + 1], h[c][4 * q + 2], h[c][4 * q + 3]};
      }
      tr.stamp(12);
      __syncthreads();   // 1: xn3 of this block and h are in LDS
      tr.stamp(13);
      // ---- phase H: every wave, chunks `wave` and `wave + 8`
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int u = r * COOP_NW + wave, cur = r * 16;
        if (r == 0) load_w1(16, bp.chunks, u + COOP_NW);
        else if (tile_owner) {   // round 0's registers are free: the own tile's W2 fragments of chunks 0..7 (W2 of chunk c sits in FF record c + FF_SKEW, tiles 8..11)
#pragma unroll
          for (int c = 0; c < COOP_W2_FIRST; ++c) {
            const uint4 *ck = reinterpret_cast<const uint4 *>(pin_ptr(reinterpret_cast<const char *>(bp.chunks + (size_t)(c + FF_SKEW) * CHUNK_TILES * TSTRIDE + (8 + wave) * TSTRIDE))) + lane;
            R[2 * c] = ck[0], R[2 * c + 1] = ck[64];
          }
        }
        v16f a = zero16(), g = zero16();   // b1' rides on the constant-one K slot (bias_slot_one)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(R[cur + 4 * c + 0]), xn_frag(2 * c), a, 0, 0, 0);
          g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(R[cur + 4 * c + 2]), xn_frag(2 * c), g, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(R[cur + 4 * c + 1]), xn_frag(2 * c + 1), a, 0, 0, 0);
          g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(R[cur + 4 * c + 3]), xn_frag(2 * c + 1), g, 0, 0, 0);
        }
        h2 aa[8], gg[8];
        HidAct hid;
        gelu16_f16_cvt(a, g, aa, gg);
        gelu16_f16_math(aa, gg, hid);
        s_hid[u][0][lane] = hid.f[0], s_hid[u][1][lane] = hid.f[1];
      }
      if (!w0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's W2 pieces have landed (long ago: loads complete in order)
      tr.stamp(14);
      __syncthreads();   // 2: hid of all chunks and the W2 copies are in LDS
      tr.stamp(15);
      // ---- phase G: wave t accumulates tile t of h, chunk after chunk (the accumulation order of the pipelined kernel), + b2
      if (tile_owner) {
        v16f ht;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f x = *hs_ptr(wave, q);
          ht[4 * q] = x[0], ht[4 * q + 1] = x[1], ht[4 * q + 2] = x[2], ht[4 * q + 3] = x[3];
        }
        // hid / LDS-resident W2 fragments: a ring three chunks deep (one chunk = two dependent MFMAs = less than one LDS round trip),
        // fenced per chunk — left alone hipcc sinks every read next to its MFMA and the chain runs at LDS latency
        uint4 hq[4][2], wq[4][2];
        unsigned hoff = CL_HID + lane * 16, woff = CL_W2 + (wave * TSTRIDE + lane) * 16;
        asm volatile("" : "+v"(hoff), "+v"(woff));   // one base register each, chunk offsets as immediates
        const uint4 *hidb = reinterpret_cast<const uint4 *>(pipe_smem + hoff), *w2b = reinterpret_cast<const uint4 *>(pipe_smem + woff);
        auto fetch = [&](int u) {
          hq[u & 3][0] = hidb[u * 128], hq[u & 3][1] = hidb[u * 128 + 64];
          if (u >= COOP_W2_FIRST) wq[u & 3][0] = w2b[(u - COOP_W2_FIRST) * 4 * TSTRIDE], wq[u & 3][1] = w2b[(u - COOP_W2_FIRST) * 4 * TSTRIDE + 64];
        };
        fetch(0), fetch(1), fetch(2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < FF_CHUNKS; ++u) {
          if (u + 3 < FF_CHUNKS) fetch(u + 3);
          const uint4 f0 = u < COOP_W2_FIRST ? R[2 * u] : wq[u & 3][0], f1 = u < COOP_W2_FIRST ? R[2 * u + 1] : wq[u & 3][1];
          ht = mma_hid(f0, hq[u & 3][0], ht);
          ht = mma_hid(f1, hq[u & 3][1], ht);
          __builtin_amdgcn_sched_barrier(0);
        }
        {
          const float *b2 = s_bc + BCONST_B2_OFF + hf * 64 + wave * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4f bb = *reinterpret_cast<const v4f *>(b2 + 4 * q);
            *hs_ptr(wave, q) = v4f{ht[4 * q] + bb[0], ht[4 * q + 1] + bb[1], ht[4 * q + 2] + bb[2], ht[4 * q + 3] + bb[3]};
          }
        }
      }
    }
    tr.stamp(16);
    __syncthreads();   // the last block's tiles are in h's home
    tr.stamp(17);
    if (w0) {
      v16f h[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f x = *hs_ptr(c, q);
          h[c][4 * q] = x[0], h[c][4 * q + 1] = x[1], h[c][4 * q + 2] = x[2], h[c][4 * q + 3] = x[3];
        }
      float eps[3];
      post_eps_tiles(h, wout, p.d.bout, eps);
      tr.stamp(18);
      PointState ps;
      ps.s = s, ps.n = n, ps.gid = 0;   // (gid: the noise was drawn by wave 7)
      pstate_load(ps_lds, pj, 32, ps, true);
      const float *s_z = reinterpret_cast<const float *>(pipe_smem + CL_Z);
      const float zr[3] = {s_z[pj], s_z[32 + pj], s_z[64 + pj]};
      const bool zok = p.mode != MODE_EPS;
      if (step_epilogue(p, ps, eps, step, t, zok ? zr : nullptr, zok ? s_z + 128 : nullptr)) break;   // (the other waves leave through the loop bound: nsteps = 1 in these modes)
      tr.stamp(19);
      pstate_store(ps_lds, pj, 32, ps, false);
      if (step + 1 < p.nsteps) enter_step(ps);
      tr.stamp(20);
    }
  }
}

// ---- k_denoise_coop16 (round 5): the co-operative kernel on SIXTEEN-point tiles (v_mfma_f32_16x16x32_bf16 / _f16).
// A 2048-point shape is 128 tiles of 16 points instead of 64 of 32: a single shape occupies 128 CUs, two shapes the whole chip, and every VALU phase of a
// tile's critical path (LayerNorms, softmax, GELU, step boundary: 10.3 k of the 13.5 k cycles of a block in k_denoise_coop) works on half the values per
// lane; an MFMA is 16 cycles instead of 32.  Same records in memory — no second pack: a 16 x 32 A fragment of this MFMA is TWO 8-byte pieces of the 32 x 32
// tile the packs hold (the K order of those tiles is the accumulator order, whose groups of four consecutive channels stay together), gathered by address.
// Layout: lane l = (point j = l & 15, group g = l >> 4); accumulator tile c (0..7), register r (0..3) = channel 16 c + 4 g + r; the B fragment of k-step s
// (32 channels) is built in-lane from tiles 2 s, 2 s + 1 (element e = register e & 3 of tile 2 s + (e >> 2)), and the gathered A fragments follow that order.
// The accumulation order over K differs from the 32 x 32 x 16 family (32 channels per MFMA instead of 16), so this variant is NOT bit-identical to the
// others: it is gated against the exact-fp32 chain and the CPU oracle at the bf16 tolerances (tests/test_gpu_small_batch16.py).
// Work of a block: phase A (wave 0: attention + LayerNorms), phase H (all 8 waves: chunks w and w + 8 of the hidden layer, both 16-unit halves of a chunk
// on one wave so that the GELU output is a complete B fragment), phase G (all 8 waves: wave w owns accumulator tile w of h; its 16 W2 fragments sit in the
// registers round 0 of phase H has freed).  The next block's attention record | c_t row | b2 are fetched by LDS-DMA during phase H into the other half of
// a double buffer.  One workgroup per CU (128 fragment registers), LDS 77 KiB.
constexpr int C16_NW = 8, C16_PTS = 16;
constexpr double C16_ROUND_MS = 1e9;   // ms per round of g_num_cus workgroups at N = 2048, T = 1000 (launch()'s cost model; 1e9 = never chosen automatically until measured)
constexpr int C16_XN = 0;                                            // 4 x 64 uint4: LN3 output, B fragments of the four k-steps
constexpr int C16_HID = C16_XN + 4 * 1024;                           // 16 x 64 uint4: GELU output, B fragment of every 32-unit chunk
constexpr int C16_AT = C16_HID + FF_CHUNKS * 1024;                   // 2 x [attention record 17 KiB | c_t row 1 KiB | b2 1 KiB]
constexpr int C16_AT_BYTES = asms_bytes(DFX_PREC_BF16) + 2048;
constexpr int C16_HS = C16_AT + 2 * C16_AT_BYTES;                    // home of h: [tile c (8)][lane] float4 = 8 KiB
constexpr int C16_CONST = C16_HS + 8 * 1024;                         // chain-invariant operands (as k_denoise_coop): W_in x-columns | pre_norm | W_out | cpart
constexpr int C16_Z = C16_CONST + 7 * 1024;                          // noise z[3][16] | posterior table row at +512
constexpr int C16_PS = C16_Z + 1024;                                 // per-point chain state: 13 x 16 floats
constexpr int C16_TOTAL = C16_PS + 1024;
static_assert(C16_TOTAL <= 160 * 1024, "LDS budget of the 16-point co-operative kernel");
static_assert(FF_CHUNKS == 2 * C16_NW && INNER == 128, "two chunks per wavefront, eight accumulator tiles");

__global__ void __launch_bounds__(C16_NW * 64, 2) k_denoise_coop16(const KParams p) {
  constexpr int PREC = DFX_PREC_BF16;
  constexpr int TSTRIDE = tile_units(PREC) * 64, AREC = asms_bytes(PREC) / 16;
  constexpr int TILE_B = TSTRIDE * 16;   // bytes per 32 x 32 tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const long long g0 = (long long)blockIdx.x * C16_PTS;
  const int s = __builtin_amdgcn_readfirstlane((int)(g0 / p.N));
  const int n = (int)(g0 - (long long)s * p.N) + j;
  const int depth = p.d.depth;
  const bool w0 = wave == 0;
  float *ps_lds = reinterpret_cast<float *>(pipe_smem + C16_PS);
  float *s_hs = reinterpret_cast<float *>(pipe_smem + C16_HS);
  uint4 *s_xn = reinterpret_cast<uint4 *>(pipe_smem + C16_XN);
  uint4 *s_hid = reinterpret_cast<uint4 *>(pipe_smem + C16_HID);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)pipe_smem);
  const unsigned voff = lane * 16;
  // byte offset of this lane's 8-byte piece inside unit 0 of a 32 x 32 tile, for the tile's rows 0..15 (half 0) / 16..31 (half 1); unit 1: + 1024
  const unsigned aoff0 = (unsigned)((j + 32 * (g & 1)) * 16 + 8 * (g >> 1)), aoff1 = aoff0 + 16 * 16;
  // index of this lane's four consecutive entries of a per-channel vector in cvec order, for accumulator tile c (channels 16 c + 4 g + 0..3)
  auto cv = [&](int c) { return (g & 1) * 64 + (c >> 1) * 16 + (c & 1) * 8 + (g >> 1) * 4; };
  // One 16 x 32 A fragment out of a 32 x 32 tile = two 8-byte pieces (elements 4 q .. 4 q + 3 of units 0 and 1, q = g >> 1).  From global memory every
  // lane loads ONE whole 16-byte unit instead — groups 0 / 1 unit 0, groups 2 / 3 unit 1 of the same row — and the two half-waves trade the piece the other
  // one needs with two v_permlane32_swap: the same bytes in half the load instructions (the first version issued 96 eight-byte loads per wavefront and block;
  // the CU's address path, shared by its eight wavefronts, took 8 k cycles per block for them).  The address space is spelled out through an integer:
  // behind pin_ptr's round trip hipcc would emit flat loads, which count against lgkmcnt as well.
  const unsigned goff0 = (unsigned)((g >> 1) * 1024 + (j + 32 * (g & 1)) * 16), goff1 = goff0 + 16 * 16;
  auto gather = [&](const char *tile, unsigned goff) -> uint4 {          // goff = goff0 (rows 0..15 of the tile) or goff1 (rows 16..31); the RAW unit: frag_fix() before use
    typedef const __attribute__((address_space(1))) unsigned long long *gp64;
    const gp64 q = (gp64)(unsigned long long)(uintptr_t)(tile + goff);
    const unsigned long long lo = q[0], hi = q[1];
    return make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
  };
  auto frag_fix = [](uint4 &f) {   // (applied where the fragment is consumed: at the load site it would wait for the data)
    const auto r0 = __builtin_amdgcn_permlane32_swap(f.x, f.z, false, false), r1 = __builtin_amdgcn_permlane32_swap(f.y, f.w, false, false);   // x, y of the upper half-wave <-> z, w of the lower
    f = make_uint4(r0[0], r1[0], r0[1], r1[1]);
  };
  auto gather_lds = [&](const unsigned char *tile, unsigned aoff) -> uint4 {   // from the attention record's LDS copy
    typedef const __attribute__((address_space(3))) unsigned long long *lp64;
    const uintptr_t la = (uintptr_t)(const __attribute__((address_space(3))) unsigned char *)tile + aoff;
    const unsigned long long a = *(lp64)la, b = *(lp64)(la + 1024);
    return make_uint4((unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32));
  };
  auto mfma_bf = [](const uint4 &a, const uint4 &b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
  };
  auto mfma_h = [](const uint4 &a, const uint4 &b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  };
  auto xg = [](float v) {   // sum over the four lane groups of a point
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
  };
  auto hs_ptr = [&](int c) -> v4f * { return reinterpret_cast<v4f *>(s_hs + (c * 64 + lane) * 4); };
  // LayerNorm statistics of h (32 values in-lane + the other three groups of the point): single pass like ln_stats_fast
  auto ln16 = [&](const v4f (&h)[8], float &mean, float &rstd) {
    float st = 0.f, qt = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) st += h[c][r], qt = fmaf(h[c][r], h[c][r], qt);
    st = xg(st), qt = xg(qt);
    mean = st * (1.0f / 128.0f);
    rstd = __builtin_amdgcn_rsqf(fmaxf(fmaf(-mean, mean, qt * (1.0f / 128.0f)), 0.f) + 1e-5f);
  };
  // normalised (affine-free) h: the bf16 B fragment of k-step k
  auto ln_frag = [&](const v4f (&h)[8], int k, float rstd, float nmr) -> uint4 {
    v8f t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = fmaf(h[2 * k + (e >> 2)][e & 3], rstd, nmr);
    return __builtin_bit_cast(uint4, __builtin_convertvector(t, v8bf));
  };

  unsigned vmask = 0;
  if (w0) {
    PointState ps0;
    ps0.live = g == 0;   // the four groups of a point hold the same state; group 0 does the stores (point_init / step_epilogue test `live` and the half-wave)
    point_init(p, ps0, s, n, ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)n, vmask);
    pstate_store(ps_lds, j, C16_PTS, ps0, true);
  }
  {   // chain-invariant small operands -> LDS
    float4 *c_winx = reinterpret_cast<float4 *>(pipe_smem + C16_CONST);
    float2 *c_pregb = reinterpret_cast<float2 *>(pipe_smem + C16_CONST + 2048);
    float4 *c_wout = reinterpret_cast<float4 *>(pipe_smem + C16_CONST + 3072);
    float *c_cp = reinterpret_cast<float *>(pipe_smem + C16_CONST + 5120);
    const int tid = threadIdx.x;
    if (tid < 128) c_winx[tid] = p.d.win_x[tid], c_pregb[tid] = p.d.pre_gb[tid], c_wout[tid] = p.d.wout[tid];
    c_cp[tid] = p.cpart[(size_t)s * NCLS * INNER + tid];
    __syncthreads();
  }
  const float4 *winx = reinterpret_cast<const float4 *>(pipe_smem + C16_CONST);
  const float2 *pregb = reinterpret_cast<const float2 *>(pipe_smem + C16_CONST + 2048);
  const float4 *wout = reinterpret_cast<const float4 *>(pipe_smem + C16_CONST + 3072);
  const uint4 *asms_s = p.as_ms + (size_t)s * depth * AREC;

  // proj_in + pre_norm of the chain state -> h's LDS home (wave 0)
  auto enter_step = [&](const PointState &ps) {
    v4f h[8];
    const float *cp = reinterpret_cast<const float *>(pipe_smem + C16_CONST + 5120) + ps.sg * INNER;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const v4f cpv = *reinterpret_cast<const v4f *>(cp + cv(c));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 w = winx[cv(c) + r];
        h[c][r] = fmaf(w.z, ps.x[2], fmaf(w.y, ps.x[1], fmaf(w.x, ps.x[0], cpv[r])));
      }
    }
    float mean, rstd;
    ln16(h, mean, rstd);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v4f o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float2 gb = pregb[cv(c) + r];
        o[r] = fmaf((h[c][r] - mean) * rstd, gb.x, gb.y);
      }
      *hs_ptr(c) = o;
    }
  };
  // the operands of block b at timestep t -> half `buf` of the double buffer: 17 KiB record | c_t row | b2 (19 pieces over waves 1..7)
  auto fetch_record = [&](int b, int t, int buf) {
    const BlockPack bp = block_pack(p, b);
    constexpr int NA = asms_bytes(PREC) / 1024;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int k = i * (C16_NW - 1) + wave - 1;
      const unsigned dst = lds0 + C16_AT + buf * C16_AT_BYTES + k * 1024;
      if (k < NA) dma1k_pinned(reinterpret_cast<const char *>(asms_s + (size_t)b * AREC) + k * 1024, voff, dst);
      else if (k == NA) dma1k_pinned(reinterpret_cast<const char *>(bp.ct + (size_t)t * CT_ROW), voff, dst);
      else if (k == NA + 1) dma1k_pinned(reinterpret_cast<const char *>(bp.bconst + BCONST_B2_OFF), voff, dst);
    }
  };

  if (w0) {
    PointState ps;
    pstate_load(ps_lds, j, C16_PTS, ps, false);
    enter_step(ps);
  } else {
    fetch_record(0, step_t(p, 0, s), 0);
  }
  uint4 R[32];   // GEMM1 fragments of the wave's two chunks (16 each); phase G: the wave's sixteen W2 fragments in R[0..15]
#ifdef DFX_TRACE   // phase stamps of wave 0 (row 0 of the trace buffer) and of wave 5 (row 1) of workgroup 0
  Tracer tr{(p.trace != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 5)) ? p.trace + (size_t)(wave ? 1 : 0) * p.trace_cap : nullptr, p.trace_cap, 0};
#else
  Tracer tr;
#endif
  int seq = 0;
  for (int step = 0; step < p.nsteps; ++step) {
    const int t = step_t(p, step, s);
    for (int b = 0; b < depth; ++b, ++seq) {
      const BlockPack bp = block_pack(p, b);
      const unsigned char *at = pipe_smem + C16_AT + (seq & 1) * C16_AT_BYTES;
      if (!w0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this block's record (requested a block ago) has landed
      // GEMM1 fragments of chunks `wave` (R[0..15]) and `wave + 8` (R[16..31]): [a | g] x 4 k-steps x 2 halves, requested by every wavefront in front of
      // barrier 0.  (The CU's address path takes ~16 cycles per wavefront-wide load and is shared by the eight wavefronts: 256 loads = 4 k cycles at the top
      // of every block.  Requesting waves 1..7's behind the barrier, under wave 0's phase A, was tried: phase A then takes 12.6 k cycles instead of 3.8 k —
      // its LDS reads and the returning load data share a path — 56 ms per chain instead of 31.8; profiles/r05_small_batch_sweep.txt.)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const char *ck = pin_ptr(reinterpret_cast<const char *>(bp.chunks + (size_t)(r * C16_NW + wave) * CHUNK_TILES * TSTRIDE));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          R[16 * r + 4 * k + 0] = gather(ck + (0 + k) * TILE_B, goff0), R[16 * r + 4 * k + 1] = gather(ck + (4 + k) * TILE_B, goff0);
          R[16 * r + 4 * k + 2] = gather(ck + (0 + k) * TILE_B, goff1), R[16 * r + 4 * k + 3] = gather(ck + (4 + k) * TILE_B, goff1);
        }
      }
      tr.stamp(10);
      __syncthreads();   // 0: this block's record is in LDS; h's home holds the previous block's result
      tr.stamp(11);
      if (w0) {          // ---- phase A
        v4f h[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) h[c] = *hs_ptr(c);
        // sim = A_s LN2(h) + sbias: rows 16 rt + 4 g + r = key r of head 4 rt + g; the B fragments are consumed as they are made
        const float *sb = reinterpret_cast<const float *>(at + 8 * TILE_B);
        v4f sim[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) sim[rt] = *reinterpret_cast<const v4f *>(sb + (g & 1) * 16 + 8 * rt + 4 * (g >> 1));
        {
          float mean, rstd;
          ln16(h, mean, rstd);
          const float nmr = -mean * rstd;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint4 xk = ln_frag(h, k, rstd, nmr);
            sim[0] = mfma_bf(gather_lds(at + k * TILE_B, aoff0), xk, sim[0]);
            sim[1] = mfma_bf(gather_lds(at + k * TILE_B, aoff1), xk, sim[1]);
          }
        }
        v8f pf;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {   // masked softmax over the four keys (attention.py:195-198), in-lane
          float sj[4], e[4], sum = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) sj[q] = (vmask >> q) & 1u ? sim[rt][q] : -3.402823466e38f;
          const float m = fmaxf(fmaxf(sj[0], sj[1]), fmaxf(sj[2], sj[3]));
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] = __expf(sj[q] - m), sum += e[q];
          const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
          for (int q = 0; q < 4; ++q) pf[4 * rt + q] = e[q] * inv;
        }
        const uint4 pa = __builtin_bit_cast(uint4, __builtin_convertvector(pf, v8bf));
        const float *ct = reinterpret_cast<const float *>(at + asms_bytes(PREC));
#pragma unroll
        for (int c = 0; c < 8; ++c) {   // h += M_s P + c_t
          h[c] = mfma_bf(gather_lds(at + (4 + (c >> 1)) * TILE_B, (c & 1) ? aoff1 : aoff0), pa, h[c]);
          const v4f cc = *reinterpret_cast<const v4f *>(ct + cv(c));
          h[c] += cc;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) *hs_ptr(c) = h[c];
        {
          float mean, rstd;
          ln16(h, mean, rstd);
          const float nmr = -mean * rstd;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            uint4 xk = ln_frag(h, k, rstd, nmr);
            if (k == 3 && p.d.w1_fold && g == 3) xk.w = (xk.w & 0x0000ffffu) | 0x3f800000u;   // channel 127 = tile 7, row 15: element 7 of k-step 3 carries the constant 1 (bias_slot_one)
            s_xn[k * 64 + lane] = xk;
          }
        }
      } else if (wave == C16_NW - 1 && b == depth - 1 && p.mode != MODE_EPS) {   // the step's noise and posterior coefficients, ready for wave 0's epilogue
        float z[3];
        if (p.noise) {
#pragma unroll
          for (int i = 0; i < 3; ++i) z[i] = p.noise[(((size_t)step * p.B + s) * 3 + i) * p.N + n];
        } else {
          philox_normal3(p.seed, ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)n, (unsigned)t, 0u, z);
        }
        float *s_z = reinterpret_cast<float *>(pipe_smem + C16_Z);
        if (g == 0) s_z[j] = z[0], s_z[16 + j] = z[1], s_z[32 + j] = z[2];
        if (lane < 8) s_z[128 + lane] = p.d.tab[(size_t)t * 8 + lane];
      }
      tr.stamp(12);
      __syncthreads();   // 1: xn3 of this block and h are in LDS
      tr.stamp(13);
      // ---- phase H.  The next block's record travels meanwhile (into the other half of the double buffer: nobody reads that half before barrier 0)
      if (!w0) {
        const int nb = b + 1 < depth ? b + 1 : 0, nstep = b + 1 < depth ? step : step + 1;
        if (nstep < p.nsteps) fetch_record(nb, step_t(p, nstep, s), (seq + 1) & 1);
      }
      uint4 xq[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) xq[k] = s_xn[k * 64 + lane];
      const float *b1t = bp.bconst;   // [u][part][hf][16] in the 32-wide register order (read only by engines without the W1 bias fold)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int u = r * C16_NW + wave;
#pragma unroll
        for (int i = 0; i < 16; ++i) frag_fix(R[16 * r + i]);
        v4f a[2], gg[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          if (p.d.w1_fold) {
            a[hh] = v4f{0.f, 0.f, 0.f, 0.f}, gg[hh] = a[hh];
          } else {   // unit 16 hh + 4 g + q of the chunk = entry (g & 1) * 16 + 8 hh + 4 (g >> 1) + q of the [hf][16] table
            a[hh] = *reinterpret_cast<const v4f *>(b1t + u * 64 + (g & 1) * 16 + 8 * hh + 4 * (g >> 1));
            gg[hh] = *reinterpret_cast<const v4f *>(b1t + u * 64 + 32 + (g & 1) * 16 + 8 * hh + 4 * (g >> 1));
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            a[hh] = mfma_bf(R[16 * r + 4 * k + 2 * hh + 0], xq[k], a[hh]);
            gg[hh] = mfma_bf(R[16 * r + 4 * k + 2 * hh + 1], xq[k], gg[hh]);
          }
        }
        if (r == 0) {   // round 0's registers are free: this wave's W2 fragments — accumulator tile `wave`, all sixteen chunks (W2 of chunk c sits in FF record c + FF_SKEW, tiles 8..11)
          const char *w2 = pin_ptr(reinterpret_cast<const char *>(bp.chunks + (size_t)FF_SKEW * CHUNK_TILES * TSTRIDE) + (8 + (wave >> 1)) * TILE_B);
          const unsigned ao = (wave & 1) ? goff1 : goff0;
#pragma unroll
          for (int c = 0; c < FF_CHUNKS; ++c) R[c] = gather(w2 + (size_t)c * CHUNK_TILES * TILE_B, ao);
        }
        // packed-fp16 GELU on the eight values of the two halves (gelu16_f16_math on four pairs): hid = a gelu(g) with the pack's scales
        h2 ap[4], gp[4], y[4], z[4], rr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gp[i] = pk_f16(gg[i >> 1][2 * (i & 1)], gg[i >> 1][2 * (i & 1) + 1]), ap[i] = pk_f16(a[i >> 1][2 * (i & 1)], a[i >> 1][2 * (i & 1) + 1]);
#define DFX_ST4(expr) _Pragma("unroll") for (int i = 0; i < 4; ++i) { expr; }
        DFX_ST4(z[i] = __builtin_elementwise_fma(gp[i], gp[i], h2c(-1.62f)));
        DFX_ST4(y[i] = ap[i] * gp[i]);
        DFX_ST4(z[i] = __builtin_elementwise_min(z[i], h2c(1.62f)));
        DFX_ST4(rr[i] = __builtin_elementwise_fma(z[i], h2c(-0.0011402554f), h2c(0.0057853916f)));
        DFX_ST4(rr[i] = __builtin_elementwise_fma(rr[i], z[i], h2c(-0.0158536041f)));
        DFX_ST4(rr[i] = __builtin_elementwise_fma(rr[i], z[i], h2c(0.0409006897f)));
        DFX_ST4(rr[i] = __builtin_elementwise_fma(rr[i], z[i], h2c(-0.1098130657f)));
        DFX_ST4(rr[i] = __builtin_elementwise_fma(rr[i], z[i], h2c(0.3885767652f)));
        DFX_ST4(asm("v_pk_fma_f16 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(z[i]) : "v"(gp[i]), "v"(rr[i])));
        DFX_ST4(y[i] = y[i] * z[i]);
#undef DFX_ST4
        s_hid[u * 64 + lane] = make_uint4(__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1]), __builtin_bit_cast(unsigned, y[2]),
                                          __builtin_bit_cast(unsigned, y[3]));
      }
      tr.stamp(14);
      __syncthreads();   // 2: hid of all chunks is in LDS
      tr.stamp(15);
      // ---- phase G: wave w accumulates tile w of h over the sixteen chunks, + b2
      {
        v4f ht = *hs_ptr(wave);
#pragma unroll
        for (int u = 0; u < FF_CHUNKS; ++u) frag_fix(R[u]);
#pragma unroll
        for (int u = 0; u < FF_CHUNKS; ++u) ht = mfma_h(R[u], s_hid[u * 64 + lane], ht);
        const float *b2 = reinterpret_cast<const float *>(at + asms_bytes(PREC) + 1024);
        const v4f bb = *reinterpret_cast<const v4f *>(b2 + cv(wave));
        *hs_ptr(wave) = ht + bb;
      }
    }
    tr.stamp(16);
    __syncthreads();   // the last block's tiles are in h's home
    tr.stamp(17);
    if (w0) {
      v4f h[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) h[c] = *hs_ptr(c);
      float mean, rstd;
      ln16(h, mean, rstd);
      const float nmr = -mean * rstd;
      float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 w = wout[cv(c) + r];
          const float v = fmaf(h[c][r], rstd, nmr);
          e0 = fmaf(w.x, v, e0), e1 = fmaf(w.y, v, e1), e2 = fmaf(w.z, v, e2);
        }
      const float eps[3] = {xg(e0) + p.d.bout[0], xg(e1) + p.d.bout[1], xg(e2) + p.d.bout[2]};
      PointState ps;
      ps.s = s, ps.n = n, ps.gid = 0, ps.live = g == 0;
      pstate_load(ps_lds, j, C16_PTS, ps, true);
      const float *s_z = reinterpret_cast<const float *>(pipe_smem + C16_Z);
      const float zr[3] = {s_z[j], s_z[16 + j], s_z[32 + j]};
      const bool zok = p.mode != MODE_EPS;
      tr.stamp(18);
      if (step_epilogue(p, ps, eps, step, t, zok ? zr : nullptr, zok ? s_z + 128 : nullptr)) break;
      tr.stamp(19);
      pstate_store(ps_lds, j, C16_PTS, ps, false);
      if (step + 1 < p.nsteps) enter_step(ps);
      tr.stamp(20);
    }
  }
}

// ---- k_denoise_coop2: the co-operative kernel with TWO 32-point tiles per workgroup (round 4; batches of 5 .. 8 shapes at N = 2048, i.e. one to two
// tiles per CU, where k_denoise_coop needs two rounds of workgroups and k_denoise_pipe<2> leaves half of every CU idle).  Same phases, same device
// functions, same MFMA order per accumulator as k_denoise_coop — bit-identical results — with the work of a block cut this way:
//   phase A   waves 0 and 1 (different SIMDs): attention + LayerNorms on tile 0 / tile 1, side by side
//   phase H   all 8 waves, chunks w and w + 8: every GEMM1 fragment (registers) feeds BOTH tiles' MFMAs; two GELUs per chunk
//   phase G   waves 0..3 own the four output tiles of tile 0, waves 4..7 those of tile 1 — the eight waves carry all of W2 for their output tile in
//             the 32 fragment registers phase H has freed (chunks 0..7 requested behind round 0, chunks 8..15 behind round 1's MFMAs: the second
//             tile's GELU covers their latency), so k_denoise_coop's 64 KiB LDS copy of W2 is gone and the second tile's h home, xn3 and GELU
//             outputs take its place (153 KiB)
// The next block's attention record, c_t row and block constants are requested behind barrier 2 (the record is only read in phase A) by all waves.
constexpr int C2_XN = 0;                                           // 2 tiles x 8 x 64 uint4
constexpr int C2_HID = C2_XN + 2 * 8 * 1024;                       // 2 tiles x [16][2][64] uint4
constexpr int C2_BC = C2_HID + 2 * FF_CHUNKS * 2048;               // 2 x block constants, by block parity
constexpr int C2_AT = C2_BC + 2 * BCONST_BYTES;                    // attention record + c_t row (one shape per workgroup)
constexpr int C2_HS = C2_AT + asms_bytes(DFX_PREC_BF16) + 1024;    // 2 tiles x 16 KiB: h's home
constexpr int C2_CONST = C2_HS + 2 * 16 * 1024;                    // chain-invariant operands (as k_denoise_coop)
constexpr int C2_Z = C2_CONST + 7 * 1024;                          // 2 tiles x (noise z[3][32] | posterior table row at +512)
constexpr int C2_PS = C2_Z + 2 * 1024;                             // 2 tiles x per-point chain state
constexpr int C2_TOTAL = C2_PS + 2 * 2048;
static_assert(C2_TOTAL <= 160 * 1024, "LDS budget of the two-tile co-operative kernel");

__global__ void __launch_bounds__(COOP_NW * 64, 2) k_denoise_coop2(const KParams p) {
  constexpr int PREC = DFX_PREC_BF16;
  constexpr int TSTRIDE = tile_units(PREC) * 64, AREC = asms_bytes(PREC) / 16;
  uint4 *s_at = reinterpret_cast<uint4 *>(pipe_smem + C2_AT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  const long long g0 = (long long)blockIdx.x * 64;
  const int s = __builtin_amdgcn_readfirstlane((int)(g0 / p.N));        // both tiles belong to shape s (N % 64 == 0: launch())
  const int nbase = (int)(g0 - (long long)s * p.N);
  const int depth = p.d.depth;
  const bool arole = wave < 2;                 // phase A / step boundary: wave w works on tile w
  const int gt = wave >> 2, ct = wave & 3;     // phase G: output tile ct of tile gt
  auto xn_base = [&](int t) { return reinterpret_cast<uint4 (*)[64]>(pipe_smem + C2_XN + t * 8 * 1024); };
  auto hid_base = [&](int t) { return reinterpret_cast<uint4 (*)[2][64]>(pipe_smem + C2_HID + t * FF_CHUNKS * 2048); };
  auto hs_ptr = [&](int t, int c, int q) -> v4f * { return reinterpret_cast<v4f *>(pipe_smem + C2_HS + t * 16 * 1024) + (c * 4 + q) * 64 + lane; };
  float *ps_lds = reinterpret_cast<float *>(pipe_smem + C2_PS + (wave & 1) * 2048);   // (waves 0, 1)
  unsigned vmask = 0;
  if (arole) {
    PointState ps0;
    const int n = nbase + wave * 32 + pj;
    point_init(p, ps0, s, n, ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)n, vmask);
    pstate_store(ps_lds, pj, 32, ps0, true);
  }
  const uint4 *asms_s = p.as_ms + (size_t)s * depth * AREC;
  {   // chain-invariant small operands -> LDS
    float4 *c_winx = reinterpret_cast<float4 *>(pipe_smem + C2_CONST);
    float2 *c_pregb = reinterpret_cast<float2 *>(pipe_smem + C2_CONST + 2048);
    float4 *c_wout = reinterpret_cast<float4 *>(pipe_smem + C2_CONST + 3072);
    float *c_cp = reinterpret_cast<float *>(pipe_smem + C2_CONST + 5120);
    const int tid = threadIdx.x;
    if (tid < 128) c_winx[tid] = p.d.win_x[tid], c_pregb[tid] = p.d.pre_gb[tid], c_wout[tid] = p.d.wout[tid];
    c_cp[tid] = p.cpart[(size_t)s * NCLS * INNER + tid];
    __syncthreads();
  }
  const float4 *winx = reinterpret_cast<const float4 *>(pipe_smem + C2_CONST) + hf * 64;
  const float2 *pregb = reinterpret_cast<const float2 *>(pipe_smem + C2_CONST + 2048) + hf * 64;
  const float4 *wout = reinterpret_cast<const float4 *>(pipe_smem + C2_CONST + 3072) + hf * 64;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)pipe_smem);
  const unsigned voff = lane * 16;

  uint4 R[32];   // phase H: GEMM1 fragments of round 0 (R[0..15]) / round 1 (R[16..31]); phase G: W2 fragments of the own output tile, chunk u in R[2 u], R[2 u + 1]
  auto load_w1 = [&](int base, const uint4 *chunks, int u) {
    const uint4 *ck = reinterpret_cast<const uint4 *>(pin_ptr(reinterpret_cast<const char *>(chunks + (size_t)u * CHUNK_TILES * TSTRIDE))) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      R[base + 4 * c + 0] = ck[(0 + c) * TSTRIDE], R[base + 4 * c + 1] = ck[(0 + c) * TSTRIDE + 64];
      R[base + 4 * c + 2] = ck[(4 + c) * TSTRIDE], R[base + 4 * c + 3] = ck[(4 + c) * TSTRIDE + 64];
    }
  };
  auto load_w2 = [&](int first, const uint4 *chunks) {   // W2 of chunks first .. first + 7, output tile ct (W2 of chunk c sits in FF record c + FF_SKEW, tiles 8..11)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 *ck = reinterpret_cast<const uint4 *>(pin_ptr(reinterpret_cast<const char *>(chunks + (size_t)(first + c + FF_SKEW) * CHUNK_TILES * TSTRIDE + (8 + ct) * TSTRIDE))) + lane;
      R[2 * (first + c)] = ck[0], R[2 * (first + c) + 1] = ck[64];
    }
  };
  // attention record, c_t row and block constants of block b at time t -> LDS: 23 pieces over the 8 waves
  auto stage_block = [&](int b, int t, int parity) {
    const BlockPack bq = block_pack(p, b);
    constexpr int NA = asms_bytes(PREC) / 1024, NB = BCONST_BYTES / 1024;
#pragma unroll
    for (int i = 0; i < (NA + 1 + NB + COOP_NW - 1) / COOP_NW; ++i) {
      const int k = i * COOP_NW + wave;
      if (k < NA) dma1k_pinned(reinterpret_cast<const char *>(asms_s + (size_t)b * AREC) + k * 1024, voff, lds0 + C2_AT + k * 1024);
      else if (k == NA) dma1k_pinned(reinterpret_cast<const char *>(bq.ct + (size_t)t * CT_ROW), voff, lds0 + C2_AT + NA * 1024);
      else if (k < NA + 1 + NB) dma1k_pinned(reinterpret_cast<const char *>(bq.bconst) + (k - NA - 1) * 1024, voff, lds0 + C2_BC + parity * BCONST_BYTES + (k - NA - 1) * 1024);
    }
  };
  auto enter_step = [&](const PointState &ps) {   // proj_in + pre_norm of the chain state -> h's home (waves 0, 1)
    v16f h[4];
    proj_in_prenorm_tiles(h, ps.x, reinterpret_cast<const float *>(pipe_smem + C2_CONST + 5120) + ps.sg * INNER + hf * 64, winx, pregb);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) *hs_ptr(wave, c, q) = v4f{h[c][4 * q], h[c][4 * q + 1], h[c][4 * q + 2], h[c][4 * q + 3]};
  };
  if (arole) {
    PointState ps;
    pstate_load(ps_lds, pj, 32, ps, false);
    enter_step(ps);
  }
  int seq = 0;
  if (p.nsteps > 0) stage_block(0, step_t(p, 0, s), 0);
  for (int step = 0; step < p.nsteps; ++step) {
    const int t = step_t(p, step, s);
    for (int b = 0; b < depth; ++b, ++seq) {
      const BlockPack bp = block_pack(p, b);
      float *s_bc = reinterpret_cast<float *>(pipe_smem + C2_BC + (seq & 1) * BCONST_BYTES);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the block's operands (requested a phase G ago)
      load_w1(0, bp.chunks, wave);   // round 0: in flight through phase A
      __syncthreads();   // 0: attention record, c_t, block constants of this block are in LDS; h's homes hold the previous block's result
      if (arole) {       // ---- phase A: wave w on tile w
        v16f h[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4f x = *hs_ptr(wave, c, q);
            h[c][4 * q] = x[0], h[c][4 * q + 1] = x[1], h[c][4 * q + 2] = x[2], h[c][4 * q + 3] = x[3];
          }
        const uint4 *rec = s_at + lane;
        attention<PREC>(h, rec, reinterpret_cast<const float *>(s_at + 8 * TSTRIDE) + hf * 16, reinterpret_cast<const float *>(s_at + AREC) + hf * 64, vmask);
        Act<PREC> xo[4];
        ln_to_act<PREC>(h, xo);
        bias_slot_one(xo, hf);
        uint4 (*s_xn)[64] = xn_base(wave);
#pragma unroll
        for (int c = 0; c < 4; ++c) s_xn[2 * c][lane] = __builtin_bit_cast(uint4, xo[c].f[0]), s_xn[2 * c + 1][lane] = __builtin_bit_cast(uint4, xo[c].f[1]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) *hs_ptr(wave, c, q) = v4f{h[c][4 * q], h[c][4 * q + 1], h[c][4 * q + 2], h[c][4 * q + 3]};
      } else if (wave >= COOP_NW - 2 && b == depth - 1 && p.mode != MODE_EPS) {   // the step's noise and posterior coefficients: wave 6 for tile 0, wave 7 for tile 1
        const int tz = wave - (COOP_NW - 2), n = nbase + tz * 32 + pj;
        float z[3];
        if (p.noise) {
#pragma unroll
          for (int i = 0; i < 3; ++i) z[i] = p.noise[(((size_t)step * p.B + s) * 3 + i) * p.N + n];
        } else {
          philox_normal3(p.seed, ((unsigned long long)p.shape0 + (unsigned)s) * (unsigned)p.N + (unsigned)n, (unsigned)t, 0u, z);
        }
        float *s_z = reinterpret_cast<float *>(pipe_smem + C2_Z + tz * 1024);
        if (hf == 0) s_z[pj] = z[0], s_z[32 + pj] = z[1], s_z[64 + pj] = z[2];
        if (lane < 8) s_z[128 + lane] = p.d.tab[(size_t)t * 8 + lane];
      }
      __syncthreads();   // 1: xn3 of both tiles and h are in LDS
      // ---- phase H: every wave, chunks `wave` and `wave + 8`, both tiles per fragment set
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int u = r * COOP_NW + wave, cur = r * 16;
        if (r == 0) load_w1(16, bp.chunks, u + COOP_NW);
        else load_w2(0, bp.chunks);   // round 0's registers are free: W2 of chunks 0..7
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          uint4 (*s_xn)[64] = xn_base(tl);
          v16f a = zero16(), g = zero16();   // b1' rides on the constant-one K slot (bias_slot_one)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const v8bf x0 = __builtin_bit_cast(v8bf, s_xn[2 * c][lane]), x1 = __builtin_bit_cast(v8bf, s_xn[2 * c + 1][lane]);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(R[cur + 4 * c + 0]), x0, a, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(R[cur + 4 * c + 2]), x0, g, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(R[cur + 4 * c + 1]), x1, a, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(R[cur + 4 * c + 3]), x1, g, 0, 0, 0);
          }
          if (r == 1 && tl == 1) {   // round 1's fragments are done with: W2 of chunks 8..15 travels under the last GELU
            __builtin_amdgcn_sched_barrier(0);
            load_w2(8, bp.chunks);
            __builtin_amdgcn_sched_barrier(0);
          }
          h2 aa[8], gg[8];
          HidAct hid;
          gelu16_f16_cvt(a, g, aa, gg);
          gelu16_f16_math(aa, gg, hid);
          uint4 (*s_hid)[2][64] = hid_base(tl);
          s_hid[u][0][lane] = hid.f[0], s_hid[u][1][lane] = hid.f[1];
        }
      }
      __syncthreads();   // 2: hid of all chunks of both tiles is in LDS; nobody reads the attention record any more
      {                  // the next block's operands (next step's c_t row behind the last block)
        const int nb = b + 1 < depth ? b + 1 : 0;
        if (b + 1 < depth) stage_block(nb, t, (seq + 1) & 1);
        else if (step + 1 < p.nsteps) stage_block(0, step_t(p, step + 1, s), (seq + 1) & 1);
      }
      // ---- phase G: wave (gt, ct) accumulates output tile ct of tile gt, chunk after chunk (the accumulation order of the pipelined kernel), + b2
      {
        v16f ht;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f x = *hs_ptr(gt, ct, q);
          ht[4 * q] = x[0], ht[4 * q + 1] = x[1], ht[4 * q + 2] = x[2], ht[4 * q + 3] = x[3];
        }
        uint4 hq[4][2];   // hid fragments: a ring three chunks deep, fenced per chunk (see k_denoise_coop)
        unsigned hoff = C2_HID + gt * FF_CHUNKS * 2048 + lane * 16;
        asm volatile("" : "+v"(hoff));
        const uint4 *hidb = reinterpret_cast<const uint4 *>(pipe_smem + hoff);
        auto fetch = [&](int u) { hq[u & 3][0] = hidb[u * 128], hq[u & 3][1] = hidb[u * 128 + 64]; };
        fetch(0), fetch(1), fetch(2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < FF_CHUNKS; ++u) {
          if (u + 3 < FF_CHUNKS) fetch(u + 3);
          ht = mma_hid(R[2 * u], hq[u & 3][0], ht);
          ht = mma_hid(R[2 * u + 1], hq[u & 3][1], ht);
          __builtin_amdgcn_sched_barrier(0);
        }
        const float *b2 = s_bc + BCONST_B2_OFF + hf * 64 + ct * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f bb = *reinterpret_cast<const v4f *>(b2 + 4 * q);
          *hs_ptr(gt, ct, q) = v4f{ht[4 * q] + bb[0], ht[4 * q + 1] + bb[1], ht[4 * q + 2] + bb[2], ht[4 * q + 3] + bb[3]};
        }
      }
    }
    __syncthreads();   // the last block's tiles are in h's homes
    if (arole) {
      v16f h[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f x = *hs_ptr(wave, c, q);
          h[c][4 * q] = x[0], h[c][4 * q + 1] = x[1], h[c][4 * q + 2] = x[2], h[c][4 * q + 3] = x[3];
        }
      float eps[3];
      post_eps_tiles(h, wout, p.d.bout, eps);
      PointState ps;
      ps.s = s, ps.n = nbase + wave * 32 + pj, ps.gid = 0;   // (gid: the noise was drawn by waves 6, 7)
      pstate_load(ps_lds, pj, 32, ps, true);
      const float *s_z = reinterpret_cast<const float *>(pipe_smem + C2_Z + wave * 1024);
      const float zr[3] = {s_z[pj], s_z[32 + pj], s_z[64 + pj]};
      const bool zok = p.mode != MODE_EPS;
      if (step_epilogue(p, ps, eps, step, t, zok ? zr : nullptr, zok ? s_z + 128 : nullptr)) break;   // (the other waves leave through the loop bound: nsteps = 1 in these modes)
      pstate_store(ps_lds, pj, 32, ps, false);
      if (step + 1 < p.nsteps) enter_step(ps);
    }
  }
}

// q_sample (anchored_diffusion.py:148-173): x_t = sqrt_acp[t] (x0 - a) + a + sqrt_1m_acp[t] L noise, per-shape t,
// anchors / variances of the point's part from the shape context (learn_anchor, learn_variance)
__global__ void k_q_sample(const float *__restrict__ part, const float *__restrict__ qtab, const int32_t *__restrict__ seg,
                           const int32_t *__restrict__ t, const float *__restrict__ x0, const float *__restrict__ noise,
                           float *__restrict__ out, int N, int T, long long total) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n = i % N, c = (i / N) % 3;
  const long long b = i / ((long long)3 * N);
  const int sg = seg[b * N + n];
  int tt = t[b];
  tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
  const float a = part[b * 32 + c * 4 + sg], L = sqrtf(part[b * 32 + 12 + c * 4 + sg]);
  out[i] = qtab[tt * 2] * (x0[i] - a) + a + qtab[tt * 2 + 1] * L * noise[i];
}

// ((target - pred)^2 * flags).mean(1).sum() / flags.sum()  (anchored_diffusion.py:840-847); float64 accumulation
__global__ void k_masked_mse(const float *__restrict__ target, const float *__restrict__ pred, const float *__restrict__ flags,
                             double *__restrict__ acc, int N, long long total) {
  double s = 0.0, f = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / N;
    const int n = i % N;
    float m = 0.f;
    const float fl = flags ? flags[i] : 1.f;
    for (int c = 0; c < 3; ++c) {
      const float dlt = target[(b * 3 + c) * N + n] - pred[(b * 3 + c) * N + n];
      m += dlt * dlt * fl;
    }
    s += (double)(flags ? m / 3.0f : m);
    f += (double)fl;
  }
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o), f += __shfl_xor(f, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(acc, s), atomicAdd(acc + 1, f);
}
__global__ void k_mse_finish(const double *acc, float *loss, int has_flags, double count) {
  *loss = (float)(has_flags ? acc[0] / acc[1] : acc[0] / count);
}

bool g_force_direct = false;
int g_force_nw = 0;       // debug: wavefronts per workgroup of the pipelined kernel (0 = by batch size)
int g_num_cus = 256;
unsigned long long *g_trace = nullptr;
int g_trace_cap = 0;

int launch(const dfx_denoiser *d, const void *shape_ctx, KParams &p, hipStream_t st) {
  ShapeCtxView v;
  shape_ctx_view(&v, const_cast<void *>(shape_ctx), p.B, d->dev.depth, d->dev.prec);
  p.d = d->dev;
  p.part = v.part;
  p.cpart = v.cpart;
  p.as_ms = v.as_ms;
  p.trace = g_trace;
  p.trace_cap = g_trace_cap;
  constexpr int NW = 4;
  const int fnw = g_force_nw == 161 ? 0 : g_force_nw;   // 161 = automatic, with k_denoise_coop16 ruled out (A/B of the launcher's choice)
  const long long waves = ((long long)p.B * p.N) / 32;
  const long long grid = (waves + NW - 1) / NW;
  if (grid > 0x7fffffffLL) return set_error(DFX_ERR_INVALID_ARG, "denoiser: B*N too large");
  // Which kernel: the pipelined one works on tiles of nw x 32 points of one shape (a partial last tile idles whole wavefronts) with
  // nw = 8, 4 or 2 wavefronts per workgroup — the per-wave instruction stream and the results are the same, fewer wavefronts spread a
  // small batch over more CUs; the co-operative one (k_denoise_coop) puts eight wavefronts on ONE 32-point tile.  One workgroup per CU
  // either way, so a launch runs in rounds of g_num_cus workgroups, and the cheapest estimate wins.  Per-round costs as measured at
  // N = 2048, T = 1000 (profiles/r02_small_batch_sweep.txt; only their ratios matter): co-operative 32.7 ms, pipelined 86.5 / 88.5 /
  // 93.5 ms for nw = 2 / 4 / 8, times 1 + 0.36 L^4 for a round that fills the fraction L of the chip's wavefront slots (the power cap).
  auto tiles = [&](int nw) { return (long long)((p.N + nw * 32 - 1) / (nw * 32)); };
  auto rounds_cost = [&](long long wgs, double base, double fill_per_wg) {
    const long long full = wgs / g_num_cus, rest = wgs % g_num_cus;
    auto f = [](double L) { return 1.0 + 0.36 * L * L * L * L; };
    return base * (full * f(g_num_cus * fill_per_wg) + (rest ? f(rest * fill_per_wg) : 0.0));
  };
  // (a bf16 engine without the W1 bias fold — DenoiserDev::w1_fold = 0: every hidden channel is an outlier of some block's W1', or the debug
  // switch — has the plain pack, which only the direct kernel and k_denoise_coop16 read: the other chain kernels take b1' from slot 127)
  const bool bf16 = d->dev.prec == DFX_PREC_BF16 && !g_force_direct && d->dev.w1_fold;
  const bool f32 = d->dev.prec == DFX_PREC_F32 && !g_force_direct;
  int nw = PIPE_NW;
  double best = 1e300;
  for (int c = 8; c >= 2; c >>= 1) {
    if (tiles(c) * c * 32 > 3LL * p.N && c > 2) continue;   // padding of a small shape
    const double cost = rounds_cost(tiles(c) * p.B, c == 8 ? 93.5 : c == 4 ? 88.5 : 86.5, c / (8.0 * g_num_cus));
    if (cost < best) best = cost, nw = c;
  }
  const bool pipe2 = bf16 && fnw == 64 && tiles(8) * 256 <= 3LL * p.N;   // two tiles per wavefront (k_denoise_pipe2): 256-point workgroup tiles
  if (fnw > 1 && fnw != 64 && fnw != 16 && fnw < 160) nw = fnw;
  if (pipe2) nw = 8;
  const long long wpg = tiles(nw);
  // (~3x faster per point than the direct kernel: taken unless the padding of a small shape eats that factor)
  const bool pipe = bf16 && wpg * nw * 32 <= 3LL * p.N;
  // the exact-fp32 chain: same tiling; ~3x the direct kernel's rate per point, so a padded small shape may still take it
  const bool pipe_f32 = f32 && fnw != 1 && wpg * nw * 32 <= 3LL * p.N;
  const bool coop = bf16 && !pipe2 && (fnw == 1 || (fnw == 16 && p.N % 64 != 0) || (fnw == 0 && (pipe ? rounds_cost(waves, 32.7, 0.0) < best : waves <= g_num_cus)));   // (16 = two tiles per workgroup: needs N % 64 == 0, else this one)
  // two tiles per co-operative workgroup (k_denoise_coop2; dfx_debug_pipe_waves(16) forces it): 48.2 ms per round of g_num_cus workgroups at N = 2048,
  // T = 1000 — between one and two rounds of k_denoise_coop (B = 5 .. 8 shapes of 2048 points) the cheapest
  const double coop_cost = rounds_cost(waves, 32.7, 0.0), coop2_cost = rounds_cost((waves + 1) / 2, 48.2, 0.0);
  const bool coop2 = bf16 && !pipe2 && p.N % 64 == 0 && (fnw == 16 || (fnw == 0 && coop2_cost < coop_cost && (!pipe || coop2_cost < best)));
  // 16-point tiles (k_denoise_coop16, round 5): 2 x the workgroups of k_denoise_coop at about half the time per round — the fastest choice while the batch
  // is at most two rounds of it (B <= 4 shapes of 2048 points); works with either W1 pack.  dfx_debug_pipe_waves(160) forces it, (161) rules it out.
  const long long tiles16 = ((long long)p.B * p.N) / 16;
  const double coop16_cost = rounds_cost(tiles16, C16_ROUND_MS, 0.0);
  const bool coop16 = d->dev.prec == DFX_PREC_BF16 && !g_force_direct && !pipe2 &&
                      (fnw == 160 || (fnw == 0 && g_force_nw != 161 && coop16_cost < (coop2 ? coop2_cost : coop ? coop_cost : bf16 ? best : 3.0 * best)));   // (an engine without the fold falls back to the direct kernel, ~3x the pipelined estimate: ADVICE r5)
  if (pipe || coop || coop2 || pipe_f32 || coop16) {
    static PerDeviceOnce attrs;
    DFX_HIP_TRY(attrs.run([] {
      hipError_t e = set_max_lds(reinterpret_cast<const void *>(k_denoise_pipe<8>), PipeCfg<8>::L_TOTAL);
      if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_denoise_pipe<4>), PipeCfg<4>::L_TOTAL);
      if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_denoise_pipe<2>), PipeCfg<2>::L_TOTAL);
      if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_denoise_coop), CL_TOTAL);
      if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_denoise_coop2), C2_TOTAL);
      if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_denoise_coop16), C16_TOTAL);
      if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_denoise_pipe2), P2_LDS);
      if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_denoise_pipe_f32<8>), PipeCfg<8>::L_TOTAL);
      if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_denoise_pipe_f32<4>), PipeCfg<4>::L_TOTAL);
      if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_denoise_pipe_f32<2>), PipeCfg<2>::L_TOTAL);
      return e;
    }));
  }
  EventTimer tm;
  tm.begin(st);
  const char *variant;
  if (coop16) variant = "k_denoise_coop16", k_denoise_coop16<<<(int)tiles16, C16_NW * 64, C16_TOTAL, st>>>(p);
  else if (pipe2) variant = "k_denoise_pipe2", k_denoise_pipe2<<<(int)(wpg * p.B), P2_NW * 64, P2_LDS, st>>>(p);
  else if (coop2 && !(fnw == 1)) variant = "k_denoise_coop2", k_denoise_coop2<<<(int)(waves / 2), COOP_NW * 64, C2_TOTAL, st>>>(p);
  else if (coop) variant = "k_denoise_coop", k_denoise_coop<<<(int)waves, COOP_NW * 64, CL_TOTAL, st>>>(p);
  else if (pipe && nw == 8) variant = "k_denoise_pipe<8>", k_denoise_pipe<8><<<(int)(wpg * p.B), 8 * 64, PipeCfg<8>::L_TOTAL, st>>>(p);
  else if (pipe && nw == 4) variant = "k_denoise_pipe<4>", k_denoise_pipe<4><<<(int)(wpg * p.B), 4 * 64, PipeCfg<4>::L_TOTAL, st>>>(p);
  else if (pipe) variant = "k_denoise_pipe<2>", k_denoise_pipe<2><<<(int)(wpg * p.B), 2 * 64, PipeCfg<2>::L_TOTAL, st>>>(p);
  else if (pipe_f32 && nw == 8) variant = "k_denoise_pipe_f32<8>", k_denoise_pipe_f32<8><<<(int)(wpg * p.B), 8 * 64, PipeCfg<8>::L_TOTAL, st>>>(p);
  else if (pipe_f32 && nw == 4) variant = "k_denoise_pipe_f32<4>", k_denoise_pipe_f32<4><<<(int)(wpg * p.B), 4 * 64, PipeCfg<4>::L_TOTAL, st>>>(p);
  else if (pipe_f32) variant = "k_denoise_pipe_f32<2>", k_denoise_pipe_f32<2><<<(int)(wpg * p.B), 2 * 64, PipeCfg<2>::L_TOTAL, st>>>(p);
  else if (d->dev.prec == DFX_PREC_BF16) variant = "k_denoise<bf16>", k_denoise<DFX_PREC_BF16, NW><<<(int)grid, NW * 64, 0, st>>>(p);
  else variant = "k_denoise<f32>", k_denoise<DFX_PREC_F32, NW><<<(int)grid, NW * 64, 0, st>>>(p);
  g_last_variant = variant;
  const int rc = check_launch("denoiser kernel");
  tm.end();
  return rc;
}

int check_common(const dfx_denoiser *d, const void *shape_ctx, const int32_t *seg, int B, int N, const char *who) {
  DFX_REQUIRE(d, "%s: null denoiser", who);
  DFX_REQUIRE(B >= 0 && N >= 0, "%s: negative size", who);
  DFX_REQUIRE(N % 32 == 0, "%s: N=%d must be a multiple of 32 (one wavefront = 32 points of one shape)", who, N);
  if ((long long)B * N == 0) return 1;
  DFX_REQUIRE(shape_ctx && seg, "%s: null pointer", who);
  return DFX_OK;
}

}  // namespace

extern "C" {

int dfx_denoise_eps(const dfx_denoiser *d, const void *shape_ctx, const float *x, const int32_t *seg, int t,
                    float *eps, int B, int N, dfx_stream_t stream) {
  const int rc = check_common(d, shape_ctx, seg, B, N, "denoise_eps");
  if (rc) return rc < 0 ? rc : DFX_OK;
  DFX_REQUIRE(x && eps, "denoise_eps: null pointer");
  DFX_REQUIRE(t >= 0 && t < d->dev.T, "denoise_eps: t=%d outside [0,%d)", t, d->dev.T);
  KParams p{};
  p.x_in = x; p.seg = seg; p.out = eps;
  p.B = B; p.N = N; p.t0 = t; p.nsteps = 1; p.ret_interval = 1; p.mode = MODE_EPS;
  return launch(d, shape_ctx, p, as_stream(stream));
}

int dfx_p_sample(const dfx_denoiser *d, const void *shape_ctx, const float *x, const int32_t *seg, int t,
                 const float *noise, uint64_t seed, uint64_t shape_offset, float *x_prev, float *pred_xstart, int B, int N,
                 dfx_stream_t stream) {
  const int rc = check_common(d, shape_ctx, seg, B, N, "p_sample");
  if (rc) return rc < 0 ? rc : DFX_OK;
  DFX_REQUIRE(x && x_prev, "p_sample: null pointer");
  DFX_REQUIRE(t >= 0 && t < d->dev.T, "p_sample: t=%d outside [0,%d)", t, d->dev.T);
  KParams p{};
  p.x_in = x; p.seg = seg; p.noise = noise; p.out = x_prev; p.xstart = pred_xstart; p.seed = seed; p.shape0 = shape_offset;
  p.B = B; p.N = N; p.t0 = t; p.nsteps = 1; p.ret_interval = 1; p.mode = MODE_PSAMPLE;
  return launch(d, shape_ctx, p, as_stream(stream));
}

// DDIM coefficient list of a launch: steps are given ASCENDING like the reference's `self.steps` (:117-122) and
// executed in reverse (:566); xt_dir_coeff in float64 like numpy (:116), cast to fp32 at use (extract_into_tensor).
static int fill_ddim(const dfx_denoiser *d, KParams &p, const int32_t *steps, int n_steps, float eta, const char *who) {
  DFX_REQUIRE(steps && n_steps >= 1 && n_steps <= DDIM_MAX_STEPS, "%s: 1..%d DDIM steps", who, DDIM_MAX_STEPS);
  const int T = d->dev.T;
  p.ddim_n = n_steps;
  p.ddim_eta = eta;
  for (int i = 0; i < n_steps; ++i) {
    const int t = steps[n_steps - 1 - i];
    DFX_REQUIRE(t >= 0 && t < T, "%s: step %d outside [0,%d)", who, t, T);
    p.ddim_t[i] = t;
    p.ddim_xdc[i] = (float)std::sqrt(1.0 - d->host_ac_pv[t] - (double)eta * (double)eta * d->host_ac_pv[(size_t)T + t]);
  }
  return DFX_OK;
}

int dfx_p_sample_ddim(const dfx_denoiser *d, const void *shape_ctx, const float *x, const int32_t *seg, int t, float eta,
                      const float *noise, uint64_t seed, uint64_t shape_offset, float *x_prev, float *pred_xstart, int B, int N,
                      dfx_stream_t stream) {
  const int rc = check_common(d, shape_ctx, seg, B, N, "p_sample_ddim");
  if (rc) return rc < 0 ? rc : DFX_OK;
  DFX_REQUIRE(x && x_prev, "p_sample_ddim: null pointer");
  KParams p{};
  const int32_t one[1] = {t};
  if (int e = fill_ddim(d, p, one, 1, eta, "p_sample_ddim")) return e;
  p.x_in = x; p.seg = seg; p.noise = noise; p.out = x_prev; p.xstart = pred_xstart; p.seed = seed; p.shape0 = shape_offset;
  p.B = B; p.N = N; p.t0 = t; p.nsteps = 1; p.ret_interval = 1; p.mode = MODE_PSAMPLE;
  return launch(d, shape_ctx, p, as_stream(stream));
}

int dfx_sample_chain_ddim(const dfx_denoiser *d, const void *shape_ctx, const int32_t *seg, const int32_t *steps,
                          int n_steps, float eta, const float *x_T_noise, const float *step_noise, uint64_t seed,
                          uint64_t shape_offset, int ret_interval, float *traj, float *pred, int B, int N, dfx_stream_t stream) {
  const int rc = check_common(d, shape_ctx, seg, B, N, "sample_chain_ddim");
  if (rc) return rc < 0 ? rc : DFX_OK;
  DFX_REQUIRE(pred, "sample_chain_ddim: null pred");
  DFX_REQUIRE(!traj || ret_interval >= 1, "sample_chain_ddim: ret_interval must be >= 1 when traj is given");
  KParams p{};
  if (int e = fill_ddim(d, p, steps, n_steps, eta, "sample_chain_ddim")) return e;
  DFX_REQUIRE(steps[0] == 0, "sample_chain_ddim: the step list must start at 0 (decode keeps the t == 0 sample as 'pred')");
  p.seg = seg; p.noise = step_noise; p.xT_noise = x_T_noise; p.out = pred; p.traj = traj; p.seed = seed; p.shape0 = shape_offset;
  p.B = B; p.N = N; p.t0 = p.ddim_t[0]; p.nsteps = n_steps; p.ret_interval = ret_interval >= 1 ? ret_interval : 1;
  p.mode = MODE_CHAIN;
  return launch(d, shape_ctx, p, as_stream(stream));
}

int dfx_denoise_eps_t(const dfx_denoiser *d, const void *shape_ctx, const float *x, const int32_t *seg, const int32_t *t,
                      float *eps, int B, int N, dfx_stream_t stream) {
  const int rc = check_common(d, shape_ctx, seg, B, N, "denoise_eps_t");
  if (rc) return rc < 0 ? rc : DFX_OK;
  DFX_REQUIRE(x && eps && t, "denoise_eps_t: null pointer");
  KParams p{};
  p.x_in = x; p.seg = seg; p.out = eps; p.t_shape = t;
  p.B = B; p.N = N; p.t0 = 0; p.nsteps = 1; p.ret_interval = 1; p.mode = MODE_EPS;
  return launch(d, shape_ctx, p, as_stream(stream));
}

int dfx_q_sample_f32(const dfx_denoiser *d, const void *shape_ctx, const int32_t *seg, const int32_t *t, const float *x_start,
                     const float *noise, float *x_t, int B, int N, dfx_stream_t stream) {
  const int rc = check_common(d, shape_ctx, seg, B, N, "q_sample");
  if (rc) return rc < 0 ? rc : DFX_OK;
  DFX_REQUIRE(t && x_start && noise && x_t, "q_sample: null pointer");
  ShapeCtxView v;
  shape_ctx_view(&v, const_cast<void *>(shape_ctx), B, d->dev.depth, d->dev.prec);
  const long long total = (long long)B * 3 * N;
  k_q_sample<<<(int)((total + 255) / 256), 256, 0, as_stream(stream)>>>(v.part, d->dev.qtab, seg, t, x_start, noise, x_t, N, d->dev.T, total);
  return check_launch("q_sample");
}

int dfx_masked_mse_f32(const float *target, const float *pred, const float *flags, double *workspace2, float *loss, int B,
                       int N, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 1 && N >= 1 && target && pred && workspace2 && loss, "masked_mse: bad argument");
  hipStream_t st = as_stream(stream);
  DFX_HIP_TRY(hipMemsetAsync(workspace2, 0, 2 * sizeof(double), st));
  const long long total = (long long)B * N;
  long long blocks = (total + 255) / 256;
  if (blocks > 64) blocks = 64;   // every wavefront ends with two float64 atomics on the same two addresses: 256 of them, not 16384 (100 us of serialised L2 atomics)
  k_masked_mse<<<(int)blocks, 256, 0, st>>>(target, pred, flags, workspace2, N, total);
  k_mse_finish<<<1, 1, 0, st>>>(workspace2, loss, flags != nullptr, (double)B * 3.0 * N);
  return check_launch("masked_mse");
}

void dfx_debug_force_direct(int on) { g_force_direct = on != 0; }
void dfx_debug_pipe_waves(int nw) { g_force_nw = (nw == 8 || nw == 4 || nw == 2 || nw == 1 || nw == 64 || nw == 16 || nw == 160 || nw == 161) ? nw : 0; }
void dfx_debug_trace(void *device_buf, int capacity) {
  g_trace = static_cast<unsigned long long *>(device_buf);
  g_trace_cap = capacity;
}

int dfx_chain_num_snapshots(int num_timesteps, int ret_interval) {
  if (num_timesteps <= 0 || ret_interval <= 0) return 0;
  return num_timesteps / ret_interval;
}

int dfx_sample_chain(const dfx_denoiser *d, const void *shape_ctx, const int32_t *seg, const float *x_T_noise,
                     const float *step_noise, uint64_t seed, uint64_t shape_offset, int ret_interval, float *traj, float *pred,
                     int B, int N, dfx_stream_t stream) {
  const int rc = check_common(d, shape_ctx, seg, B, N, "sample_chain");
  if (rc) return rc < 0 ? rc : DFX_OK;
  DFX_REQUIRE(pred, "sample_chain: null pred");
  DFX_REQUIRE(!traj || ret_interval >= 1, "sample_chain: ret_interval must be >= 1 when traj is given");
  KParams p{};
  p.seg = seg; p.noise = step_noise; p.xT_noise = x_T_noise; p.out = pred; p.traj = traj; p.seed = seed; p.shape0 = shape_offset;
  p.B = B; p.N = N; p.t0 = d->dev.T - 1; p.nsteps = d->dev.T; p.ret_interval = ret_interval >= 1 ? ret_interval : 1;
  p.mode = MODE_CHAIN;
  return launch(d, shape_ctx, p, as_stream(stream));
}

}  // extern "C"
