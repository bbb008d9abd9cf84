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
#include "denoiser_internal.h"

#pragma clang fp contract(fast)

using namespace dfx;

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

enum { MODE_EPS = 0, MODE_PSAMPLE = 1, MODE_CHAIN = 2 };

struct KParams {
  DenoiserDev d;
  const float *part;     // shape ctx regions
  const float *cpart;
  const float *sbias;
  const uint4 *as_ms;
  const float *x_in;     // (B,3,N)        eps / p_sample
  const int32_t *seg;    // (B,N)
  const float *noise;    // p_sample: (B,3,N) | chain: (nsteps,B,3,N) | null -> Philox
  const float *xT_noise; // chain: (B,3,N) | null -> Philox
  float *out;            // eps: (B,3,N) | p_sample: (B,3,N) | chain: pred (B,N,3)
  float *traj;           // chain snapshots (n_keep,B,N,3) or null
  float *xstart;         // p_sample: optional pred_xstart (B,3,N)
  unsigned long long seed;
  int B, N, t0, nsteps, ret_interval, mode;
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

template <int PREC>
__device__ __forceinline__ void ln_to_act(const v16f (&h)[4], Act<PREC> (&xn)[4]) {
  float mean, rstd;
  ln_stats(h, mean, rstd);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v16f t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = (h[c][r] - mean) * rstd;
    xn[c].set(t);
  }
}

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
  const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
  float sa, ca;
  sincosf(6.28318530717958647692f * u1, &sa, &ca);
  z[0] = ra * ca;
  z[1] = ra * sa;
  z[2] = rb * cosf(6.28318530717958647692f * u3);
}

// ----------------------------------------------------------------------------------------------
template <int PREC, int NW>
__global__ void __launch_bounds__(NW * 64) k_denoise(const KParams p) {
  constexpr int TU = tile_units(PREC);         // 16-byte units per 32x32 weight tile
  constexpr int TSTRIDE = TU * 64;             // uint4 elements per tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = lane >> 5, pj = lane & 31;
  const long long g0 = ((long long)blockIdx.x * NW + wave) * 32;
  if (g0 >= (long long)p.B * p.N) return;
  const int s = __builtin_amdgcn_readfirstlane((int)(g0 / p.N));
  const int n = (int)(g0 - (long long)s * p.N) + pj;
  const unsigned long long gid = (unsigned long long)g0 + pj;
  const int depth = p.d.depth;

  // ---- per-point constants through seg (replaces gather_operation, part_encoders.py:417-428) ----
  const float *part = p.part + (size_t)s * 32;
  const int sg = p.seg[(size_t)s * p.N + n];
  float anc[3], var[3], L[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    anc[i] = part[i * 4 + sg];
    var[i] = part[12 + i * 4 + sg];
    L[i] = sqrtf(var[i]);  // anchored_diffusion.py:306 / :562
  }
  unsigned vmask = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) vmask |= (part[24 + j] != 0.f ? 1u : 0u) << j;  // mask.to(bool), attention.py:196
  vmask = __builtin_amdgcn_readfirstlane(vmask);

  const float *cpart = p.cpart + ((size_t)s * NCLS + sg) * INNER + hf * 64;
  const uint4 *asms_s = p.as_ms + (size_t)s * depth * 8 * TSTRIDE + lane;
  const float *sbias_s = p.sbias + (size_t)s * depth * 32 + hf * 16;

  // ---- x_t ----
  float x[3];
  if (p.mode == MODE_CHAIN) {
    float z[3];
    if (p.xT_noise) {
#pragma unroll
      for (int i = 0; i < 3; ++i) z[i] = p.xT_noise[((size_t)s * 3 + i) * p.N + n];
    } else {
      philox_normal3(p.seed, gid, (unsigned)p.d.T, 1u, z);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = L[i] * z[i] + anc[i];  // anchored_diffusion.py:563-564
    if (p.traj && p.d.T % p.ret_interval == 0 && hf == 0) {
      float *o = p.traj + ((size_t)s * p.N + n) * 3;  // snapshot index 0 <-> t = T
      o[0] = x[0]; o[1] = x[1]; o[2] = x[2];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = p.x_in[((size_t)s * 3 + i) * p.N + n];
  }

  for (int step = 0; step < p.nsteps; ++step) {
    const int t = p.t0 - step;
    v16f h[4];
    // ---- proj_in (13 -> 128): x-columns on the VALU, the 10 per-part-constant inputs pre-folded ----
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 cp = *reinterpret_cast<const float4 *>(cpart + c * 16 + r4 * 4);
        const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float4 w = p.d.win_x[hf * 64 + c * 16 + r4 * 4 + e];
          h[c][r4 * 4 + e] = fmaf(w.z, x[2], fmaf(w.y, x[1], fmaf(w.x, x[0], cpv[e])));
        }
      }
    // ---- pre_norm (affine kept: h is the residual stream) ----
    {
      float mean, rstd;
      ln_stats(h, mean, rstd);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float2 gb = p.d.pre_gb[hf * 64 + c * 16 + r];
          h[c][r] = fmaf((h[c][r] - mean) * rstd, gb.x, gb.y);
        }
    }
    // ---- transformer blocks (attention.py:296-306) ----
    for (int b = 0; b < depth; ++b) {
      const BlockPack &bp = p.d.blk[b];
      Act<PREC> xn[4];
      // -- cross attention to the 4 part tokens --
      ln_to_act<PREC>(h, xn);
      const uint4 *as = asms_s + (size_t)b * 8 * TSTRIDE;
      v16f sim;
      {
        const float *sb = sbias_s + b * 32;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4 t4 = *reinterpret_cast<const float4 *>(sb + r4 * 4);
          sim[r4 * 4 + 0] = t4.x; sim[r4 * 4 + 1] = t4.y; sim[r4 * 4 + 2] = t4.z; sim[r4 * 4 + 3] = t4.w;
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) mma_tile<PREC>(sim, as + c * TSTRIDE, xn[c]);
      // softmax over the 4 keys of each head: registers 4g..4g+3 = keys 0..3 of head 2g+hf
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
        const float inv = 1.0f / sum;
#pragma unroll
        for (int j = 0; j < 4; ++j) sim[4 * g + j] = e[j] * inv;
      }
      {
        Act<PREC> pa;
        pa.set(sim);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) mma_tile<PREC>(h[ct], as + (4 + ct) * TSTRIDE, pa);
      }
      add_cvec(h, bp.ct + (size_t)t * INNER + hf * 64);
      // -- GEGLU feed-forward, 16 hidden chunks of 32 units; hidden never leaves registers --
      ln_to_act<PREC>(h, xn);
      const uint4 *w1 = bp.w1 + lane;
      const uint4 *w2 = bp.w2 + lane;
      const float *b1 = bp.b1 + hf * 16;
#pragma unroll 1
      for (int u = 0; u < FF_CHUNKS; ++u) {
        v16f a, g;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4 ta = *reinterpret_cast<const float4 *>(b1 + (u * 2 + 0) * 32 + r4 * 4);
          const float4 tg = *reinterpret_cast<const float4 *>(b1 + (u * 2 + 1) * 32 + r4 * 4);
          a[r4 * 4 + 0] = ta.x; a[r4 * 4 + 1] = ta.y; a[r4 * 4 + 2] = ta.z; a[r4 * 4 + 3] = ta.w;
          g[r4 * 4 + 0] = tg.x; g[r4 * 4 + 1] = tg.y; g[r4 * 4 + 2] = tg.z; g[r4 * 4 + 3] = tg.w;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          mma_tile<PREC>(a, w1 + ((u * 2 + 0) * 4 + c) * TSTRIDE, xn[c]);
          mma_tile<PREC>(g, w1 + ((u * 2 + 1) * 4 + c) * TSTRIDE, xn[c]);
        }
        v16f hid;
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[r] = a[r] * gelu_erf(g[r]);
        Act<PREC> ha;
        ha.set(hid);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) mma_tile<PREC>(h[ct], w2 + (u * 4 + ct) * TSTRIDE, ha);
      }
      add_cvec(h, bp.b2 + hf * 64);
    }
    // ---- post_norm (affine folded) + proj_out (128 -> 3) on the VALU ----
    float eps[3];
    {
      float mean, rstd;
      ln_stats(h, mean, rstd);
      float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float4 w = p.d.wout[hf * 64 + c * 16 + r];
          const float v = (h[c][r] - mean) * rstd;
          e0 = fmaf(w.x, v, e0);
          e1 = fmaf(w.y, v, e1);
          e2 = fmaf(w.z, v, e2);
        }
      eps[0] = e0 + xhalf(e0) + p.d.bout[0];
      eps[1] = e1 + xhalf(e1) + p.d.bout[1];
      eps[2] = e2 + xhalf(e2) + p.d.bout[2];
    }
    if (p.mode == MODE_EPS) {
      if (hf == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) p.out[((size_t)s * 3 + i) * p.N + n] = eps[i];
      }
      return;
    }
    // ---- posterior (anchored_diffusion.py:306-319,365-367,378-380,401-409,175-213,476-483), reference op order ----
    float z[3];
    if (p.noise) {
#pragma unroll
      for (int i = 0; i < 3; ++i) z[i] = p.noise[(((size_t)step * p.B + s) * 3 + i) * p.N + n];
    } else {
      philox_normal3(p.seed, gid, (unsigned)t, 0u, z);
    }
    {
#pragma clang fp contract(off)
      const float *tb = p.d.tab + (size_t)t * 8;
      const float sra = tb[0], srm1 = tb[1], c1 = tb[2], c2 = tb[3], c3 = tb[4], pv = tb[5];
      const float nz = t != 0 ? 1.0f : 0.0f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float x0 = sra * (x[i] - anc[i]) + anc[i] - srm1 * L[i] * eps[i];
        if (p.xstart && hf == 0) p.xstart[((size_t)s * 3 + i) * p.N + n] = x0;
        const float mu = c1 * x0 + c2 * x[i] + c3 * anc[i];
        const float mv = pv * var[i];
        x[i] = mu + nz * sqrtf(mv) * z[i];
      }
    }
    if (p.mode == MODE_PSAMPLE) {
      if (hf == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) p.out[((size_t)s * 3 + i) * p.N + n] = x[i];
      }
      return;
    }
    // ---- chain bookkeeping of AnchorDiffAE.decode (anchor_gen.py:160-167) ----
    if (hf == 0) {
      if (t == 0) {
        float *o = p.out + ((size_t)s * p.N + n) * 3;
        o[0] = x[0]; o[1] = x[1]; o[2] = x[2];
      } else if (p.traj && t % p.ret_interval == 0) {
        const int k = p.d.T / p.ret_interval - t / p.ret_interval;
        float *o = p.traj + (((size_t)k * p.B + s) * p.N + n) * 3;
        o[0] = x[0]; o[1] = x[1]; o[2] = x[2];
      }
    }
  }
}

int launch(const dfx_denoiser *d, const void *shape_ctx, KParams &p, hipStream_t st) {
  ShapeCtxView v;
  shape_ctx_view(&v, const_cast<void *>(shape_ctx), p.B, d->dev.depth, d->dev.prec);
  p.d = d->dev;
  p.part = v.part;
  p.cpart = v.cpart;
  p.sbias = v.sbias;
  p.as_ms = v.as_ms;
  constexpr int NW = 4;
  const long long waves = ((long long)p.B * p.N) / 32;
  const long long grid = (waves + NW - 1) / NW;
  if (grid > 0x7fffffffLL) return set_error(DFX_ERR_INVALID_ARG, "denoiser: B*N too large");
  EventTimer tm;
  tm.begin(st);
  if (d->dev.prec == DFX_PREC_BF16) k_denoise<DFX_PREC_BF16, NW><<<(int)grid, NW * 64, 0, st>>>(p);
  else k_denoise<DFX_PREC_F32, NW><<<(int)grid, NW * 64, 0, st>>>(p);
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
                 const float *noise, uint64_t seed, float *x_prev, float *pred_xstart, int B, int N,
                 dfx_stream_t stream) {
  const int rc = check_common(d, shape_ctx, seg, B, N, "p_sample");
  if (rc) return rc < 0 ? rc : DFX_OK;
  DFX_REQUIRE(x && x_prev, "p_sample: null pointer");
  DFX_REQUIRE(t >= 0 && t < d->dev.T, "p_sample: t=%d outside [0,%d)", t, d->dev.T);
  KParams p{};
  p.x_in = x; p.seg = seg; p.noise = noise; p.out = x_prev; p.xstart = pred_xstart; p.seed = seed;
  p.B = B; p.N = N; p.t0 = t; p.nsteps = 1; p.ret_interval = 1; p.mode = MODE_PSAMPLE;
  return launch(d, shape_ctx, p, as_stream(stream));
}

int dfx_chain_num_snapshots(int num_timesteps, int ret_interval) {
  if (num_timesteps <= 0 || ret_interval <= 0) return 0;
  return num_timesteps / ret_interval;
}

int dfx_sample_chain(const dfx_denoiser *d, const void *shape_ctx, const int32_t *seg, const float *x_T_noise,
                     const float *step_noise, uint64_t seed, int ret_interval, float *traj, float *pred, int B,
                     int N, dfx_stream_t stream) {
  const int rc = check_common(d, shape_ctx, seg, B, N, "sample_chain");
  if (rc) return rc < 0 ? rc : DFX_OK;
  DFX_REQUIRE(pred, "sample_chain: null pred");
  DFX_REQUIRE(!traj || ret_interval >= 1, "sample_chain: ret_interval must be >= 1 when traj is given");
  KParams p{};
  p.seg = seg; p.noise = step_noise; p.xT_noise = x_T_noise; p.out = pred; p.traj = traj; p.seed = seed;
  p.B = B; p.N = N; p.t0 = d->dev.T - 1; p.nsteps = d->dev.T; p.ret_interval = ret_interval >= 1 ? ret_interval : 1;
  p.mode = MODE_CHAIN;
  return launch(d, shape_ctx, p, as_stream(stream));
}

}  // extern "C"
