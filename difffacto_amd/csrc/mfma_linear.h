// Generic row-batched fp32 linear layer on the exact fp32 matrix pipe (v_mfma_f32_32x32x2_f32), shared by the latent
// sampler (latents_kernels.hip) and the PointNet++ shared-MLP path (sa_kernels.hip).
//
// Evaluated transposed like the denoiser: output channels on the MFMA M axis (accumulator registers), rows / tokens /
// points on the N axis (lanes), so that the epilogues are in-lane.  The K axis is consumed 8 at a time with one 16-byte
// load per operand and lane: lane (j, hf) holds k = k0 + 4 hf + s for the s-th MFMA of a block.
#pragma once
#include "dfx_common.h"
#include <algorithm>
#include <type_traits>

namespace dfx {
namespace lin {
extern int g_lin_split_k;   // -1 automatic (by tile count), 0 never, 1 whenever K >= 128 (dfx_debug_lin_split_k)


typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_RESID = 2, EPI_COUPLING = 3, EPI_GEGLU = 4 };

struct LinArgs {
  const float *X; long long x_gs; int ldx;   // activations (M, K) rows, leading dimension ldx, group stride
  const float *W; long long w_gs;            // weights (N or 2N, K) row-major (nn.Linear layout)
  const float *b; long long b_gs;            // bias (N or 2N) or nullptr
  const float *Wtab[4], *btab[4];            // k_lin only: per-group weight / bias pointers (groups <= 4) when Wtab[0] != nullptr,
                                             // for groups whose parameters are separate tensors (the per-part flows)
  float *Y; long long y_gs; int ldy;         // output (M, N); COUPLING: the half of x updated in place
  const float *R; int ldr; int r_mod;        // RESID: Y = acc + b + R[(r_mod ? m % r_mod : m)][n]
  long long r_gs;                            // k_lin only: group stride of R
  int M, N, K;                               // N = output columns (dual epilogues read 2N weight rows)
  float *stats;                              // k_lin_wide_lds<.., STATS = true> only: [gridDim.y][N][3] per-workgroup (count, mean, M2) of every
                                             // output column over the workgroup's rows (BatchNorm batch statistics without a pass over Y)
};

// (count, mean, sum of squared deviations) of two disjoint sets of values -> of their union (Chan et al.); exact in real arithmetic and as
// accurate as a two-pass computation; either side may be empty
__host__ __device__ inline void stats_merge(float &n, float &mean, float &m2, float nb, float mb, float qb) {
  if (nb == 0.f) return;
  if (n == 0.f) {
    n = nb, mean = mb, m2 = qb;
    return;
  }
  const float nn = n + nb, d = mb - mean;
  mean += d * (nb / nn);
  m2 += qb + d * d * (n * nb / nn);
  n = nn;
}

// SPLIT > 1: split-K.  A call with few rows (the flows' and the aligner's 128-512 rows, the time-embedding MLP) is 100-odd single-wavefront
// workgroups, each ONE dependent chain of K / 32 trips (~1 us of L2 latency per trip): SPLIT wavefronts per tile take every SPLIT-th trip, wavefronts
// 1 .. SPLIT - 1 hand their accumulators over through LDS and wavefront 0 adds them in wave order and runs the epilogue (fixed order: deterministic;
// the sum over K is grouped differently from SPLIT = 1, i.e. equal up to fp32 rounding).
template <int EPI, int SPLIT = 1>
static __global__ __launch_bounds__(64 * SPLIT) void k_lin(LinArgs a) {
  constexpr bool DUAL = (EPI == EPI_COUPLING || EPI == EPI_GEGLU);
  const int lane = threadIdx.x & 63, wave = SPLIT > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0, j = lane & 31, hf = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, g = blockIdx.z;
  const int mrow = min(m0 + j, a.M - 1), nrow = min(n0 + j, a.N - 1);   // clamped rows are never stored
  const float *xp = a.X + g * a.x_gs + (size_t)mrow * a.ldx + 4 * hf;
  const float *wp = (a.Wtab[0] ? a.Wtab[g] : a.W + g * a.w_gs) + (size_t)nrow * a.K + 4 * hf;
  const float *wq = wp + (size_t)a.N * a.K;
  v16f acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
  auto block = [&](const v4f xv, const v4f wv, const v4f uv) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[s], xv[s], acc0, 0, 0, 0);
    if (DUAL) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(uv[s], xv[s], acc1, 0, 0, 0);
    }
  };
  auto ld = [](const float *p) { return *reinterpret_cast<const v4f *>(p); };
  // 4 K-blocks of 8 per trip (8-12 loads, 16-32 MFMAs); the loads of trip t + 1 are issued before the MFMAs of trip t (two register
  // sets used alternately; the request behind the last trip re-reads it: the same number of loads in flight on every path) — the
  // few-row calls of the flows are one dependent chain of trips per wavefront, ~1.2 us of memory latency each
  struct Trip {
    v4f xv[4], wv[4], uv[4];
  };
  auto load_trip = [&](int k, Trip &t) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      t.xv[u] = ld(xp + k + 8 * u), t.wv[u] = ld(wp + k + 8 * u);
      t.uv[u] = DUAL ? ld(wq + k + 8 * u) : v4f{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto run_trip = [&](const Trip &t) {
#pragma unroll
    for (int u = 0; u < 4; ++u) block(t.xv[u], t.wv[u], t.uv[u]);
  };
  const int ntrip_all = a.K / 32;
  // this wavefront's trips: wave, wave + SPLIT, ... (SPLIT = 1: all of them, in order)
  const int ntrip = ntrip_all > wave ? (ntrip_all - wave + SPLIT - 1) / SPLIT : 0;
  auto kof = [&](int t) { return 32 * (wave + t * SPLIT); };
  if (ntrip > 0) {
    Trip ta, tb;
    load_trip(kof(0), ta);
    int t = 0;
    for (; t + 1 < ntrip; t += 2) {
      load_trip(kof(t + 1), tb);
      __builtin_amdgcn_sched_barrier(0);
      run_trip(ta);
      load_trip(kof(min(t + 2, ntrip - 1)), ta);
      __builtin_amdgcn_sched_barrier(0);
      run_trip(tb);
    }
    if (t < ntrip) run_trip(ta);
  }
  if (wave == 0)
    for (int k = 32 * ntrip_all; k < a.K; k += 8) block(ld(xp + k), ld(wp + k), DUAL ? ld(wq + k) : v4f{0.f, 0.f, 0.f, 0.f});
  if constexpr (SPLIT > 1) {
    __shared__ float red[SPLIT - 1][DUAL ? 2 : 1][16][64];
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[wave - 1][0][r][lane] = acc0[r];
        if (DUAL) red[wave - 1][DUAL ? 1 : 0][r][lane] = acc1[r];
      }
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < SPLIT - 1; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc0[r] += red[w][0][r][lane];
        if (DUAL) acc1[r] += red[w][DUAL ? 1 : 0][r][lane];
      }
  }
  const int m = m0 + j;
  if (m >= a.M) return;
  const float *bp = a.Wtab[0] ? a.btab[g] : (a.b ? a.b + g * a.b_gs : nullptr);
  float *yp = a.Y + g * a.y_gs + (size_t)m * a.ldy;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * hf;   // 32x32 C/D layout
    if (n >= a.N) continue;
    float v = acc0[r] + (bp ? bp[n] : 0.f);
    if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
    if (EPI == EPI_RESID) v += a.R[g * a.r_gs + (size_t)(a.r_mod ? m % a.r_mod : m) * a.ldr + n];
    if (EPI == EPI_COUPLING) {   // flow.py:30-31,40: y1 = (x2 - shift) / sigmoid(s + 2)
      const float shift = acc1[r] + (bp ? bp[a.N + n] : 0.f);
      const float scale = 1.f / (1.f + expf(-(v + 2.f)));
      v = (yp[n] - shift) / scale;
    }
    if (EPI == EPI_GEGLU) {      // attention.py:55-57: x * F.gelu(gate), exact erf form
      const float gate = acc1[r] + (bp ? bp[a.N + n] : 0.f);
      v = v * (0.5f * gate * (1.f + erff(gate * 0.70710678118654752440f)));
    }
    yp[n] = v;
  }
}

// Wide variant for the plain epilogues when N is a multiple of 128: one wavefront = 32 rows x 128 output channels (four
// accumulator tiles share every activation fragment), four wavefronts per workgroup = 128 rows on the same weight tile
// (shared through the L1), 16-byte stores.  2-3x the throughput of k_lin on the large row counts of the PointNet paths.
template <int EPI>
static __global__ __launch_bounds__(256) void k_lin_wide(LinArgs a) {
  static_assert(EPI == EPI_NONE || EPI == EPI_RELU || EPI == EPI_RESID, "plain epilogues only");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hf = lane >> 5;
  const int n0 = blockIdx.x * 128, m0 = (blockIdx.y * 4 + wave) * 32, g = blockIdx.z;
  if (m0 >= a.M) return;
  const int mrow = min(m0 + j, a.M - 1);
  const float *xp = a.X + g * a.x_gs + (size_t)mrow * a.ldx + 4 * hf;
  const float *wp = a.W + g * a.w_gs + (size_t)(n0 + j) * a.K + 4 * hf;
  const size_t wt = (size_t)32 * a.K;   // next 32-channel tile
  v16f acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto ld = [](const float *p) { return *reinterpret_cast<const v4f *>(p); };
  // The operands of K step k + 8 are requested before the 16 MFMAs (1024 matrix-pipe cycles) of step k, two register sets used
  // alternately (left to itself the compiler waits for each step's five loads right where it issues them).  The request behind the
  // last step re-reads the last one: an unconditional load keeps the count of outstanding loads the same on every path.
  auto operands = [&](int k, v4f &xv, v4f (&wv)[4]) {
    xv = ld(xp + k);
#pragma unroll
    for (int t = 0; t < 4; ++t) wv[t] = ld(wp + t * wt + k);
  };
  auto step = [&](const v4f &xv, const v4f (&wv)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t][s], xv[s], acc[t], 0, 0, 0);
  };
  {
    v4f xa, xb, wa[4], wb[4];
    operands(0, xa, wa);
    int k = 0;
    for (; k + 8 < a.K; k += 16) {
      operands(k + 8, xb, wb);
      __builtin_amdgcn_sched_barrier(0);
      step(xa, wa);
      operands(min(k + 16, a.K - 8), xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      step(xb, wb);
    }
    if (k < a.K) step(xa, wa);   // K / 8 odd: the last step's operands were requested by the last pair (or above)
  }
  const int m = m0 + j;
  if (m >= a.M) return;
  const float *bp = a.b ? a.b + g * a.b_gs : nullptr;
  float *yp = a.Y + g * a.y_gs + (size_t)m * a.ldy;
  const float *rp = EPI == EPI_RESID ? a.R + (size_t)(a.r_mod ? m % a.r_mod : m) * a.ldr : nullptr;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 32 * t + 8 * q + 4 * hf;   // registers 4q..4q+3 of the 32x32 C/D layout: four consecutive channels
      v4f v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = acc[t][4 * q + e] + (bp ? bp[n + e] : 0.f);
        if (EPI == EPI_RELU) y = fmaxf(y, 0.f);
        if (EPI == EPI_RESID) y += rp[n + e];
        v[e] = y;
      }
      *reinterpret_cast<v4f *>(yp + n) = v;
    }
}

// The same tile with the weights resident in LDS.  k_lin_wide's five loads per K step all have the fragment shape (32 rows x 32
// bytes: 32 cache lines per instruction), four of them for weights that every wavefront of the grid re-reads: at 262 144 rows the
// kernel sits on the texture-address path at ~43 % of the matrix peak whatever the prefetch distance.  Here a workgroup of eight
// wavefronts stages its 128-channel weight block ONCE, fragment-ready ([tile][K step][lane] float4: K / 2 KiB for 128 channels, K <= 272; 64-channel blocks up to K = 544), and walks over
// row tiles of 256 with it: per K step one global load (the activations, one step ahead) and four conflict-free ds_read_b128.
// STATS: the workgroup also leaves (count, mean, M2) of every output column over ITS rows in a.stats — a lane holds 16 rows of one channel, so a
// tile's share is 16 in-lane adds per pass; tiles, lane halves, wavefronts are merged with stats_merge in a fixed order (deterministic).
template <int EPI, int NT, bool STATS = false>   // NT = 4: 128 channels per block, K <= 272; NT = 2: 64 channels, K <= 544 (the block fits LDS either way)
static __global__ __launch_bounds__(512) void k_lin_wide_lds(LinArgs a) {
  static_assert(EPI == EPI_NONE || EPI == EPI_RELU || EPI == EPI_RESID, "plain epilogues only");
  static_assert(!STATS || EPI == EPI_NONE, "statistics of the plain output");
  extern __shared__ __align__(16) float lin_smem[];
  v4f *wl = reinterpret_cast<v4f *>(lin_smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hf = lane >> 5;
  const int n0 = blockIdx.x * (32 * NT), g = blockIdx.z, U = a.K >> 3;
  auto ld = [](const float *p) { return *reinterpret_cast<const v4f *>(p); };
  {
    // (eight loads in flight per thread: one load, one wait, one LDS write per trip made the staging a chain of 16-68 memory latencies
    // in front of every workgroup's work)
    const float *wbase = a.W + g * a.w_gs;
    const int total = NT * U * 64;
    for (int e0 = threadIdx.x; e0 < total; e0 += 8 * 512) {
      v4f v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = min(e0 + q * 512, total - 1), l = e & 63, tu = e >> 6, u = tu % U, t = tu / U;
        v[q] = ld(wbase + (size_t)(n0 + 32 * t + (l & 31)) * a.K + 8 * u + 4 * (l >> 5));
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (e0 + q * 512 < total) wl[e0 + q * 512] = v[q];
    }
  }
  __syncthreads();
  const float *bp = a.b ? a.b + g * a.b_gs : nullptr;
  const v4f *wlane = wl + lane;
  float sn = 0.f, smean[NT], sm2[NT];   // STATS: running (count, mean, M2) of this lane's 16-row column pieces
#pragma unroll
  for (int t = 0; t < NT; ++t) smean[t] = 0.f, sm2[t] = 0.f;
  for (long long rt = blockIdx.y; rt * 256 < a.M; rt += gridDim.y) {   // (no barrier inside: wavefronts past the end just skip)
    const int m0 = (int)(rt * 256) + wave * 32;
    if (m0 >= a.M) continue;
    const int mrow = min(m0 + j, a.M - 1);
    const float *xp = a.X + g * a.x_gs + (size_t)mrow * a.ldx + 4 * hf;
    v16f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto step = [&](const v4f &xv, const v4f (&wv)[NT]) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[s], wv[t][s], acc[t], 0, 0, 0);   // rows on the M axis
    };
    {
      // The activations come from HBM, the weights from LDS: the activation fragments are requested XD K steps ahead (a step is 16 MFMAs = 0.43 us
      // of matrix pipe, an HBM round trip under load is longer), the weight fragments one step ahead.  Requests past the end re-read the last
      // step: the same number of loads in flight on every path.  Same MFMA order as before: the same bits.
      constexpr int XD = 4;   // (same box, PointNetV2 training forward + backward: one step ahead 4.565 ms, 4: 4.51, 8: 4.55)
      v4f xr[XD], wa[NT], wb[NT];
#pragma unroll
      for (int q = 0; q < XD; ++q) xr[q] = ld(xp + 8 * min(q, U - 1));
      auto weights = [&](int u, v4f (&wv)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) wv[t] = wlane[(t * U + u) * 64];
      };
      weights(0, wa);
      int u = 0;
      for (; u + XD <= U; u += XD) {
#pragma unroll
        for (int q = 0; q < XD; ++q) {
          const v4f xv = xr[q];
          xr[q] = ld(xp + 8 * min(u + q + XD, U - 1));
          if (q & 1) {
            weights(min(u + q + 1, U - 1), wa);
            __builtin_amdgcn_sched_barrier(0);
            step(xv, wb);
          } else {
            weights(min(u + q + 1, U - 1), wb);
            __builtin_amdgcn_sched_barrier(0);
            step(xv, wa);
          }
        }
      }
      // tail (U % XD steps): the fragments are already in xr[0 ..]; (XD is even, so the weight buffers alternate from wa again)
#pragma unroll
      for (int q = 0; q < XD - 1; ++q) {
        if (u + q < U) {
          if (q & 1) {
            weights(min(u + q + 1, U - 1), wa);
            step(xr[q], wb);
          } else {
            weights(min(u + q + 1, U - 1), wb);
            step(xr[q], wa);
          }
        }
      }
    }

    // Rows on the MFMA's M axis (accumulator registers), channels on the lanes: a store instruction writes 32 consecutive channels
    // = one full 128-byte line of two rows.  (With the channels in the registers — the orientation of k_lin_wide — a store was 16 bytes
    // per lane at the row stride: 32 lines touched per instruction, each a quarter written: 10 % of the kernel in
    // tools/ubench/mfma_f32_feed.hip.)
    float *ybase = a.Y + g * a.y_gs;
    auto epilogue = [&](auto full) {   // full: all 32 rows of the tile exist (every tile but the last one of a ragged M): no per-store mask
      float cnt = 0.f;
      if (STATS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt += (decltype(full)::value || m0 + (r & 3) + 8 * (r >> 2) + 4 * hf < a.M) ? 1.f : 0.f;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = n0 + 32 * t + j;
        const float bias = bp ? bp[n] : 0.f;
        float ssum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
          if (decltype(full)::value || m < a.M) {
            float y = acc[t][r] + bias;
            if (EPI == EPI_RELU) y = fmaxf(y, 0.f);
            if (EPI == EPI_RESID) y += a.R[(size_t)(a.r_mod ? m % a.r_mod : m) * a.ldr + n];
            ybase[(size_t)m * a.ldy + n] = y;
            if (STATS) ssum += y;
          }
        }
        if (STATS && cnt > 0.f) {
          const float mt = ssum / cnt;
          float q = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (decltype(full)::value || m < a.M) {
              const float d = (acc[t][r] + bias) - mt;
              q = fmaf(d, d, q);
            }
          }
          float nn = sn;
          stats_merge(nn, smean[t], sm2[t], cnt, mt, q);
        }
      }
      if (STATS) sn += cnt;
    };
    if (m0 + 32 <= a.M) epilogue(std::true_type{});
    else epilogue(std::false_type{});
  }
  if constexpr (STATS) {
    // lane halves (rows 4 hf + ... of the same channel), then the eight wavefronts in order, through LDS
    __shared__ float sred[8][NT * 32][3];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float on = __shfl_xor(sn, 32), om = __shfl_xor(smean[t], 32), oq = __shfl_xor(sm2[t], 32);
      float n2 = sn, mean = smean[t], m2 = sm2[t];
      if (hf == 0) {   // (half 0 = the first operand for both lanes' results: one order)
        stats_merge(n2, mean, m2, on, om, oq);
        sred[wave][t * 32 + j][0] = n2, sred[wave][t * 32 + j][1] = mean, sred[wave][t * 32 + j][2] = m2;
      }
    }
    __syncthreads();
    if (threadIdx.x < NT * 32) {
      float n2 = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) stats_merge(n2, mean, m2, sred[w][threadIdx.x][0], sred[w][threadIdx.x][1], sred[w][threadIdx.x][2]);
      float *o = a.stats + ((size_t)blockIdx.y * a.N + n0 + threadIdx.x) * 3;
      o[0] = n2, o[1] = mean, o[2] = m2;
    }
  }
}

// The wide LDS-resident kernel with the statistics epilogue, when its conditions hold (returns the number of partial rows = gridDim.y, or 0:
// the caller then takes the plain launch and a separate statistics pass).  stats: room for (M + 255) / 256 x N x 3 floats at most.
inline int launch_wide_lds_stats(hipStream_t st, const LinArgs &a) {
  if (!(a.N % 128 == 0 && a.M >= 8192 && a.K % 8 == 0 && a.K <= 272 && a.ldx % 4 == 0 && a.stats)) return 0;
  const size_t lds = (size_t)a.K / 8 * 4096;
  static PerDeviceOnce attrs;
  (void)attrs.run([] { return set_max_lds(reinterpret_cast<const void *>(k_lin_wide_lds<EPI_NONE, 4, true>), 34 * 4096); });
  const int cols = a.N / 128, per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024 - 16 * 1024) / lds));
  const int rows = (int)std::min<long long>((a.M + 255) / 256, std::max(1, 256 * per_cu / cols));
  k_lin_wide_lds<EPI_NONE, 4, true><<<dim3(cols, rows, 1), 512, lds, st>>>(a);
  return rows;
}

template <int EPI>
inline void launch(hipStream_t st, int groups, const LinArgs &a) {
  constexpr bool plain = (EPI == EPI_NONE || EPI == EPI_RELU || EPI == EPI_RESID);
  if constexpr (plain) {
    // (the wide tile only when its grid gives every CU a workgroup: a wavefront's K loop is one dependent chain, 16 MFMAs per step
    // there against 4 in k_lin — the aligner's 512-row layers ran as 8-32 workgroups of the wide kernel, 54 us per call)
    if (a.N % 128 == 0 && a.M >= 512 && (long long)(a.N / 128) * ((a.M + 127) / 128) * groups >= 256 && a.ldy % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 && (a.y_gs % 4) == 0) {
      if (a.K % 8 == 0 && a.K <= 544 && a.M >= 8192 && a.ldx % 4 == 0) {   // enough row tiles per workgroup to pay for staging the weights
        const bool narrow = a.K > 272;   // 64-channel blocks when the 128-channel one would not fit
        const size_t lds = (size_t)a.K / 8 * (narrow ? 2048 : 4096);
        static PerDeviceOnce attrs;   // (a failure here surfaces as the launch error the caller's check_launch reports)
        (void)attrs.run([] {
          hipError_t e = set_max_lds(reinterpret_cast<const void *>(k_lin_wide_lds<EPI, 4>), 34 * 4096);
          if (e == hipSuccess) e = set_max_lds(reinterpret_cast<const void *>(k_lin_wide_lds<EPI, 2>), 68 * 2048);
          return e;
        });
        const int cols = a.N / (narrow ? 64 : 128), per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024 - 1024) / lds));
        const int rows = (int)std::min<long long>((a.M + 255) / 256, std::max(1, 256 * per_cu / (cols * groups)));
        if (narrow) k_lin_wide_lds<EPI, 2><<<dim3(cols, rows, groups), 512, lds, st>>>(a);
        else k_lin_wide_lds<EPI, 4><<<dim3(cols, rows, groups), 512, lds, st>>>(a);
        return;
      }
      dim3 grid(a.N / 128, (a.M + 127) / 128, groups);
      k_lin_wide<EPI><<<grid, 256, 0, st>>>(a);
      return;
    }
  }
  dim3 grid((a.N + 31) / 32, (a.M + 31) / 32, groups);
  // at most one tile per CU and a long K: four wavefronts per tile (split-K) — the call is a chain of round trips, not of MFMAs (measured per call of
  // the latent sampler at B = 128: flows 7.1 / 8.8 / 13.7 -> 6.4 / 7.9 / 11.1 us, the aligner's K = 1024 projection 20.3 -> 14.1; with 384-512 tiles
  // the split costs 1-2 us instead)
  // (the split regroups the fp32 sum over K, and the choice depends on the call's row count: results of the few-row calls — the latent front end, the
  // time-embedding MLP — are reproducible for a given batch size, not across batch sizes.  dfx_debug_lin_split_k(1) takes the split whenever K >= 128,
  // whatever the tile count — one grouping for every batch size, for sharded runs that must match a single-process run bit for bit; 0 never splits)
  const int mode = g_lin_split_k;
  const bool split = a.K >= 128 && (mode == 1 || (mode < 0 && (long long)grid.x * grid.y * groups <= 256));
  if (split) k_lin<EPI, 4><<<grid, 256, 0, st>>>(a);
  else k_lin<EPI><<<grid, 64, 0, st>>>(a);
}

}  // namespace lin
}  // namespace dfx
