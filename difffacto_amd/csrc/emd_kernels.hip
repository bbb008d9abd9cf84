// Approximate earth mover's distance by auction — replaces the `emd` extension
// (python/difffacto/metrics/emd/emd_cuda.cu: forward :236-284 = `iters` x {calc_unass_* :29-100, Bid :102-186,
// GetMax :188-201, Assign :203-223} + CalcDist :225-234; backward :286-316; bound in emd_module.py:17-51).
//
// The reference launches seven kernels per auction iteration (70 000 launches at the evaluation setting of 10 000
// iterations).  Here ONE persistent workgroup (16 wavefronts) runs the whole auction of one cloud pair, and the loop ends as
// soon as nothing is unassigned (after that no reference kernel changes any state).
//   LDS (n <= 2688, 60 B per point; 120 KiB at the evaluation size n = 2048): the targets with their prices as 16-byte records
//   (one ds_read_b128 per evaluated pair), the bidders' coordinates, both assignments, the bids and TWO unassigned lists — the
//   list is carried from one iteration to the next (losers + displaced owners) instead of being rebuilt by a scan over n points.
//   Larger n (<= 8192): only the targets in LDS, the state in the caller's workspace (every phase then pays an L2 round trip).
//   Bid: a wavefront per unassigned point, lanes strided over the targets, branch-free running (best, second best, first index),
//   merged over the wavefront by three single-word DPP reductions (the merge keeps the reference's result: first maximum in index
//   order, :152-178).  The scan is VALU-bound (correctly rounded sqrtf + the reference's double-precision `3.0 - d - price`).
//   Tail (<= 8 unassigned, 85 % of the iterations): 8 wavefronts share the bidders' scans, wavefront 0 merges their partials and
//   does GetMax and Assign in registers, one lane per bidder — two workgroup barriers per iteration.
//   GetMax's data race (several bidders within 1e-6 of the maximum all write max_idx, :195-199) is resolved deterministically:
//   the largest bidder index wins — the oracle uses the same rule.
// 117 ms -> 54 ms for 32 pairs x 2048 points x 10 000 iterations over the round (state through L2, one wavefront per bid, LDS
// shuffles, 8 barriers); a pair owns a compute unit, so throughput comes from launching >= 256 pairs (evaluation.py: 1024).
#include "dfx_common.h"
#include <type_traits>

namespace {

constexpr int EMD_THREADS = 1024;
constexpr int EMD_STATE_LDS_MAX_N = 2688;   // 60 B per point <= 157.5 KiB
bool g_emd_state_global = false;            // debug: keep the auction state in global memory whatever n

#ifdef DFX_EMD_PROBE   // tools/ubench/emd_phase_probe.hip: cycles per phase of the U == 1 iterations, seen by wavefront 0
__device__ long long g_emd_probe[24];   // 0..7 phases of U == 1 | 8..11 cycles, 12..15 count of iterations with U = 1, 2..8, 9..16, > 16 | 16 kernel
#define EMD_T(i) do { if (U == 1 && wave == 0) { const long long t_ = clock64(); if (lane == 0 && blockIdx.x == 0) g_emd_probe[i] += t_ - tp_; tp_ = clock64(); } } while (0)
#else
#define EMD_T(i) do { } while (0)
#endif

struct Best {
  float best, better;
  int idx;
};

// Merge of two (best, second best, first index of the best) triples over disjoint target sets; ties take the smaller index, a triple
// that has seen nothing (idx -1) loses them.  Branch-free (the auction's tail is a chain of these): bitwise | and & on purpose.
__device__ __forceinline__ Best merge(const Best &a, const Best &b) {
  const bool bw = (b.best > a.best) | ((b.best == a.best) & ((unsigned)b.idx < (unsigned)a.idx));
  Best r;
  r.best = bw ? b.best : a.best, r.idx = bw ? b.idx : a.idx;
  r.better = fmaxf(fmaxf(a.better, b.better), fminf(a.best, b.best));   // the loser's best or one of the second bests
  return r;
}

// Correctly rounded sqrtf (what sqrtf() is, and what the reference's CUDA sqrtf is) with the compiler's range handling moved out of
// the way: v_sqrt_f32 is within 1 ulp, the two residuals pick between the result and its neighbours.  The compiler's expansion
// also rescales inputs below 2^-96 (the residuals would lose bits there) and patches 0 / inf through a class test — 7 of its 16
// instructions; here 0, inf and NaN fall through the residual tests unchanged (their neighbours give NaN or zero residuals), and a
// scan that met an input in (0, 2^-96) is redone with the library function (`tiny`: running minimum of bits(s) - 1, unsigned).
__device__ __forceinline__ float sqrt_rn_normal(float s, unsigned &tiny) {
  tiny = min(tiny, (unsigned)(__float_as_int(s) - 1));
  const float r = __builtin_amdgcn_sqrtf(s);
  const float rd = __int_as_float(__float_as_int(r) - 1), ru = __int_as_float(__float_as_int(r) + 1);
  const float e1 = fmaf(-rd, r, s), e2 = fmaf(-ru, r, s);
  float q = e1 <= 0.f ? rd : r;
  q = e2 > 0.f ? ru : q;
  return q;
}
constexpr unsigned EMD_TINY = 0x0f800000u - 1u;   // bits(2^-96) - 1

// lane i receives lane ((i - R) mod 16) of its row of 16 (DPP row_ror: no LDS round trip, unlike ds_bpermute)
template <int R>
__device__ __forceinline__ Best row_ror(const Best &m) {
  const int a = __float_as_int(m.best), b = __float_as_int(m.better), c = m.idx;
  Best t;
  t.best = __int_as_float(__builtin_amdgcn_update_dpp(a, a, 0x120 + R, 0xF, 0xF, false));
  t.better = __int_as_float(__builtin_amdgcn_update_dpp(b, b, 0x120 + R, 0xF, 0xF, false));
  t.idx = __builtin_amdgcn_update_dpp(c, c, 0x120 + R, 0xF, 0xF, false);
  return t;
}

// The merge over the whole wavefront, wave-uniform.
// Three single-word reductions instead of a tree of triple merges (87 instructions): the largest value; the smallest index among the
// lanes that have it; the second largest = the winner lane's own second best against everybody else's best.  Every step is ONE
// v_max / v_min with the lane permute as operand modifier (inline asm: hipcc makes a mov + op pair of each; a DPP read needs two wait
// states behind the VALU write of its source, hence the s_nop in front of every step).  (The merge of the <= 8 partials in wavefront 0
// stays a tree of `merge`: its steps depend on the number of bidders, and as separate asm statements they were no faster.)
#define DFX_DPP1(op, ctrl) "s_nop 1\n " op " %0, %0, %0 " ctrl "\n"
#define DFX_WAVE_STEPS(op)                                                                                                       \
  DFX_DPP1(op, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") DFX_DPP1(op, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")   \
  DFX_DPP1(op, "row_half_mirror row_mask:0xf bank_mask:0xf") DFX_DPP1(op, "row_mirror row_mask:0xf bank_mask:0xf")                \
  DFX_DPP1(op, "row_bcast:15 row_mask:0xa bank_mask:0xf") DFX_DPP1(op, "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1\n"
__device__ __forceinline__ float wave_max_f32(float v) {   // wave-uniform
  asm volatile(DFX_WAVE_STEPS("v_max_f32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  asm volatile(DFX_WAVE_STEPS("v_min_u32_dpp") : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
#undef DFX_WAVE_STEPS
#undef DFX_DPP1
__device__ __forceinline__ Best wave_merge(const Best &m) {
  Best r;
  r.best = wave_max_f32(m.best);
  r.idx = (int)wave_min_u32(m.best == r.best ? (unsigned)m.idx : 0xffffffffu);   // (a lane that has seen nothing carries idx -1: it loses)
  r.better = wave_max_f32(((m.best == r.best) & (m.idx == r.idx)) ? m.better : m.best);
  return r;
}

template <bool STATE_LDS>
__global__ __launch_bounds__(EMD_THREADS) void k_emd(const float *__restrict__ xyz1, const float *__restrict__ xyz2, float eps,
                                                    int iters, float *__restrict__ dist, int32_t *__restrict__ assignment,
                                                    int32_t *__restrict__ wsi, float *__restrict__ wsf, int n) {
  extern __shared__ __align__(16) float lds[];   // (16-byte aligned: a ds_read_b128 off its alignment stalls) {x2, y2, z2, price}[n]  [ | as ass_inv bid max_idx list0 (int) | bid_inc max_inc (float) | x1[n] y1[n] z1[n] | list1 ]
  float4 *T = reinterpret_cast<float4 *>(lds);   // a target and its price in one 16-byte LDS read
  __shared__ int cnt[2];
  __shared__ Best wbest[EMD_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *A = xyz1 + (size_t)b * n * 3, *Bp = xyz2 + (size_t)b * n * 3;
  int32_t *as_out = assignment + (size_t)b * n;
  int32_t *sl = reinterpret_cast<int32_t *>(lds + 4 * n);
  int32_t *as = STATE_LDS ? sl : as_out;
  int32_t *ass_inv = STATE_LDS ? sl + n : wsi + (size_t)b * 5 * n, *bid = ass_inv + n, *max_idx = bid + n, *list0 = max_idx + n;
  float *bid_inc = STATE_LDS ? reinterpret_cast<float *>(sl + 5 * n) : wsf + (size_t)b * 2 * n, *max_inc = bid_inc + n;
  float *XA = reinterpret_cast<float *>(sl + 7 * n);   // STATE_LDS: xyz1 as x[n] y[n] z[n] (a bid starts with its point's coordinates)
  int32_t *list1 = STATE_LDS ? sl + 10 * n : list0 + n;
  for (int k = tid; k < n; k += EMD_THREADS) {
    T[k] = make_float4(Bp[k * 3], Bp[k * 3 + 1], Bp[k * 3 + 2], 0.f);
    as[k] = -1, ass_inv[k] = -1, max_idx[k] = -1, max_inc[k] = 0.f;   // emd_module.py:28-37 (max_increments starts at 0)
    list0[k] = k;
    if (STATE_LDS) XA[k] = A[k * 3], XA[n + k] = A[k * 3 + 1], XA[2 * n + k] = A[k * 3 + 2];
  }
  if (tid == 0) cnt[0] = n, cnt[1] = 0;
  __syncthreads();
  // The unassigned list is kept from one iteration to the next instead of being rebuilt by a scan over all n points (the reference's
  // calc_unass_* kernels, :29-100): what is unassigned after Assign = this iteration's losing bidders + the owners the winners
  // displaced.  (The order of the list does not matter: bids only read prices, and GetMax / Assign resolve by value.)
#ifdef DFX_EMD_PROBE
  long long ti_ = 0, tk_ = clock64();
  int cls_ = 0;
#endif
  for (int it = 0; it < iters; ++it) {
    const bool last = it == iters - 1;
    const int c = it & 1, U = cnt[c];
    if (U == 0) break;
    const int32_t *list = c ? list1 : list0;
    int32_t *nlist = c ? list0 : list1;
    int *ncnt = &cnt[c ^ 1];
#ifdef DFX_EMD_PROBE
    long long tp_ = clock64();
    if (tid == 0 && blockIdx.x == 0 && it > 0) {
      g_emd_probe[8 + cls_] += tp_ - ti_, g_emd_probe[12 + cls_] += 1;
    }
    ti_ = tp_, cls_ = U == 1 ? 0 : U <= 8 ? 1 : U <= 16 ? 2 : 3;
#endif
    if (tid == 0) *ncnt = 0;   // (read by everyone at the start of the previous iteration; filled after this iteration's first barrier;
                               // tid 0 is lane 0 of wavefront 0, which also stores the count of the <= 8 bidder path: same lane, program order)
    constexpr int NWAVES = EMD_THREADS / 64;
    // ---- Bid (:102-186): a wavefront walks the targets for one unassigned point, lanes strided, keeping (best, second best, first index
    // of the best); the lanes' triples are merged with the reference's result (first maximum in index order, :152-178) ----
    auto bid_scan = [&](int j, int first, int stride_log2) {
      Best m{-1e9f, -1e9f, -1};
      const float x1 = STATE_LDS ? XA[j] : A[j * 3], y1 = STATE_LDS ? XA[n + j] : A[j * 3 + 1], z1 = STATE_LDS ? XA[2 * n + j] : A[j * 3 + 2];
      EMD_T(0);
      // (VALU-bound: branch-free update, the strides every lane has in full unrolled by four)
      unsigned tiny = 0xffffffffu;
      auto scan = [&](auto lib) {
        auto eval = [&](int k) {
#pragma clang fp contract(off)
          const float4 t = T[k];
          const float dx = t.x - x1, dy = t.y - y1, dz = t.z - z1;
          const float s = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
          const float r = decltype(lib)::value ? sqrtf(s) : sqrt_rn_normal(s, tiny);
          const float d = (float)(3.0 - (double)r - (double)t.w);   // `3.0` is a double literal in the reference (:151)
          m.better = __builtin_amdgcn_fmed3f(m.best, m.better, d);   // best >= better: the median = (d > best ? best : d > better ? d : better)
          m.idx = d > m.best ? k : m.idx;
          m.best = fmaxf(m.best, d);
        };
        const int stride = 1 << stride_log2, full = n >> stride_log2;
        int k = first + lane, t = 0;
        for (; t + 4 <= full; t += 4, k += 4 * stride) eval(k), eval(k + stride), eval(k + 2 * stride), eval(k + 3 * stride);
        for (; t < full; ++t, k += stride) eval(k);
        if (k < n) eval(k);
      };
      scan(std::false_type{});
      if (__builtin_expect(__any(tiny < EMD_TINY), 0)) {   // a squared distance in (0, 2^-96): once more with the library's sqrtf
        m = Best{-1e9f, -1e9f, -1};
        scan(std::true_type{});
      }
      EMD_T(1);
      m = wave_merge(m);
      EMD_T(2);
      return m;
    };
    if (U <= 8) {
      // ---- the tail of an auction: thousands of iterations with a handful of unassigned points.  The iteration is as long as the
      // instructions its wavefronts issue (four SIMDs; a merge over a wavefront costs as much as two targets per lane; one wavefront
      // alone on a SIMD issues dependent instructions at half the rate two reach), so the bids use EIGHT wavefronts — two per SIMD:
      // G = 8, 4, 2, 1 wavefronts per bidder for U = 1, 2, 3-4, 5-8 — and wavefront 0
      // finishes the iteration alone, one lane per bidder: merge of the partials, GetMax and Assign follow each other in registers /
      // in program order, no workgroup barrier between them ----
      const int lp = U == 1 ? 0 : U == 2 ? 1 : U <= 4 ? 2 : 3, P = 1 << lp, lg = 3 - lp, G = 1 << lg;   // P = 1, 2, 4, 8 >= U; G = 8 / P
      if (wave < U * G) {
        const Best m = bid_scan(list[wave >> lg], (wave & (G - 1)) * 64, 6 + lg);
        if (lane == 0) wbest[wave] = m;
      }
      __syncthreads();
      EMD_T(3);
      if (wave == 0) {
        // partial q of bidder u sits in lane q * P + u of row 0: rotations to the LEFT by P, 2P, .. < 8 (lane i receives lane i + s) merge
        // a bidder's partials into lane u
        Best m{-1e9f, -1e9f, -1};
        if ((lane & (P - 1)) < U && (lane >> lp) < G) m = wbest[(lane & (P - 1)) * G + (lane >> lp)];
        if (P <= 1) m = merge(m, row_ror<15>(m));
        if (P <= 2) m = merge(m, row_ror<14>(m));
        if (P <= 4) m = merge(m, row_ror<12>(m));
        int j = -1, k = -2 - lane;
        float inc = 0.f;
        if (lane < U) {
#pragma clang fp contract(off)
          j = list[lane], k = m.idx, inc = m.best - m.better + eps;
        }
        EMD_T(4);
        // GetMax among at most 8 lanes, in registers (max_inc / max_idx are scratch of one iteration — every target that receives bids has
        // a winner, which puts them back to -1e9 / -1, and every increment is > 0 —, so this path neither reads nor writes them):
        // the target's largest increment, then the largest bidder index within 1e-6 of it (:195-199 with the race resolved as in the oracle)
        float mi = inc;
        for (int v = 0; v < U; ++v) {
          const int kv = __builtin_amdgcn_readlane(k, v);
          const float iv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc), v));
          mi = kv == k ? fmaxf(mi, iv) : mi;
        }
        const int cj = ((double)inc - 1e-6 <= (double)mi && (double)mi <= (double)inc + 1e-6) ? j : -1;
        int wj = cj;
        for (int v = 0; v < U; ++v) {
          const int kv = __builtin_amdgcn_readlane(k, v), cv = __builtin_amdgcn_readlane(cj, v);
          wj = kv == k ? max(wj, cv) : wj;
        }
        EMD_T(5);
        int push = -1;   // the point this lane leaves unassigned: itself when it lost, the owner it displaced when it won
        if (lane < U) {
          if (last || wj == j) {
            const int prev = ass_inv[k];
            if (!last && prev != -1) as[prev] = -1, push = prev;
            ass_inv[k] = j;
            as[j] = k;
            if (!last) T[k].w += inc;
          } else {
            push = j;
          }
        }
        const unsigned long long pm = __ballot(push >= 0);
        if (push >= 0) nlist[__popcll(pm & ((1ull << lane) - 1))] = push;
        if (lane == 0) *ncnt = __popcll(pm);
        EMD_T(6);
      }
      __syncthreads();
      EMD_T(7);
      continue;
    }
    // ---- more than 8 bidders: one wavefront per bidder, 16 at a time ----
    for (int u0 = 0; u0 < U; u0 += NWAVES) {
      const int u = u0 + wave;
      if (u < U) {
        const int j = list[u];
        const Best m = bid_scan(j, 0, 6);
        if (lane == 0) {
#pragma clang fp contract(off)
          const float inc = m.best - m.better + eps;
          bid[j] = m.idx;
          bid_inc[j] = inc;
          atomicMax(reinterpret_cast<int *>(max_inc) + m.idx, __float_as_int(inc));   // inc > 0 vs stored 0 / -1e9: int order = float order
        }
      }
    }
    __syncthreads();
    // ---- GetMax ----
    for (int u = tid; u < U; u += EMD_THREADS) {
      const int j = list[u], k = bid[j];
      const float bi = bid_inc[j], mi = max_inc[k];
      if ((double)bi - 1e-6 <= (double)mi && (double)mi <= (double)bi + 1e-6) atomicMax(max_idx + k, j);
    }
    __syncthreads();
    // ---- Assign ----
    for (int u = tid; u < U; u += EMD_THREADS) {
      const int j = list[u], k = bid[j];
      if (last || max_idx[k] == j) {
        const int prev = ass_inv[k];
        if (!last && prev != -1) as[prev] = -1, nlist[atomicAdd(ncnt, 1)] = prev;
        ass_inv[k] = j;
        as[j] = k;
        // (max_idx: the winner's target forgets this round's bidder index — the reference's separate pass :203-223 clears exactly the
        // winners' targets; a loser of the same target compares its own index with j or with -1: it loses either way)
        if (!last) T[k].w += bid_inc[j], max_inc[k] = -1e9f, max_idx[k] = -1;
      } else {
        nlist[atomicAdd(ncnt, 1)] = j;
      }
    }
    __syncthreads();
  }
#ifdef DFX_EMD_PROBE
  if (tid == 0 && blockIdx.x == 0) g_emd_probe[16] = clock64() - tk_;
#endif
  __syncthreads();
  for (int j = tid; j < n; j += EMD_THREADS) {
#pragma clang fp contract(off)
    const int k = as[j];
    if (STATE_LDS) as_out[j] = k;
    const float dx = A[j * 3] - T[k].x, dy = A[j * 3 + 1] - T[k].y, dz = A[j * 3 + 2] - T[k].z;
    dist[(size_t)b * n + j] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  }
}

__global__ void k_emd_grad(const float *__restrict__ xyz1, const float *__restrict__ xyz2, const float *__restrict__ grad_dist,
                           const int32_t *__restrict__ assignment, float *__restrict__ grad_xyz1, int n, long long total) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long b = i / n;
  const int k = assignment[i];
  const float g = grad_dist[i] * 2;
  for (int c = 0; c < 3; ++c) grad_xyz1[i * 3 + c] = g * (xyz1[i * 3 + c] - xyz2[(b * n + k) * 3 + c]);
}

}  // namespace

extern "C" {

size_t dfx_emd_workspace_bytes(int B, int n) { return (size_t)(B > 0 ? B : 0) * (size_t)(n > 0 ? n : 0) * 7 * 4; }

int dfx_emd_forward_f32(const float *xyz1, const float *xyz2, float *dist, int32_t *assignment, void *workspace, int B, int n,
                        float eps, int iters, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && n >= 1 && iters >= 1, "emd_forward: bad sizes");
  if (B == 0) return DFX_OK;
  DFX_REQUIRE(n <= 8192, "emd_forward: n = %d > 8192 (the targets of one cloud live in LDS)", n);
  DFX_REQUIRE(xyz1 && xyz2 && dist && assignment && workspace, "emd_forward: null pointer");
  static dfx::PerDeviceOnce attrs;
  DFX_HIP_TRY(attrs.run([] {
    hipError_t e = dfx::set_max_lds(reinterpret_cast<const void *>(k_emd<false>), 8192 * 16);
    if (e == hipSuccess) e = dfx::set_max_lds(reinterpret_cast<const void *>(k_emd<true>), EMD_STATE_LDS_MAX_N * 60);
    return e;
  }));
  int32_t *wsi = static_cast<int32_t *>(workspace);
  float *wsf = reinterpret_cast<float *>(wsi + (size_t)B * 5 * n);
  if (n <= EMD_STATE_LDS_MAX_N && !g_emd_state_global)   // targets + prices (16 B), the auction state (32 B) and the bidders (12 B) per point in LDS
    k_emd<true><<<B, EMD_THREADS, n * 60, dfx::as_stream(stream)>>>(xyz1, xyz2, eps, iters, dist, assignment, wsi, wsf, n);
  else
    k_emd<false><<<B, EMD_THREADS, n * 16, dfx::as_stream(stream)>>>(xyz1, xyz2, eps, iters, dist, assignment, wsi, wsf, n);
  return dfx::check_launch("emd_forward");
}

void dfx_debug_emd_state_global(int on) { g_emd_state_global = on != 0; }

int dfx_emd_backward_f32(const float *xyz1, const float *xyz2, const float *grad_dist, const int32_t *assignment,
                         float *grad_xyz1, int B, int n, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && n >= 1, "emd_backward: bad sizes");
  if (B == 0) return DFX_OK;
  DFX_REQUIRE(xyz1 && xyz2 && grad_dist && assignment && grad_xyz1, "emd_backward: null pointer");
  const long long total = (long long)B * n;
  k_emd_grad<<<(int)((total + 255) / 256), 256, 0, dfx::as_stream(stream)>>>(xyz1, xyz2, grad_dist, assignment, grad_xyz1, n, total);
  return dfx::check_launch("emd_backward");
}

}  // extern "C"
