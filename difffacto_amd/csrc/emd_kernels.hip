// Approximate earth mover's distance by auction — replaces the `emd` extension
// (python/difffacto/metrics/emd/emd_cuda.cu: forward :236-284 = `iters` x {calc_unass_* :29-100, Bid :102-186,
// GetMax :188-201, Assign :203-223} + CalcDist :225-234; backward :286-316; bound in emd_module.py:17-51).
//
// The reference launches seven kernels per auction iteration (70 000 launches at the evaluation setting of 10 000
// iterations).  Here ONE persistent workgroup (16 wavefronts) runs the whole auction of one cloud pair: the targets'
// coordinates and prices stay in LDS (16 B per point: n <= 8192 in 128 KiB), the four phases are separated by workgroup
// barriers, and the loop ends as soon as nothing is unassigned (after that no reference kernel changes any state).
// Bids: one wavefront per unassigned point, lanes scan the targets strided and merge (best, second best, first index of
// the best) with DPP-free shuffles; the merge keeps the reference's result (first maximum in index order, :152-178).
// GetMax's data race (several bidders within 1e-6 of the maximum all write max_idx, :195-199) is resolved
// deterministically: the largest bidder index wins (integer atomicMax) — the oracle uses the same rule.
#include "dfx_common.h"

namespace {

constexpr int EMD_THREADS = 1024;

struct Best {
  float best, better;
  int idx;
};

// a then b, where every index in a is smaller than every index in b is NOT required: ties take the smaller index
__device__ __forceinline__ Best merge(const Best &a, const Best &b) {
  Best r;
  if (b.best > a.best || (b.best == a.best && b.idx >= 0 && (a.idx < 0 || b.idx < a.idx))) {
    r.best = b.best, r.idx = b.idx, r.better = fmaxf(a.best, b.better);
  } else {
    r.best = a.best, r.idx = a.idx, r.better = fmaxf(a.better, b.best);
  }
  return r;
}

__global__ __launch_bounds__(EMD_THREADS) void k_emd(const float *__restrict__ xyz1, const float *__restrict__ xyz2, float eps,
                                                    int iters, float *__restrict__ dist, int32_t *__restrict__ assignment,
                                                    int32_t *__restrict__ wsi, float *__restrict__ wsf, int n) {
  extern __shared__ float lds[];   // x2[n] y2[n] z2[n] price[n]
  float *X2 = lds, *Y2 = lds + n, *Z2 = lds + 2 * n, *price = lds + 3 * n;
  __shared__ int cnt;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *A = xyz1 + (size_t)b * n * 3, *Bp = xyz2 + (size_t)b * n * 3;
  int32_t *as = assignment + (size_t)b * n;
  int32_t *ass_inv = wsi + (size_t)b * 4 * n, *bid = ass_inv + n, *max_idx = bid + n, *list = max_idx + n;
  float *bid_inc = wsf + (size_t)b * 2 * n, *max_inc = bid_inc + n;
  for (int k = tid; k < n; k += EMD_THREADS) {
    X2[k] = Bp[k * 3], Y2[k] = Bp[k * 3 + 1], Z2[k] = Bp[k * 3 + 2], price[k] = 0.f;
    as[k] = -1, ass_inv[k] = -1, max_idx[k] = -1, max_inc[k] = 0.f;   // emd_module.py:28-37 (max_increments starts at 0)
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const bool last = it == iters - 1;
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int j = tid; j < n; j += EMD_THREADS)
      if (as[j] == -1) list[atomicAdd(&cnt, 1)] = j;
    __syncthreads();
    const int U = cnt;
    if (U == 0) break;
    // ---- Bid: one wavefront per unassigned point ----
    for (int u = wave; u < U; u += EMD_THREADS / 64) {
      const int j = list[u];
      const float x1 = A[j * 3], y1 = A[j * 3 + 1], z1 = A[j * 3 + 2];
      Best m{-1e9f, -1e9f, -1};
      for (int k = lane; k < n; k += 64) {
#pragma clang fp contract(off)
        const float dx = X2[k] - x1, dy = Y2[k] - y1, dz = Z2[k] - z1;
        const float s = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const float d = (float)(3.0 - (double)sqrtf(s) - (double)price[k]);   // `3.0` is a double literal in the reference (:151)
        if (d > m.best) m.better = m.best, m.best = d, m.idx = k;
        else if (d > m.better) m.better = d;
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        Best t;
        t.best = __shfl_xor(m.best, o, 64), t.better = __shfl_xor(m.better, o, 64), t.idx = __shfl_xor(m.idx, o, 64);
        m = merge(m, t);
      }
      if (lane == 0) {
#pragma clang fp contract(off)
        const float inc = m.best - m.better + eps;
        bid[j] = m.idx;
        bid_inc[j] = inc;
        atomicMax(reinterpret_cast<int *>(max_inc) + m.idx, __float_as_int(inc));   // inc > 0 vs stored 0 / -1e9: int order = float order
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
        if (!last && prev != -1) as[prev] = -1;
        ass_inv[k] = j;
        as[j] = k;
        if (!last) price[k] += bid_inc[j], max_inc[k] = -1e9f;
      }
    }
    __syncthreads();
    if (!last)
      for (int u = tid; u < U; u += EMD_THREADS) {   // winners' targets: forget this round's bidder index
        const int k = bid[list[u]];
        if (as[list[u]] == k) max_idx[k] = -1;
      }
    __syncthreads();
  }
  __syncthreads();
  for (int j = tid; j < n; j += EMD_THREADS) {
#pragma clang fp contract(off)
    const int k = as[j];
    const float dx = A[j * 3] - X2[k], dy = A[j * 3 + 1] - Y2[k], dz = A[j * 3 + 2] - Z2[k];
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

size_t dfx_emd_workspace_bytes(int B, int n) { return (size_t)(B > 0 ? B : 0) * (size_t)(n > 0 ? n : 0) * 6 * 4; }

int dfx_emd_forward_f32(const float *xyz1, const float *xyz2, float *dist, int32_t *assignment, void *workspace, int B, int n,
                        float eps, int iters, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && n >= 1 && iters >= 1, "emd_forward: bad sizes");
  if (B == 0) return DFX_OK;
  DFX_REQUIRE(n <= 8192, "emd_forward: n = %d > 8192 (the targets of one cloud live in LDS)", n);
  DFX_REQUIRE(xyz1 && xyz2 && dist && assignment && workspace, "emd_forward: null pointer");
  const int lds_bytes = n * 16;
  static bool attr = false;
  if (!attr) {
    DFX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_emd), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16));
    attr = true;
  }
  int32_t *wsi = static_cast<int32_t *>(workspace);
  float *wsf = reinterpret_cast<float *>(wsi + (size_t)B * 4 * n);
  k_emd<<<B, EMD_THREADS, lds_bytes, dfx::as_stream(stream)>>>(xyz1, xyz2, eps, iters, dist, assignment, wsi, wsf, n);
  return dfx::check_launch("emd_forward");
}

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
