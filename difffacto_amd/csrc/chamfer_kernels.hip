// Chamfer-L2 distance kernels for gfx950 (needed to REPORT Chamfer deltas on the GPU box).
//
// Semantics of the reference extension python/difffacto/metrics/chamfer_dist/chamfer.cu:
//   forward  :15-145  dist1[b,i] = min_k |xyz1[b,i] - xyz2[b,k]|^2, idx1 = arg min (strict '<' in k order: the first
//                     minimum wins, also across its 512-point tiles, :137), and symmetrically dist2 / idx2
//   backward :173-201 grad_xyz1[i] += 2 g1[i] (x1_i - x2_idx1[i]);  grad_xyz2[idx1[i]] -= the same   (atomicAdd)
// Squared distances use the mul,fma,fma evaluation order stated in oracle/pointnet2.c.
// Mapping: one or two query points per thread, the other cloud streams through LDS in tiles of 16-byte records (broadcast reads);
// HBM traffic = both clouds once per 256-query workgroup (L2-resident), outputs once.  Bound: VALU (N*M fma chains).
#include "dfx_common.h"

namespace {

__device__ __forceinline__ float sq3(float a, float b, float c) {
  float t = __fmul_rn(a, a);
  t = __fmaf_rn(b, b, t);
  t = __fmaf_rn(c, c, t);
  return t;
}

constexpr int CD_TILE = 1024;   // 16 KiB of records: eight workgroups per CU

// Q query points per thread (2 whenever that still leaves every CU a workgroup: one LDS read and one loop step per two distances —
// 4.19 -> 4.44 T pair-distances/s at 128 x 2048 x 2048, 5.3 T at 1024 clouds), the reference cloud through LDS as 16-byte records (one broadcast ds_read_b128 per
// reference point instead of three ds_read_b32), the first reference point taken outside the loop (the reference's "k == 0 or closer"
// test, chamfer.cu:137, is then the plain strict '<').
template <int Q>
__global__ void __launch_bounds__(256) chamfer_nn_kernel(const float *__restrict__ query, const float *__restrict__ ref,
                                                         float *__restrict__ dist, int32_t *__restrict__ idx, int n,
                                                         int m, int wg_per_cloud) {
  __shared__ float4 sp[CD_TILE];
  const int b = blockIdx.x / wg_per_cloud;
  const int j0 = (blockIdx.x % wg_per_cloud) * (256 * Q) + threadIdx.x;
  const float *Qp = query + (size_t)b * n * 3;
  const float *R = ref + (size_t)b * m * 3;
  float qx[Q], qy[Q], qz[Q], best[Q];
  int besti[Q];
  const float r0x = R[0], r0y = R[1], r0z = R[2];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = min(j0 + 256 * q, n - 1);   // (threads past the end work on the last point and store nothing)
    qx[q] = Qp[j * 3 + 0], qy[q] = Qp[j * 3 + 1], qz[q] = Qp[j * 3 + 2];
    best[q] = sq3(r0x - qx[q], r0y - qy[q], r0z - qz[q]), besti[q] = 0;
  }
  for (int base = 0; base < m; base += CD_TILE) {
    const int cnt = min(CD_TILE, m - base);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += 256) sp[k] = make_float4(R[(size_t)(base + k) * 3], R[(size_t)(base + k) * 3 + 1], R[(size_t)(base + k) * 3 + 2], 0.f);
    __syncthreads();
#pragma unroll 4
    for (int k = base == 0 ? 1 : 0; k < cnt; ++k) {
      const float4 r = sp[k];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float d = sq3(r.x - qx[q], r.y - qy[q], r.z - qz[q]);
        const bool c = d < best[q];
        best[q] = c ? d : best[q], besti[q] = c ? base + k : besti[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = j0 + 256 * q;
    if (j < n) dist[(size_t)b * n + j] = best[q], idx[(size_t)b * n + j] = besti[q];
  }
}

__global__ void __launch_bounds__(256) chamfer_grad_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                           const float *__restrict__ grad_dist1,
                                                           const int32_t *__restrict__ idx1, float *__restrict__ grad_xyz1,
                                                           float *__restrict__ grad_xyz2, int n, int m, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / n;
    const int j2 = idx1[i];
    const float *p1 = xyz1 + i * 3;
    const float *p2 = xyz2 + (b * m + j2) * 3;
    const float g = grad_dist1[i] * 2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = g * (p1[c] - p2[c]);
      atomicAdd(grad_xyz1 + i * 3 + c, v);
      atomicAdd(grad_xyz2 + (b * m + j2) * 3 + c, -v);
    }
  }
}

}  // namespace

extern "C" {

int dfx_chamfer_forward_f32(const float *xyz1, const float *xyz2, float *dist1, float *dist2, int32_t *idx1,
                            int32_t *idx2, int B, int N, int M, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && N >= 0 && M >= 0, "chamfer_forward: negative size");
  if (B == 0) return DFX_OK;
  DFX_REQUIRE(N > 0 && M > 0, "chamfer_forward: empty cloud");
  DFX_REQUIRE(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2, "chamfer_forward: null pointer");
  hipStream_t st = dfx::as_stream(stream);
  auto run = [&](const float *q, const float *r, float *d, int32_t *ix, int n, int m) {
    if ((long long)B * ((n + 511) / 512) >= 256) {   // a workgroup per CU even with two query points per thread
      const int w = (n + 511) / 512;
      chamfer_nn_kernel<2><<<B * w, 256, 0, st>>>(q, r, d, ix, n, m, w);
    } else {
      const int w = (n + 255) / 256;
      chamfer_nn_kernel<1><<<B * w, 256, 0, st>>>(q, r, d, ix, n, m, w);
    }
  };
  run(xyz1, xyz2, dist1, idx1, N, M);
  run(xyz2, xyz1, dist2, idx2, M, N);
  return dfx::check_launch("chamfer_forward");
}

int dfx_chamfer_backward_f32(const float *xyz1, const float *xyz2, const int32_t *idx1, const int32_t *idx2,
                             const float *grad_dist1, const float *grad_dist2, float *grad_xyz1, float *grad_xyz2,
                             int B, int N, int M, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && N >= 0 && M >= 0, "chamfer_backward: negative size");
  if (B == 0 || N == 0 || M == 0) return DFX_OK;
  DFX_REQUIRE(xyz1 && xyz2 && idx1 && idx2 && grad_dist1 && grad_dist2 && grad_xyz1 && grad_xyz2,
              "chamfer_backward: null pointer");
  hipStream_t st = dfx::as_stream(stream);
  DFX_HIP_TRY(hipMemsetAsync(grad_xyz1, 0, sizeof(float) * (size_t)B * N * 3, st));
  DFX_HIP_TRY(hipMemsetAsync(grad_xyz2, 0, sizeof(float) * (size_t)B * M * 3, st));
  const long long t1 = (long long)B * N, t2 = (long long)B * M;
  auto grid = [](long long t) { long long g = (t + 255) / 256; return (int)(g > 8192 ? 8192 : g); };
  chamfer_grad_kernel<<<grid(t1), 256, 0, st>>>(xyz1, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2, N, M, t1);
  chamfer_grad_kernel<<<grid(t2), 256, 0, st>>>(xyz2, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1, M, N, t2);
  return dfx::check_launch("chamfer_backward");
}

}  // extern "C"
