/* Plain-C restatement of the pointnet2_ops CUDA kernels (ORACLE — test infrastructure only).
 *
 * Each function restates one kernel of the reference
 *   pointnet2_ops_lib/pointnet2_ops/_ext-src/src/{sampling_gpu,ball_query_gpu,group_points_gpu,interpolate_gpu}.cu
 * as a serial loop nest with the same index order, the same comparisons and the same tie rules.
 *
 * PARITY UNPINNED for these kernels: the reference holds no tests/golden vectors for them and
 * its CUDA sources cannot be built here.  Two stated assumptions replace what cannot be observed:
 *   (1) float expressions of the form a*a + b*b + c*c are evaluated the way nvcc's default
 *       -fmad=true contracts them: t = a*a; t = fma(b,b,t); t = fma(c,c,t)  (helper sq3()),
 *       and p1*w1 + p2*w2 + p3*w3 likewise (mul, fma, fma).
 *   (2) opt_n_threads() uses the host's double log() exactly as include/cuda_utils.h:15-19.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TOTAL_THREADS 512 /* include/cuda_utils.h:13 */

/* include/cuda_utils.h:15-19 */
int oracle_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > TOTAL_THREADS) t = TOTAL_THREADS;
  if (t < 1) t = 1;
  return t;
}

static inline float sq3(float a, float b, float c) {
  float t = a * a;
  t = fmaf(b, b, t);
  t = fmaf(c, c, t);
  return t;
}

/* sampling_gpu.cu:8-20  out[b,c,j] = points[b,c,idx[b,j]] */
void oracle_gather_points(int b, int c, int n, int m, const float *points, const int32_t *idx, float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        int a = idx[i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* sampling_gpu.cu:34-47  scatter-add (atomicAdd in the reference: summation order unspecified there) */
void oracle_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx,
                               float *grad_points) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        int a = idx[i * m + j];
        grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * m + j];
      }
}

/* sampling_gpu.cu:59-173.  temp must be pre-filled with 1e10 by the caller (sampling.cpp:74-76).
 * Emulates the block of `block_size` threads: thread tid scans k = tid, tid+bs, ... keeping the first
 * strictly-greater candidate, then the shared-memory tree (__update :59-65, "v2 > v1 ? i2 : i1"). */
void oracle_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int32_t *idxs) {
  if (m <= 0) return;
  const int bs = oracle_opt_n_threads(n);
  float *dists = (float *)malloc(sizeof(float) * bs);
  int *dists_i = (int *)malloc(sizeof(int) * bs);
  for (int bi = 0; bi < b; ++bi) {
    const float *ds = dataset + (size_t)bi * n * 3;
    float *tp = temp + (size_t)bi * n;
    int32_t *out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += bs) {
          const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
          const float mag = sq3(x2, y2, z2);
          if (mag <= 1e-3) continue; /* :100-101 (double literal, float promoted) */
          const float d = sq3(x2 - x1, y2 - y1, z2 - z1);
          const float d2 = fminf(d, tp[k]);
          tp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1) {
        for (int tid = 0; tid < s; ++tid) {
          const float v1 = dists[tid], v2 = dists[tid + s];
          const int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/* ball_query_gpu.cu:9-44.  idx must be zero-filled by the caller (ball_query.cpp:19-21). */
void oracle_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                       int32_t *idx) {
  const float radius2 = radius * radius;
  for (int bi = 0; bi < b; ++bi) {
    const float *X = xyz + (size_t)bi * n * 3;
    const float *Q = new_xyz + (size_t)bi * m * 3;
    int32_t *I = idx + (size_t)bi * m * nsample;
    for (int j = 0; j < m; ++j) {
      const float nx = Q[j * 3 + 0], ny = Q[j * 3 + 1], nz = Q[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float d2 = sq3(nx - X[k * 3 + 0], ny - X[k * 3 + 1], nz - X[k * 3 + 2]);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) I[j * nsample + l] = k;
          I[j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* group_points_gpu.cu:8-28  out[b,c,j,k] = points[b,c,idx[b,j,k]] */
void oracle_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int32_t *idx,
                         float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *P = points + (size_t)bi * n * c;
    const int32_t *I = idx + (size_t)bi * npoints * nsample;
    float *O = out + (size_t)bi * npoints * nsample * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) O[((size_t)l * npoints + j) * nsample + k] = P[(size_t)l * n + I[j * nsample + k]];
  }
}

/* group_points_gpu.cu:43-64 */
void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                              const int32_t *idx, float *grad_points) {
  for (int bi = 0; bi < b; ++bi) {
    const float *G = grad_out + (size_t)bi * npoints * nsample * c;
    const int32_t *I = idx + (size_t)bi * npoints * nsample;
    float *P = grad_points + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) P[(size_t)l * n + I[j * nsample + k]] += G[((size_t)l * npoints + j) * nsample + k];
  }
}

/* interpolate_gpu.cu:9-59 (best* are doubles, d is float; strict '<' keeps the earliest on ties) */
void oracle_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *U = unknown + (size_t)bi * n * 3;
    const float *K = known + (size_t)bi * m * 3;
    float *D = dist2 + (size_t)bi * n * 3;
    int32_t *I = idx + (size_t)bi * n * 3;
    for (int j = 0; j < n; ++j) {
      const float ux = U[j * 3 + 0], uy = U[j * 3 + 1], uz = U[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float d = sq3(ux - K[k * 3 + 0], uy - K[k * 3 + 1], uz - K[k * 3 + 2]);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      D[j * 3 + 0] = (float)best1; D[j * 3 + 1] = (float)best2; D[j * 3 + 2] = (float)best3;
      I[j * 3 + 0] = besti1; I[j * 3 + 1] = besti2; I[j * 3 + 2] = besti3;
    }
  }
}

/* interpolate_gpu.cu:72-101 */
void oracle_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                              const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *P = points + (size_t)bi * m * c;
    const int32_t *I = idx + (size_t)bi * n * 3;
    const float *Wt = weight + (size_t)bi * n * 3;
    float *O = out + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        float t = P[(size_t)l * m + I[j * 3 + 0]] * Wt[j * 3 + 0];
        t = fmaf(P[(size_t)l * m + I[j * 3 + 1]], Wt[j * 3 + 1], t);
        t = fmaf(P[(size_t)l * m + I[j * 3 + 2]], Wt[j * 3 + 2], t);
        O[(size_t)l * n + j] = t;
      }
  }
}

/* interpolate_gpu.cu:116-143 */
void oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx,
                                   const float *weight, float *grad_points) {
  for (int bi = 0; bi < b; ++bi) {
    const float *G = grad_out + (size_t)bi * n * c;
    const int32_t *I = idx + (size_t)bi * n * 3;
    const float *Wt = weight + (size_t)bi * n * 3;
    float *P = grad_points + (size_t)bi * m * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float g = G[(size_t)l * n + j];
        P[(size_t)l * m + I[j * 3 + 0]] += g * Wt[j * 3 + 0];
        P[(size_t)l * m + I[j * 3 + 1]] += g * Wt[j * 3 + 1];
        P[(size_t)l * m + I[j * 3 + 2]] += g * Wt[j * 3 + 2];
      }
  }
}

/* ---- Chamfer-L2 (python/difffacto/metrics/chamfer_dist/chamfer.cu) ------------------------------------------ */
/* forward :15-145: nearest neighbour of every query point, first minimum in k order (strict '<', :47,:137) */
void oracle_chamfer_nn(int b, int n, int m, const float *query, const float *ref, float *dist, int32_t *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *Q = query + (size_t)bi * n * 3;
    const float *R = ref + (size_t)bi * m * 3;
    for (int j = 0; j < n; ++j) {
      float best = 0.f;
      int besti = 0;
      for (int k = 0; k < m; ++k) {
        const float d = sq3(R[k * 3 + 0] - Q[j * 3 + 0], R[k * 3 + 1] - Q[j * 3 + 1], R[k * 3 + 2] - Q[j * 3 + 2]);
        if (k == 0 || d < best) {
          best = d;
          besti = k;
        }
      }
      dist[(size_t)bi * n + j] = best;
      idx[(size_t)bi * n + j] = besti;
    }
  }
}

/* backward :173-201 (one direction; the caller runs it for both) */
void oracle_chamfer_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *grad_dist1,
                         const int32_t *idx1, float *grad_xyz1, float *grad_xyz2) {
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < n; ++j) {
      const size_t i = (size_t)bi * n + j;
      const int j2 = idx1[i];
      const float g = grad_dist1[i] * 2;
      for (int c = 0; c < 3; ++c) {
        const float v = g * (xyz1[i * 3 + c] - xyz2[((size_t)bi * m + j2) * 3 + c]);
        grad_xyz1[i * 3 + c] += v;
        grad_xyz2[((size_t)bi * m + j2) * 3 + c] -= v;
      }
    }
}

/* ---- approximate EMD by auction (python/difffacto/metrics/emd/emd_cuda.cu, emd_module.py:17-41) -------------------- */
/* One iteration = calc_unass_* :29-100 (list the unassigned points), Bid :102-186, GetMax :188-201, Assign :203-223;
 * after `iters` iterations CalcDist :225-234.  Sequential restatement with the kernels' scan order:
 *   - value of target k for bidder j: d = 3.0 - sqrtf(|x2_k - x1_j|^2) - price[k], evaluated in DOUBLE and rounded to
 *     float (`3.0` is a double literal, :151); the squared norm with nvcc's contraction (mul, fma, fma);
 *   - best = FIRST maximum in k order (strict '>' in the per-thread scans :152-160 and in the ordered merge :171-178),
 *     better = the second largest value; increment = best - better + eps (float);
 *   - GetMax lets every bidder whose increment is within 1e-6 (double compare, :195) of the target's maximum write
 *     max_idx: a data race in the reference; here the LARGEST bidder index wins (stated deviation, ties are rare);
 *   - the last iteration assigns every still-unassigned bidder to its bid without evicting (:209-211).
 * Iterations stop early once nothing is unassigned (no kernel changes any state after that). */
static void emd_forward_impl(int b, int n, const float *xyz1, const float *xyz2, float eps, int iters, float *dist,
                             int32_t *assignment, int32_t *unassigned) {
  int32_t *ass_inv = (int32_t *)malloc(sizeof(int32_t) * n), *bid = (int32_t *)malloc(sizeof(int32_t) * n),
          *max_idx = (int32_t *)malloc(sizeof(int32_t) * n);
  float *price = (float *)malloc(sizeof(float) * n), *bid_inc = (float *)malloc(sizeof(float) * n),
        *max_inc = (float *)malloc(sizeof(float) * n);
  for (int bi = 0; bi < b; ++bi) {
    const float *A = xyz1 + (size_t)bi * n * 3, *Bp = xyz2 + (size_t)bi * n * 3;
    int32_t *as = assignment + (size_t)bi * n;
    for (int j = 0; j < n; ++j) as[j] = -1, ass_inv[j] = -1, price[j] = 0.f, max_inc[j] = 0.f, max_idx[j] = 0, bid[j] = 0, bid_inc[j] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int last = it == iters - 1;
      int any = 0;
      for (int j = 0; j < n; ++j) {
        if (as[j] != -1) continue;
        any = 1;
        float best = -1e9f, better = -1e9f;
        int best_i = -1;
        const float x1 = A[j * 3], y1 = A[j * 3 + 1], z1 = A[j * 3 + 2];
        for (int k = 0; k < n; ++k) {
          const float x2 = Bp[k * 3] - x1, y2 = Bp[k * 3 + 1] - y1, z2 = Bp[k * 3 + 2] - z1;
          const float s = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
          const float d = (float)(3.0 - (double)sqrtf(s) - (double)price[k]);
          if (d > best) better = best, best = d, best_i = k;
          else if (d > better) better = d;
        }
        bid[j] = best_i;
        bid_inc[j] = best - better + eps;
        if (bid_inc[j] > max_inc[best_i]) max_inc[best_i] = bid_inc[j];
      }
      if (unassigned) {   /* (tests: how many points bid in iteration `it` of the first pair) */
        int u = 0;
        for (int j = 0; j < n; ++j) u += as[j] == -1;
        if (bi == 0) unassigned[it] = u;
      }
      if (!any) break;
      for (int j = 0; j < n; ++j) {   /* GetMax: ascending j, the largest matching bidder index stays */
        if (as[j] != -1) continue;
        const float mi = max_inc[bid[j]];
        if ((double)bid_inc[j] - 1e-6 <= (double)mi && (double)mi <= (double)bid_inc[j] + 1e-6) max_idx[bid[j]] = j;
      }
      for (int j = 0; j < n; ++j) {   /* Assign: decisions use the state before this phase (threads are independent) */
        if (as[j] != -1) continue;
        const int k = bid[j];
        if (last || max_idx[k] == j) {
          const int prev = ass_inv[k];
          if (!last && prev != -1) as[prev] = -1;
          ass_inv[k] = j;
          as[j] = k;
          price[k] += bid_inc[j];
          max_inc[k] = -1e9f;
        }
      }
    }
    for (int j = 0; j < n; ++j) {
      const int k = as[j];
      const float dx = A[j * 3] - Bp[k * 3], dy = A[j * 3 + 1] - Bp[k * 3 + 1], dz = A[j * 3 + 2] - Bp[k * 3 + 2];
      dist[(size_t)bi * n + j] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    }
  }
  free(ass_inv); free(bid); free(max_idx); free(price); free(bid_inc); free(max_inc);
}

void oracle_emd_forward(int b, int n, const float *xyz1, const float *xyz2, float eps, int iters, float *dist,
                        int32_t *assignment) {
  emd_forward_impl(b, n, xyz1, xyz2, eps, iters, dist, assignment, NULL);
}

/* the same auction; unassigned[it] = number of bidders of iteration it of the first pair (0 from convergence on; caller zero-fills) */
void oracle_emd_forward_trace(int b, int n, const float *xyz1, const float *xyz2, float eps, int iters, float *dist,
                              int32_t *assignment, int32_t *unassigned) {
  emd_forward_impl(b, n, xyz1, xyz2, eps, iters, dist, assignment, unassigned);
}

/* NmDistanceGradKernel :286-301: grad_xyz1[j] += 2 grad_dist[j] (x1_j - x2_assignment[j]); grad_xyz2 stays zero */
void oracle_emd_backward(int b, int n, const float *xyz1, const float *xyz2, const float *grad_dist, const int32_t *assignment,
                         float *grad_xyz1) {
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < n; ++j) {
      const size_t i = (size_t)bi * n + j;
      const int k = assignment[i];
      const float g = grad_dist[i] * 2;
      for (int c = 0; c < 3; ++c) grad_xyz1[i * 3 + c] += g * (xyz1[i * 3 + c] - xyz2[((size_t)bi * n + k) * 3 + c]);
    }
}
