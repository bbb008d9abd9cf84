// LDS-tiled GEMMs on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate) over fp32 operands in HBM, for the
// training path (train_kernels.hip): operands are rounded to bf16 (RNE) on their way into LDS, results stay fp32.
//
//   NT   C (M, N) = X (M, K) W (N, K)^T + bias + residual       linear forward and dX (W = transposed weight copy)
//   TN   C (O, I) = dY (R, O)^T X (R, I), column sums of dY     weight / bias gradient, one partial tile per slab of rows
//
// Workgroup = 256 threads, 128 x 128 output tile, K consumed 32 at a time through two LDS buffers (one barrier per
// K tile, the next tile's global loads in flight during the MFMAs).  Evaluated transposed like every linear layer of
// NT puts the rows on the MFMA M axis (accumulator registers) and the output channels on the lanes, so that one store
// instruction writes 128-byte row segments; TN puts dY's columns (o) on the M axis and X's columns (i) on the lanes.  Each wavefront owns a
// 64 x 64 quadrant: two fragments of either operand per 16-deep step feed four MFMAs.  LDS rows are 32 bf16 + 8 padding
// (80 bytes: the 16-byte fragment reads of 32 consecutive rows fall on distinct bank groups).
// The TN variant transposes while staging: a thread loads an 8-row x 4-column block (rows = the reduction index) and
// writes, per column, the eight values as one 16-byte bf16 fragment.
#pragma once
#include "dfx_common.h"

namespace dfx {
namespace gemm {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int TILE = 128, TK = 32, LROW = 40;   // LDS row: 32 bf16 + 8 pad

struct GemmArgs {
  const float *A; int lda;   // NT: X (M, K)          TN: dY (R, O)
  const float *B; int ldb;   // NT: W (N, K)          TN: X (R, I)
  const float *bias;         // NT: (N) or nullptr
  const float *R; int ldr;   // NT: residual (M, N) or nullptr (may alias C)
  float *C; int ldc;         // NT: (M, N)            TN: partial tiles [slab][O][I], ldc = I
  float *bpart;              // TN: [slab][O] column sums of dY (exact fp32) or nullptr
  int M, N, K;               // NT: rows, outputs, inputs      TN: M = O, N = I, K = R (all rows)
  int rows_per_slab;         // TN: multiple of 32
  int a_bf16, b_bf16;        // the operand is already stored as bf16 (lda / ldb in elements): NT: A only; TN: A and / or B
  int c_bf16;                // NT: store the result as bf16 (ldc in elements)
};

// (A GEGLU-backward epilogue on the dX product was tried and dropped: 664 us fused against 244 + 388 us for the product and
// the element-wise kernel, which streams at 5 TB/s; the dword-per-lane epilogue accesses do not.)
template <bool TN>
__global__ __launch_bounds__(256) void k_gemm_bf16(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[4 * TILE * LROW];   // 40 KiB: W and X tiles x 2 buffers; NT epilogue staging
  __bf16(*Ws)[TILE * LROW] = reinterpret_cast<__bf16(*)[TILE * LROW]>(smem);
  __bf16(*Xs)[TILE * LROW] = reinterpret_cast<__bf16(*)[TILE * LROW]>(smem + 2 * TILE * LROW);
  __shared__ float bsum[4][TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, kq = lane >> 5;
  // tile origins: w0 on the M axis of the MFMA (NT: output channel n; TN: o), x0 on the N axis (NT: row m; TN: i)
  const int w0 = (TN ? blockIdx.y : blockIdx.x) * TILE, x0 = (TN ? blockIdx.x : blockIdx.y) * TILE;
  long long kbeg = 0, kend = a.K;
  if (TN) {
    kbeg = (long long)blockIdx.z * a.rows_per_slab;
    kend = kbeg + a.rows_per_slab < a.K ? kbeg + a.rows_per_slab : a.K;
  }
  const int nk = (int)((kend - kbeg + TK - 1) / TK);

  v4f gw[TN ? 8 : 4], gx[TN ? 8 : 4];   // staging registers (TN: a thread stages ONE operand, 8 x float4; NT: both, 4 + 4)
  v2u gb[8];                            // TN, bf16-stored operand: 8 rows x 4 columns = 8 x 8 bytes
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  auto load_tile = [&](int kt) {
    const long long k0 = kbeg + (long long)kt * TK;
    if (!TN) {
      const int row = tid >> 1, ks = (tid & 1) * 16;
      const int xr = x0 + row < a.M ? x0 + row : a.M - 1;
      const float *pw = a.B + (size_t)(w0 + row) * a.ldb + k0 + ks;
#pragma unroll
      for (int e = 0; e < 4; ++e) gw[e] = reinterpret_cast<const v4f *>(pw)[e];
      if (a.a_bf16) {   // 16 bf16 = 32 bytes, copied to LDS as they are
        const __bf16 *px = reinterpret_cast<const __bf16 *>(a.A) + (size_t)xr * a.lda + k0 + ks;
        gx[0] = reinterpret_cast<const v4f *>(px)[0], gx[1] = reinterpret_cast<const v4f *>(px)[1];
      } else {
        const float *px = a.A + (size_t)xr * a.lda + k0 + ks;
#pragma unroll
        for (int e = 0; e < 4; ++e) gx[e] = reinterpret_cast<const v4f *>(px)[e];
      }
    } else {
      const int tt = tid & 127, cg = tt & 31, rg = tt >> 5;
      const bool isw = tid < 128;
      const int ld = isw ? a.lda : a.ldb, c0 = (isw ? w0 : x0) + 4 * cg;
      if (isw ? a.a_bf16 : a.b_bf16) {   // four bf16 columns = 8 bytes per row
        const __bf16 *p = reinterpret_cast<const __bf16 *>(isw ? a.A : a.B) + c0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const long long r = k0 + 8 * rg + e;
          gb[e] = r < kend ? *reinterpret_cast<const v2u *>(p + (size_t)r * ld) : v2u{0u, 0u};
        }
      } else {
        const float *p = (isw ? a.A : a.B) + c0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const long long r = k0 + 8 * rg + e;
          gw[e] = r < kend ? *reinterpret_cast<const v4f *>(p + (size_t)r * ld) : v4f{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
    if (!TN) {
      const int row = tid >> 1, ks = (tid & 1) * 16;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const v8f fw = {gw[2 * h][0], gw[2 * h][1], gw[2 * h][2], gw[2 * h][3], gw[2 * h + 1][0], gw[2 * h + 1][1], gw[2 * h + 1][2], gw[2 * h + 1][3]};
        *reinterpret_cast<v8bf *>(&Ws[buf][row * LROW + ks + 8 * h]) = __builtin_convertvector(fw, v8bf);
        if (a.a_bf16) {
          *reinterpret_cast<v4f *>(&Xs[buf][row * LROW + ks + 8 * h]) = gx[h];
        } else {
          const v8f fx = {gx[2 * h][0], gx[2 * h][1], gx[2 * h][2], gx[2 * h][3], gx[2 * h + 1][0], gx[2 * h + 1][1], gx[2 * h + 1][2], gx[2 * h + 1][3]};
          *reinterpret_cast<v8bf *>(&Xs[buf][row * LROW + ks + 8 * h]) = __builtin_convertvector(fx, v8bf);
        }
      }
    } else {
      const int tt = tid & 127, cg = tt & 31, rg = tt >> 5;
      __bf16 *S = tid < 128 ? Ws[buf] : Xs[buf];
      if (tid < 128 ? a.a_bf16 : a.b_bf16) {
        // rows e hold columns (0, 1) in word 0 and (2, 3) in word 1: gather column c of the eight rows with byte permutes
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
          v4u o;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            o[k] = __builtin_amdgcn_perm(c < 2 ? gb[2 * k + 1][0] : gb[2 * k + 1][1], c < 2 ? gb[2 * k][0] : gb[2 * k][1], sel);
          *reinterpret_cast<v4u *>(&S[(4 * cg + c) * LROW + 8 * rg]) = o;
          if (tid < 128 && a.bpart) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) t += __builtin_bit_cast(float, o[k] << 16) + __builtin_bit_cast(float, o[k] & 0xffff0000u);
            colsum[c] += t;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const v8f f = {gw[0][c], gw[1][c], gw[2][c], gw[3][c], gw[4][c], gw[5][c], gw[6][c], gw[7][c]};
          *reinterpret_cast<v8bf *>(&S[(4 * cg + c) * LROW + 8 * rg]) = __builtin_convertvector(f, v8bf);
          if (tid < 128) colsum[c] += (f[0] + f[1]) + (f[2] + f[3]) + ((f[4] + f[5]) + (f[6] + f[7]));
        }
      }
    }
  };

  v16f acc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.f;

  if (nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      v8bf wf[2], xf[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        wf[p] = *reinterpret_cast<const v8bf *>(&Ws[buf][(64 * wc + 32 * p + li) * LROW + 16 * s + 8 * kq]);
        xf[p] = *reinterpret_cast<const v8bf *>(&Xs[buf][(64 * wr + 32 * p + li) * LROW + 16 * s + 8 * kq]);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[p][q] = TN ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[p], xf[q], acc[p][q], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[q], wf[p], acc[p][q], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (!TN) {
    // NT epilogue through LDS: rows sit on the MFMA M axis (registers) and output channels on the lanes, so a wavefront
    // writes its quadrant into a staging tile two rows x 32 consecutive floats per instruction (conflict-free); the
    // workgroup then streams the tile out with 16-byte non-temporal stores, a full 512-byte (fp32) / 256-byte (bf16) row
    // segment per 32 / 16 threads, bias and residual added on the way.  64 rows per pass (the staging tile is the 40 KiB
    // of the operand buffers: 64 x 132 floats).
    constexpr int SROW = TILE + 4;
    float *stage = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (wr == h) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg)
              stage[(32 * q + (rg & 3) + 8 * (rg >> 2) + 4 * kq) * SROW + 64 * wc + 32 * p + li] = acc[p][q][rg];
      }
      __syncthreads();
      const int mrow0 = x0 + 64 * h;
      if (a.c_bf16) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = it * 256 + tid, row = idx >> 4, c8 = (idx & 15) * 8, m = mrow0 + row;
          if (m < a.M) {
            const v4f s0 = *reinterpret_cast<const v4f *>(&stage[row * SROW + c8]), s1 = *reinterpret_cast<const v4f *>(&stage[row * SROW + c8 + 4]);
            v8f v = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            if (a.bias) {
              const v4f b0 = *reinterpret_cast<const v4f *>(a.bias + w0 + c8), b1 = *reinterpret_cast<const v4f *>(a.bias + w0 + c8 + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += b0[e], v[4 + e] += b1[e];
            }
            if (a.R) {
              const v4f r0 = *reinterpret_cast<const v4f *>(a.R + (size_t)m * a.ldr + w0 + c8), r1 = *reinterpret_cast<const v4f *>(a.R + (size_t)m * a.ldr + w0 + c8 + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += r0[e], v[4 + e] += r1[e];
            }
            __builtin_nontemporal_store(__builtin_bit_cast(v4f, __builtin_convertvector(v, v8bf)),
                                        reinterpret_cast<v4f *>(reinterpret_cast<__bf16 *>(a.C) + (size_t)m * a.ldc + w0 + c8));
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int idx = it * 256 + tid, row = idx >> 5, c4 = (idx & 31) * 4, m = mrow0 + row;
          if (m < a.M) {
            v4f v = *reinterpret_cast<const v4f *>(&stage[row * SROW + c4]);
            if (a.bias) {
              const v4f bv = *reinterpret_cast<const v4f *>(a.bias + w0 + c4);
              v += bv;
            }
            if (a.R) v += *reinterpret_cast<const v4f *>(a.R + (size_t)m * a.ldr + w0 + c4);   // may be C itself: same thread, read before write
            __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(a.C + (size_t)m * a.ldc + w0 + c4));
          }
        }
      }
      __syncthreads();
    }
  } else {
    float *pp = a.C + (size_t)blockIdx.z * a.M * a.N;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = x0 + 64 * wr + 32 * q + li;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int o = w0 + 64 * wc + 32 * p + (rg & 3) + 8 * (rg >> 2) + 4 * kq;
          pp[(size_t)o * a.N + i] = acc[p][q][rg];
        }
      }
    if (a.bpart && blockIdx.x == 0) {   // column sums of dY over this slab: 4 row groups x 32 column groups of 4
      if (tid < 128) {
        const int tt = tid, cg = tt & 31, rg = tt >> 5;
#pragma unroll
        for (int c = 0; c < 4; ++c) bsum[rg][4 * cg + c] = colsum[c];
      }
      __syncthreads();
      if (tid < TILE) a.bpart[(size_t)blockIdx.z * a.M + w0 + tid] = (bsum[0][tid] + bsum[1][tid]) + (bsum[2][tid] + bsum[3][tid]);
    }
  }
}

// NT usable: N % 128 == 0, K % 32 == 0, 16-byte aligned rows; TN usable: O % 128 == 0, I % 128 == 0, rows_per_slab % 32 == 0
inline bool nt_ok(const GemmArgs &a) {
  return a.N % TILE == 0 && a.K % TK == 0 && a.M >= 256 && a.lda % (a.a_bf16 ? 8 : 4) == 0 && a.ldb % 4 == 0 && a.ldc % 4 == 0 &&
         ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B) | reinterpret_cast<uintptr_t>(a.C)) & 15) == 0 &&
         (!a.R || (a.ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(a.R) & 15) == 0)) &&
         (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) && (!a.c_bf16 || a.ldc % 8 == 0);
}
inline bool tn_ok(const GemmArgs &a) {
  return a.M % TILE == 0 && a.N % TILE == 0 && a.rows_per_slab % TK == 0 && a.lda % 4 == 0 && a.ldb % 4 == 0 &&   // (8-byte rows of 4 bf16 need ld % 4 too)
         ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15) == 0;
}
inline void launch_nt(hipStream_t st, const GemmArgs &a) {
  k_gemm_bf16<false><<<dim3(a.N / TILE, (a.M + TILE - 1) / TILE), 256, 0, st>>>(a);
}
inline void launch_tn(hipStream_t st, const GemmArgs &a, int nslab) {
  k_gemm_bf16<true><<<dim3(a.N / TILE, a.M / TILE, nslab), 256, 0, st>>>(a);
}

}  // namespace gemm
}  // namespace dfx
