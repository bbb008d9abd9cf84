// PointNet++ set-abstraction / feature-propagation layers in eval mode: the part of
// pointnet2_ops that the reference leaves to PyTorch — QueryAndGroup / GroupAll + shared MLP (Conv2d 1x1 +
// BatchNorm2d + ReLU stack) + max over the neighbourhood, and the FP module's interpolate + concat + MLP.
//
//   reference                                                                   here
//   pointnet2_utils.py:296-333 QueryAndGroup ([xyz - centre | features])        k_sa_fused / k_sa_rows (gather in-kernel)
//   pointnet2_utils.py:349-381 GroupAll                                         same kernels, M = 1, ns = N, no centring
//   pointnet2_modules.py:9-19  build_shared_mlp (conv + BN(eval) + ReLU)        folded at create time: y = relu(W' x + b')
//   pointnet2_modules.py:62-70 max_pool2d over nsample                          wave max (DPP) + atomicMax (values >= 0)
//   pointnet2_modules.py:170-209 PointnetFPModule                               k_fp_rows + linear layers
//
// Two native paths behind one entry point:
//  * fused (2-3 layers, hidden widths <= 128, output <= 256: SA1 / SA2 of PointNet2SSG): one wavefront = one tile of
//    32 neighbours of one centre; the gathered inputs feed the first layer's MFMAs directly, the layer outputs stay in
//    accumulator registers and chain into the next layer as the B operand (weights pre-permuted at create time, the
//    same trick as the denoiser), the last layer is reduced over the 32 lanes and merged across tiles with an integer
//    atomicMax (post-ReLU values are >= 0, so the IEEE bit patterns order like integers).  The (B, C+3, M, ns) grouped
//    tensor and the per-neighbour activations never exist in memory.
//  * general (any widths / depth, e.g. SA3 with 1024 outputs, and the FP module): rows materialised once, one
//    row-batched MFMA linear layer per MLP layer (mfma_linear.h), then a max over the neighbourhood.
// All fp32 on v_mfma_f32_32x32x2_f32.
#include <algorithm>
#include <vector>

#include "dfx_common.h"
#include <type_traits>
#include "mfma_linear.h"

using namespace dfx::lin;

namespace {

constexpr int MAX_LAYERS = 4;

__device__ __forceinline__ constexpr int rho(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// ---- create-time folding / packing --------------------------------------------------------------------------------
// W'(n, k) = W(n, k) * inv(n), b'(n) = (bconv(n) - mean(n)) * inv(n) + beta(n), inv = gamma / sqrt(var + eps)
// natural layout: (N, Kpad) row-major zero-padded (general path);
// packed layout (fused path): [out tile][K unit][lane][4]: lane (i, hf), element e of unit u holds W'(32 ot + i, k(u, e, hf))
//   first layer:  k = 8 u + 2 e + hf              (B operand built from gathers in channel order)
//   later layers: k = 32 (u >> 2) + rho(4 (u & 3) + e, hf)   (B operand = accumulator registers of the previous layer)
__global__ void k_fold(const float *__restrict__ w, const float *__restrict__ cb, const float *__restrict__ gamma,
                       const float *__restrict__ beta, const float *__restrict__ mean, const float *__restrict__ var,
                       float eps, float *__restrict__ wn, float *__restrict__ bn_, float *__restrict__ wp, int N, int K,
                       int Kpad, int first) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int Npad = (N + 31) & ~31;
  if (t >= Npad * Kpad) return;
  const int n = t / Kpad, k = t % Kpad;
  float inv = 1.f, v = 0.f;
  if (n < N) {
    if (gamma) inv = gamma[n] / sqrtf(var[n] + eps);
    if (k < K) v = w[(size_t)n * K + k] * inv;
    if (k == 0) bn_[n] = gamma ? ((cb ? cb[n] : 0.f) - mean[n]) * inv + beta[n] : (cb ? cb[n] : 0.f);
    wn[(size_t)n * Kpad + k] = v;
  } else if (k == 0) {
    bn_[n] = 0.f;
  }
  if (wp) {   // scatter into the packed position: invert k -> (u, e, hf)
    int u, e, hf;
    if (first) {
      u = k >> 3, e = (k & 7) >> 1, hf = k & 1;
    } else {
      const int kk = k & 31, r_hf = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);   // kk = (r&3) + 8 (r>>2) + 4 hf
      u = 4 * (k >> 5) + (r >> 2), e = r & 3, hf = r_hf;
    }
    const int ot = n >> 5, i = n & 31, U = Kpad >> 3;
    wp[(((size_t)ot * U + u) * 64 + (i + 32 * hf)) * 4 + e] = v;
  }
}

// ---- general path ----------------------------------------------------------------------------------------------------
// X[(b M + m) ns + j][k] = grouped input channel k of neighbour j of centre m (zero for k >= C0)
__global__ void k_sa_rows(const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ feat,
                          const int32_t *__restrict__ idx, float *__restrict__ X, int N, int M, int ns, int C, int use_xyz,
                          int Kpad, long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int k = t % Kpad;
  const long long row = t / Kpad;
  const int j = row % ns, m = (row / ns) % M, b = row / ((long long)ns * M);
  const int pid = idx ? idx[((size_t)b * M + m) * ns + j] : j;
  const int nx = use_xyz ? 3 : 0;
  float v = 0.f;
  if (k < nx) {
    v = xyz[((size_t)b * N + pid) * 3 + k];
    if (new_xyz) v -= new_xyz[((size_t)b * M + m) * 3 + k];
  } else if (k < nx + C) {
    v = feat[((size_t)b * C + (k - nx)) * N + pid];
  }
  X[t] = v;
}

// out[b][c][m] = max_j Y[(b M + m) ns + j][c]
__global__ void k_rowmax(const float *__restrict__ Y, float *__restrict__ out, int M, int ns, int Cout, int ld,
                         long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = t % Cout;
  const long long bm = t / Cout;
  const float *y = Y + bm * ns * ld + c;
  float v = y[0];
  for (int j = 1; j < ns; ++j) v = fmaxf(v, y[(size_t)j * ld]);
  const int m = bm % M;
  const long long b = bm / M;
  out[((size_t)b * Cout + c) * M + m] = v;
}

// FP module rows: X[b n + i][k] = k < C2 ? sum_q w_q known_feats[b][k][idx_q] : unknow_feats[b][k - C2][i]
// with w_q = (1 / (sqrt(d2_q) + 1e-8)) / sum_q (pointnet2_modules.py:190-194; ThreeNN returns sqrt, pointnet2_utils.py:125)
__global__ void k_fp_rows(const float *__restrict__ d2, const int32_t *__restrict__ idx, const float *__restrict__ kf,
                          const float *__restrict__ uf, float *__restrict__ X, int n, int m, int C1, int C2, int Kpad,
                          long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int k = t % Kpad;
  const long long row = t / Kpad;
  const int i = row % n;
  const long long b = row / n;
  float v = 0.f;
  if (k < C2) {
    if (idx) {
      const float r0 = 1.0f / (sqrtf(d2[row * 3 + 0]) + 1e-8f), r1 = 1.0f / (sqrtf(d2[row * 3 + 1]) + 1e-8f),
                  r2 = 1.0f / (sqrtf(d2[row * 3 + 2]) + 1e-8f);
      const float norm = (r0 + r1) + r2;   // torch.sum over 3 elements: sequential
      const float *p = kf + ((size_t)b * C2 + k) * m;
      float acc = p[idx[row * 3 + 0]] * (r0 / norm);
      acc = fmaf(p[idx[row * 3 + 1]], r1 / norm, acc);   // interpolate_gpu.cu:72-101 (stated nvcc -fmad contraction)
      acc = fmaf(p[idx[row * 3 + 2]], r2 / norm, acc);
      v = acc;
    } else {
      v = kf[((size_t)b * C2 + k)];   // known is None: known_feats (B, C2, 1) expanded (pointnet2_modules.py:196-199)
    }
  } else if (k < C2 + C1) {
    v = uf[((size_t)b * C1 + (k - C2)) * n + i];
  }
  X[t] = v;
}

// (rows = B n, ld) -> (B, C, n)
__global__ void k_rows_to_channels(const float *__restrict__ Y, float *__restrict__ out, int n, int C, int ld, long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int i = t % n, c = (t / n) % C;
  const long long b = t / ((long long)n * C);
  out[t] = Y[(b * n + i) * ld + c];
}

// ---- fused path ----------------------------------------------------------------------------------------------------
struct FusedArgs {
  const float *xyz, *new_xyz, *feat;
  const int32_t *idx;
  float *out;                 // (B, Cout, M), zero-initialised
  const float *wp[3], *b[3];  // packed weights / folded bias per layer
  int L;                      // 2 or 3
  int U0;                     // K units (of 8) of the first layer
  int nt[3];                  // output tiles (of 32 channels) per layer
  int cout;                   // real output channels of the last layer
  int N, M, ns, C, use_xyz, tiles_per_centre;
  long long total_tiles;
};

__device__ __forceinline__ v16f bias_tile(const float *b, int t, int hf) {
  v16f a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = b[32 * t + rho(r, hf)];
  return a;
}

__device__ __forceinline__ v16f relu16(v16f a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
  return a;
}

// Weight fragments of the register-chained layers: group s = (output tile ot, input tile t) = 4 x 64 float4, consecutive in s
// = ot * nin + t.  They come from L2 (every wavefront of the grid reads the same few KiB), ~600 cycles away: the group of step
// s + 1 is requested before the 16 MFMAs (1024 matrix-pipe cycles) of step s start, across output tiles and across layers — the
// compiler on its own kept two fragments in flight, one group of 4 MFMAs ahead, and drained the queue at every input tile.
__device__ __forceinline__ void load_group(const v4f *w, v4f (&u)[4]) {
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) u[r4] = w[r4 * 64];
}

__device__ __forceinline__ void mfma_group(const v4f (&u)[4], const v16f &x, v16f &acc) {
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u[r4][e], x[4 * r4 + e], acc, 0, 0, 0);
}

// max over the 32 lanes of each half-wave for the 16 accumulator registers of a tile, left in lanes 16..31 / 48..63: DPP permutes inside
// the rows of 16 (no LDS-crossbar round trips: 80 ds_bpermute per output tile before), then row 0 -> 1 and row 2 -> 3 by row_bcast:15.
// Written as ONE asm block of v_max_f32 with the permute as operand modifier: from `fmaxf(x, update_dpp(x))` hipcc makes a
// v_mov_b32_dpp + v_max_f32 pair per step.  A DPP read needs two wait states behind the VALU write of its source: inside the block
// a register's steps are 16 instructions apart; the leading s_nop covers what the compiler put in front of it.
#define DFX_DPPMAX(n, ctrl) "v_max_f32_dpp %" #n ", %" #n ", %" #n " " ctrl "\n"
#define DFX_DPPMAX16(ctrl)                                                                                                          \
  DFX_DPPMAX(0, ctrl) DFX_DPPMAX(1, ctrl) DFX_DPPMAX(2, ctrl) DFX_DPPMAX(3, ctrl) DFX_DPPMAX(4, ctrl) DFX_DPPMAX(5, ctrl)          \
  DFX_DPPMAX(6, ctrl) DFX_DPPMAX(7, ctrl) DFX_DPPMAX(8, ctrl) DFX_DPPMAX(9, ctrl) DFX_DPPMAX(10, ctrl) DFX_DPPMAX(11, ctrl)        \
  DFX_DPPMAX(12, ctrl) DFX_DPPMAX(13, ctrl) DFX_DPPMAX(14, ctrl) DFX_DPPMAX(15, ctrl)
__device__ __forceinline__ void half_max16(float (&f)[16]) {
  asm volatile("s_nop 1\n"
               DFX_DPPMAX16("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
               DFX_DPPMAX16("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
               DFX_DPPMAX16("row_half_mirror row_mask:0xf bank_mask:0xf")
               DFX_DPPMAX16("row_mirror row_mask:0xf bank_mask:0xf")
               DFX_DPPMAX16("row_bcast:15 row_mask:0xa bank_mask:0xf")
               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]),
                 "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]));
}
#undef DFX_DPPMAX16
#undef DFX_DPPMAX

// One output tile of a register-chained layer: acc += sum over the nin input tiles, the fragments of the step after each one
// requested first.  Two fragment sets used alternately (a copy "cur = next" would make every step wait for the loads it has
// just issued); `wa` holds step s0's fragments on entry and the fragments of step s0 + nin on exit (for odd nin after one copy).
template <int NIN>   // the number of input tiles at compile time (0: run time — every `t < nin` is then a branch whose merge makes the
                     // compiler wait for ALL outstanding loads, the prefetched ones included)
__device__ __forceinline__ void chain_steps(const v16f (&in)[4], int nin_rt, const v4f *w, int s0, int S, const v4f *after, v16f &acc,
                                            v4f (&wa)[4], v4f (&wb)[4]) {
  const int nin = NIN ? NIN : nin_rt;
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < nin) {
      v4f(&use)[4] = (t & 1) ? wb : wa;
      v4f(&pre)[4] = (t & 1) ? wa : wb;
      const int s = s0 + t;
      // behind a layer's last step: the first group of the next layer (or, after the very last one, any valid group: an
      // unconditional request keeps the outstanding-load count the same on every path, so the waits can be counted ones)
      load_group(s + 1 < S ? w + (size_t)(s + 1) * 256 : (after ? after : w), pre);
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the requests behind 12 of the 16 MFMAs to save registers)
      mfma_group(use, in[t], acc);
    }
  if (nin & 1) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) wa[r4] = wb[r4];
  }
}

// last layer: one output tile at a time -> ReLU -> max over this tile's neighbours -> atomic max across tiles.
// `wa` holds the fragments of the layer's first group on entry.
// COMBINE (the tiles of a centre all sit in one workgroup: 1, 2 or 4 tiles per centre): the four wavefronts leave their 32 channel
// maxima of the output tile in LDS, the first wavefront of each centre merges them and stores — no atomics (they cost SA1 270 us of
// 1550: 16.8 M read-modify-writes in L2) and no zero-fill of `out`.  Otherwise: atomic max across tiles into the zero-filled output.
template <int NIN, bool COMBINE>
__device__ __forceinline__ void pooled_last_layer(const FusedArgs &a, const v16f (&hin)[4], int nin_rt, int li, int lane, bool valid,
                                                  bool store, int b, int m, v4f (&wa)[4], v4f (&wb)[4]) {
  const int nin = NIN ? NIN : nin_rt;
  const int j = lane & 31, hf = lane >> 5, wave = threadIdx.x >> 6;
  const v4f *w = reinterpret_cast<const v4f *>(a.wp[li]) + lane;
  const int S = a.nt[li] * nin;
  __shared__ unsigned red[2][4][32];
  for (int ot = 0; ot < a.nt[li]; ++ot) {
    v16f acc = bias_tile(a.b[li], ot, hf);
    chain_steps<NIN>(hin, nin, w, ot * nin, S, nullptr, acc, wa, wb);
    const v16f y = relu16(acc);
    unsigned mine = 0;   // COMBINE: lane 16 + r (48 + r) keeps the maximum of accumulator register r = channel rho(r, hf) of the tile
    float pooled[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pooled[r] = valid ? y[r] : 0.f;   // padding neighbours contribute 0 <= the true maximum
    half_max16(pooled);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = pooled[r];
      // post-ReLU values are >= 0: their bit patterns order like unsigned integers (sign bit masked: -0)
      const unsigned bits = __float_as_uint(v) & 0x7fffffffu;
      if (COMBINE) {
        mine = (lane & 15) == r ? bits : mine;
      } else {
        const int c = 32 * ot + rho(r, hf);
        if (j == 16 && c < a.cout && store) atomicMax(reinterpret_cast<unsigned *>(a.out) + ((size_t)b * a.cout + c) * a.M + m, bits);
      }
    }
    if (COMBINE) {
      if (j >= 16) red[ot & 1][wave][rho(lane & 15, hf)] = mine;
      __syncthreads();   // (one per output tile: the buffer of tile ot + 2 is written after the barrier of ot + 1, behind the reads of ot)
      if (wave % a.tiles_per_centre == 0 && lane < 32) {
        unsigned u = red[ot & 1][wave][lane];
        for (int q = 1; q < a.tiles_per_centre; ++q) u = max(u, red[ot & 1][wave + q][lane]);
        const int c = 32 * ot + lane;
        if (store && c < a.cout) reinterpret_cast<unsigned *>(a.out)[((size_t)b * a.cout + c) * a.M + m] = u;
      }
    }
  }
}

template <int N0, int N1, bool COMBINE>   // tiles of the first / middle layer's output at compile time (0: run time)
__global__ __launch_bounds__(256) void k_sa_fused(FusedArgs a) {
  const int lane = threadIdx.x & 63, j = lane & 31, hf = lane >> 5;
  long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool store = tile < a.total_tiles;
  if (!store) {
    if (!COMBINE) return;
    tile = a.total_tiles - 1;   // (the workgroup's barriers need every wavefront: the spare ones redo the last tile and store nothing)
  }
  const int tl = tile % a.tiles_per_centre;
  const long long bm = tile / a.tiles_per_centre;
  const int m = bm % a.M;
  const int b = bm / a.M;
  const int nb = tl * 32 + j;
  const bool valid = nb < a.ns;
  const int pid = valid ? (a.idx ? a.idx[((size_t)b * a.M + m) * a.ns + nb] : nb) : 0;
  const int nx = a.use_xyz ? 3 : 0, C0 = nx + a.C, nt0 = N0 ? N0 : a.nt[0];
  v4f wa[4], wb[4];
  load_group(reinterpret_cast<const v4f *>(a.wp[1]) + lane, wa);   // first group of the second layer: under the gathers
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) wb[r4] = wa[r4];

  // ---- layer 0: B operand straight from the gathers, channel k = 8 u + 2 e + hf ----
  v16f h0[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < nt0) h0[t] = bias_tile(a.b[0], t, hf);
  const v4f *w0 = reinterpret_cast<const v4f *>(a.wp[0]) + lane;
  // (the gathers are two dependent trips to memory — index, then coordinate / feature — and the compiler issues them where they are
  // used: the operands of K unit u + 1 are requested before the MFMAs of unit u, two register sets used alternately)
  auto operands = [&](int u, float (&x)[4], v4f (&wv)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 8 * u + 2 * e + hf;
      float v = 0.f;
      if (valid && k < C0) {
        if (k < nx) {
          v = a.xyz[((size_t)b * a.N + pid) * 3 + k];
          if (a.new_xyz) v -= a.new_xyz[((size_t)b * a.M + m) * 3 + k];
        } else {
          v = a.feat[((size_t)b * a.C + (k - nx)) * a.N + pid];
        }
      }
      x[e] = v;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < nt0) wv[t] = w0[((size_t)t * a.U0 + u) * 64];
  };
  auto unit = [&](const float (&x)[4], const v4f (&wv)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < nt0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) h0[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t][e], x[e], h0[t], 0, 0, 0);
      }
  };
  {
    float xa[4], xb[4];
    v4f va[4], vb[4];
    operands(0, xa, va);
    int u = 0;
    for (; u + 1 < a.U0; u += 2) {
      operands(u + 1, xb, vb);
      __builtin_amdgcn_sched_barrier(0);
      unit(xa, va);
      operands(min(u + 2, a.U0 - 1), xa, va);
      __builtin_amdgcn_sched_barrier(0);
      unit(xb, vb);
    }
    if (u < a.U0) unit(xa, va);   // odd number of units: the last one's operands were requested by the last pair (or above)
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < nt0) h0[t] = relu16(h0[t]);

  // ---- optional middle layer (register-chained), then the pooled last layer ----
  if (a.L == 3) {
    v16f h1[4];
    const v4f *w1 = reinterpret_cast<const v4f *>(a.wp[1]) + lane, *w2 = reinterpret_cast<const v4f *>(a.wp[2]) + lane;
    const int nin = N0 ? N0 : a.nt[0], nt1 = N1 ? N1 : a.nt[1], S = nt1 * nin;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
      if (ot < nt1) {
        v16f acc = bias_tile(a.b[1], ot, hf);
        chain_steps<N0>(h0, nin, w1, ot * nin, S, w2, acc, wa, wb);
        h1[ot] = relu16(acc);
      }
    pooled_last_layer<N1, COMBINE>(a, h1, nt1, 2, lane, valid, store, b, m, wa, wb);
  } else {
    pooled_last_layer<N0, COMBINE>(a, h0, a.nt[0], 1, lane, valid, store, b, m, wa, wb);
  }
}

// PointNetV2 pooling (pointnet.py:194-198): pooled[b][j][c] = max_n Y[b N + n][c] * attn[b][n][j] * scale
// (points outside part j contribute 0 * x = 0, exactly as in the reference)
// One workgroup = one cloud x 64 channels, its 16 wavefronts take the points n = p, p + 16, ... (a wavefront reads 256 contiguous
// bytes per point; one thread per (cloud, channel) walking all N points alone — 1 wavefront per SIMD, a dependent load per point —
// ran at 0.63 TB/s: 853 of the 2000 us of the encoder's forward); the 16 partial maxima meet in LDS.  max is exact: any order.
template <int A>
__global__ __launch_bounds__(1024) void k_masked_max(const float *__restrict__ Y, const float *__restrict__ attn, float *__restrict__ pooled,
                                                   int N, int C, int ld, float scale) {
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  const long long b = blockIdx.y;
  __shared__ float red[16][A][64];
  float mx[A];
#pragma unroll
  for (int j = 0; j < A; ++j) mx[j] = -3.402823466e38f;
  if (c < C) {
    const float *y = Y + b * N * ld + c;
    const float *w = attn + b * N * A;
#pragma unroll 4
    for (int n = ph; n < N; n += 16) {
      const float v = y[(size_t)n * ld];
#pragma unroll
      for (int j = 0; j < A; ++j) {
#pragma clang fp contract(off)
        mx[j] = fmaxf(mx[j], v * w[n * A + j] * scale);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < A; ++j) red[ph][j][lane] = mx[j];
  __syncthreads();
  if (ph < A && c < C) {
    float m = red[0][ph][lane];
#pragma unroll
    for (int p = 1; p < 16; ++p) m = fmaxf(m, red[p][ph][lane]);
    pooled[(b * A + ph) * C + c] = m;
  }
}

inline int nblk(long long n, int bs = 256) { return (int)((n + bs - 1) / bs); }

}  // namespace

struct dfx_shared_mlp {
  int L = 0;
  int ch[MAX_LAYERS + 1] = {0};
  int kpad[MAX_LAYERS] = {0};
  float *wbuf = nullptr;
  size_t wn[MAX_LAYERS] = {0}, bn[MAX_LAYERS] = {0}, wp[MAX_LAYERS] = {0};
  unsigned relu_mask = ~0u;   // bit l: ReLU after layer l
  bool fused_ok = false;
  float *ws = nullptr;
  size_t ws_floats = 0;
  int reserve(size_t floats) {
    if (floats <= ws_floats) return DFX_OK;
    if (ws) (void)hipFree(ws);
    ws = nullptr, ws_floats = 0;
    if (hipMalloc(&ws, floats * sizeof(float)) != hipSuccess)
      return dfx::set_error(DFX_ERR_ALLOC, "shared_mlp: %zu bytes of workspace", floats * sizeof(float));
    ws_floats = floats;
    return DFX_OK;
  }
};

namespace {

// X (rows, ld = kpad[0]) -> last layer output Y (rows, ld_out); returns pointer + ld through refs.  Uses ws beyond `used`.
int run_layers(dfx_shared_mlp *h, float *X, long long rows, float *bufA, float *bufB, float **Yout, int *ld_out,
               hipStream_t st) {
  float *in = X, *out = bufA;
  for (int l = 0; l < h->L; ++l) {
    const int N = h->ch[l + 1], ldy = (l + 1 < h->L) ? h->kpad[l + 1] : ((N + 7) & ~7);
    if (ldy != N) DFX_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)rows * ldy, st));   // zero K padding
    LinArgs a{};
    a.M = (int)rows, a.N = N, a.K = h->kpad[l];
    a.X = in, a.ldx = h->kpad[l];
    a.W = h->wbuf + h->wn[l], a.b = h->wbuf + h->bn[l];
    a.Y = out, a.ldy = ldy;
    if ((h->relu_mask >> l) & 1u) launch<EPI_RELU>(st, 1, a);
    else launch<EPI_NONE>(st, 1, a);
    in = out;
    out = (out == bufA) ? bufB : bufA;
    *ld_out = ldy;
  }
  *Yout = in;
  return dfx::check_launch("shared_mlp layers");
}

}  // namespace

struct dfx_pointnet_v2 {
  int A = 0, zdim = 0;
  float scale = 1.f;
  dfx_shared_mlp *trunk = nullptr;
  float *hbuf = nullptr;                // folded head weights
  size_t hw[2][3] = {{0}}, hb[2][3] = {{0}};
  float *ws = nullptr;
  size_t ws_floats = 0;
};

extern "C" {

int dfx_shared_mlp_create(dfx_shared_mlp **out, int n_layers, const int32_t *channels, const float *const *conv_w,
                          const float *const *conv_b, const float *const *bn_w, const float *const *bn_b,
                          const float *const *bn_mean, const float *const *bn_var, float eps, uint32_t relu_mask,
                          dfx_stream_t stream) {
  DFX_REQUIRE(out && channels && conv_w, "shared_mlp_create: null argument");
  *out = nullptr;
  DFX_REQUIRE(n_layers >= 1 && n_layers <= MAX_LAYERS, "shared_mlp_create: %d layers outside [1,%d]", n_layers, MAX_LAYERS);
  for (int l = 0; l <= n_layers; ++l) DFX_REQUIRE(channels[l] >= 1 && channels[l] <= 4096, "shared_mlp_create: bad width %d", channels[l]);
  hipStream_t st = dfx::as_stream(stream);
  dfx_shared_mlp *h = new dfx_shared_mlp();
  h->L = n_layers;
  h->relu_mask = relu_mask;
  size_t cur = 0;
  const unsigned all = (1u << n_layers) - 1u;
  bool fused = (n_layers == 2 || n_layers == 3) && (relu_mask & all) == all;   // the pooled atomicMax needs values >= 0
  for (int l = 0; l <= n_layers; ++l) h->ch[l] = channels[l];
  for (int l = 0; l < n_layers; ++l) {
    // general path: K padded to 8; fused path: later layers consume whole 32-channel tiles of the previous layer
    h->kpad[l] = (channels[l] + 7) & ~7;
    const int npad = (channels[l + 1] + 31) & ~31;
    h->wn[l] = cur, cur += (size_t)npad * h->kpad[l];
    h->bn[l] = cur, cur += npad;
    if (l + 1 < n_layers && channels[l + 1] > 128) fused = false;
    if (l + 1 == n_layers && channels[l + 1] > 256) fused = false;
  }
  if (channels[0] > 160) fused = false;
  h->fused_ok = fused;
  int kp_fused[MAX_LAYERS];
  if (fused)
    for (int l = 0; l < n_layers; ++l) {
      kp_fused[l] = l == 0 ? h->kpad[0] : ((channels[l] + 31) & ~31);
      h->wp[l] = cur, cur += (size_t)((channels[l + 1] + 31) & ~31) * kp_fused[l];
    }
  if (hipMalloc(&h->wbuf, cur * sizeof(float)) != hipSuccess) {
    delete h;
    return dfx::set_error(DFX_ERR_ALLOC, "shared_mlp_create: %zu bytes", cur * sizeof(float));
  }
  (void)hipMemsetAsync(h->wbuf, 0, cur * sizeof(float), st);
  for (int l = 0; l < n_layers; ++l) {
    const bool has_bn = bn_w && bn_w[l];
    if (!conv_w[l] || (has_bn && !(bn_b && bn_b[l] && bn_mean && bn_mean[l] && bn_var && bn_var[l]))) {
      (void)hipFree(h->wbuf);
      delete h;
      return dfx::set_error(DFX_ERR_INVALID_ARG, "shared_mlp_create: null parameter of layer %d", l);
    }
    const int N = channels[l + 1], K = channels[l], npad = (N + 31) & ~31;
    const float *cb = conv_b ? conv_b[l] : nullptr;
    // natural layout (Kpad = multiple of 8)
    k_fold<<<nblk((long long)npad * h->kpad[l]), 256, 0, st>>>(conv_w[l], cb, has_bn ? bn_w[l] : nullptr, has_bn ? bn_b[l] : nullptr,
                                                               has_bn ? bn_mean[l] : nullptr, has_bn ? bn_var[l] : nullptr, eps,
                                                               h->wbuf + h->wn[l], h->wbuf + h->bn[l],
                                                               (fused && kp_fused[l] == h->kpad[l]) ? h->wbuf + h->wp[l] : nullptr, N, K,
                                                               h->kpad[l], l == 0);
    if (fused && kp_fused[l] != h->kpad[l]) {   // packed layout needs the 32-padded K: second pass into a scratch natural copy
      float *scratch = nullptr;
      if (hipMalloc(&scratch, ((size_t)npad * kp_fused[l] + npad) * sizeof(float)) != hipSuccess) {
        (void)hipFree(h->wbuf);
        delete h;
        return dfx::set_error(DFX_ERR_ALLOC, "shared_mlp_create: scratch");
      }
      k_fold<<<nblk((long long)npad * kp_fused[l]), 256, 0, st>>>(conv_w[l], cb, has_bn ? bn_w[l] : nullptr, has_bn ? bn_b[l] : nullptr,
                                                                  has_bn ? bn_mean[l] : nullptr, has_bn ? bn_var[l] : nullptr, eps, scratch,
                                                                  scratch + (size_t)npad * kp_fused[l], h->wbuf + h->wp[l], N, K,
                                                                  kp_fused[l], l == 0);
      (void)hipStreamSynchronize(st);
      (void)hipFree(scratch);
    }
  }
  if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
    (void)hipFree(h->wbuf);
    delete h;
    return dfx::set_error(DFX_ERR_HIP, "shared_mlp_create: fold kernels failed");
  }
  *out = h;
  return DFX_OK;
}

void dfx_shared_mlp_destroy(dfx_shared_mlp *h) {
  if (!h) return;
  if (h->wbuf) (void)hipFree(h->wbuf);
  if (h->ws) (void)hipFree(h->ws);
  delete h;
}

int dfx_shared_mlp_is_fused(const dfx_shared_mlp *h) { return h && h->fused_ok ? 1 : 0; }

int dfx_sa_forward_f32(dfx_shared_mlp *h, const float *xyz, const float *new_xyz, const float *features,
                       const int32_t *idx, int use_xyz, float *out, int B, int N, int M, int ns, int C, int force_general,
                       dfx_stream_t stream) {
  DFX_REQUIRE(h && B >= 0 && N > 0 && M > 0 && ns > 0 && C >= 0, "sa_forward: bad sizes");
  if (B == 0) return DFX_OK;
  DFX_REQUIRE(xyz && out, "sa_forward: null pointer");
  DFX_REQUIRE(C == 0 || features, "sa_forward: features missing");
  DFX_REQUIRE(use_xyz || C > 0, "sa_forward: nothing to group (no features and use_xyz = 0)");
  DFX_REQUIRE((use_xyz ? 3 : 0) + C == h->ch[0], "sa_forward: %d input channels, the MLP expects %d", (use_xyz ? 3 : 0) + C, h->ch[0]);
  DFX_REQUIRE(idx || (M == 1 && ns == N), "sa_forward: without idx (GroupAll) M must be 1 and ns = N");
  hipStream_t st = dfx::as_stream(stream);
  const int Cout = h->ch[h->L];
  if (h->fused_ok && !force_general) {
    FusedArgs a{};
    a.xyz = xyz, a.new_xyz = new_xyz, a.feat = features, a.idx = idx, a.out = out;
    a.L = h->L, a.U0 = h->kpad[0] >> 3, a.cout = Cout;
    for (int l = 0; l < h->L; ++l) a.wp[l] = h->wbuf + h->wp[l], a.b[l] = h->wbuf + h->bn[l], a.nt[l] = (h->ch[l + 1] + 31) >> 5;
    a.N = N, a.M = M, a.ns = ns, a.C = C, a.use_xyz = use_xyz;
    a.tiles_per_centre = (ns + 31) >> 5;
    a.total_tiles = (long long)B * M * a.tiles_per_centre;
    DFX_REQUIRE(a.total_tiles / 4 + 1 < 0x7fffffffLL, "sa_forward: too many tiles");
    const int grid = (int)((a.total_tiles + 3) / 4), n0 = a.nt[0], n1 = a.L == 3 ? a.nt[1] : 0;
    const bool combine = a.tiles_per_centre == 1 || a.tiles_per_centre == 2 || a.tiles_per_centre == 4;   // a centre's tiles in one workgroup
    if (!combine) DFX_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * Cout * M, st));
    // the channel widths of the reference's encoders at compile time (64-64-128, 128-128-256, two-layer 64 / 128); others at run time
#define DFX_SA_CASE(A, B_)                                                       \
  do {                                                                           \
    if (combine) k_sa_fused<A, B_, true><<<grid, 256, 0, st>>>(a);              \
    else k_sa_fused<A, B_, false><<<grid, 256, 0, st>>>(a);                     \
  } while (0)
    if (a.L == 3 && n0 == 2 && n1 == 2) DFX_SA_CASE(2, 2);
    else if (a.L == 3 && n0 == 4 && n1 == 4) DFX_SA_CASE(4, 4);
    else if (a.L == 2 && n0 == 2) DFX_SA_CASE(2, 0);
    else if (a.L == 2 && n0 == 4) DFX_SA_CASE(4, 0);
    else DFX_SA_CASE(0, 0);
#undef DFX_SA_CASE
    return dfx::check_launch("sa_forward (fused)");
  }
  const long long rows = (long long)B * M * ns;
  size_t wmax = 0;
  for (int l = 0; l < h->L; ++l) wmax = std::max<size_t>(wmax, (size_t)((h->ch[l + 1] + 7) & ~7));
  const size_t nX = (size_t)rows * h->kpad[0], nY = (size_t)rows * wmax;
  if (int e = h->reserve(nX + 2 * nY)) return e;
  float *X = h->ws, *bufA = X + nX, *bufB = bufA + nY;
  k_sa_rows<<<nblk((long long)nX), 256, 0, st>>>(xyz, new_xyz, features, idx, X, N, M, ns, C, use_xyz, h->kpad[0], (long long)nX);
  float *Y;
  int ld;
  if (int e = run_layers(h, X, rows, bufA, bufB, &Y, &ld, st)) return e;
  const long long tot = (long long)B * M * Cout;
  k_rowmax<<<nblk(tot), 256, 0, st>>>(Y, out, M, ns, Cout, ld, tot);
  return dfx::check_launch("sa_forward (general)");
}

int dfx_fp_forward_f32(dfx_shared_mlp *h, const float *unknown, const float *known, const float *unknow_feats,
                       const float *known_feats, float *out, int B, int n, int m, int C1, int C2, dfx_stream_t stream) {
  DFX_REQUIRE(h && B >= 0 && n > 0 && C1 >= 0 && C2 > 0, "fp_forward: bad sizes");
  if (B == 0) return DFX_OK;
  DFX_REQUIRE(known_feats && out && (C1 == 0 || unknow_feats), "fp_forward: null pointer");
  DFX_REQUIRE(C1 + C2 == h->ch[0], "fp_forward: %d input channels, the MLP expects %d", C1 + C2, h->ch[0]);
  DFX_REQUIRE(!known || (unknown && m >= 1), "fp_forward: bad known/unknown");
  hipStream_t st = dfx::as_stream(stream);
  const long long rows = (long long)B * n;
  size_t wmax = 0;
  for (int l = 0; l < h->L; ++l) wmax = std::max<size_t>(wmax, (size_t)((h->ch[l + 1] + 7) & ~7));
  const size_t nX = (size_t)rows * h->kpad[0], nY = (size_t)rows * wmax, nD = (size_t)rows * 3;
  if (int e = h->reserve(nX + 2 * nY + 2 * nD)) return e;
  float *X = h->ws, *bufA = X + nX, *bufB = bufA + nY, *d2 = bufB + nY;
  int32_t *idx = reinterpret_cast<int32_t *>(d2 + nD);
  if (known) {
    if (int e = dfx_three_nn_f32(unknown, known, d2, idx, B, n, m, stream)) return e;
  }
  k_fp_rows<<<nblk((long long)nX), 256, 0, st>>>(d2, known ? idx : nullptr, known_feats, unknow_feats, X, n, m, C1, C2, h->kpad[0],
                                                 (long long)nX);
  float *Y;
  int ld;
  if (int e = run_layers(h, X, rows, bufA, bufB, &Y, &ld, st)) return e;
  const int Cout = h->ch[h->L];
  const long long tot = rows * Cout;
  k_rows_to_channels<<<nblk(tot), 256, 0, st>>>(Y, out, n, Cout, ld, tot);
  return dfx::check_launch("fp_forward");
}

int dfx_pointnet_v2_create(dfx_pointnet_v2 **out, const dfx_pointnet_v2_weights *w, dfx_stream_t stream) {
  DFX_REQUIRE(out && w, "pointnet_v2_create: null argument");
  *out = nullptr;
  DFX_REQUIRE(w->num_anchors >= 1 && w->num_anchors <= 8, "pointnet_v2_create: num_anchors %d outside [1,8]", w->num_anchors);
  DFX_REQUIRE(w->zdim >= 8 && w->zdim % 8 == 0, "pointnet_v2_create: zdim %d must be a multiple of 8", w->zdim);
  hipStream_t st = dfx::as_stream(stream);
  dfx_pointnet_v2 *h = new dfx_pointnet_v2();
  h->A = w->num_anchors, h->zdim = w->zdim, h->scale = w->reweight_by_anchor ? (float)w->num_anchors : 1.f;
  const int32_t ch[5] = {3, 128, 128, 256, 512};
  if (int e = dfx_shared_mlp_create(&h->trunk, 4, ch, w->conv_w, w->conv_b, w->bn_w, w->bn_b, w->bn_mean, w->bn_var, w->bn_eps,
                                    0x7u /* no ReLU after bn4 (pointnet.py:193) */, stream)) {
    delete h;
    return e;
  }
  const int A = h->A, cin[3] = {512, 256, 128}, cout[3] = {256, 128, w->zdim};
  size_t cur = 0;
  for (int k = 0; k < 2; ++k)
    for (int l = 0; l < 3; ++l) {   // rows padded to a multiple of 32: k_fold zero-fills the padding rows / biases
      const size_t npad = ((size_t)A * cout[l] + 31) & ~(size_t)31;
      h->hw[k][l] = cur, cur += npad * cin[l], h->hb[k][l] = cur, cur += npad;
    }
  if (hipMalloc(&h->hbuf, cur * sizeof(float)) != hipSuccess) {
    dfx_shared_mlp_destroy(h->trunk);
    delete h;
    return dfx::set_error(DFX_ERR_ALLOC, "pointnet_v2_create: %zu bytes", cur * sizeof(float));
  }
  bool ok = true;
  for (int k = 0; k < 2 && ok; ++k)
    for (int l = 0; l < 3 && ok; ++l) {
      const bool bn = l < 2;
      ok = w->head_w[k][l] && w->head_b[k][l] && (!bn || (w->head_bn_w[k][l] && w->head_bn_b[k][l] && w->head_bn_mean[k][l] && w->head_bn_var[k][l]));
      if (!ok) break;
      const int N = A * cout[l], K = cin[l];   // grouped conv weight (A*Cout, Cin): group g = rows g*Cout.. (contiguous)
      k_fold<<<nblk((long long)((N + 31) & ~31) * K), 256, 0, st>>>(w->head_w[k][l], w->head_b[k][l], bn ? w->head_bn_w[k][l] : nullptr,
                                                                  bn ? w->head_bn_b[k][l] : nullptr, bn ? w->head_bn_mean[k][l] : nullptr,
                                                                  bn ? w->head_bn_var[k][l] : nullptr, w->bn_eps, h->hbuf + h->hw[k][l],
                                                                  h->hbuf + h->hb[k][l], nullptr, N, K, K, 0);
    }
  if (!ok || hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
    (void)hipFree(h->hbuf);
    dfx_shared_mlp_destroy(h->trunk);
    delete h;
    return dfx::set_error(ok ? DFX_ERR_HIP : DFX_ERR_INVALID_ARG, "pointnet_v2_create: %s", ok ? "fold kernels failed" : "null head parameter");
  }
  *out = h;
  return DFX_OK;
}

void dfx_pointnet_v2_destroy(dfx_pointnet_v2 *h) {
  if (!h) return;
  dfx_shared_mlp_destroy(h->trunk);
  if (h->hbuf) (void)hipFree(h->hbuf);
  if (h->ws) (void)hipFree(h->ws);
  delete h;
}

int dfx_pointnet_v2_forward_f32(dfx_pointnet_v2 *h, const float *x, const float *attn, float *m, float *v, int B, int N,
                                dfx_stream_t stream) {
  DFX_REQUIRE(h && B >= 0 && N > 0, "pointnet_v2_forward: bad sizes");
  if (B == 0) return DFX_OK;
  DFX_REQUIRE(x && attn && m && v, "pointnet_v2_forward: null pointer");
  hipStream_t st = dfx::as_stream(stream);
  dfx_shared_mlp *t = h->trunk;
  const long long rows = (long long)B * N;
  const int A = h->A;
  // trunk: rows (B N, 8) -> 128 -> 128 -> 256 -> 512
  const size_t nX = (size_t)rows * t->kpad[0], nY = (size_t)rows * 512;
  if (int e = t->reserve(nX + 2 * nY)) return e;
  float *X = t->ws, *bufA = X + nX, *bufB = bufA + nY;
  k_sa_rows<<<nblk((long long)nX), 256, 0, st>>>(x, nullptr, nullptr, nullptr, X, N, 1, N, 0, 1, t->kpad[0], (long long)nX);
  float *Y;
  int ld;
  if (int e = run_layers(t, X, rows, bufA, bufB, &Y, &ld, st)) return e;
  // pooled (B, A, 512) + head activations
  const size_t nP = (size_t)B * A * 512, nH1 = (size_t)B * A * 256, nH2 = (size_t)B * A * 128;
  if (nP + nH1 + nH2 > h->ws_floats) {
    if (h->ws) (void)hipFree(h->ws);
    h->ws = nullptr, h->ws_floats = 0;
    if (hipMalloc(&h->ws, (nP + nH1 + nH2) * sizeof(float)) != hipSuccess) return dfx::set_error(DFX_ERR_ALLOC, "pointnet_v2_forward: workspace");
    h->ws_floats = nP + nH1 + nH2;
  }
  float *P = h->ws, *H1 = P + nP, *H2 = H1 + nH1;
  switch (A) {
#define DFX_CASE(a) case a: k_masked_max<a><<<dim3(512 / 64, B), 1024, 0, st>>>(Y, attn, P, N, 512, ld, h->scale); break;
    DFX_CASE(1) DFX_CASE(2) DFX_CASE(3) DFX_CASE(4) DFX_CASE(5) DFX_CASE(6) DFX_CASE(7) DFX_CASE(8)
#undef DFX_CASE
  }
  // per-part heads: grouped 1x1 convolutions = one linear layer per part (blockIdx.z = part)
  const int cin[3] = {512, 256, 128}, cout[3] = {256, 128, h->zdim};
  for (int k = 0; k < 2; ++k) {
    const float *in = P;
    float *outs[3] = {H1, H2, k == 0 ? m : v};
    for (int l = 0; l < 3; ++l) {
      LinArgs a{};
      a.M = B, a.N = cout[l], a.K = cin[l];
      a.X = in, a.ldx = A * cin[l], a.x_gs = cin[l];
      a.W = h->hbuf + h->hw[k][l], a.w_gs = (long long)cout[l] * cin[l];
      a.b = h->hbuf + h->hb[k][l], a.b_gs = cout[l];
      a.Y = outs[l], a.ldy = A * cout[l], a.y_gs = cout[l];
      if (l < 2) launch<EPI_RELU>(st, A, a);
      else launch<EPI_NONE>(st, A, a);
      in = outs[l];
    }
  }
  return dfx::check_launch("pointnet_v2_forward");
}

}  // extern "C"
