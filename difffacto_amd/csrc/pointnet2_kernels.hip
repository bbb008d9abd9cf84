// pointnet2_ops primitives for gfx950 (MI355X): gather, FPS, ball query, grouping, 3-NN, 3-interpolate.
//
// Semantics follow the reference CUDA kernels
//   pointnet2_ops_lib/pointnet2_ops/_ext-src/src/{sampling_gpu,ball_query_gpu,group_points_gpu,interpolate_gpu}.cu
// (index order, strict comparisons, tie rules, fill conventions) but none of their structure:
//   * these are HBM/latency-bound integer+fp32 gather/scan kernels -> coalesced dword streams,
//     64-wide wavefront ballots/shuffles, LDS-resident clouds; no MFMA.
//   * FPS: one workgroup per cloud, point coordinates + running min-distance in REGISTERS
//     (the reference keeps `temp` in global memory), (dist,key) packed in one u64 so the
//     block-wide arg-max is a plain max, 1 barrier per selected point.
//   * ball query: one wavefront per centre, 64 candidates per step, ordered compaction by
//     ballot + prefix popcount (the reference scans N serially in one thread per centre).
// Distances use the explicit mul,fma,fma evaluation order stated in oracle/pointnet2.c.
#include "dfx_common.h"

namespace {

constexpr int WAVE = 64;

__device__ __forceinline__ float sq3(float a, float b, float c) {
  float t = __fmul_rn(a, a);
  t = __fmaf_rn(b, b, t);
  t = __fmaf_rn(c, c, t);
  return t;
}

// ------------------------------------------------------------------------------------------
// gather_points: out[b,c,j] = points[b,c,idx[b,j]]          (sampling_gpu.cu:8-20)
// One thread per output element, coalesced along j; idx re-read per channel hits L1/L2.
__global__ void __launch_bounds__(256) gather_points_kernel(const float *__restrict__ points,
                                                            const int32_t *__restrict__ idx,
                                                            float *__restrict__ out, int C, int N, int M,
                                                            long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % M);
    const long long bc = i / M;
    const long long b = bc / C;
    const int a = idx[b * M + j];
    out[i] = points[bc * N + a];
  }
}

// gather_points_grad: grad_points[b,c,idx[b,j]] += grad_out[b,c,j]   (sampling_gpu.cu:34-47)
__global__ void __launch_bounds__(256) gather_points_grad_kernel(const float *__restrict__ grad_out,
                                                                 const int32_t *__restrict__ idx,
                                                                 float *__restrict__ grad_points, int C, int N,
                                                                 int M, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % M);
    const long long bc = i / M;
    const long long b = bc / C;
    const int a = idx[b * M + j];
    atomicAdd(grad_points + bc * N + a, grad_out[i]);
  }
}

// ------------------------------------------------------------------------------------------
// Furthest point sampling                                    (sampling_gpu.cu:59-173)
//
// Reference tie rule, restated: with bs = opt_n_threads(N) threads, thread tid scans k = tid,
// tid+bs, ... keeping the FIRST strictly greater candidate, and the shared-memory tree keeps the
// lower slot on ties at every level (stride bs/2 ... 1).  Hence the winner among equal distances
// is the candidate with the smallest (bitrev(k mod bs), k div bs).  We pack
//   val = (bits(d2) << 32) | ~key32,   key32 = bitrev(k mod bs) << 22 | (k div bs)
// so that the arg-max with that tie rule is a plain u64 max (d2 >= 0, so its bit pattern orders
// like the float).  val == 0 means "no candidate" -> index 0, as in the reference (besti = 0).
constexpr unsigned FPS_SKIP_THRESH_BITS = 0x3a83126eu;  // largest float <= 1e-3 (double): mag <= 1e-3 skip (:100-101)

__device__ __forceinline__ unsigned long long fps_pack(float d2, int k, int log2bs) {
  const unsigned tidr = (unsigned)k & ((1u << log2bs) - 1u);
  const unsigned rev = log2bs ? (__brev(tidr) >> (32 - log2bs)) : 0u;
  const unsigned key = (rev << 22) | ((unsigned)k >> log2bs);
  return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(~key);
}

__device__ __forceinline__ int fps_unpack(unsigned long long v, int log2bs) {
  if (v == 0ull) return 0;
  const unsigned key = ~(unsigned)(v & 0xffffffffull);
  const unsigned rev = key >> 22;
  const unsigned tidr = log2bs ? (__brev(rev) >> (32 - log2bs)) : 0u;
  return (int)(((key & 0x3fffffu) << log2bs) | tidr);
}

// 64-bit max across the wavefront with DPP lane permutes (VALU, ~6 x 5 instructions) instead of ds_bpermute
// shuffles (~100 cycles of LDS-crossbar latency each): the selection loop is a chain of M dependent arg-max
// reductions, so this latency IS the kernel's run time.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v) {
  const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  const unsigned lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  const unsigned long long o = ((unsigned long long)hi2 << 32) | lo2;
  return o > v ? o : v;
}

// after this every lane of each 16-lane row holds the row maximum
__device__ __forceinline__ unsigned long long row_max_u64(unsigned long long v) {
  v = dpp_max_u64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_max_u64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_max_u64<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_max_u64<0x140, 0xf>(v);  // row_mirror
  return v;
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), lane) << 32) |
         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
}

// wave-uniform maximum of the 64 lanes
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = row_max_u64(v);
  v = dpp_max_u64<0x142, 0xA>(v);  // row_bcast:15 -> rows 1 and 3
  v = dpp_max_u64<0x143, 0xC>(v);  // row_bcast:31 -> rows 2 and 3
  return readlane_u64(v, 63);
}

// The arg-max in two single-word reductions instead of one over 64-bit (distance, key) pairs: first the largest distance (floats
// >= 0), then the largest key among the candidates that have it — the same winner as the u64 maximum, but every step is ONE
// v_max with the lane permute as operand modifier (a 64-bit step is two DPP moves, a 64-bit compare and two selects, and hipcc makes a
// mov + max pair even of a 32-bit one); the selection loop is a chain of M such reductions.  Inline asm: a DPP read needs two wait
// states behind the VALU write of its source, which the assembler does not insert — hence the s_nop in front of every step.
#define DFX_DPP1(op, ctrl) "s_nop 1\n " op " %0, %0, %0 " ctrl "\n"
#define DFX_ROW_STEPS(op)                                                                                                  \
  DFX_DPP1(op, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") DFX_DPP1(op, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") \
  DFX_DPP1(op, "row_half_mirror row_mask:0xf bank_mask:0xf") DFX_DPP1(op, "row_mirror row_mask:0xf bank_mask:0xf")
#define DFX_WAVE_STEPS(op) DFX_ROW_STEPS(op) DFX_DPP1(op, "row_bcast:15 row_mask:0xa bank_mask:0xf") DFX_DPP1(op, "row_bcast:31 row_mask:0xc bank_mask:0xf")
__device__ __forceinline__ float wave_max_f32(float v) {   // wave-uniform
  asm volatile(DFX_WAVE_STEPS("v_max_f32_dpp") "s_nop 1\n" : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  asm volatile(DFX_WAVE_STEPS("v_max_u32_dpp") "s_nop 1\n" : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float row0_max_f32(float v) {   // maximum of lanes 0..15, wave-uniform
  asm volatile(DFX_ROW_STEPS("v_max_f32_dpp") "s_nop 1\n" : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
}
__device__ __forceinline__ unsigned row0_max_u32(unsigned v) {
  asm volatile(DFX_ROW_STEPS("v_max_u32_dpp") "s_nop 1\n" : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 0);
}
#undef DFX_WAVE_STEPS
#undef DFX_ROW_STEPS
#undef DFX_DPP1

// Register-resident variant: N <= NT*PPT.  LDS holds the cloud as SoA for the winner lookup when
// COORDS_LDS (N*12 bytes), else the 3 winner coordinates come from global (L2-resident).
template <int NT, int PPT, bool COORDS_LDS>
__global__ void __launch_bounds__(NT) fps_resident_kernel(const float *__restrict__ dataset,
                                                          int32_t *__restrict__ idxs, int N, int M, int log2bs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = NT / WAVE;
  static_assert(NW <= 16, "second-level reduction uses one 16-lane row");
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem);  // [2][16]
  float *sx = reinterpret_cast<float *>(smem + 2 * 16 * sizeof(unsigned long long));
  float *sy = sx + N;
  float *sz = sy + N;

  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = tid / WAVE;
  const float *ds = dataset + (size_t)blockIdx.x * N * 3;
  int32_t *out = idxs + (size_t)blockIdx.x * M;

  // per-point state in registers: coordinates (as PAIRS: the distance update runs on packed fp32 — v_pk_add / v_pk_mul / v_pk_fma do
  // two points per instruction with the scalar instructions' IEEE results), running min distance, and the (constant) tie-break key
  typedef float v2f __attribute__((ext_vector_type(2)));
  static_assert(PPT % 2 == 0, "points per thread come in pairs");
  v2f px[PPT / 2], py[PPT / 2], pz[PPT / 2];
  float tmp[PPT];
  unsigned nkey[PPT];  // low half of fps_pack(): ~key32 ; 0 marks "skipped / absent" (never wins)
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * NT;
    float x = 0.f, y = 0.f, z = 0.f;
    // skipped / absent points: running distance -1 and key 0 — min(d, -1) = -1 keeps it, the maximum (which starts at 0) never sees it,
    // and it never EQUALS the maximum either, so the key selection below needs no validity test (the reference never reads a skipped
    // point's temp entry)
    tmp[i] = -1.f;
    nkey[i] = 0u;
    if (k < N) {
      x = ds[k * 3 + 0];
      y = ds[k * 3 + 1];
      z = ds[k * 3 + 2];
      const float mag = sq3(x, y, z);
      if (!(mag <= __uint_as_float(FPS_SKIP_THRESH_BITS))) {
        nkey[i] = (unsigned)(fps_pack(0.f, k, log2bs) & 0xffffffffull);
        tmp[i] = 1e10f;  // sampling.cpp:74-76
      }
      if (COORDS_LDS) {
        sx[k] = x;
        sy[k] = y;
        sz[k] = z;
      }
    }
    px[i >> 1][i & 1] = x;
    py[i >> 1][i & 1] = y;
    pz[i >> 1][i & 1] = z;
  }
  if (tid < 32) slots[tid] = 0ull;
  int old = 0;
  if (tid == 0) out[0] = 0;
  __syncthreads();
  // A thread's keys fall with i when the reference's block size divides NT (k mod bs is then the same for all of a thread's points and
  // k div bs grows with i): the largest key among the points at the maximum is the one of the SMALLEST such i — a compare + select
  // per point, walked downwards, instead of compare + select + max.
  const bool mono = (NT & ((1 << log2bs) - 1)) == 0;

  for (int j = 1; j < M; ++j) {
    float x1, y1, z1;
    if (COORDS_LDS) {
      x1 = sx[old];
      y1 = sy[old];
      z1 = sz[old];
    } else {
      x1 = ds[old * 3 + 0];
      y1 = ds[old * 3 + 1];
      z1 = ds[old * 3 + 2];
    }
    const v2f nx = {-x1, -x1}, ny = {-y1, -y1}, nz = {-z1, -z1};
    float m = 0.f;   // distances are >= 0; absent / skipped points carry -1
#pragma unroll
    for (int i = 0; i < PPT / 2; ++i) {
      // sq3(px - x1, py - y1, pz - z1) for two points: the same roundings as the scalar chain (sub = add of the negated value; mul;
      // fma; fma), in three v_pk_add_f32, one v_pk_mul_f32 and two v_pk_fma_f32
      const v2f dx = px[i] + nx, dy = py[i] + ny, dz = pz[i] + nz;
      v2f t = dx * dx;
      t = __builtin_elementwise_fma(dy, dy, t);
      t = __builtin_elementwise_fma(dz, dz, t);
      // (plain v_min_f32: through fminf hipcc first re-quiets the extracted halves with a v_max x, x each — 8 more instructions per step)
      float da, db;
      asm("v_min_f32 %0, %1, %2" : "=v"(da) : "v"(t[0]), "v"(tmp[2 * i]));
      asm("v_min_f32 %0, %1, %2" : "=v"(db) : "v"(t[1]), "v"(tmp[2 * i + 1]));
      tmp[2 * i] = da;
      tmp[2 * i + 1] = db;
      m = fmaxf(fmaxf(m, da), db);   // v_max3_f32
    }
    const float M = wave_max_f32(m);
    unsigned k = 0u;   // the largest key among this lane's points at the wavefront's largest distance (key 0: none)
    if (mono) {
#pragma unroll
      for (int i = PPT - 1; i >= 0; --i) k = tmp[i] == M ? nkey[i] : k;
    } else {
#pragma unroll
      for (int i = 0; i < PPT; ++i) k = max(k, tmp[i] == M ? nkey[i] : 0u);
    }
    const unsigned K = wave_max_u32(k);
    unsigned long long *slot = slots + (j & 1) * 16;
    if (lane == 0) slot[wave] = ((unsigned long long)__float_as_uint(M) << 32) | K;
    __syncthreads();
    const unsigned long long sv = lane < NW ? slot[lane] : 0ull;
    const float sd = __uint_as_float((unsigned)(sv >> 32));
    const float M2 = row0_max_f32(sd);
    const unsigned K2 = row0_max_u32(sd == M2 ? (unsigned)sv : 0u);
    const unsigned long long v = ((unsigned long long)__float_as_uint(M2) << 32) | K2;
    old = fps_unpack(v, log2bs);
    if (tid == 0) out[j] = old;
  }
}

// Streaming variant for clouds that do not fit the register-resident kernels: `temp` lives in
// global memory exactly like the reference's scratch tensor.
template <int NT>
__global__ void __launch_bounds__(NT) fps_streaming_kernel(const float *__restrict__ dataset,
                                                           float *__restrict__ temp, int32_t *__restrict__ idxs,
                                                           int N, int M, int log2bs) {
  constexpr int NW = NT / WAVE;
  __shared__ unsigned long long slots[2 * NW];
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = tid / WAVE;
  const float *ds = dataset + (size_t)blockIdx.x * N * 3;
  float *tp = temp + (size_t)blockIdx.x * N;
  int32_t *out = idxs + (size_t)blockIdx.x * M;
  for (int k = tid; k < N; k += NT) tp[k] = 1e10f;
  int old = 0;
  if (tid == 0) out[0] = 0;
  __syncthreads();
  for (int j = 1; j < M; ++j) {
    const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
    unsigned long long best = 0ull;
    for (int k = tid; k < N; k += NT) {
      const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
      const float mag = sq3(x2, y2, z2);
      if (mag <= __uint_as_float(FPS_SKIP_THRESH_BITS)) continue;
      const float d = sq3(x2 - x1, y2 - y1, z2 - z1);
      const float d2 = fminf(d, tp[k]);
      tp[k] = d2;
      const unsigned long long v = fps_pack(d2, k, log2bs);
      best = v > best ? v : best;
    }
    best = wave_max_u64(best);
    unsigned long long *slot = slots + (j & 1) * NW;
    if (lane == 0) slot[wave] = best;
    __syncthreads();
    unsigned long long v = lane < NW ? slot[lane] : 0ull;
    v = readlane_u64(row_max_u64(v), 0);
    old = fps_unpack(v, log2bs);
    if (tid == 0) out[j] = old;
  }
}

// ------------------------------------------------------------------------------------------
// Ball query                                                 (ball_query_gpu.cu:9-44)
// Workgroup = 4 wavefronts; the cloud is staged once per workgroup in LDS (SoA, conflict-free
// stride-1 reads) when it fits; each wavefront owns centres j = first + wave, +4, ...
constexpr int BQ_WAVES = 4;
constexpr int BQ_CENTRES_PER_WG = 32;

template <bool CLOUD_LDS>
__global__ void __launch_bounds__(BQ_WAVES *WAVE) ball_query_kernel(const float *__restrict__ new_xyz,
                                                                     const float *__restrict__ xyz,
                                                                     int32_t *__restrict__ idx, int N, int M,
                                                                     float radius2, int nsample, int wg_per_cloud) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *sx = reinterpret_cast<float *>(smem);
  float *sy = sx + N;
  float *sz = sy + N;
  const int b = blockIdx.x / wg_per_cloud;
  const int first = (blockIdx.x % wg_per_cloud) * BQ_CENTRES_PER_WG;
  const float *X = xyz + (size_t)b * N * 3;
  const float *Q = new_xyz + (size_t)b * M * 3;
  int32_t *I = idx + (size_t)b * M * nsample;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  if (CLOUD_LDS) {
    // coalesced AoS read -> SoA LDS, eight loads in flight per thread (one load, one wait, one LDS write per trip made the fill a chain
    // of 24 memory latencies in front of every workgroup's work)
    constexpr int NT = BQ_WAVES * WAVE;
    for (int i0 = tid; i0 < N * 3; i0 += 8 * NT) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = X[min(i0 + u * NT, N * 3 - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * NT;
        if (i < N * 3) {
          const int k = i / 3, c = i - 3 * k;
          (c == 0 ? sx : (c == 1 ? sy : sz))[k] = v[u];
        }
      }
    }
    __syncthreads();
  }
  const int last = min(first + BQ_CENTRES_PER_WG, M);
  for (int j = first + wave; j < last; j += BQ_WAVES) {
    const float nx = Q[j * 3 + 0], ny = Q[j * 3 + 1], nz = Q[j * 3 + 2];
    int32_t *row = I + (size_t)j * nsample;
    int cnt = 0, firstk = 0;
    // four chunks of 64 points per step: their 12 LDS reads and distance tests are independent, and a step without any hit (the
    // common one: ~9 of 2048 points fall inside r = 0.2) costs one test of the OR of the four ballots; hits are recorded chunk by
    // chunk in index order exactly as before (slots past nsample are not written, the first hit is the first in index order)
    auto test = [&](int k) {   // (branch-free: past the end the last point is read and the result masked — a predicate per chunk
                               // put every chunk's reads behind their own exec-mask branch and wait)
      const int kc = min(k, N - 1);
      float x, y, z;
      if (CLOUD_LDS) {
        x = sx[kc]; y = sy[kc]; z = sz[kc];
      } else {
        x = X[kc * 3 + 0]; y = X[kc * 3 + 1]; z = X[kc * 3 + 2];
      }
      return (sq3(nx - x, ny - y, nz - z) < radius2) & (k < N);
    };
    for (int base = 0; base < N && cnt < nsample; base += 4 * WAVE) {
      bool hit[4];
      unsigned long long mask[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) hit[c] = test(base + c * WAVE + lane);
#pragma unroll
      for (int c = 0; c < 4; ++c) mask[c] = __ballot(hit[c]);
      if ((mask[0] | mask[1] | mask[2] | mask[3]) == 0ull) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (mask[c] && cnt < nsample) {
          if (cnt == 0) firstk = base + c * WAVE + __builtin_ctzll(mask[c]);
          const int slot = cnt + __builtin_popcountll(mask[c] & ((1ull << lane) - 1ull));
          if (hit[c] && slot < nsample) row[slot] = base + c * WAVE + lane;
          cnt += __builtin_popcountll(mask[c]);
        }
      }
    }
    // pad: first hit fills the unused slots; no hit at all -> zeros (ball_query.cpp:19-21)
    const int filled = min(cnt, nsample);
    for (int l = filled + lane; l < nsample; l += WAVE) row[l] = cnt ? firstk : 0;
  }
}

// ------------------------------------------------------------------------------------------
// group_points: out[b,c,j,k] = points[b,c,idx[b,j,k]]         (group_points_gpu.cu:8-28)
// One thread per (b, j*nsample+k) walks the channels: idx read once (coalesced), every store
// coalesced along the innermost (j,k) dimension.
__global__ void __launch_bounds__(256) group_points_kernel(const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx,
                                                           float *__restrict__ out, int C, int N, int S,
                                                           long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / S;
    const int s = (int)(i - b * S);
    const int a = idx[i];
    const float *P = points + b * C * N + a;
    float *O = out + b * C * S + s;
    for (int l = 0; l < C; ++l) O[(long long)l * S] = P[(long long)l * N];
  }
}

__global__ void __launch_bounds__(256) group_points_grad_kernel(const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx,
                                                                float *__restrict__ grad_points, int C, int N,
                                                                int S, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / S;
    const int s = (int)(i - b * S);
    const int a = idx[i];
    float *P = grad_points + b * C * N + a;
    const float *G = grad_out + b * C * S + s;
    for (int l = 0; l < C; ++l) atomicAdd(P + (long long)l * N, G[(long long)l * S]);
  }
}

// ------------------------------------------------------------------------------------------
// three_nn                                                     (interpolate_gpu.cu:9-59)
// One thread per unknown point; the known cloud streams through LDS in SoA tiles.
constexpr int NN_TILE = 2048;
__global__ void __launch_bounds__(256) three_nn_kernel(const float *__restrict__ unknown,
                                                       const float *__restrict__ known, float *__restrict__ dist2,
                                                       int32_t *__restrict__ idx, int n, int m, int wg_per_cloud) {
  __shared__ float sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
  const int b = blockIdx.x / wg_per_cloud;
  const int j = (blockIdx.x % wg_per_cloud) * 256 + threadIdx.x;
  const float *U = unknown + (size_t)b * n * 3;
  const float *K = known + (size_t)b * m * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (j < n) {
    ux = U[j * 3 + 0]; uy = U[j * 3 + 1]; uz = U[j * 3 + 2];
  }
  // best* are doubles in the reference (init 1e40, compared with the float d): keep that.
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int base = 0; base < m; base += NN_TILE) {
    const int cnt = min(NN_TILE, m - base);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * 3; i += 256) {
      const float v = K[(size_t)base * 3 + i];
      const int k = i / 3, c = i - 3 * k;
      (c == 0 ? sx : (c == 1 ? sy : sz))[k] = v;
    }
    __syncthreads();
    if (j < n) {
      for (int k = 0; k < cnt; ++k) {
        const float d = sq3(ux - sx[k], uy - sy[k], uz - sz[k]);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = base + k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = base + k;
        } else if (d < best3) {
          best3 = d; besti3 = base + k;
        }
      }
    }
  }
  if (j < n) {
    float *D = dist2 + ((size_t)b * n + j) * 3;
    int32_t *I = idx + ((size_t)b * n + j) * 3;
    D[0] = (float)best1; D[1] = (float)best2; D[2] = (float)best3;
    I[0] = besti1; I[1] = besti2; I[2] = besti3;
  }
}

// three_interpolate                                            (interpolate_gpu.cu:72-101)
__global__ void __launch_bounds__(256) three_interpolate_kernel(const float *__restrict__ points,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out, int c, int m, int n,
                                                                long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % n);
    const long long bl = i / n;  // b*c + l
    const long long b = bl / c;
    const float *W = weight + (b * n + j) * 3;
    const int32_t *I = idx + (b * n + j) * 3;
    const float *P = points + bl * m;
    float t = __fmul_rn(P[I[0]], W[0]);
    t = __fmaf_rn(P[I[1]], W[1], t);
    t = __fmaf_rn(P[I[2]], W[2], t);
    out[i] = t;
  }
}

// three_interpolate_grad                                       (interpolate_gpu.cu:116-143)
__global__ void __launch_bounds__(256) three_interpolate_grad_kernel(const float *__restrict__ grad_out,
                                                                     const int32_t *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     float *__restrict__ grad_points, int c, int n,
                                                                     int m, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % n);
    const long long bl = i / n;
    const long long b = bl / c;
    const float *W = weight + (b * n + j) * 3;
    const int32_t *I = idx + (b * n + j) * 3;
    float *P = grad_points + bl * m;
    const float g = grad_out[i];
    atomicAdd(P + I[0], g * W[0]);
    atomicAdd(P + I[1], g * W[1]);
    atomicAdd(P + I[2], g * W[2]);
  }
}

inline int grid_for(long long total, int per_block = 256) {
  long long g = (total + per_block - 1) / per_block;
  const long long cap = 256LL * 8 * 4;  // 256 CUs x 8 blocks, grid-stride the rest
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

inline int ilog2_floor(int v) {
  int l = 0;
  while ((1 << (l + 1)) <= v) ++l;
  return l;
}

}  // namespace

extern "C" {

int dfx_gather_points_f32(const float *points, const int32_t *idx, float *out, int B, int C, int N, int M,
                          dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0, "gather_points: negative size");
  const long long total = (long long)B * C * M;
  if (total == 0) return DFX_OK;
  DFX_REQUIRE(points && idx && out, "gather_points: null pointer");
  gather_points_kernel<<<grid_for(total), 256, 0, dfx::as_stream(stream)>>>(points, idx, out, C, N, M, total);
  return dfx::check_launch("gather_points");
}

int dfx_gather_points_grad_f32(const float *grad_out, const int32_t *idx, float *grad_points, int B, int C, int N,
                               int M, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0, "gather_points_grad: negative size");
  const long long total = (long long)B * C * M;
  if ((long long)B * C * N == 0) return DFX_OK;
  DFX_REQUIRE(grad_points, "gather_points_grad: null pointer");
  DFX_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)B * C * N, dfx::as_stream(stream)));
  if (total == 0) return DFX_OK;
  DFX_REQUIRE(grad_out && idx, "gather_points_grad: null pointer");
  gather_points_grad_kernel<<<grid_for(total), 256, 0, dfx::as_stream(stream)>>>(grad_out, idx, grad_points, C, N, M,
                                                                                total);
  return dfx::check_launch("gather_points_grad");
}

int dfx_fps_max_resident(void) { return 1024 * 16; }

namespace { int g_fps_nt = 0, g_fps_ppt = 0; }
void dfx_debug_fps_shape(int threads, int points_per_thread) { g_fps_nt = threads, g_fps_ppt = points_per_thread; }

int dfx_furthest_point_sampling_f32(const float *xyz, float *tmp, int32_t *idx, int B, int N, int M,
                                    dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && N >= 0 && M >= 0, "fps: negative size");
  if (B == 0 || M == 0) return DFX_OK;
  DFX_REQUIRE(N >= 1, "fps: empty cloud with npoint > 0");
  DFX_REQUIRE(xyz && idx, "fps: null pointer");
  hipStream_t st = dfx::as_stream(stream);
  // opt_n_threads(N) of the reference (cuda_utils.h:15-19) decides the tie rule: largest power of two <= N, <= 512
  int log2bs = ilog2_floor(N);
  if (log2bs > 9) log2bs = 9;
#define DFX_FPS_LAUNCH(NT, PPT, LDSC)                                                                          \
  do {                                                                                                         \
    const size_t sh = 2 * 16 * sizeof(unsigned long long) + ((LDSC) ? (size_t)N * 12 : 0);                     \
    fps_resident_kernel<NT, PPT, LDSC><<<B, NT, sh, st>>>(xyz, idx, N, M, log2bs);                             \
  } while (0)
  // Workgroup shape: the selection loop is a chain of M dependent arg-max reductions whose length is the instructions a wavefront
  // issues per step — ~8 per point plus the two reduction levels; every size takes the fastest shape of `tools/experiments/sweep_fps_shape.py`
  // (with the 64-bit reductions, ~70 instructions per wavefront, fewer and fatter wavefronts won: 512 x 16 at N = 8192; with the
  // single-word ones the per-wavefront cost is small again and 1024 x 8 is back in front).
  int nt = 0, ppt = 0;
  if (g_fps_nt && (long long)g_fps_nt * g_fps_ppt >= N) nt = g_fps_nt, ppt = g_fps_ppt;   // debug / sweep override
  else if (N <= 512) nt = 256, ppt = 2;
  else if (N <= 1024) nt = 256, ppt = 4;
  else if (N <= 2048) nt = 512, ppt = 4;
  else if (N <= 4096) nt = 512, ppt = 8;
  else if (N <= 8192) nt = 1024, ppt = 8;
  else if (N <= 16384) nt = 1024, ppt = 16;
  const bool ldsc = N <= 8192;
#define DFX_FPS_CASE(NT, PPT)                                                                  \
  if (nt == NT && ppt == PPT) {                                                                \
    if (ldsc) DFX_FPS_LAUNCH(NT, PPT, true);                                                   \
    else DFX_FPS_LAUNCH(NT, PPT, false);                                                       \
  } else
  DFX_FPS_CASE(256, 2) DFX_FPS_CASE(256, 4) DFX_FPS_CASE(256, 8) DFX_FPS_CASE(256, 16) DFX_FPS_CASE(256, 32)
  DFX_FPS_CASE(512, 2) DFX_FPS_CASE(512, 4) DFX_FPS_CASE(512, 8) DFX_FPS_CASE(512, 16)
  DFX_FPS_CASE(1024, 2) DFX_FPS_CASE(1024, 4) DFX_FPS_CASE(1024, 8) DFX_FPS_CASE(1024, 16)
  if (nt) return dfx::set_error(DFX_ERR_INVALID_ARG, "fps: no kernel for %d threads x %d points per thread", nt, ppt);
#undef DFX_FPS_CASE
  else {
    DFX_REQUIRE(tmp, "fps: N=%d > %d needs the (B,N) float scratch `tmp`", N, dfx_fps_max_resident());
    fps_streaming_kernel<1024><<<B, 1024, 0, st>>>(xyz, tmp, idx, N, M, log2bs);
  }
#undef DFX_FPS_LAUNCH
  return dfx::check_launch("furthest_point_sampling");
}

int dfx_ball_query_f32(const float *new_xyz, const float *xyz, int32_t *idx, int B, int N, int M, float radius,
                       int nsample, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && N >= 0 && M >= 0 && nsample >= 0, "ball_query: negative size");
  if ((long long)B * M * nsample == 0) return DFX_OK;
  DFX_REQUIRE(new_xyz && idx && (xyz || N == 0), "ball_query: null pointer");
  const int wg_per_cloud = (M + BQ_CENTRES_PER_WG - 1) / BQ_CENTRES_PER_WG;
  const float r2 = radius * radius;
  const size_t cloud_bytes = (size_t)N * 12;
  if (cloud_bytes <= 144 * 1024) {
    ball_query_kernel<true><<<B * wg_per_cloud, BQ_WAVES * WAVE, cloud_bytes, dfx::as_stream(stream)>>>(
        new_xyz, xyz, idx, N, M, r2, nsample, wg_per_cloud);
  } else {
    ball_query_kernel<false><<<B * wg_per_cloud, BQ_WAVES * WAVE, 0, dfx::as_stream(stream)>>>(
        new_xyz, xyz, idx, N, M, r2, nsample, wg_per_cloud);
  }
  return dfx::check_launch("ball_query");
}

int dfx_group_points_f32(const float *points, const int32_t *idx, float *out, int B, int C, int N, int npoints,
                         int nsample, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && C >= 0 && N >= 0 && npoints >= 0 && nsample >= 0, "group_points: negative size");
  const long long S = (long long)npoints * nsample;
  const long long total = (long long)B * S;
  if (total == 0 || C == 0) return DFX_OK;
  DFX_REQUIRE(points && idx && out, "group_points: null pointer");
  DFX_REQUIRE(S < (1LL << 31), "group_points: npoints*nsample too large");
  group_points_kernel<<<grid_for(total), 256, 0, dfx::as_stream(stream)>>>(points, idx, out, C, N, (int)S, total);
  return dfx::check_launch("group_points");
}

int dfx_group_points_grad_f32(const float *grad_out, const int32_t *idx, float *grad_points, int B, int C, int N,
                              int npoints, int nsample, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && C >= 0 && N >= 0 && npoints >= 0 && nsample >= 0, "group_points_grad: negative size");
  if ((long long)B * C * N == 0) return DFX_OK;
  DFX_REQUIRE(grad_points, "group_points_grad: null pointer");
  DFX_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)B * C * N, dfx::as_stream(stream)));
  const long long S = (long long)npoints * nsample;
  const long long total = (long long)B * S;
  if (total == 0) return DFX_OK;
  DFX_REQUIRE(grad_out && idx, "group_points_grad: null pointer");
  group_points_grad_kernel<<<grid_for(total), 256, 0, dfx::as_stream(stream)>>>(grad_out, idx, grad_points, C, N,
                                                                               (int)S, total);
  return dfx::check_launch("group_points_grad");
}

int dfx_three_nn_f32(const float *unknown, const float *known, float *dist2, int32_t *idx, int B, int n, int m,
                     dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && n >= 0 && m >= 0, "three_nn: negative size");
  if ((long long)B * n == 0) return DFX_OK;
  DFX_REQUIRE(unknown && dist2 && idx && (known || m == 0), "three_nn: null pointer");
  const int wg_per_cloud = (n + 255) / 256;
  three_nn_kernel<<<B * wg_per_cloud, 256, 0, dfx::as_stream(stream)>>>(unknown, known, dist2, idx, n, m,
                                                                       wg_per_cloud);
  return dfx::check_launch("three_nn");
}

int dfx_three_interpolate_f32(const float *points, const int32_t *idx, const float *weight, float *out, int B,
                              int c, int m, int n, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && c >= 0 && m >= 0 && n >= 0, "three_interpolate: negative size");
  const long long total = (long long)B * c * n;
  if (total == 0) return DFX_OK;
  DFX_REQUIRE(points && idx && weight && out, "three_interpolate: null pointer");
  three_interpolate_kernel<<<grid_for(total), 256, 0, dfx::as_stream(stream)>>>(points, idx, weight, out, c, m, n,
                                                                               total);
  return dfx::check_launch("three_interpolate");
}

int dfx_three_interpolate_grad_f32(const float *grad_out, const int32_t *idx, const float *weight,
                                   float *grad_points, int B, int c, int n, int m, dfx_stream_t stream) {
  DFX_REQUIRE(B >= 0 && c >= 0 && m >= 0 && n >= 0, "three_interpolate_grad: negative size");
  if ((long long)B * c * m == 0) return DFX_OK;
  DFX_REQUIRE(grad_points, "three_interpolate_grad: null pointer");
  DFX_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)B * c * m, dfx::as_stream(stream)));
  const long long total = (long long)B * c * n;
  if (total == 0) return DFX_OK;
  DFX_REQUIRE(grad_out && idx && weight, "three_interpolate_grad: null pointer");
  three_interpolate_grad_kernel<<<grid_for(total), 256, 0, dfx::as_stream(stream)>>>(grad_out, idx, weight,
                                                                                    grad_points, c, n, m, total);
  return dfx::check_launch("three_interpolate_grad");
}

}  // extern "C"
