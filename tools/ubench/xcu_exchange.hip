// Micro-benchmark behind DESIGN.md 5.1b: what does it cost to split ONE 32-point tile's feed-forward over G CUs?
// Each of the G workgroups of a group (512 threads = 8 wavefronts, one workgroup per CU) owns 1 / G of the hidden units; per transformer block
// it produces a 128 x 32 fp32 partial sum (16 KiB) that the others need before the next block's attention can start.  One exchange =
//     write the own 16 KiB partial (whole-line stores, all 8 wavefronts) -> release a per-(group, member) sequence flag (L2-coherent atomic)
//     -> wait for the other G - 1 flags -> read their partials ((G - 1) x 16 KiB, behind an agent-scope acquire) and add.
// Reports shader cycles per exchange (s_memtime of wave 0, averaged over the launch) for G = 2, 4, 8 with the members of a group on ONE XCD
// (workgroup ids congruent mod 8: ids are dealt round-robin over the 8 XCDs, so they share an L2) or on DIFFERENT XCDs (consecutive ids: the
// exchange crosses the fabric / MALL), with 1 or 32 groups running at once, and the same loop WITHOUT the flag wait (store + load cost alone).
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/xcu_exchange.hip -o tools/ubench/_build/xcu_exchange && tools/ubench/_build/xcu_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int PART_FLOATS = 128 * 32;   // one partial: 16 KiB

// The LEAN variant (one XCD only): no memory-model fences at all — the stores are left to reach the XCD's L2 on their own (the vector cache is write-through:
// s_waitcnt vmcnt(0) = acknowledged by L2), the flag is a plain store behind them, the reader polls and reads with sc0 sc1 loads (served by L2, past its own
// vector cache).  Physically coherent only because both workgroups sit on the same XCD (one L2); it is the floor of any same-XCD hand-off, not a portable protocol.
__device__ __forceinline__ float4 load_l2(const float4 *p) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned load_l2_u32(const unsigned *p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__global__ void __launch_bounds__(512, 1) k_exchange_lean(float *parts, unsigned *flags, unsigned long long *cycles, float *sink, int G, int ngroups, int iters) {
  const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int grp = (k / G) * 8 + x, mem = k % G;
  if (grp >= ngroups) return;
  const int tid = threadIdx.x;
  float4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  float4 mine[2];
  mine[0] = float4{tid * 1.0f, 1.f, 2.f, 3.f}, mine[1] = float4{tid * 2.0f, 1.f, 2.f, 3.f};
  float *gp = parts + (size_t)grp * G * 2 * PART_FLOATS;
  unsigned *gf = flags + (size_t)grp * 64;
  unsigned long long t0 = 0;
  for (int it = 1; it <= iters; ++it) {
    if (it == 2 && tid == 0) t0 = __builtin_readcyclecounter();
    float *buf = gp + (size_t)(it & 1) * G * PART_FLOATS;
    float4 *dst = reinterpret_cast<float4 *>(buf + (size_t)mem * PART_FLOATS);
    mine[0].x += 1.f, mine[1].x += 1.f;
    dst[tid] = mine[0], dst[512 + tid] = mine[1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wavefront's stores acknowledged by L2
    __syncthreads();
    if (tid == 0) {
      gf[mem] = (unsigned)it;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (tid < G && tid != mem) {
      while (load_l2_u32(gf + tid) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    for (int m = 0; m < G; ++m) {
      if (m == mem) continue;
      const float4 *src = reinterpret_cast<const float4 *>(buf + (size_t)m * PART_FLOATS);
      const float4 a = load_l2(src + tid), b = load_l2(src + 512 + tid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc[0].x += a.x, acc[0].y += a.y, acc[0].z += a.z, acc[0].w += a.w;
      acc[1].x += b.x, acc[1].y += b.y, acc[1].z += b.z, acc[1].w += b.w;
    }
    mine[0].y += acc[0].x * 1e-30f;
  }
  if (tid == 0) cycles[blockIdx.x] = (__builtin_readcyclecounter() - t0) / (unsigned long long)(iters - 1);
  if (acc[0].x + acc[1].y == 12345.678f) sink[0] = acc[0].x;
  // (a wrong sum would mean a stale read: checked by the host through `sink2`)
  if (tid == 0 && mem == 0) sink[1 + grp % 8] = acc[0].x;
}

template <bool WAIT>
__global__ void __launch_bounds__(512, 1) k_exchange(float *parts, unsigned *flags, unsigned long long *cycles, float *sink, int G, int same_xcd,
                                                     int ngroups, int iters) {
  // group / member of this workgroup
  int grp, mem;
  if (same_xcd) {   // members = ids congruent mod 8: group g of XCD x = ids x + 8 (g' G + m)
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    grp = (k / G) * 8 + x, mem = k % G;
  } else {
    grp = blockIdx.x / G, mem = blockIdx.x % G;
  }
  if (grp >= ngroups) return;
  const int tid = threadIdx.x;
  float4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  float4 mine[2];
  mine[0] = float4{tid * 1.0f, 1.f, 2.f, 3.f}, mine[1] = float4{tid * 2.0f, 1.f, 2.f, 3.f};
  float *gp = parts + (size_t)grp * G * 2 * PART_FLOATS;   // [2 parities][G members][16 KiB]
  unsigned *gf = flags + (size_t)grp * 64;
  unsigned long long t0 = 0;
  for (int it = 1; it <= iters; ++it) {
    if (it == 2 && tid == 0) t0 = __builtin_readcyclecounter();
    float *buf = gp + (size_t)(it & 1) * G * PART_FLOATS;
    // own partial: 512 threads x 2 x 16 bytes = 16 KiB
    float4 *dst = reinterpret_cast<float4 *>(buf + (size_t)mem * PART_FLOATS);
    mine[0].x += 1.f, mine[1].x += 1.f;
    dst[tid] = mine[0], dst[512 + tid] = mine[1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(gf + mem, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (WAIT) {
      if (tid < G && tid != mem) {
        while (__hip_atomic_load(gf + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    for (int m = 0; m < G; ++m) {
      if (m == mem) continue;
      const float4 *src = reinterpret_cast<const float4 *>(buf + (size_t)m * PART_FLOATS);
      const float4 a = src[tid], b = src[512 + tid];   // (behind the acquire fence: the CU's vector cache has been invalidated)
      acc[0].x += a.x, acc[0].y += a.y, acc[0].z += a.z, acc[0].w += a.w;
      acc[1].x += b.x, acc[1].y += b.y, acc[1].z += b.z, acc[1].w += b.w;
    }
    mine[0].y += acc[0].x * 1e-30f;   // (the next partial depends on what was read: no overlap across exchanges, like the residual stream)
  }
  if (tid == 0) cycles[blockIdx.x] = (__builtin_readcyclecounter() - t0) / (unsigned long long)(iters - 1);
  if (acc[0].x + acc[1].y == 12345.678f) sink[0] = acc[0].x;
}

template <bool WAIT>
static void run(int G, int same_xcd, int ngroups, int iters) {
  float *parts, *sink;
  unsigned *flags;
  unsigned long long *cyc;
  const int nwg = same_xcd ? ((ngroups + 7) / 8) * G * 8 : ngroups * G;
  hipMalloc(&parts, (size_t)ngroups * G * 2 * PART_FLOATS * sizeof(float));
  hipMalloc(&flags, (size_t)ngroups * 64 * sizeof(unsigned));
  hipMalloc(&cyc, nwg * sizeof(unsigned long long));
  hipMalloc(&sink, 64);
  hipMemset(flags, 0, (size_t)ngroups * 64 * sizeof(unsigned));
  hipMemset(cyc, 0, nwg * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0);
  k_exchange<WAIT><<<nwg, 512>>>(parts, flags, cyc, sink, G, same_xcd, ngroups, iters);
  hipEventRecord(e1);
  if (hipEventSynchronize(e1) != hipSuccess) {
    printf("G=%d same_xcd=%d groups=%d: launch failed\n", G, same_xcd, ngroups);
    return;
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0, mx = 0;
  int n = 0;
  for (auto v : h)
    if (v) s += (double)v, mx = v > mx ? (double)v : mx, ++n;
  printf("G=%d  %-14s groups=%-3d %-22s %8.0f cycles per exchange (max %8.0f; %.2f us by the event clock)\n", G, same_xcd ? "one XCD" : "across XCDs", ngroups,
         WAIT ? "store+flag+wait+load" : "store+load, no wait", n ? s / n : 0.0, mx, ms * 1e3 / iters);
  (void)hipFree(parts), (void)hipFree(flags), (void)hipFree(cyc), (void)hipFree(sink);
}

static void run_lean(int G, int ngroups, int iters) {
  float *parts, *sink;
  unsigned *flags;
  unsigned long long *cyc;
  const int nwg = ((ngroups + 7) / 8) * G * 8;
  hipMalloc(&parts, (size_t)ngroups * G * 2 * PART_FLOATS * sizeof(float));
  hipMalloc(&flags, (size_t)ngroups * 64 * sizeof(unsigned));
  hipMalloc(&cyc, nwg * sizeof(unsigned long long));
  hipMalloc(&sink, 64);
  hipMemset(flags, 0, (size_t)ngroups * 64 * sizeof(unsigned));
  hipMemset(cyc, 0, nwg * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0);
  k_exchange_lean<<<nwg, 512>>>(parts, flags, cyc, sink, G, ngroups, iters);
  hipEventRecord(e1);
  if (hipEventSynchronize(e1) != hipSuccess) {
    printf("lean G=%d groups=%d: launch failed\n", G, ngroups);
    return;
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  float hs[16];
  hipMemcpy(hs, sink, sizeof(hs), hipMemcpyDeviceToHost);
  // member 0 of group 0 adds, per exchange, the .x of thread 0's first float4 of every other member: tid * 1 + it = it  ->  (G - 1) * sum_{it = 1..iters} it
  float expect = 0.f;   // (the device's own order of fp32 additions)
  for (int it = 1; it <= iters; ++it)
    for (int m = 1; m < G; ++m) expect += (float)it;
  double s = 0, mx = 0;
  int n = 0;
  for (auto v : h)
    if (v) s += (double)v, mx = v > mx ? (double)v : mx, ++n;
  printf("G=%d  %-14s groups=%-3d %-22s %8.0f cycles per exchange (max %8.0f; %.2f us by the event clock)  [sum check %s: %.0f / %.0f]\n", G, "one XCD, LEAN", ngroups,
         "store+flag+wait+load", n ? s / n : 0.0, mx, ms * 1e3 / iters, hs[1] == expect ? "ok" : "STALE READS", (double)hs[1], (double)expect);
  (void)hipFree(parts), (void)hipFree(flags), (void)hipFree(cyc), (void)hipFree(sink);
}

int main() {
  const int iters = 5000;   // = the exchanges of one T = 1000 chain (5 blocks per step)
  for (int G : {2, 4, 8})
    for (int same : {1, 0})
      for (int ng : {1, 32}) {
        if (same && (ng + 7) / 8 * G * 8 > 256) continue;   // one workgroup per CU
        if (!same && ng * G > 256) continue;
        run<true>(G, same, ng, iters);
      }
  for (int G : {2, 4}) {
    run_lean(G, 1, iters);
    run_lean(G, 32, iters);
  }
  run<false>(4, 1, 1, iters);
  run<false>(4, 1, 32, iters);
  return 0;
}
