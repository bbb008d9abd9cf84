// Where a tail iteration of the EMD auction (one unassigned point) spends its cycles, seen by wavefront 0 of pair 0:
//   hipcc --offload-arch=gfx950 -O3 -I difffacto_amd/csrc -I include tools/ubench/emd_phase_probe.hip -o tools/ubench/_build/emd_phase_probe
// (the product kernel compiled with its probe points: clock64() deltas accumulated per phase; the probes drain the LDS queue, so the
// phases add up to a little more than the unprobed iteration)
#define DFX_EMD_PROBE 1
#include "../../difffacto_amd/csrc/emd_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdarg>
namespace dfx { int set_error(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc(10, stderr); return code; } }
int main(int argc, char **argv) {
  const bool gauss = argc > 1;   // any argument: clouds clustered around the centre (sum of 4 uniforms, sigma ~ 0.07) instead of uniform
  const int B = 32, n = 2048, iters = 10000;
  std::vector<float> a((size_t)B * n * 3), b(a.size());
  srand(1);
  auto draw = [&]() { float u = rand() / (float)RAND_MAX; if (gauss) { u = (u + rand() / (float)RAND_MAX + rand() / (float)RAND_MAX + rand() / (float)RAND_MAX) * 0.25f; u = 0.5f + (u - 0.5f) * 0.5f; } return u; };
  for (auto &v : a) v = draw();
  for (auto &v : b) v = draw();
  float *da, *db, *dd;
  int32_t *das;
  void *ws;
  hipMalloc(&da, a.size() * 4), hipMalloc(&db, a.size() * 4), hipMalloc(&dd, (size_t)B * n * 4), hipMalloc(&das, (size_t)B * n * 4);
  hipMalloc(&ws, dfx_emd_workspace_bytes(B, n));
  hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), a.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    long long z[24] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(g_emd_probe), z, sizeof(z));
    hipEventRecord(e0);
    if (dfx_emd_forward_f32(da, db, dd, das, ws, B, n, 0.002f, iters, nullptr) != 0) return 1;
    hipEventRecord(e1), hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_emd_probe), sizeof(z));
    const char *name[8] = {"list + own xyz", "scan", "wave reduce", "barrier (partials)", "merge + bid", "GetMax", "Assign", "barrier (end)"};
    printf("%.1f ms the call; pair 0: %lld cycles in the loop\n", ms, z[16]);
    const char *cls[4] = {"U = 1", "U = 2..8", "U = 9..16", "U > 16"};
    for (int i = 0; i < 4; ++i) printf("  %-10s %6lld iterations, %8.0f cycles each, %5.1f %% of the loop\n", cls[i], z[12 + i], z[12 + i] ? (double)z[8 + i] / z[12 + i] : 0., 100. * z[8 + i] / z[16]);
    z[15] = z[12];
    long long tot = 0;
    for (int i = 0; i < 8; ++i) printf("  %-20s %7.0f cycles\n", name[i], (double)z[i] / z[15]), tot += z[i];
    printf("  %-20s %7.0f cycles\n", "sum", (double)tot / z[15]);
  }
  return 0;
}
