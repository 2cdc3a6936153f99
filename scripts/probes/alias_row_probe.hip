// Where one long row's alias build spends its cycles (sum | classify | pairing | leftovers), one wave.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DGLX_ALIAS_PROFILE \
//   -I include -I graph-learn_amd/csrc scripts/probes/alias_row_probe.hip -o /tmp/alias_row_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "glx_common.h"

__global__ __launch_bounds__(64) void one_row(const float* w, int32_t n, GlxAlias* tab, GlxAlias* stk) {
  glx_alias_build_row_wave(w, n, tab, stk);
}

int main(int argc, char** argv) {
  const int32_t n = argc > 1 ? atoi(argv[1]) : 138719;
  std::vector<float> w(n);
  unsigned s = 12345;
  for (auto& x : w) {
    s = s * 1664525u + 1013904223u;
    x = 0.01f + 0.99f * (float)(s >> 8) / 16777216.0f;
  }
  float* d_w;
  GlxAlias *tab, *stk;
  hipMalloc(&d_w, n * 4);
  hipMalloc(&tab, n * 8);
  hipMalloc(&stk, n * 8);
  hipMemcpy(d_w, w.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    one_row<<<1, 64>>>(d_w, n, tab, stk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long t[8];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(glx_alias_prof), sizeof(t));
    const double tot = (double)(t[4] - t[0]);
    printf("n = %d: %.3f ms (%.1f ns / entry); cycles: sum %.0f%% classify %.0f%% pairing %.0f%% leftovers %.0f%% (total %.0f cycles, %.1f per entry)\n",
           n, ms, ms * 1e6 / n, 100 * (t[1] - t[0]) / tot, 100 * (t[2] - t[1]) / tot, 100 * (t[3] - t[2]) / tot,
           100 * (t[4] - t[3]) / tot, tot, tot / n);
  }
  return 0;
}
