// tools/build_probe.hip -- where the time of the grid build goes: the build body of grid_common.h with
// PP_PHASE recording the 100 MHz clock at every phase boundary (workgroup 0 and the last one), on
// B = 32 x 2 sets of 16384 uniform-sphere points, 4 slabs per set, as in chamfer_grid.hip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Ipytorch_points_amd/csrc tools/build_probe.hip -o /tmp/build_probe
#include <hip/hip_runtime.h>
__device__ unsigned long long g_phase[2][16];
#define PP_PHASE(n)                                                                          \
  do {                                                                                       \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                \
      g_phase[blockIdx.x == 0 ? 0 : 1][n] = wall_clock64();                                  \
  } while (0)
#include "grid_common.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ __launch_bounds__(pp::kBuildThreads) void probe_kernel(const float* xyz, pp::GridSet* gs, unsigned* cs,
                                                                  pp::f4* sorted, int nsets, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_cnt[];
  const int V = pp::xcd_virtual_block(blockIdx.x, (nsets * pp::kBuildSlabs + 7) / 8);
  if (V >= nsets * pp::kBuildSlabs) return;
  const int set = V / pp::kBuildSlabs, slab = V % pp::kBuildSlabs;
  pp::grid_build_set<false, true>(xyz + (size_t)set * n * 3, n, gs + set, cs + (size_t)set * (pp::kGridCells + 1),
                     sorted + (size_t)set * n, nullptr, s_cnt, nullptr, nullptr, slab, pp::kBuildSlabs);
}

int main() {
  const int nsets = 64, n = 16384;
  std::vector<float> h((size_t)nsets * n * 3);
  srand(1);
  for (size_t i = 0; i < (size_t)nsets * n; ++i) {
    float x, y, z, r;
    do {
      x = rand() / (float)RAND_MAX * 2 - 1; y = rand() / (float)RAND_MAX * 2 - 1; z = rand() / (float)RAND_MAX * 2 - 1;
      r = x * x + y * y + z * z;
    } while (r > 1.0f || r < 1e-4f);
    r = 1.0f / sqrtf(r);
    h[3 * i] = x * r; h[3 * i + 1] = y * r; h[3 * i + 2] = z * r;
  }
  float* d; pp::GridSet* gs; unsigned* cs; pp::f4* sorted;
  hipMalloc(&d, h.size() * 4); hipMalloc(&gs, nsets * sizeof(pp::GridSet));
  hipMalloc(&cs, (size_t)nsets * (pp::kGridCells + 1) * 4); hipMalloc(&sorted, (size_t)nsets * n * 16);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const size_t lds = pp::grid_build_lds_bytes(pp::kBuildSlabs);
  hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(a);
    probe_kernel<<<8 * ((nsets * pp::kBuildSlabs + 7) / 8), pp::kBuildThreads, lds>>>(d, gs, cs, sorted, nsets, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long ph[2][16];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph));
    printf("iter %d: %.1f us (events); phases in us:", it, ms * 1000);
    for (int w = 0; w < 2; ++w) {
      printf("\n   wg %s:", w ? "last " : "first");
      for (int k = 1; k <= 10; ++k) printf(" p%d %.2f", k - 1, (double)(ph[w][k] - ph[w][k - 1]) / 100.0);
      printf("  total %.2f", (double)(ph[w][10] - ph[w][0]) / 100.0);
    }
    printf("\n");
  }
  pp::GridSet g0; hipMemcpy(&g0, gs, sizeof(g0), hipMemcpyDeviceToHost);
  printf("set 0: grid %d x %d x %d, h %g\n", g0.gx, g0.gy, g0.gz, g0.h);
  return 0;
}
