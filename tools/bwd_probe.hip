// tools/bwd_probe.hip -- where the time of the Chamfer backward (nmdist_bwd_lds64_kernel) goes: chamfer.hip
// compiled with PP_PHASE recording the 100 MHz clock at the phase boundaries of workgroup 0 and the last one.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Ipytorch_points_amd/csrc tools/bwd_probe.hip -o tools/bwd_probe
#include <hip/hip_runtime.h>
__device__ unsigned long long g_phase[2][16];
#define PP_PHASE(n)                                                                          \
  do {                                                                                       \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                \
      g_phase[blockIdx.x == 0 ? 0 : 1][n] = wall_clock64();                                  \
  } while (0)
#include "../pytorch_points_amd/csrc/chamfer.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main() {
  const int B = 32, N = 16384, M = 16384;
  std::vector<float> h((size_t)B * N * 3), g((size_t)B * N, 1.0f / (B * N));
  std::vector<int> idx((size_t)B * N);
  srand(1);
  for (auto& v : h) v = rand() / (float)RAND_MAX;
  for (auto& v : idx) v = rand() % M;  // a random assignment: every target gets ~1 scattered term
  float *x1, *x2, *g1, *g2, *o1, *o2; int *i1, *i2;
  hipMalloc(&x1, h.size() * 4); hipMalloc(&x2, h.size() * 4); hipMalloc(&o1, h.size() * 4); hipMalloc(&o2, h.size() * 4);
  hipMalloc(&g1, g.size() * 4); hipMalloc(&g2, g.size() * 4); hipMalloc(&i1, idx.size() * 4); hipMalloc(&i2, idx.size() * 4);
  hipMemcpy(x1, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(x2, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(g1, g.data(), g.size() * 4, hipMemcpyHostToDevice); hipMemcpy(g2, g.data(), g.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(i1, idx.data(), idx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(i2, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(a);
    int rc = pp_nmdistance_backward_f32(x1, x2, g1, g2, i1, i2, o1, o2, B, N, M, 3, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long ph[2][16];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph));
    printf("iter %d rc %d: %.1f us (events); phases in us:", it, rc, ms * 1000);
    for (int w = 0; w < 2; ++w) {
      printf("\n   wg %s:", w ? "last " : "first");
      for (int k = 1; k <= 5; ++k) printf(" p%d %.2f", k - 1, (double)(ph[w][k] - ph[w][k - 1]) / 100.0);
      printf("  total %.2f", (double)(ph[w][5] - ph[w][0]) / 100.0);
    }
    printf("\n");
  }
  return 0;
}
