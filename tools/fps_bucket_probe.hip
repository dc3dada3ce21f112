// tools/fps_bucket_probe.hip -- where a pick of the bucketed FPS kernel spends its time: fps_bucket.hip compiled with
// phase marks (shader clock, accumulated per wave of workgroup 0 over all steps) and touched-bucket counts at config 3.
// Intervals per step:  0->1 this wave's best bucket (DPP reduction if one of its buckets changed) + LDS write
//   1->2 barrier   2->3 16-value reduction, winner's coordinates from LDS, pick stored   3->4 box test + ballot
//   4->5 touched buckets re-evaluated (loads, update, reduction each)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Ipytorch_points_amd/csrc tools/fps_bucket_probe.hip -o tools/fps_bucket_probe
#include <hip/hip_runtime.h>
__device__ unsigned long long g_probe[16][16];
#ifdef PP_FPSB_NOMARKS  // timing only (with -DPP_FPSB_DOUBLE=<bits>: what one more copy of a link of the chain costs)
#define PP_FPSB_PROBE_DECL
#define PP_FPSB_MARK(n)
#define PP_FPSB_TOUCHED(mask)
#define PP_FPSB_END()
#else
#define PP_FPSB_PROBE_DECL                                   \
  unsigned long long pr_t[6] = {0, 0, 0, 0, 0, 0};           \
  unsigned long long pr_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PP_FPSB_MARK(n)                                                            \
  do {                                                                             \
    pr_t[n] = __builtin_amdgcn_s_memtime();                                        \
    if ((n) > 0) pr_acc[(n)-1] += pr_t[n] - pr_t[(n)-1];                           \
  } while (0)
#define PP_FPSB_TOUCHED(mask)                                \
  do {                                                       \
    const int c__ = __builtin_popcountll(mask);              \
    pr_acc[6] += c__;                                        \
    pr_acc[7 + (c__ > 4 ? 4 : c__)] += 1;                    \
  } while (0)
#define PP_FPSB_END()                                                         \
  do {                                                                        \
    if (blockIdx.x == 0 && lane == 0)                                         \
      for (int i__ = 0; i__ < 12; ++i__) g_probe[wave][i__] = pr_acc[i__];    \
  } while (0)
#endif
#include "../pytorch_points_amd/csrc/fps_bucket.hip"
#include "../pytorch_points_amd/csrc/fps.hip"
#include "../pytorch_points_amd/csrc/api.hip"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, N = argc > 2 ? atoi(argv[2]) : 65536, npoint = argc > 3 ? atoi(argv[3]) : 4096;
  std::vector<float> h((size_t)B * N * 3);
  srand(1);
  for (size_t i = 0; i < (size_t)B * N; ++i) {
    float x, y, z, r;
    do {
      x = rand() / (float)RAND_MAX * 2 - 1; y = rand() / (float)RAND_MAX * 2 - 1; z = rand() / (float)RAND_MAX * 2 - 1;
      r = x * x + y * y + z * z;
    } while (r > 1.0f || r < 1e-4f);
    r = 1.0f / sqrtf(r);
    h[3 * i] = x * r; h[3 * i + 1] = y * r; h[3 * i + 2] = z * r;
  }
  std::vector<float> big((size_t)B * N, 1e10f);
  float *x, *temp; int* idx; void* ws;
  const size_t wsb = pp_furthest_sampling_workspace_bytes(B, N, npoint);
  hipMalloc(&x, h.size() * 4); hipMalloc(&temp, big.size() * 4); hipMalloc(&idx, (size_t)B * npoint * 4); hipMalloc(&ws, wsb + 256);
  hipMemset(ws, 0, 256);
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) {
    hipMemcpy(temp, big.data(), big.size() * 4, hipMemcpyHostToDevice);
    hipEventRecord(a);
    const int rc = pp_furthest_sampling_f32(x, temp, idx, B, N, npoint, 0, ws, wsb, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long pr[16][16];
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr));
    printf("rc %d: %.3f ms = %.3f us/pick (events)\n", rc, ms, ms * 1e3 / (npoint - 1));
#ifdef PP_FPSB_NOMARKS
    continue;
#endif
    if (it < 2) continue;
    const double steps = npoint - 2;
    for (int w = 0; w < 16; ++w) {
      printf(" wave %2d cycles/step:", w);
      double tot = 0;
      for (int k = 0; k < 5; ++k) { printf(" p%d %6.0f", k, pr[w][k] / steps); tot += pr[w][k] / steps; }
      printf("  sum %6.0f | touched/step %.2f; steps with 0/1/2/3/4+ touched: %.3f %.3f %.3f %.3f %.3f\n", tot, pr[w][6] / steps,
             pr[w][7] / steps, pr[w][8] / steps, pr[w][9] / steps, pr[w][10] / steps, pr[w][11] / steps);
    }
  }
  return 0;
}
