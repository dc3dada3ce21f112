// tools/fps_probe.hip -- where a pick of the FPS cluster kernel spends its time: fps.hip compiled with the
// per-step marks storing the 100 MHz clock (thread 0 of workgroup 0, last 64 steps) at config 3
// (B=16, N=65536 -> 4096).  Intervals:
//   0->1 coordinates of the last pick (dependent scalar loads) + distance update of the slice + per-thread best
//   1->2 wave reduction, LDS, barrier            2->3 workgroup reduction, publish the granule
//   3->4 poll the cluster's granules             4->5 final reduction, LDS, barrier       5->0' back to the top
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Ipytorch_points_amd/csrc tools/fps_probe.hip -o tools/fps_probe
#include <hip/hip_runtime.h>
__shared__ unsigned long long s_probe_ts[64 * 8];
__device__ unsigned long long g_probe_ts[64 * 8];
#define PP_FPS_MARK(n)                                                                                   \
  do {                                                                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0) s_probe_ts[(j & 63) * 8 + (n)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define PP_FPS_MARK_END()                                                               \
  do {                                                                                  \
    if (blockIdx.x == 0 && threadIdx.x == 0)                                            \
      for (int i__ = 0; i__ < 64 * 8; ++i__) g_probe_ts[i__] = s_probe_ts[i__];         \
  } while (0)
#include "../pytorch_points_amd/csrc/fps.hip"
#include "../pytorch_points_amd/csrc/api.hip"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main() {
  const int B = 16, N = 65536, npoint = 4096;
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
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) {
    hipMemcpy(temp, big.data(), big.size() * 4, hipMemcpyHostToDevice);
    hipEventRecord(a);
    const int rc = pp_furthest_sampling_f32(x, temp, idx, B, N, npoint, 0, ws, wsb, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long ts[64 * 8];
    hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_probe_ts), sizeof(ts));
    double acc[6] = {0, 0, 0, 0, 0, 0};
    int n = 0;
    for (int s = 0; s < 63; ++s) {  // step slot s and the next one (skip the wrap between the oldest and newest)
      const int nx = s + 1;
      if (ts[nx * 8] <= ts[s * 8]) continue;
      for (int k = 0; k < 5; ++k) acc[k] += (double)(ts[s * 8 + k + 1] - ts[s * 8 + k]);
      acc[5] += (double)(ts[nx * 8] - ts[s * 8 + 5]);
      ++n;
    }
    printf("rc %d: %.3f ms = %.3f us/pick (events);  per step over %d steps, us:", rc, ms, ms * 1e3 / (npoint - 1), n);
    double tot = 0;
    for (int k = 0; k < 6; ++k) { printf(" p%d %.3f", k, acc[k] / n / 100.0); tot += acc[k] / n / 100.0; }
    printf("  sum %.3f\n", tot);
  }
  return 0;
}
