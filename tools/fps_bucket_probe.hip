// tools/fps_bucket_probe.hip -- where a round of the bucketed FPS kernel spends its time: fps_bucket.hip compiled with
// phase marks (shader clock, accumulated per wave of workgroup 0 over all rounds), touched-bucket counts and the
// picks-per-round histogram at config 3.  Intervals per round of the batched chain (round 6):
//   0->1 this wave's two best buckets (two reductions if one of them was visited) + the post to LDS
//   1->2 barrier   2->3 own candidates against the others (a pair per lane), FAIL raised   3->4 barrier
//   4->5 picks read, own picks stored, box tests of every pick + ballot   5->6 touched buckets re-evaluated
// (the one-pick chain, PP_PROBE_CHAIN=1: 0->1 best bucket + post, 1->2 barrier, 2->3 pick read, 3->4 box test, 4->5 visits)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Ipytorch_points_amd/csrc tools/fps_bucket_probe.hip -o tools/fps_bucket_probe
#include <hip/hip_runtime.h>
__device__ unsigned long long g_probe[16][40];
__device__ unsigned long long g_at[16];
#ifdef PP_FPSB_NOMARKS  // timing only (with -DPP_FPSB_DOUBLE=<bits>: what one more copy of a link of the chain costs)
#define PP_FPSB_PROBE_DECL
#define PP_FPSB_MARK(n)
#define PP_FPSB_TOUCHED(mask)
#define PP_FPSB_PICKS(k)
#define PP_FPSB_ELIG(e, k)
#define PP_FPSB_AT(j)
#define PP_FPSB_END()
#else
// pr_acc: 0..5 phase clocks, 6 touched buckets, 12 rounds; pr_hist (LDS: indexed by a run-time value -- a register
// array would go to scratch memory, whose vmcnt(0) waits would be charged to the phase): 0..4 rounds with 0/1/2/3/4+
// touched buckets, 8..25 rounds with k picks
#define PP_FPSB_PROBE_DECL                                   \
  unsigned long long pr_t[7] = {0, 0, 0, 0, 0, 0, 0};        \
  unsigned long long pr_acc[13] = {0};                       \
  unsigned long long pr_first = 0; int pr_next = 3;          \
  __shared__ unsigned pr_hist[16][32];                       \
  if (lane < 32) pr_hist[wave][lane] = 0u;
#define PP_FPSB_MARK(n)                                                            \
  do {                                                                             \
    pr_t[n] = __builtin_amdgcn_s_memtime();                                        \
    if ((n) > 0) pr_acc[(n)-1] += pr_t[n] - pr_t[(n)-1];                           \
  } while (0)
#define PP_FPSB_TOUCHED(mask)                                \
  do {                                                       \
    const int c__ = __builtin_popcountll(mask);              \
    pr_acc[6] += c__;                                        \
    if (lane == 0) pr_hist[wave][c__ > 4 ? 4 : c__] += 1u;   \
    pr_acc[12] += 1;                                         \
  } while (0)
#define PP_FPSB_PICKS(k) do { if (lane == 0) pr_hist[wave][8 + ((k) > 17 ? 17 : (k))] += 1u; } while (0)
// eligible candidates of the round (above BOUND) against the picks it took: summed, and the rounds that took them all
#define PP_FPSB_ELIG(e, k) do { pr_acc[7] += (e); pr_acc[8] += ((k) == (e)) ? 1 : 0; } while (0)
// the clock when the chain reaches pick 2^e (e = 3 .. 12), wave 0 of workgroup 0: where in the call the time goes
#define PP_FPSB_AT(j)                                                                                 \
  do {                                                                                                \
    if (pr_first == 0) pr_first = pr_t[0];                                                            \
    while (pr_next <= 12 && (j) >= (1 << pr_next)) {                                                   \
      if (blockIdx.x == 0 && t == 0) g_at[pr_next] = pr_t[0] - pr_first;                              \
      ++pr_next;                                                                                      \
    }                                                                                                 \
  } while (0)
#define PP_FPSB_END()                                                         \
  do {                                                                        \
    if (blockIdx.x == 0 && lane == 0) {                                       \
      for (int i__ = 0; i__ < 7; ++i__) g_probe[wave][i__] = pr_acc[i__];     \
      g_probe[wave][12] = pr_acc[12];                                         \
      g_probe[wave][32] = pr_acc[7]; g_probe[wave][33] = pr_acc[8];           \
      for (int i__ = 0; i__ < 5; ++i__) g_probe[wave][7 + i__] = pr_hist[wave][i__];   \
      for (int i__ = 0; i__ < 18; ++i__) g_probe[wave][13 + i__] = pr_hist[wave][8 + i__]; \
    }                                                                         \
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
  if (getenv("PP_PROBE_CHAIN")) pp_debug_set_fps_bucket_chain(atoi(getenv("PP_PROBE_CHAIN")));
  if (getenv("PP_PROBE_SORT")) pp_debug_set_fps_bucket_sort(atoi(getenv("PP_PROBE_SORT")));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) {
    hipMemcpy(temp, big.data(), big.size() * 4, hipMemcpyHostToDevice);
    hipEventRecord(a);
    const int rc = pp_furthest_sampling_f32(x, temp, idx, B, N, npoint, 0, ws, wsb, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long pr[16][40];
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr));
    printf("rc %d: %.3f ms = %.3f us/pick (events)\n", rc, ms, ms * 1e3 / (npoint - 1));
#ifdef PP_FPSB_NOMARKS
    continue;
#endif
    if (it < 2) continue;
    unsigned long long at[16];
    hipMemcpyFromSymbol(at, HIP_SYMBOL(g_at), sizeof(at));
    printf(" cycles from the chain's start to pick 2^e:");
    for (int e = 3; e <= 12; ++e) printf(" %d:%llu", 1 << e, at[e]);
    printf("\n");
    const double rounds = (double)pr[0][12];
    printf(" %.0f rounds for %d picks: %.2f picks per round; rounds with k picks:", rounds, npoint - 1, (npoint - 1) / rounds);
    for (int k = 0; k <= 17; ++k)
      if (pr[0][13 + k]) printf(" %d:%llu", k, pr[0][13 + k]);
    printf("\n");
    printf(" eligible candidates per round (above BOUND): %.2f; rounds that took every eligible one: %.3f of the rounds\n",
           pr[0][32] / rounds, pr[0][33] / rounds);
    for (int w = 0; w < 16; ++w) {
      printf(" wave %2d cycles/round:", w);
      double tot = 0;
      for (int k = 0; k < 6; ++k) { printf(" p%d %6.0f", k, pr[w][k] / rounds); tot += pr[w][k] / rounds; }
      printf("  sum %6.0f | touched/round %.2f; rounds with 0/1/2/3/4+ touched: %.3f %.3f %.3f %.3f %.3f\n", tot,
             pr[w][6] / rounds, pr[w][7] / rounds, pr[w][8] / rounds, pr[w][9] / rounds, pr[w][10] / rounds,
             pr[w][11] / rounds);
    }
  }
  return 0;
}
