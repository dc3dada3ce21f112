// dispatch_probe.hip -- how fast does the chip start workgroups that leave at once?  Kernels of 256 threads that read one
// word and return, without / with scratch memory (a private array indexed at run time) and without / with 24 KB of LDS,
// timed over grid sizes.  hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <bool SCRATCH, int LDSW, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const int* __restrict__ in, int* __restrict__ out, int never) {
  __shared__ int s[LDSW ? LDSW : 1];
  const int v = in[0];
  if (v != never) return;  // (always taken: the rest only makes the compiler keep the resources)
  if constexpr (SCRATCH) {
    int priv[64];
    for (int i = 0; i < 64; ++i) priv[i] = in[i + threadIdx.x];
    if (LDSW) s[threadIdx.x] = priv[in[threadIdx.x] & 63];
    __syncthreads();
    out[threadIdx.x] = priv[in[threadIdx.x + 1] & 63];
  } else {
    if (LDSW) s[threadIdx.x] = in[threadIdx.x + 7];
    __syncthreads();
    out[threadIdx.x] = LDSW ? s[(threadIdx.x + 1) % THREADS] : 1;
  }
}
template <bool SCRATCH, int LDSW, int THREADS>
static void run(const char* name, const int* in, int* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  printf("%-34s", name);
  for (int g : {256, 768, 1024, 2048, 4096, 8192}) {
    for (int i = 0; i < 5; ++i) probe<SCRATCH, LDSW, THREADS><<<g, THREADS>>>(in, out, 12345);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 50; ++i) probe<SCRATCH, LDSW, THREADS><<<g, THREADS>>>(in, out, 12345);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("  %5d wg: %6.2f us", g, ms * 1000 / 50);
  }
  printf("\n");
}
int main() {
  int *in, *out;
  hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 20);
  hipMemset(in, 0, 1 << 20);
  run<false, 0, 256>("256 thr, no scratch, no LDS", in, out);
  run<true, 0, 256>("256 thr, scratch, no LDS", in, out);
  run<false, 6144, 256>("256 thr, no scratch, 24 KB LDS", in, out);
  run<true, 6144, 256>("256 thr, scratch, 24 KB LDS", in, out);
  run<false, 12288, 512>("512 thr, no scratch, 48 KB LDS", in, out);
  run<true, 12288, 512>("512 thr, scratch, 48 KB LDS", in, out);
  run<false, 0, 1024>("1024 thr, no scratch, no LDS", in, out);
  return 0;
}
