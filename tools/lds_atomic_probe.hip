// lds_atomic_probe.hip -- LDS throughput per CU for random-address ds_read_b32, ds_add_f32,
// ds_add_u32 (no return) and ds_add_rtn_f32, 16 waves per CU, 64 KiB region.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned seed) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < 16384; i += 1024) s[i] = 0.f;
  __syncthreads();
  if (KIND == 5) __builtin_amdgcn_s_setreg((1 | (4 << 6) | (1 << 11)), 0);  // MODE[5:4] (fp32 denorm) = 0: flush in and out
  unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  float acc = 0;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned a = (x >> 10) & 16383u;
    if (KIND == 0) acc += s[a];
    else if (KIND == 1) atomicAdd(&s[a], 1.0f);
    else if (KIND == 2) atomicAdd(reinterpret_cast<unsigned*>(s) + a, 1u);
    else if (KIND == 3) acc += atomicAdd(&s[a], 1.0f);
    else if (KIND == 4) s[a] = acc + i;   // plain scattered store
    else if (KIND == 5) atomicAdd(&s[a], 1.0f);   // with fp32 denormals flushed (MODE set below)
    else if (KIND == 6) atomicAdd(reinterpret_cast<unsigned long long*>(s) + (a >> 1), 1ull);
    else if (KIND == 7) atomicAdd(reinterpret_cast<double*>(s) + (a >> 1), 1.0);
    else if (KIND == 8) atomicMax(reinterpret_cast<int*>(s) + a, (int)i);
  }
  __syncthreads();
  out[blockIdx.x * 1024 + threadIdx.x] = acc + s[threadIdx.x];
}
template <int KIND> void run(const char* name) {
  float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4096;
  k<KIND><<<256, 1024, 65536>>>(out, 16, 1);
  (void)hipEventRecord(e0);
  k<KIND><<<256, 1024, 65536>>>(out, iters, 1);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double lanes = 256.0 * 1024 * iters;
  printf("%-18s %.3f ms  %.2f G lane-ops/s chip  = %.2f lanes/clk/CU at 2.2 GHz\n", name, ms, lanes / ms / 1e6, lanes / 256 / (ms * 1e-3 * 2.2e9));
}
int main() {
  run<0>("ds_read_b32"); run<1>("ds_add_f32"); run<2>("ds_add_u32"); run<3>("ds_add_rtn_f32"); run<4>("ds_write_b32");
  run<5>("ds_add_f32 ftz"); run<6>("ds_add_u64"); run<7>("ds_add_f64"); run<8>("ds_max_i32");
  return 0;
}
