// group_probe.hip -- ablation of the LDS-staged group_points kernel at config 4
// (B=32 C=128 N=16384 P=262144): which phase holds it below the HBM write roof?
// MODE 0 full; 1 no LDS gather (stores only + row staging); 2 no stores (gather + staging);
// 3 no row staging (gather from a once-filled LDS + stores); 4 stores only, no staging no gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
constexpr int V = 8, KR = 8;
template <int MODE, int THREADS, int VV>
__global__ __launch_bounds__(THREADS) void k(const float* __restrict__ points, const int* __restrict__ idx,
                                         float* __restrict__ out, int B, int C, int N, long long P, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float s_row[];
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / chunks), chunk = y % chunks;
  if (b >= B) return;
  const int t = threadIdx.x;
  const long long p0 = (long long)chunk * (THREADS * 4 * VV) + t * 4;
  i4 ii[VV];
#pragma unroll
  for (int v = 0; v < VV; ++v) ii[v] = *reinterpret_cast<const i4*>(idx + (size_t)b * P + p0 + (long long)v * THREADS * 4);
  const int n4 = N >> 2;
  const f4* __restrict__ row = reinterpret_cast<const f4*>(points + (size_t)b * C * N);
  constexpr int KK = 4096 / THREADS;
  f4 pre[KK];
  int ee[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) { ee[kk] = min(t + THREADS * kk, n4 - 1); pre[kk] = row[ee[kk]]; }
  f4 acc = {0, 0, 0, 0};
  for (int c = 0; c < C; ++c) {
    if (MODE != 3 && MODE != 4) {
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) reinterpret_cast<f4*>(s_row)[ee[kk]] = pre[kk];
      __syncthreads();
      const f4* __restrict__ nrow = row + (size_t)(c + 1 < C ? c + 1 : c) * n4;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) pre[kk] = nrow[ee[kk]];
    }
    float* __restrict__ o = out + ((size_t)b * C + c) * P;
#pragma unroll
    for (int v = 0; v < VV; ++v) {
      f4 r;
      if (MODE == 1 || MODE == 4) { r = pre[v % KK]; r.x += c; }
      else { r.x = s_row[ii[v].x]; r.y = s_row[ii[v].y]; r.z = s_row[ii[v].z]; r.w = s_row[ii[v].w]; }
      if (MODE == 2) acc += r;
      else *reinterpret_cast<f4*>(o + p0 + (long long)v * THREADS * 4) = r;
    }
  }
  if (MODE == 2) *reinterpret_cast<f4*>(out + (size_t)b * C * P + p0) = acc;
}
template <int MODE, int THREADS, int VV>
void run(const char* name, const float* pts, const int* idx, float* out, int B, int C, int N, long long P) {
  const long long per_block = (long long)THREADS * 4 * VV;
  const int chunks = (int)(P / per_block);
  const int blocks = 8 * ((B + 7) / 8) * chunks;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<MODE, THREADS, VV>, hipFuncAttributeMaxDynamicSharedMemorySize, N * 4);
  k<MODE, THREADS, VV><<<blocks, THREADS, N * 4>>>(pts, idx, out, B, C, N, P, chunks);
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0);
    k<MODE, THREADS, VV><<<blocks, THREADS, N * 4>>>(pts, idx, out, B, C, N, P, chunks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%-44s threads=%4d V=%2d blocks=%5d  %.3f ms  (%.2f TB/s of output)\n", name, THREADS, VV, blocks, best,
         4.0 * B * C * P / best / 1e9);
}

// ---- DMA-ring form (the shipped v3) with ablation modes: 0 full, 1 no gather, 2 no stores, 3 no DMA
template <int MODE, int VV>
__global__ __launch_bounds__(1024) void kd(const float* __restrict__ points, const int* __restrict__ idx,
                                           float* __restrict__ out, int B, int C, int N, long long P, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float s_ring[];
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / chunks), chunk = y % chunks;
  if (b >= B) return;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const long long p0 = (long long)chunk * (1024 * 4 * VV) + t * 4;
  i4 ii[VV];
#pragma unroll
  for (int v = 0; v < VV; ++v) ii[v] = *reinterpret_cast<const i4*>(idx + (size_t)b * P + p0 + (long long)v * 4096);
  const int n4 = N >> 2;
  const int passes = n4 / 1024;
  const f4* __restrict__ row0 = reinterpret_cast<const f4*>(points + (size_t)b * C * N);
  auto issue_row = [&](int c, int slot) {
    if (MODE == 3) return;
    const f4* __restrict__ row = row0 + (size_t)c * n4;
    for (int k = 0; k < passes; ++k) {
      float* dst = s_ring + (size_t)slot * N + (size_t)(k * 1024 + wave * 64) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + k * 1024 + t),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  issue_row(0, 0);
  f4 acc = {0, 0, 0, 0};
  for (int c = 0; c < C; ++c) {
    if (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(VV) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (c + 1 < C) issue_row(c + 1, (c + 1) & 1);
    const float* cur = s_ring + (size_t)(c & 1) * N;
    float* __restrict__ o = out + ((size_t)b * C + c) * P;
#pragma unroll
    for (int v = 0; v < VV; ++v) {
      f4 r;
      if (MODE == 1) { r.x = c; r.y = t; r.z = v; r.w = 1; }
      else { r.x = cur[ii[v].x]; r.y = cur[ii[v].y]; r.z = cur[ii[v].z]; r.w = cur[ii[v].w]; }
      if (MODE == 2) acc += r;
      else *reinterpret_cast<f4*>(o + p0 + (long long)v * 4096) = r;
    }
  }
  if (MODE == 2) *reinterpret_cast<f4*>(out + (size_t)b * C * P + p0) = acc;
}
template <int MODE, int VV>
void rund(const char* name, const float* pts, const int* idx, float* out, int B, int C, int N, long long P) {
  const int chunks = (int)(P / (1024LL * 4 * VV));
  const int blocks = 8 * ((B + 7) / 8) * chunks;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)kd<MODE, VV>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * N * 4);
  kd<MODE, VV><<<blocks, 1024, 2 * N * 4>>>(pts, idx, out, B, C, N, P, chunks);
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0);
    kd<MODE, VV><<<blocks, 1024, 2 * N * 4>>>(pts, idx, out, B, C, N, P, chunks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("DMA ring %-34s V=%2d blocks=%5d  %.3f ms  (%.2f TB/s of output)\n", name, VV, blocks, best, 4.0 * B * C * P / best / 1e9);
}
int main() {
  const int B = 32, C = 128, N = 16384; const long long P = 262144;
  float *pts, *out; int* idx;
  (void)hipMalloc(&pts, sizeof(float) * B * C * N); (void)hipMalloc(&out, sizeof(float) * B * C * P); (void)hipMalloc(&idx, sizeof(int) * B * P);
  std::vector<int> h(B * P); srand(1);
  for (auto& v : h) v = rand() % N;
  (void)hipMemcpy(idx, h.data(), sizeof(int) * B * P, hipMemcpyHostToDevice);
  (void)hipMemset(pts, 0, sizeof(float) * B * C * N);
  rund<0, 8>("full", pts, idx, out, B, C, N, P);
  rund<1, 8>("no gather (DMA + stores)", pts, idx, out, B, C, N, P);
  rund<2, 8>("no stores (DMA + gather)", pts, idx, out, B, C, N, P);
  rund<3, 8>("no DMA (gather + stores)", pts, idx, out, B, C, N, P);
  rund<0, 4>("full", pts, idx, out, B, C, N, P);
  run<0, 512, 8>("full", pts, idx, out, B, C, N, P);
  run<1, 512, 8>("no LDS gather (staging + stores)", pts, idx, out, B, C, N, P);
  run<2, 512, 8>("no stores (staging + gather)", pts, idx, out, B, C, N, P);
  run<3, 512, 8>("no staging (gather + stores)", pts, idx, out, B, C, N, P);
  run<4, 512, 8>("stores only", pts, idx, out, B, C, N, P);
  run<0, 1024, 8>("full", pts, idx, out, B, C, N, P);
  run<0, 1024, 4>("full", pts, idx, out, B, C, N, P);
  run<0, 256, 8>("full", pts, idx, out, B, C, N, P);
  run<0, 256, 16>("full", pts, idx, out, B, C, N, P);
  run<4, 256, 8>("stores only", pts, idx, out, B, C, N, P);
  run<4, 1024, 8>("stores only", pts, idx, out, B, C, N, P);
  return 0;
}
