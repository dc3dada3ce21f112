// fps.hip -- iterative furthest point sampling for gfx950.
// Replaces the reference's furthest_point_sampling_forward_kernel + launcher
// (_ext/sampling_cuda.cu:162-233, 235-325).
//
// Semantics: idx[b,0] = seed; for j = 1..npoint-1, with `old` the last
// pick: every point k gets d2 = min(dist3(x_k, x_old), temp[k]), temp[k] = d2, and the next pick
// is the point with the largest d2.  The reference resolves exact ties through its thread
// decomposition (T = opt_n_threads(N) threads, thread t owns k = t mod T in ascending order, strict
// '>', then a binary tree in which the lower slot wins): the winner among equal d2 is the point
// with the smallest (k mod T), then the smallest k.  That order is reproduced here for ANY
// decomposition by reducing a 64-bit key
//     key(k) = float_bits(d2) << 32 | (0xFFFFFFFF - ((k mod T) * ceil(N/T) + k div T))
// with an unsigned max: d2 >= 0, so its bit pattern orders like its value.
//
// v1 decomposition: one 1024-thread workgroup per batch element, thread t owns k = t + 1024*i;
// the running minima (temp) live in registers for the whole call (read once, written once), the
// coordinates are re-read from L2 every step, one barrier per step (double-buffered LDS slots).
#include "pp_common.h"

namespace {

using pp::dist3;

constexpr int kFpsThreads = 1024;
constexpr int kFpsWaves = kFpsThreads / 64;

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}

struct TieOrder {
  int t_mask;   // T - 1
  int t_shift;  // log2(T)
  int rows;     // ceil(N / T)
  __device__ __forceinline__ unsigned rank(int k) const {
    return (unsigned)((k & t_mask) * rows + (k >> t_shift));
  }
  __device__ __forceinline__ int unrank(unsigned r) const {
    return (int)((r % (unsigned)rows) << t_shift) + (int)(r / (unsigned)rows);
  }
};

// R > 0: temp of this thread's R points in registers.  R == 0: temp stays in global memory
// (any N); slower, only used when N > 1024 * 64.
template <int R>
__global__ __launch_bounds__(kFpsThreads) void fps_block_kernel(const float* __restrict__ xyz,
                                                                float* __restrict__ temp,
                                                                int* __restrict__ idx, int N,
                                                                int npoint, int seed,
                                                                TieOrder order) {
  __shared__ unsigned long long s_key[2][kFpsWaves];
  const int b = blockIdx.x;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  float* __restrict__ tmp = temp + (size_t)b * N;
  int* __restrict__ out = idx + (size_t)b * npoint;
  const int t = threadIdx.x;
  const int wave = pp::wave_id_uniform();

  float td[R > 0 ? R : 1];
  if (R > 0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int k = t + kFpsThreads * i;
      td[i] = k < N ? tmp[k] : 0.0f;
    }
  }
  int old = seed;
  if (t == 0) out[0] = old;
  for (int j = 1; j < npoint; ++j) {
    const float ox = p[3 * (size_t)old + 0], oy = p[3 * (size_t)old + 1], oz = p[3 * (size_t)old + 2];
    unsigned long long best = 0ull;
    if (R > 0) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int k = t + kFpsThreads * i;
        if (k < N) {
          const float d = dist3(p[3 * (size_t)k], p[3 * (size_t)k + 1], p[3 * (size_t)k + 2], ox, oy, oz);
          const float d2 = __builtin_fminf(d, td[i]);
          td[i] = d2;
          const unsigned long long key =
              ((unsigned long long)__float_as_uint(d2) << 32) | (0xFFFFFFFFu - order.rank(k));
          best = key > best ? key : best;
        }
      }
    } else {
      for (int k = t; k < N; k += kFpsThreads) {
        const float d = dist3(p[3 * (size_t)k], p[3 * (size_t)k + 1], p[3 * (size_t)k + 2], ox, oy, oz);
        const float tdv = tmp[k];
        const float d2 = __builtin_fminf(d, tdv);
        if (d2 != tdv) tmp[k] = d2;
        const unsigned long long key =
            ((unsigned long long)__float_as_uint(d2) << 32) | (0xFFFFFFFFu - order.rank(k));
        best = key > best ? key : best;
      }
    }
    best = wave_max_u64(best);
    if ((t & 63) == 0) s_key[j & 1][wave] = best;
    __syncthreads();
    unsigned long long m = s_key[j & 1][0];
#pragma unroll
    for (int w = 1; w < kFpsWaves; ++w) {
      const unsigned long long o = s_key[j & 1][w];
      m = o > m ? o : m;
    }
    old = __builtin_amdgcn_readfirstlane(order.unrank(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFull)));
    if (t == 0) out[j] = old;
  }
  if (R > 0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int k = t + kFpsThreads * i;
      if (k < N) tmp[k] = td[i];
    }
  }
}

template <int R>
void launch_fps(const float* xyz, float* temp, int* idx, int B, int N, int npoint, int seed,
                TieOrder order, hipStream_t s) {
  fps_block_kernel<R><<<dim3(B), dim3(kFpsThreads), 0, s>>>(xyz, temp, idx, N, npoint, seed, order);
}

}  // namespace

extern "C" size_t pp_furthest_sampling_workspace_bytes(int B, int N, int npoint) {
  (void)B; (void)N; (void)npoint;
  return 0;
}

extern "C" int pp_furthest_sampling_f32(const float* xyz, float* temp, int* idx, int B, int N,
                                        int npoint, int seed_idx, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  if (B < 0 || N < 0 || npoint < 0) return PP_EINVAL;
  if (B == 0 || npoint <= 0) return PP_OK;  // ref: `if (m <= 0) return;` (sampling_cuda.cu:166)
  if (N == 0 || !xyz || !temp || !idx) return PP_EINVAL;
  if (seed_idx < 0 || seed_idx >= N) return PP_EINVAL;  // the reference would read out of bounds
  hipStream_t s = (hipStream_t)stream;
  const int T = pp_opt_n_threads(N);
  TieOrder order;
  order.t_mask = T - 1;
  order.t_shift = __builtin_ctz((unsigned)T);
  order.rows = (N + T - 1) / T;
  if ((long long)T * order.rows > 0xFFFFFFFELL) return PP_EINVAL;
  const int per_thread = (N + kFpsThreads - 1) / kFpsThreads;
  if (per_thread <= 1) launch_fps<1>(xyz, temp, idx, B, N, npoint, seed_idx, order, s);
  else if (per_thread <= 2) launch_fps<2>(xyz, temp, idx, B, N, npoint, seed_idx, order, s);
  else if (per_thread <= 4) launch_fps<4>(xyz, temp, idx, B, N, npoint, seed_idx, order, s);
  else if (per_thread <= 8) launch_fps<8>(xyz, temp, idx, B, N, npoint, seed_idx, order, s);
  else if (per_thread <= 16) launch_fps<16>(xyz, temp, idx, B, N, npoint, seed_idx, order, s);
  else if (per_thread <= 32) launch_fps<32>(xyz, temp, idx, B, N, npoint, seed_idx, order, s);
  else if (per_thread <= 64) launch_fps<64>(xyz, temp, idx, B, N, npoint, seed_idx, order, s);
  else launch_fps<0>(xyz, temp, idx, B, N, npoint, seed_idx, order, s);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
