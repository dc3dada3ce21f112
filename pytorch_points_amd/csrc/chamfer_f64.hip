// chamfer_f64.hip -- nndistance forward / backward for double and half clouds.
//
// The reference dispatches its Chamfer kernels over the floating types (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// _ext/nmdistance_cuda.cu:125,193,210): `scalar_t` is the type of the coordinates, of the distances and of the
// gradients; the indices stay int.  fp32 is the tuned path of this library (chamfer.hip, chamfer_grid.hip); this
// file serves double with the same arithmetic as the reference's kernel instantiated for double,
//     d = t_0*t_0; d = fma(t_1, t_1, d); ...   (t_c = ref_c - query_c; nvcc contracts `d += tmp*tmp`, :31-35)
//     first minimum in index order (strict `<` inside a chunk :36, strict `>` between chunks :41)
//     g = graddist*2; v = g*(xa - xb): +v to the own row, -v to the matched row (:176-181)
// through an every-pair scan: one lane per query, the reference point wave-uniform (scalar loads), the query
// in registers for C <= 8.  A correctness path (0.4e12 pairs/s of fp64 compare-select), not a tuned one.
//
// Round 4: the same kernels instantiated for HALF (`scalar_t = at::Half`, the third type of the reference's dispatch).
// c10::Half has no fused arithmetic: every operator converts to float, computes and rounds the result back to half
// (c10/util/Half-inl.h), so `tmp = buf - xyz; d += tmp * tmp` is three separately rounded half operations per
// coordinate -- a float product of two halves is exact and a float sum of two halves rounds to the same half as the
// exact sum (24 >= 2 * 11 + 2 bits), so native v_sub_f16 / v_mul_f16 / v_add_f16 (this file is compiled with
// -ffp-contract=off) give the same bits.  Comparisons in half, distances stored as half, indices int.  Backward:
// g = graddist * 2, v = g * (xa - xb), each rounded to half (:176-181); own rows by plain stores, scattered terms by
// packed half atomics (global_atomic_pk_add_f16: rounded per addition, in arrival order -- as the reference's
// CAS-loop atomicAdd on at::Half is).
#include <hip/hip_fp16.h>

#include "pp_common.h"

namespace {

// CT > 0: compile-time point dimension, query in registers; CT == 0: any C, query re-read (L1) per pair
// d <- d + t * t in the reference's arithmetic for the type: double contracts to a fused multiply-add (nvcc), half
// cannot (see above)
__device__ __forceinline__ double sq_acc(double d, double t, bool first) { return first ? t * t : __builtin_fma(t, t, d); }
__device__ __forceinline__ _Float16 sq_acc(_Float16 d, _Float16 t, bool first) {
  const _Float16 p = t * t;
  return first ? p : (_Float16)(d + p);
}
template <typename T>
__device__ __forceinline__ T type_inf() {
  return (T)__builtin_inff();
}

template <typename T, int CT>
__global__ __launch_bounds__(256) void nmdist_fwd_f64_kernel(const T* __restrict__ xyz1,
                                                             const T* __restrict__ xyz2,
                                                             T* __restrict__ dist1, int* __restrict__ idx1,
                                                             T* __restrict__ dist2, int* __restrict__ idx2,
                                                             int N, int M, int C, int tiles1, int tiles2) {
  const int c = CT > 0 ? CT : C;
  const int per_b = tiles1 + tiles2;
  const int b = blockIdx.x / per_b;
  const int r = blockIdx.x - b * per_b;
  const bool second = r >= tiles1;
  const int tile = second ? r - tiles1 : r;
  const int nq = second ? M : N;
  const int nr = second ? N : M;
  const T* __restrict__ qry = (second ? xyz2 : xyz1) + (size_t)b * nq * c;
  const T* __restrict__ ref = (second ? xyz1 : xyz2) + (size_t)b * nr * c;
  const int j = tile * 256 + threadIdx.x;
  if (j >= nq) return;
  const T* qp = qry + (size_t)j * c;
  T q[CT > 0 ? CT : 1];
  if (CT > 0) {
#pragma unroll
    for (int e = 0; e < CT; ++e) q[e] = qp[e];
  }
  T best = type_inf<T>();
  int bi = 0;
  for (int k = 0; k < nr; ++k) {
    const T* rp = ref + (size_t)k * c;  // wave-uniform
    T d = (T)0;
    if (CT > 0) {
#pragma unroll
      for (int e = 0; e < CT; ++e) {
        const T t = (T)(rp[e] - q[e]);
        d = sq_acc(d, t, e == 0);
      }
    } else {
      for (int e = 0; e < c; ++e) {
        const T t = (T)(rp[e] - qp[e]);
        d = sq_acc(d, t, e == 0);
      }
    }
    // k == 0 always takes (a NaN distance of the first point is kept, as `k==0 || d<best` does, :36)
    const bool lt = (k == 0) | (d < best);
    best = lt ? d : best;
    bi = lt ? k : bi;
  }
  ((second ? dist2 : dist1) + (size_t)b * nq)[j] = best;
  ((second ? idx2 : idx1) + (size_t)b * nq)[j] = bi;
}

template <typename T>
__global__ __launch_bounds__(256) void fill_zero_f64_kernel(T* __restrict__ d, int* __restrict__ i, long long n) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    d[t] = (T)0;
    i[t] = 0;
  }
}

// gb[e] -= v for one element of a row of the other cloud
__device__ __forceinline__ void scatter_sub(double* p, double v) { unsafeAtomicAdd(p, -v); }
__device__ __forceinline__ void scatter_sub(_Float16* p, _Float16 v) {
  // the aligned pair of halves that holds *p, the other half of the pair adds +0 (x + 0 = x in every rounding)
  const bool odd = ((size_t)p & 2) != 0;
  __half2* pair = reinterpret_cast<__half2*>(reinterpret_cast<char*>(p) - (odd ? 2 : 0));
  const __half nv = __float2half(-(float)v), z = __float2half(0.0f);
  unsafeAtomicAdd(pair, odd ? __halves2half2(z, nv) : __halves2half2(nv, z));
}

// Backward, two passes on one stream (the own-row pass overwrites: the reference's zero_() is folded in):
//   own:     gradA[j] = g * (xA[j] - xB[idxA[j]])      plain stores, one writer per row
//   scatter: gradB[idxA[j]] -= the same value           global fp64 atomics (global_atomic_add_f64)
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void nmdist_bwd_f64_kernel(
    const T* __restrict__ xyz1, const T* __restrict__ xyz2, const T* __restrict__ gd1,
    const T* __restrict__ gd2, const int* __restrict__ idx1, const int* __restrict__ idx2,
    T* __restrict__ gx1, T* __restrict__ gx2, int N, int M, int C, long long total1, long long total2) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total1 + total2) return;
  const bool second = t >= total1;
  const long long p = second ? t - total1 : t;  // flat (b, j)
  const int na = second ? M : N, nb = second ? N : M;
  const long long b = p / na;
  const T* __restrict__ xa = (second ? xyz2 : xyz1) + p * C;
  const int j2 = (second ? idx2 : idx1)[p];
  T* __restrict__ ga = (second ? gx2 : gx1) + p * C;
  if (j2 < 0) {  // no neighbour (labeled variant, :175)
    if (!SCATTER)
      for (int e = 0; e < C; ++e) ga[e] = (T)0;
    return;
  }
  const T g = (T)((second ? gd2 : gd1)[p] * (T)2);
  const T* __restrict__ xb = (second ? xyz1 : xyz2) + (b * nb + j2) * C;
  T* __restrict__ gb = (second ? gx1 : gx2) + (b * nb + j2) * C;
  for (int e = 0; e < C; ++e) {
    const T diff = (T)(xa[e] - xb[e]);
    const T v = (T)(g * diff);
    if (SCATTER)
      scatter_sub(gb + e, v);
    else
      ga[e] = v;
  }
}

}  // namespace

template <typename T>
static int typed_forward(const T* xyz1, const T* xyz2, T* dist1, int* idx1, T* dist2, int* idx2, int B, int N, int M, int C,
                         void* stream) {
  if (B < 0 || N < 0 || M < 0 || C < 1) return PP_EINVAL;
  if (B == 0 || (N == 0 && M == 0)) return PP_OK;
  if ((N > 0 && (!dist1 || !idx1)) || (M > 0 && (!dist2 || !idx2))) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0 || M == 0) {  // no pairs: zero-filled outputs, as pp_nmdistance_forward_f32
    const long long n = (long long)B * (N == 0 ? M : N);
    fill_zero_f64_kernel<T><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(N == 0 ? dist2 : dist1,
                                                                                     N == 0 ? idx2 : idx1, n);
    PP_RETURN_IF_LAUNCH_FAILED();
    return PP_OK;
  }
  if (!xyz1 || !xyz2) return PP_EINVAL;
  const int tiles1 = (N + 255) / 256, tiles2 = (M + 255) / 256;
  const long long total = (long long)B * (tiles1 + tiles2);
  if (total > 0x7fffff00LL) return PP_EINVAL;
  const dim3 grid((unsigned)total), block(256);
#define PP_F64_FWD(CT) \
  nmdist_fwd_f64_kernel<T, CT><<<grid, block, 0, s>>>(xyz1, xyz2, dist1, idx1, dist2, idx2, N, M, C, tiles1, tiles2)
  switch (C) {
    case 1: PP_F64_FWD(1); break;
    case 2: PP_F64_FWD(2); break;
    case 3: PP_F64_FWD(3); break;
    case 4: PP_F64_FWD(4); break;
    case 6: PP_F64_FWD(6); break;
    case 8: PP_F64_FWD(8); break;
    default: PP_F64_FWD(0); break;
  }
#undef PP_F64_FWD
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

template <typename T>
static int typed_backward(const T* xyz1, const T* xyz2, const T* graddist1, const T* graddist2, const int* idx1,
                          const int* idx2, T* gradxyz1, T* gradxyz2, int B, int N, int M, int C, void* stream) {
  if (B < 0 || N < 0 || M < 0 || C < 1) return PP_EINVAL;
  const long long t1 = (long long)B * N, t2 = (long long)B * M;
  if (t1 + t2 == 0) return PP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0 || M == 0) {  // no pairs: gradients are zero
    T* g = N == 0 ? gradxyz2 : gradxyz1;
    if (!g) return PP_EINVAL;
    return (int)hipMemsetAsync(g, 0, (size_t)((N == 0 ? t2 : t1) * C) * sizeof(T), s);
  }
  if (!xyz1 || !xyz2 || !graddist1 || !graddist2 || !idx1 || !idx2 || !gradxyz1 || !gradxyz2) return PP_EINVAL;
  const long long blocks = (t1 + t2 + 255) / 256;
  if (blocks > 0x7fffff00LL) return PP_EINVAL;
  nmdist_bwd_f64_kernel<T, false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
      xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, C, t1, t2);
  PP_RETURN_IF_LAUNCH_FAILED();
  nmdist_bwd_f64_kernel<T, true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
      xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, C, t1, t2);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_nmdistance_forward_f64(const double* xyz1, const double* xyz2, double* dist1, int* idx1,
                                         double* dist2, int* idx2, int B, int N, int M, int C, void* stream) {
  return typed_forward<double>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, stream);
}

extern "C" int pp_nmdistance_backward_f64(const double* xyz1, const double* xyz2, const double* graddist1,
                                          const double* graddist2, const int* idx1, const int* idx2,
                                          double* gradxyz1, double* gradxyz2, int B, int N, int M, int C,
                                          void* stream) {
  return typed_backward<double>(xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, B, N, M, C, stream);
}

// half: IEEE binary16 words (torch.float16); opaque pointers in the C ABI
extern "C" int pp_nmdistance_forward_f16(const void* xyz1, const void* xyz2, void* dist1, int* idx1, void* dist2,
                                         int* idx2, int B, int N, int M, int C, void* stream) {
  return typed_forward<_Float16>((const _Float16*)xyz1, (const _Float16*)xyz2, (_Float16*)dist1, idx1, (_Float16*)dist2, idx2,
                                 B, N, M, C, stream);
}

extern "C" int pp_nmdistance_backward_f16(const void* xyz1, const void* xyz2, const void* graddist1,
                                          const void* graddist2, const int* idx1, const int* idx2, void* gradxyz1,
                                          void* gradxyz2, int B, int N, int M, int C, void* stream) {
  return typed_backward<_Float16>((const _Float16*)xyz1, (const _Float16*)xyz2, (const _Float16*)graddist1,
                                  (const _Float16*)graddist2, idx1, idx2, (_Float16*)gradxyz1, (_Float16*)gradxyz2, B, N, M, C,
                                  stream);
}
