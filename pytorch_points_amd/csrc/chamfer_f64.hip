// chamfer_f64.hip -- nndistance forward / backward for double clouds.
//
// The reference dispatches its Chamfer kernels over the floating types (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// _ext/nmdistance_cuda.cu:125,193,210): `scalar_t` is the type of the coordinates, of the distances and of the
// gradients; the indices stay int.  fp32 is the tuned path of this library (chamfer.hip, chamfer_grid.hip); this
// file serves double with the same arithmetic as the reference's kernel instantiated for double,
//     d = t_0*t_0; d = fma(t_1, t_1, d); ...   (t_c = ref_c - query_c; nvcc contracts `d += tmp*tmp`, :31-35)
//     first minimum in index order (strict `<` inside a chunk :36, strict `>` between chunks :41)
//     g = graddist*2; v = g*(xa - xb): +v to the own row, -v to the matched row (:176-181)
// through an every-pair scan: one lane per query, the reference point wave-uniform (scalar loads), the query
// in registers for C <= 8.  A correctness path (0.4e12 pairs/s of fp64 compare-select), not a tuned one; half
// stays unsupported (the Python layer raises TypeError).
#include "pp_common.h"

namespace {

// CT > 0: compile-time point dimension, query in registers; CT == 0: any C, query re-read (L1) per pair
template <int CT>
__global__ __launch_bounds__(256) void nmdist_fwd_f64_kernel(const double* __restrict__ xyz1,
                                                             const double* __restrict__ xyz2,
                                                             double* __restrict__ dist1, int* __restrict__ idx1,
                                                             double* __restrict__ dist2, int* __restrict__ idx2,
                                                             int N, int M, int C, int tiles1, int tiles2) {
  const int c = CT > 0 ? CT : C;
  const int per_b = tiles1 + tiles2;
  const int b = blockIdx.x / per_b;
  const int r = blockIdx.x - b * per_b;
  const bool second = r >= tiles1;
  const int tile = second ? r - tiles1 : r;
  const int nq = second ? M : N;
  const int nr = second ? N : M;
  const double* __restrict__ qry = (second ? xyz2 : xyz1) + (size_t)b * nq * c;
  const double* __restrict__ ref = (second ? xyz1 : xyz2) + (size_t)b * nr * c;
  const int j = tile * 256 + threadIdx.x;
  if (j >= nq) return;
  const double* qp = qry + (size_t)j * c;
  double q[CT > 0 ? CT : 1];
  if (CT > 0) {
#pragma unroll
    for (int e = 0; e < CT; ++e) q[e] = qp[e];
  }
  double best = __builtin_inf();
  int bi = 0;
  for (int k = 0; k < nr; ++k) {
    const double* rp = ref + (size_t)k * c;  // wave-uniform
    double d = 0.0;
    if (CT > 0) {
#pragma unroll
      for (int e = 0; e < CT; ++e) {
        const double t = rp[e] - q[e];
        d = e == 0 ? t * t : __builtin_fma(t, t, d);
      }
    } else {
      for (int e = 0; e < c; ++e) {
        const double t = rp[e] - qp[e];
        d = e == 0 ? t * t : __builtin_fma(t, t, d);
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

__global__ __launch_bounds__(256) void fill_zero_f64_kernel(double* __restrict__ d, int* __restrict__ i, long long n) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    d[t] = 0.0;
    i[t] = 0;
  }
}

// Backward, two passes on one stream (the own-row pass overwrites: the reference's zero_() is folded in):
//   own:     gradA[j] = g * (xA[j] - xB[idxA[j]])      plain stores, one writer per row
//   scatter: gradB[idxA[j]] -= the same value           global fp64 atomics (global_atomic_add_f64)
template <bool SCATTER>
__global__ __launch_bounds__(256) void nmdist_bwd_f64_kernel(
    const double* __restrict__ xyz1, const double* __restrict__ xyz2, const double* __restrict__ gd1,
    const double* __restrict__ gd2, const int* __restrict__ idx1, const int* __restrict__ idx2,
    double* __restrict__ gx1, double* __restrict__ gx2, int N, int M, int C, long long total1, long long total2) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total1 + total2) return;
  const bool second = t >= total1;
  const long long p = second ? t - total1 : t;  // flat (b, j)
  const int na = second ? M : N, nb = second ? N : M;
  const long long b = p / na;
  const double* __restrict__ xa = (second ? xyz2 : xyz1) + p * C;
  const int j2 = (second ? idx2 : idx1)[p];
  double* __restrict__ ga = (second ? gx2 : gx1) + p * C;
  if (j2 < 0) {  // no neighbour (labeled variant, :175)
    if (!SCATTER)
      for (int e = 0; e < C; ++e) ga[e] = 0.0;
    return;
  }
  const double g = (second ? gd2 : gd1)[p] * 2;
  const double* __restrict__ xb = (second ? xyz1 : xyz2) + (b * nb + j2) * C;
  double* __restrict__ gb = (second ? gx1 : gx2) + (b * nb + j2) * C;
  for (int e = 0; e < C; ++e) {
    const double v = g * (xa[e] - xb[e]);
    if (SCATTER)
      unsafeAtomicAdd(gb + e, -v);
    else
      ga[e] = v;
  }
}

}  // namespace

extern "C" int pp_nmdistance_forward_f64(const double* xyz1, const double* xyz2, double* dist1, int* idx1,
                                         double* dist2, int* idx2, int B, int N, int M, int C, void* stream) {
  if (B < 0 || N < 0 || M < 0 || C < 1) return PP_EINVAL;
  if (B == 0 || (N == 0 && M == 0)) return PP_OK;
  if ((N > 0 && (!dist1 || !idx1)) || (M > 0 && (!dist2 || !idx2))) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0 || M == 0) {  // no pairs: zero-filled outputs, as pp_nmdistance_forward_f32
    const long long n = (long long)B * (N == 0 ? M : N);
    fill_zero_f64_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(N == 0 ? dist2 : dist1,
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
  nmdist_fwd_f64_kernel<CT><<<grid, block, 0, s>>>(xyz1, xyz2, dist1, idx1, dist2, idx2, N, M, C, tiles1, tiles2)
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

extern "C" int pp_nmdistance_backward_f64(const double* xyz1, const double* xyz2, const double* graddist1,
                                          const double* graddist2, const int* idx1, const int* idx2,
                                          double* gradxyz1, double* gradxyz2, int B, int N, int M, int C,
                                          void* stream) {
  if (B < 0 || N < 0 || M < 0 || C < 1) return PP_EINVAL;
  const long long t1 = (long long)B * N, t2 = (long long)B * M;
  if (t1 + t2 == 0) return PP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0 || M == 0) {  // no pairs: gradients are zero
    double* g = N == 0 ? gradxyz2 : gradxyz1;
    if (!g) return PP_EINVAL;
    return (int)hipMemsetAsync(g, 0, (size_t)((N == 0 ? t2 : t1) * C) * sizeof(double), s);
  }
  if (!xyz1 || !xyz2 || !graddist1 || !graddist2 || !idx1 || !idx2 || !gradxyz1 || !gradxyz2) return PP_EINVAL;
  const long long blocks = (t1 + t2 + 255) / 256;
  if (blocks > 0x7fffff00LL) return PP_EINVAL;
  nmdist_bwd_f64_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
      xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, C, t1, t2);
  PP_RETURN_IF_LAUNCH_FAILED();
  nmdist_bwd_f64_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
      xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, C, t1, t2);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
