// sampling.hip -- gather_points, ball_query, group_points, three_nn, three_interpolate for gfx950.
// Replaces the reference's _ext/sampling_cuda.cu (gather :9-84, ball_query :340-397,
// group_points :447-513) and _ext/interpolate_gpu.cu (three_nn :9-74, three_interpolate :77-160).
#include "pp_common.h"

namespace {

using pp::dist3;

// ------------------------------------------------------------------------------------------------
// gather_points: out[b,c,m] = points[b,c,idx[b,m]]            (ref sampling_cuda.cu:9-25)
// One thread per (b, m) column quad walks all channels: idx is read once, stores are contiguous.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_fwd_kernel(const float* __restrict__ points,
                                                         const int* __restrict__ idx,
                                                         float* __restrict__ out, int C, int N,
                                                         int M, int c_per_block) {
  const int b = blockIdx.z;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int i = idx[(size_t)b * M + m];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c)
    out[((size_t)b * C + c) * M + m] = points[((size_t)b * C + c) * N + i];
}

// gather backward: grad_points[b,c,idx[b,m]] += grad_out[b,c,m]   (ref sampling_cuda.cu:47-64)
__global__ __launch_bounds__(256) void gather_bwd_kernel(const float* __restrict__ grad_out,
                                                         const int* __restrict__ idx,
                                                         float* __restrict__ grad_points, int C,
                                                         int N, int M, int c_per_block) {
  const int b = blockIdx.z;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int i = idx[(size_t)b * M + m];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c)
    atomicAdd(grad_points + ((size_t)b * C + c) * N + i, grad_out[((size_t)b * C + c) * M + m]);
}

// ------------------------------------------------------------------------------------------------
// ball_query (ref sampling_cuda.cu:340-376): per centre the first `nsample` indices k (ascending)
// with dist3 < radius^2 (strict), padded with the first hit; rows with no hit are all zero.
//
// One lane per centre; the scanned point is wave-uniform (scalar loads -> SGPR operands), so the
// per-pair cost is 3 sub + 1 mul + 2 fma + 1 compare and a scalar branch over the rare hit path.
// Each wave stages its 64 rows (64 x nsample ints, contiguous in the output) in LDS and writes
// them out once with fully coalesced stores, including the padding -- the output needs no
// pre-zeroing.  The wave leaves the scan as soon as all of its centres are full.
// ------------------------------------------------------------------------------------------------
constexpr int kBqGroup = 8;

template <bool STAGE_LDS>
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ new_xyz,
                                                         const float* __restrict__ xyz,
                                                         int* __restrict__ idx, int N, int M,
                                                         float radius2, int nsample,
                                                         int tiles_per_b) {
  extern __shared__ __attribute__((aligned(16))) int s_rows[];  // [4 waves][64][nsample]
  const int b = blockIdx.x / tiles_per_b;
  const int tile = blockIdx.x - b * tiles_per_b;
  const int wave = pp::wave_id_uniform();
  const int lane = threadIdx.x & 63;
  const int m0 = tile * 256 + wave * 64;  // first centre of this wave
  if (m0 >= M) return;                    // wave-uniform
  const int m = m0 + lane;
  const bool valid = m < M;
  const int mc = valid ? m : M - 1;
  const float* __restrict__ q = new_xyz + ((size_t)b * M + mc) * 3;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  int* __restrict__ grow = idx + ((size_t)b * M + m) * nsample;  // this lane's output row
  int* srow = s_rows + ((size_t)wave * 64 + lane) * nsample;

  int cnt = valid ? 0 : nsample;  // out-of-range lanes count as full
  int first = 0;
  auto hit = [&](int k) {
    if (cnt == 0) first = k;
    if (STAGE_LDS)
      srow[cnt] = k;
    else
      grow[cnt] = k;
    ++cnt;
  };
  const int ngroups = N / kBqGroup;
  int k = 0;
  for (int g = 0; g < ngroups; ++g, k += kBqGroup) {
    if (__all(cnt >= nsample)) break;
    const float* __restrict__ rp = p + (size_t)k * 3;  // wave-uniform
    float rr[kBqGroup * 3];
#pragma unroll
    for (int e = 0; e < kBqGroup * 3; ++e) rr[e] = rp[e];
    float d[kBqGroup];
    bool any = false;
#pragma unroll
    for (int e = 0; e < kBqGroup; ++e) {
      d[e] = dist3(qx, qy, qz, rr[3 * e], rr[3 * e + 1], rr[3 * e + 2]);
      any |= d[e] < radius2;
    }
    if (__any(any)) {
#pragma unroll
      for (int e = 0; e < kBqGroup; ++e)
        if (d[e] < radius2 && cnt < nsample) hit(k + e);
    }
  }
  if (!__all(cnt >= nsample)) {
    for (k = ngroups * kBqGroup; k < N; ++k) {
      const float d = dist3(qx, qy, qz, p[3 * (size_t)k], p[3 * (size_t)k + 1], p[3 * (size_t)k + 2]);
      if (d < radius2 && cnt < nsample) hit(k);
    }
  }

  if (STAGE_LDS) {
    // rows of the wave's 64 centres are one contiguous run of 64*nsample ints in the output
    const int* wrows = s_rows + (size_t)wave * 64 * nsample;
    int* __restrict__ gout = idx + ((size_t)b * M + m0) * nsample;
    const int nrows = min(64, M - m0);
    const int total = nrows * nsample;
    for (int f = lane; f < total; f += 64) {
      const int row = f / nsample;
      const int slot = f - row * nsample;
      const int rc = __shfl(cnt, row);
      const int rf = __shfl(first, row);
      // slot < rc: a recorded hit; otherwise the pad (first hit, or 0 when the ball is empty)
      gout[f] = slot < rc ? wrows[f] : rf;
    }
  } else if (valid) {
    for (int s = cnt; s < nsample; ++s) grow[s] = first;
  }
}

// ------------------------------------------------------------------------------------------------
// group_points: out[b,c,j,k] = points[b,c,idx[b,j,k]]        (ref sampling_cuda.cu:447-467)
// A thread owns 4 consecutive (j,k) positions: one 16-byte idx load, then per channel four
// gathers and one 16-byte store.  The reference launches B blocks; this fills the chip.
// ------------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ __launch_bounds__(256) void group_points_kernel(const float* __restrict__ points,
                                                           const int* __restrict__ idx,
                                                           float* __restrict__ out, int C, int N,
                                                           long long P, int c_per_block) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (VEC4) {
    const long long p = t * 4;
    if (p >= P) return;
    const int4 ii = *reinterpret_cast<const int4*>(idx + (size_t)b * P + p);
    for (int c = c0; c < c1; ++c) {
      const float* __restrict__ row = points + ((size_t)b * C + c) * N;
      float4 v;
      v.x = row[ii.x];
      v.y = row[ii.y];
      v.z = row[ii.z];
      v.w = row[ii.w];
      *reinterpret_cast<float4*>(out + ((size_t)b * C + c) * P + p) = v;
    }
  } else {
    if (t >= P) return;
    const int i = idx[(size_t)b * P + t];
    for (int c = c0; c < c1; ++c)
      out[((size_t)b * C + c) * P + t] = points[((size_t)b * C + c) * N + i];
  }
}

// group_points backward: grad_points[b,c,idx[b,j,k]] += grad_out[b,c,j,k]  (ref :482-503)
__global__ __launch_bounds__(256) void group_points_grad_kernel(const float* __restrict__ grad_out,
                                                                const int* __restrict__ idx,
                                                                float* __restrict__ grad_points,
                                                                int C, int N, long long P,
                                                                int c_per_block) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= P) return;
  const int i = idx[(size_t)b * P + t];
  for (int c = c0; c < c1; ++c)
    atomicAdd(grad_points + ((size_t)b * C + c) * N + i, grad_out[((size_t)b * C + c) * P + t]);
}

// ------------------------------------------------------------------------------------------------
// three_nn (ref interpolate_gpu.cu:9-52): the three smallest dist3 and their indices, ascending,
// strict < at every rank (earlier index wins ties).  The reference keeps the bests as double
// initialised to 1e40 and stores them as float: identical to float bests initialised to +inf.
// One lane per query, known point wave-uniform.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ unknown,
                                                       const float* __restrict__ known,
                                                       float* __restrict__ dist2,
                                                       int* __restrict__ idx, int N, int M,
                                                       int tiles_per_b) {
  const int b = blockIdx.x / tiles_per_b;
  const int tile = blockIdx.x - b * tiles_per_b;
  const int n = tile * 256 + threadIdx.x;
  const bool valid = n < N;
  const int nc = valid ? n : N - 1;
  const float* __restrict__ u = unknown + ((size_t)b * N + nc) * 3;
  const float* __restrict__ kn = known + (size_t)b * M * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
  int i1 = 0, i2 = 0, i3 = 0;
  for (int k = 0; k < M; ++k) {
    const float d = dist3(ux, uy, uz, kn[3 * (size_t)k], kn[3 * (size_t)k + 1], kn[3 * (size_t)k + 2]);
    if (__any(d < b3)) {
      const bool l1 = d < b1, l2 = d < b2, l3 = d < b3;
      b3 = l2 ? b2 : (l3 ? d : b3);
      i3 = l2 ? i2 : (l3 ? k : i3);
      b2 = l1 ? b1 : (l2 ? d : b2);
      i2 = l1 ? i1 : (l2 ? k : i2);
      b1 = l1 ? d : b1;
      i1 = l1 ? k : i1;
    }
  }
  if (valid) {
    float* od = dist2 + ((size_t)b * N + n) * 3;
    int* oi = idx + ((size_t)b * N + n) * 3;
    od[0] = b1; od[1] = b2; od[2] = b3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
  }
}

// three_interpolate (ref interpolate_gpu.cu:77-97): out[b,c,n] = w0*p[i0] + w1*p[i1] + w2*p[i2],
// canonical rounding fma(w2,p2, fma(w0,p0, w1*p1)).  A thread keeps (idx, weight) of one n and
// walks a slab of channels.
__global__ __launch_bounds__(256) void three_interpolate_kernel(const float* __restrict__ points,
                                                                const int* __restrict__ idx,
                                                                const float* __restrict__ weight,
                                                                float* __restrict__ out, int C,
                                                                int M, int N, int c_per_block) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int* id = idx + ((size_t)b * N + n) * 3;
  const float* w = weight + ((size_t)b * N + n) * 3;
  const int i0 = id[0], i1 = id[1], i2 = id[2];
  const float w0 = w[0], w1 = w[1], w2 = w[2];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c) {
    const float* __restrict__ p = points + ((size_t)b * C + c) * M;
    out[((size_t)b * C + c) * N + n] = __builtin_fmaf(w2, p[i2], __builtin_fmaf(w0, p[i0], w1 * p[i1]));
  }
}

// three_interpolate backward (ref interpolate_gpu.cu:120-142)
__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    const float* __restrict__ grad_out, const int* __restrict__ idx,
    const float* __restrict__ weight, float* __restrict__ grad_points, int C, int N, int M,
    int c_per_block) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int* id = idx + ((size_t)b * N + n) * 3;
  const float* w = weight + ((size_t)b * N + n) * 3;
  const int i0 = id[0], i1 = id[1], i2 = id[2];
  const float w0 = w[0], w1 = w[1], w2 = w[2];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c) {
    const float g = grad_out[((size_t)b * C + c) * N + n];
    float* __restrict__ gp = grad_points + ((size_t)b * C + c) * M;
    atomicAdd(gp + i0, g * w0);
    atomicAdd(gp + i1, g * w1);
    atomicAdd(gp + i2, g * w2);
  }
}

// channels per block so that the grid has roughly >= 8 blocks per CU without re-reading idx more
// often than needed
int pick_c_per_block(long long col_blocks, int B, int C) {
  const long long want = 256LL * 8;
  long long have = col_blocks * B;
  int splits = 1;
  while (splits < C && have * splits < want) splits *= 2;
  if (splits > C) splits = C;
  return (C + splits - 1) / splits;
}

bool grid_ok(long long x, long long y, long long z) {
  return x > 0 && y > 0 && z > 0 && x <= 0x7fffffffLL && y <= 65535 && z <= 65535;
}

}  // namespace

extern "C" int pp_gather_forward_f32(const float* points, const int* idx, float* out, int B, int C,
                                     int N, int M, void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || M == 0) return PP_OK;
  if (!points || !idx || !out || N == 0) return PP_EINVAL;
  const long long cols = (M + 255) / 256;
  const int cpb = pick_c_per_block(cols, B, C);
  const long long gy = (C + cpb - 1) / cpb;
  if (!grid_ok(cols, gy, B)) return PP_EINVAL;
  gather_fwd_kernel<<<dim3((unsigned)cols, (unsigned)gy, (unsigned)B), dim3(256), 0,
                      (hipStream_t)stream>>>(points, idx, out, C, N, M, cpb);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_gather_backward_f32(const float* grad_out, const int* idx, float* grad_points,
                                      int B, int C, int N, int M, void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || M == 0) return PP_OK;
  if (!grad_out || !idx || !grad_points || N == 0) return PP_EINVAL;
  const long long cols = (M + 255) / 256;
  const int cpb = pick_c_per_block(cols, B, C);
  const long long gy = (C + cpb - 1) / cpb;
  if (!grid_ok(cols, gy, B)) return PP_EINVAL;
  gather_bwd_kernel<<<dim3((unsigned)cols, (unsigned)gy, (unsigned)B), dim3(256), 0,
                      (hipStream_t)stream>>>(grad_out, idx, grad_points, C, N, M, cpb);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_ball_query_f32(const float* new_xyz, const float* xyz, int* idx, int B, int N,
                                 int M, float radius, int nsample, void* stream) {
  if (B < 0 || N < 0 || M < 0 || nsample < 0) return PP_EINVAL;
  if (B == 0 || M == 0 || nsample == 0) return PP_OK;
  if (!new_xyz || !idx || (N > 0 && !xyz)) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float radius2 = radius * radius;  // fp32, as the reference (sampling_cuda.cu:354)
  const int tiles = (M + 255) / 256;
  const long long blocks = (long long)B * tiles;
  if (blocks > 0x7fffffffLL) return PP_EINVAL;
  const size_t lds = (size_t)4 * 64 * nsample * sizeof(int);
  if (lds <= 64 * 1024) {
    ball_query_kernel<true><<<dim3((unsigned)blocks), dim3(256), lds, s>>>(
        new_xyz, xyz, idx, N, M, radius2, nsample, tiles);
  } else {
    ball_query_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
        new_xyz, xyz, idx, N, M, radius2, nsample, tiles);
  }
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_group_points_f32(const float* points, const int* idx, float* out, int B, int C,
                                   int N, int npoint, int nsample, void* stream) {
  if (B < 0 || C < 0 || N < 0 || npoint < 0 || nsample < 0) return PP_EINVAL;
  const long long P = (long long)npoint * nsample;
  if (B == 0 || C == 0 || P == 0) return PP_OK;
  if (!points || !idx || !out || N == 0) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool vec4 = (P % 4 == 0) && ((uintptr_t)idx % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const long long threads = vec4 ? P / 4 : P;
  const long long cols = (threads + 255) / 256;
  const int cpb = pick_c_per_block(cols, B, C);
  const long long gy = (C + cpb - 1) / cpb;
  if (!grid_ok(cols, gy, B)) return PP_EINVAL;
  const dim3 grid((unsigned)cols, (unsigned)gy, (unsigned)B);
  if (vec4)
    group_points_kernel<true><<<grid, dim3(256), 0, s>>>(points, idx, out, C, N, P, cpb);
  else
    group_points_kernel<false><<<grid, dim3(256), 0, s>>>(points, idx, out, C, N, P, cpb);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_group_points_grad_f32(const float* grad_out, const int* idx, float* grad_points,
                                        int B, int C, int N, int npoint, int nsample,
                                        void* stream) {
  if (B < 0 || C < 0 || N < 0 || npoint < 0 || nsample < 0) return PP_EINVAL;
  const long long P = (long long)npoint * nsample;
  if (B == 0 || C == 0 || P == 0) return PP_OK;
  if (!grad_out || !idx || !grad_points || N == 0) return PP_EINVAL;
  const long long cols = (P + 255) / 256;
  const int cpb = pick_c_per_block(cols, B, C);
  const long long gy = (C + cpb - 1) / cpb;
  if (!grid_ok(cols, gy, B)) return PP_EINVAL;
  group_points_grad_kernel<<<dim3((unsigned)cols, (unsigned)gy, (unsigned)B), dim3(256), 0,
                             (hipStream_t)stream>>>(grad_out, idx, grad_points, C, N, P, cpb);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_three_nn_f32(const float* unknown, const float* known, float* dist2, int* idx,
                               int B, int N, int M, void* stream) {
  if (B < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || N == 0) return PP_OK;
  if (!unknown || !dist2 || !idx || (M > 0 && !known)) return PP_EINVAL;
  const int tiles = (N + 255) / 256;
  const long long blocks = (long long)B * tiles;
  if (blocks > 0x7fffffffLL) return PP_EINVAL;
  three_nn_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
      unknown, known, dist2, idx, N, M, tiles);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_three_interpolate_f32(const float* points, const int* idx, const float* weight,
                                        float* out, int B, int C, int M, int N, void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || N == 0) return PP_OK;
  if (!points || !idx || !weight || !out || M == 0) return PP_EINVAL;
  const long long cols = (N + 255) / 256;
  const int cpb = pick_c_per_block(cols, B, C);
  const long long gy = (C + cpb - 1) / cpb;
  if (!grid_ok(cols, gy, B)) return PP_EINVAL;
  three_interpolate_kernel<<<dim3((unsigned)cols, (unsigned)gy, (unsigned)B), dim3(256), 0,
                             (hipStream_t)stream>>>(points, idx, weight, out, C, M, N, cpb);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_three_interpolate_grad_f32(const float* grad_out, const int* idx,
                                             const float* weight, float* grad_points, int B,
                                             int C, int N, int M, void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || N == 0) return PP_OK;
  if (!grad_out || !idx || !weight || !grad_points || M == 0) return PP_EINVAL;
  const long long cols = (N + 255) / 256;
  const int cpb = pick_c_per_block(cols, B, C);
  const long long gy = (C + cpb - 1) / cpb;
  if (!grid_ok(cols, gy, B)) return PP_EINVAL;
  three_interpolate_grad_kernel<<<dim3((unsigned)cols, (unsigned)gy, (unsigned)B), dim3(256), 0,
                                  (hipStream_t)stream>>>(grad_out, idx, weight, grad_points, C, N,
                                                         M, cpb);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
