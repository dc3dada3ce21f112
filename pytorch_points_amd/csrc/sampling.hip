// sampling.hip -- gather_points, ball_query, group_points, three_nn, three_interpolate for gfx950.
// Replaces the reference's _ext/sampling_cuda.cu (gather :9-84, ball_query :340-397,
// group_points :447-513) and _ext/interpolate_gpu.cu (three_nn :9-74, three_interpolate :77-160).
#include "grid_common.h"

namespace pp {  // scatter.hip
size_t ssa_workspace_bytes(int B, long long P, int R, int Nd, bool weighted);
int ssa_run(const float* src, const int* dst, const float* weight, float* out, int B, int C,
            long long P, int R, int Nd, long long src_bstride, void* workspace, hipStream_t s, bool ordered);
}  // namespace pp

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

// gather_points through the LDS: a workgroup stages the row points[b,c,:] (coalesced 16-byte loads)
// and gathers from LDS, CPB channels in turn, the next row's loads in flight while this one is
// gathered.  Reads 4 N bytes per row instead of one 64-byte sector per gathered element: the better
// deal when N <= 16 M.  XCD mapping as group_points: idx of one batch element is read by one L2.
template <int KR, bool VEC>
__global__ __launch_bounds__(1024) void gather_fwd_lds_kernel(const float* __restrict__ points,
                                                              const int* __restrict__ idx,
                                                              float* __restrict__ out, int B, int C, int N,
                                                              int M, int cpb, int groups) {
  extern __shared__ __attribute__((aligned(16))) float s_grow[];
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / groups);
  if (b >= B) return;
  const int c0 = (y % groups) * cpb;
  const int c1 = min(C, c0 + cpb);
  const int t = threadIdx.x;
  const int* __restrict__ ib = idx + (size_t)b * M;
  if constexpr (VEC) {
    const int n4 = N >> 2;
    pp::f4 pre[KR];
    auto load_row = [&](int c) {
      const pp::f4* __restrict__ row = reinterpret_cast<const pp::f4*>(points + ((size_t)b * C + c) * N);
#pragma unroll
      for (int k = 0; k < KR; ++k) pre[k] = row[min(t + 1024 * k, n4 - 1)];
    };
    load_row(c0);
    for (int c = c0; c < c1; ++c) {
      __syncthreads();  // the previous row has been gathered
#pragma unroll
      for (int k = 0; k < KR; ++k)
        if (t + 1024 * k < n4) reinterpret_cast<pp::f4*>(s_grow)[t + 1024 * k] = pre[k];
      __syncthreads();
      if (c + 1 < c1) load_row(c + 1);
      float* __restrict__ o = out + ((size_t)b * C + c) * M;
      for (int m = t * 4; m < M; m += 4096) {
        const pp::i4 i = *reinterpret_cast<const pp::i4*>(ib + m);
        pp::f4 r = {s_grow[i.x], s_grow[i.y], s_grow[i.z], s_grow[i.w]};
        *reinterpret_cast<pp::f4*>(o + m) = r;
      }
    }
  } else {  // no alignment assumed: 4-byte accesses, all coalesced
    for (int c = c0; c < c1; ++c) {
      __syncthreads();
      const float* __restrict__ row = points + ((size_t)b * C + c) * N;
      for (int e = t; e < N; e += 1024) s_grow[e] = row[e];
      __syncthreads();
      float* __restrict__ o = out + ((size_t)b * C + c) * M;
      for (int m = t; m < M; m += 1024) o[m] = s_grow[ib[m]];
    }
  }
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
                                                         int tiles_per_b, const pp::GridSet* __restrict__ skip) {
  extern __shared__ __attribute__((aligned(16))) int s_rows[];  // [4 waves][64][nsample]
  const int b = blockIdx.x / tiles_per_b;
  const int tile = blockIdx.x - b * tiles_per_b;
  if (skip && skip[b].pad[0]) return;  // this batch element was handled by the grid search
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
    for (int f0 = 0; f0 < total; f0 += 64) {  // uniform trip count: a shuffle needs its source lane active
      const int f = min(f0 + lane, total - 1);
      const int row = f / nsample;
      const int slot = f - row * nsample;
      const int rc = __shfl(cnt, row);
      const int rf = __shfl(first, row);
      // slot < rc: a recorded hit; otherwise the pad (first hit, or 0 when the ball is empty)
      const int v = slot < rc ? wrows[f] : rf;
      if (f0 + lane < total) gout[f] = v;
    }
  } else if (valid) {
    for (int s = cnt; s < nsample; ++s) grow[s] = first;
  }
}

// ------------------------------------------------------------------------------------------------
// ball_query v2: a workgroup = 64 centres x 4 waves; wave w scans the w-th quarter of the cloud
// (4x the waves of v1 at the same problem size -- v1 leaves the chip at 2 waves per SIMD at
// config 4) and records its in-radius indices, ascending, in its own LDS list (capped at nsample:
// later entries can never be output).  The row is then the concatenation of the four lists in
// quarter order, cut at nsample and padded with the first hit -- exactly the sequential scan's
// result.  Lists are stored as IT = uint16 when N <= 65536 (32 KiB per workgroup at nsample = 64).
// Per group of 8 scanned points the eight `d < r^2` tests become eight wave masks (v_cmp into
// SGPRs); points nobody hit are skipped with scalar branches.
// ------------------------------------------------------------------------------------------------
template <typename IT>
__global__ __launch_bounds__(256) void ball_query_split_kernel(const float* __restrict__ new_xyz,
                                                               const float* __restrict__ xyz,
                                                               int* __restrict__ idx, int N, int M,
                                                               float radius2, int nsample,
                                                               int tiles_per_b,
                                                               const pp::GridSet* __restrict__ skip) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  if (skip && skip[blockIdx.x / tiles_per_b].pad[0]) return;  // handled by the grid search
  IT* s_list = reinterpret_cast<IT*>(s_raw);                               // [4][64][nsample]
  int* s_cnt = reinterpret_cast<int*>(s_raw + (size_t)4 * 64 * nsample * sizeof(IT));  // [4][64]
  const int b = blockIdx.x / tiles_per_b;
  const int tile = blockIdx.x - b * tiles_per_b;
  const int wave = pp::wave_id_uniform();
  const int lane = threadIdx.x & 63;
  const int m0 = tile * 64;
  const int m = m0 + lane;
  const bool valid = m < M;
  const int mc = valid ? m : M - 1;
  const float* __restrict__ q = new_xyz + ((size_t)b * M + mc) * 3;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  IT* mylist = s_list + ((size_t)wave * 64 + lane) * nsample;

  // quarter [k0, k1), boundaries on multiples of the group size
  const int ngroups = N / kBqGroup;
  const int g0 = (int)(((long long)ngroups * wave) / 4), g1 = (int)(((long long)ngroups * (wave + 1)) / 4);
  int cnt = valid ? 0 : nsample;
  auto load_group = [&](float (&rr)[kBqGroup * 3], int g) {
    const float* __restrict__ rp = p + (size_t)g * (kBqGroup * 3);  // wave-uniform -> s_load
#pragma unroll
    for (int e = 0; e < kBqGroup * 3; ++e) rr[e] = rp[e];
  };
  auto scan_group = [&](const float (&rr)[kBqGroup * 3], int g) {
    const int k = g * kBqGroup;
    unsigned long long hit[kBqGroup];
    unsigned long long any = 0;
#pragma unroll
    for (int e = 0; e < kBqGroup; ++e) {
      const float d = dist3(qx, qy, qz, rr[3 * e], rr[3 * e + 1], rr[3 * e + 2]);
      hit[e] = __ballot(d < radius2);
      any |= hit[e];
    }
    if (any) {
#pragma unroll
      for (int e = 0; e < kBqGroup; ++e) {
        if (hit[e]) {
          if (((hit[e] >> lane) & 1ull) && cnt < nsample) {
            mylist[cnt] = (IT)(k + e);
            ++cnt;
          }
        }
      }
    }
  };
  // two groups per trip, the scalar loads of the next group issued before the current one is
  // scanned (ping-pong SGPR sets): the s_load latency is covered by this wave's own VALU work
  if (g0 < g1) {
    float ra[kBqGroup * 3], rb[kBqGroup * 3];
    load_group(ra, g0);
    int g = g0;
    for (; g + 1 < g1; g += 2) {
      if (__all(cnt >= nsample)) break;
      load_group(rb, g + 1);
      scan_group(ra, g);
      // scalar loads return out of order, so the only wait is lgkmcnt(0): make rb "used" here,
      // before the next loads are issued, or waiting for rb would also wait for them
      asm volatile("" ::"s"(rb[0]), "s"(rb[kBqGroup * 3 - 1]));
      load_group(ra, g + 2 < g1 ? g + 2 : g1 - 1);
      scan_group(rb, g + 1);
      asm volatile("" ::"s"(ra[0]), "s"(ra[kBqGroup * 3 - 1]));
    }
    if (g < g1 && !__all(cnt >= nsample)) scan_group(ra, g);
  }
  if (wave == 3 && !__all(cnt >= nsample)) {  // tail points (N mod 8), highest indices
    for (int k = ngroups * kBqGroup; k < N; ++k) {
      const float d = dist3(qx, qy, qz, p[3 * (size_t)k], p[3 * (size_t)k + 1], p[3 * (size_t)k + 2]);
      if (d < radius2 && cnt < nsample) {
        mylist[cnt] = (IT)k;
        ++cnt;
      }
    }
  }
  s_cnt[wave * 64 + lane] = valid ? cnt : 0;
  __syncthreads();

  // 64 rows x nsample ints, contiguous in the output
  int* __restrict__ gout = idx + ((size_t)b * M + m0) * nsample;
  const int nrows = min(64, M - m0);
  const int total = nrows * nsample;
  for (int f = threadIdx.x; f < total; f += 256) {
    const int row = f / nsample;
    const int slot = f - row * nsample;
    const int c0 = s_cnt[row], c1 = s_cnt[64 + row], c2 = s_cnt[128 + row], c3 = s_cnt[192 + row];
    const IT* l0 = s_list + (size_t)row * nsample;
    const IT* l1 = l0 + (size_t)64 * nsample;
    const IT* l2 = l1 + (size_t)64 * nsample;
    const IT* l3 = l2 + (size_t)64 * nsample;
    // first hit of the row = head of the first non-empty list (0 if the ball is empty)
    int first = 0;
    if (c0 > 0) first = (int)l0[0];
    else if (c1 > 0) first = (int)l1[0];
    else if (c2 > 0) first = (int)l2[0];
    else if (c3 > 0) first = (int)l3[0];
    int v = first;
    int sidx = slot;
    if (sidx < c0) v = (int)l0[sidx];
    else if ((sidx -= c0) < c1) v = (int)l1[sidx];
    else if ((sidx -= c1) < c2) v = (int)l2[sidx];
    else if ((sidx -= c2) < c3) v = (int)l3[sidx];
    gout[f] = v;
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
                                                           long long P, int c_per_block, long long obs) {
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
      *reinterpret_cast<float4*>(out + (size_t)b * obs + (size_t)c * P + p) = v;
    }
  } else {
    if (t >= P) return;
    const int i = idx[(size_t)b * P + t];
    for (int c = c0; c < c1; ++c)
      out[(size_t)b * obs + (size_t)c * P + t] = points[((size_t)b * C + c) * N + i];
  }
}

// v2: the channel row points[b,c,:] (N floats) is staged in LDS once per workgroup and the gathers
// become ds_read_b32 (random LDS addresses: ~3-4-way bank conflicts, still several times the rate
// of 64-distinct-line global gathers).  A 512-thread workgroup owns 512*4*V consecutive (j,k)
// positions of one batch element, keeps their indices in registers for all C channels (idx is
// read from HBM exactly once) and streams 16-byte stores.  Two workgroups per CU (<= 64 KiB of LDS
// each) overlap one's row load + barrier with the other's gather + store phase.  Workgroups of
// one batch element share blockIdx % 8 (one XCD's L2 serves the row re-reads).
// A 16-byte global load the compiler does not track: hipcc's waitcnt pass treats a vmcnt queue that
// holds both loads and stores as unordered and drains it (vmcnt(0)) at the first use of a load
// result -- which here would wait for the previous channel's 8 HBM stores every iteration.  The
// hardware retires vector-memory operations in issue order, so the loads (issued before the
// stores) are complete once at most `V` younger operations are outstanding: vm_wait<V>().
__device__ __forceinline__ void load16_untracked(pp::f4& dst, const pp::f4* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N, int KR>
__device__ __forceinline__ void vm_wait(pp::f4 (&r)[KR]) {
  // names every destination so that no use of them can be scheduled above the wait
  if constexpr (KR == 1)
    asm volatile("s_waitcnt vmcnt(%c1)" : "+v"(r[0]) : "i"(N));
  else if constexpr (KR == 2)
    asm volatile("s_waitcnt vmcnt(%c2)" : "+v"(r[0]), "+v"(r[1]) : "i"(N));
  else if constexpr (KR == 4)
    asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "i"(N));
  else
    asm volatile("s_waitcnt vmcnt(%c8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "i"(N));
}

template <int V, int KR, bool FULL>
__device__ __forceinline__ void group_points_lds_loop(const pp::f4* __restrict__ row, float* __restrict__ out_b,
                                                      float* s_row, const pp::i4 (&ii)[V], int C, int n4,
                                                      long long P, long long p0, int t) {
  // The row loads are unconditional (indices clamped to the row: surplus lanes re-load and
  // re-write the last float4 with the same value).  Row c+1 is fetched into registers BEFORE the
  // stores of row c are issued; in the FULL instantiation exactly V stores follow the loads, so
  // vmcnt(V) retires the loads and leaves the stores in flight across the barriers.
  static_assert(KR == 1 || KR == 2 || KR == 4 || KR == 8, "");
  int ee[KR];
  pp::f4 pre[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    ee[k] = min(t + 512 * k, n4 - 1);
    load16_untracked(pre[k], row + ee[k]);
  }
  vm_wait<0, KR>(pre);
  for (int c = 0; c < C; ++c) {
    __syncthreads();  // everyone is done gathering from the previous row
    // FULL: exactly V stores were issued after the loads of this row; otherwise the count varies
    // per wave and the queue is drained.  (c == 0: retired before the loop.)
    if (c > 0) vm_wait<FULL ? V : 0, KR>(pre);
#pragma unroll
    for (int k = 0; k < KR; ++k) reinterpret_cast<pp::f4*>(s_row)[ee[k]] = pre[k];
    __syncthreads();
    // last iteration: harmless re-load of the same row (keeps the loop body branch-free)
    const pp::f4* __restrict__ nrow = row + (size_t)(c + 1 < C ? c + 1 : c) * n4;
#pragma unroll
    for (int k = 0; k < KR; ++k) load16_untracked(pre[k], nrow + ee[k]);
    float* __restrict__ o = out_b + (size_t)c * P;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const long long p = p0 + (long long)v * 2048;
      pp::f4 r;
      r.x = s_row[ii[v].x];
      r.y = s_row[ii[v].y];
      r.z = s_row[ii[v].z];
      r.w = s_row[ii[v].w];
      if (FULL || p < P) *reinterpret_cast<pp::f4*>(o + p) = r;
    }
  }
  vm_wait<0, KR>(pre);  // the surplus prefetch of the last iteration
}

template <int V, int KR>
__global__ __launch_bounds__(512, 4) void group_points_lds_kernel(const float* __restrict__ points,
                                                                  const int* __restrict__ idx,
                                                                  float* __restrict__ out, int B, int C,
                                                                  int N, long long P, int chunks, long long obs) {
  extern __shared__ __attribute__((aligned(16))) float s_row[];
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / chunks);
  const int chunk = y % chunks;
  if (b >= B) return;
  const int t = threadIdx.x;
  const long long p0 = (long long)chunk * (512 * 4 * V) + t * 4;
  const bool full = (long long)(chunk + 1) * (512 * 4 * V) <= P;  // every position of the chunk exists
  pp::i4 ii[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const long long p = p0 + (long long)v * 2048;
    ii[v] = p < P ? *reinterpret_cast<const pp::i4*>(idx + (size_t)b * P + p) : pp::i4{0, 0, 0, 0};
  }
  const pp::f4* __restrict__ row = reinterpret_cast<const pp::f4*>(points + (size_t)b * C * N);
  float* __restrict__ out_b = out + (size_t)b * obs;
  if (full)
    group_points_lds_loop<V, KR, true>(row, out_b, s_row, ii, C, N >> 2, P, p0, t);
  else
    group_points_lds_loop<V, KR, false>(row, out_b, s_row, ii, C, N >> 2, P, p0, t);
}

// ------------------------------------------------------------------------------------------------
// v5: the channel rows arrive by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write) into ONE row
// buffer per 512-thread workgroup, two workgroups per CU: nothing overlaps inside a workgroup (DMA row c, wait,
// barrier, gather + store, barrier); the two workgroups of a CU overlap each other and are not in step (0.89 against
// 0.92 ms at config 4 for the 1024-thread ring of two buffers it replaced -- tools/archive/group_points_dma_ring_r2_r3
// -- ; the stores alone, without any barrier, would take 0.75 ms).  Full chunks of 512 * 4 * V positions only.
// ------------------------------------------------------------------------------------------------
constexpr int kDma1Threads = 512;
template <int V, bool NT = false>
__global__ __launch_bounds__(kDma1Threads) void group_points_dma1_kernel(const float* __restrict__ points,
                                                                         const int* __restrict__ idx,
                                                                         float* __restrict__ out, int B, int C,
                                                                         int N, long long P, int chunks, int passes,
                                                                         int cgroups, int c_per_group, long long obs) {
  extern __shared__ __attribute__((aligned(16))) float s_row1[];
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int per_b = chunks * cgroups;
  const int b = x + 8 * (y / per_b);
  const int rem = y % per_b;
  const int chunk = rem / cgroups;
  const int c_begin = (rem % cgroups) * c_per_group;
  const int c_end = min(C, c_begin + c_per_group);
  if (b >= B || c_begin >= c_end) return;
  const int t = threadIdx.x;
  const int wave = pp::wave_id_uniform();
  const long long p0 = (long long)chunk * (kDma1Threads * 4 * V) + t * 4;
  unsigned ii[V][4];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const pp::i4 q = *reinterpret_cast<const pp::i4*>(idx + (size_t)b * P + p0 + (long long)v * (kDma1Threads * 4));
    ii[v][0] = q.x; ii[v][1] = q.y; ii[v][2] = q.z; ii[v][3] = q.w;
  }
  const int n4 = N >> 2;
  const pp::f4* __restrict__ row0 = reinterpret_cast<const pp::f4*>(points + (size_t)b * C * N);
  float* __restrict__ out_b = out + (size_t)b * obs;
  for (int c = c_begin; c < c_end; ++c) {
    const pp::f4* __restrict__ row = row0 + (size_t)c * n4;
    for (int k = 0; k < passes; ++k) {
      const int e = k * kDma1Threads + t;
      const int src = e < n4 ? e : n4 - 1;
      float* dst = s_row1 + (size_t)(k * kDma1Threads + wave * 64) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + src),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float* __restrict__ o = out_b + (size_t)c * P;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      pp::f4 r;
      r.x = s_row1[ii[v][0]];
      r.y = s_row1[ii[v][1]];
      r.z = s_row1[ii[v][2]];
      r.w = s_row1[ii[v][3]];
      if (NT)
        __builtin_nontemporal_store(r, reinterpret_cast<pp::f4*>(o + p0 + (long long)v * (kDma1Threads * 4)));
      else
        *reinterpret_cast<pp::f4*>(o + p0 + (long long)v * (kDma1Threads * 4)) = r;
    }
    __builtin_amdgcn_s_barrier();  // every wave has read the row: the next one may land
    asm volatile("" ::: "memory");
  }
}

template <int V, bool NT = false>
bool launch_group_dma1(const float* points, const int* idx, float* out, int B, int C, int N, long long P,
                       long long obs, hipStream_t s) {
  const long long per_block = (long long)kDma1Threads * 4 * V;
  if (P % per_block != 0) return false;
  const long long chunks = P / per_block;
  const long long base = 8LL * ((B + 7) / 8) * chunks;
  int cgroups = 1;
  while (base * cgroups < 512 && cgroups * 2 <= C) cgroups *= 2;
  const int c_per_group = (C + cgroups - 1) / cgroups;
  const long long blocks = base * cgroups;
  const int n4 = N / 4;
  const int passes = (n4 + kDma1Threads - 1) / kDma1Threads;
  const size_t lds = (size_t)passes * kDma1Threads * 16;
  if (lds > 72 * 1024 || chunks > 0x7fffffLL || blocks > 0x7fffffffLL) return false;
  static pp::DeviceFlags lds_ok;
  if (pp::allow_big_lds(group_points_dma1_kernel<V, NT>, 80 * 1024, lds_ok) != hipSuccess) return false;
  group_points_dma1_kernel<V, NT><<<dim3((unsigned)blocks), dim3(kDma1Threads), lds, s>>>(
      points, idx, out, B, C, N, P, (int)chunks, passes, cgroups, c_per_group, obs);
  return true;
}

template <int V>
bool launch_group_lds(const float* points, const int* idx, float* out, int B, int C, int N,
                      long long P, long long obs, hipStream_t s) {
  const long long per_block = 512LL * 4 * V;
  const long long chunks = (P + per_block - 1) / per_block;
  const long long blocks = 8LL * ((B + 7) / 8) * chunks;
  if (chunks > 0x7fffffLL || blocks > 0x7fffffffLL) return false;
  // KR = float4 per thread per staging pass: 512 * KR * 16 B >= row bytes
  const dim3 grid((unsigned)blocks), block(512);
  const size_t lds = (size_t)N * sizeof(float);
  const int n4 = N / 4;
  if (n4 <= 512 * 1)
    group_points_lds_kernel<V, 1><<<grid, block, lds, s>>>(points, idx, out, B, C, N, P, (int)chunks, obs);
  else if (n4 <= 512 * 2)
    group_points_lds_kernel<V, 2><<<grid, block, lds, s>>>(points, idx, out, B, C, N, P, (int)chunks, obs);
  else if (n4 <= 512 * 4)
    group_points_lds_kernel<V, 4><<<grid, block, lds, s>>>(points, idx, out, B, C, N, P, (int)chunks, obs);
  else
    group_points_lds_kernel<V, 8><<<grid, block, lds, s>>>(points, idx, out, B, C, N, P, (int)chunks, obs);
  return true;
}

// LDS-staged form without any alignment requirement (P, N, the batch stride or the pointers not
// multiples of 4 elements) and for rows up to 152 KiB: 4-byte loads and stores, all coalesced; a thread
// keeps V positions (stride 1024) for all channels.  Slower than the 16-byte forms above, but the
// alternative for such shapes is the global-gather kernel at ~0.5 TB/s.
template <int V>
__global__ __launch_bounds__(1024) void group_points_lds_scalar_kernel(const float* __restrict__ points,
                                                                       const int* __restrict__ idx,
                                                                       float* __restrict__ out, int B, int C,
                                                                       int N, long long P, int chunks,
                                                                       long long obs) {
  extern __shared__ __attribute__((aligned(16))) float s_srow[];
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / chunks);
  const int chunk = y % chunks;
  if (b >= B) return;
  const int t = threadIdx.x;
  const long long p0 = (long long)chunk * (1024 * V) + t;
  int ii[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const long long p = p0 + 1024LL * v;
    ii[v] = p < P ? idx[(size_t)b * P + p] : 0;
  }
  const float* __restrict__ rows = points + (size_t)b * C * N;
  float* __restrict__ out_b = out + (size_t)b * obs;
  for (int c = 0; c < C; ++c) {
    __syncthreads();  // the previous row has been gathered
    const float* __restrict__ row = rows + (size_t)c * N;
    for (int e = t; e < N; e += 1024) s_srow[e] = row[e];
    __syncthreads();
    float* __restrict__ o = out_b + (size_t)c * P;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const long long p = p0 + 1024LL * v;
      if (p < P) o[p] = s_srow[ii[v]];
    }
  }
}

// group_points backward: grad_points[b,c,idx[b,j,k]] += grad_out[b,c,j,k]  (ref :482-503)
__global__ __launch_bounds__(256) void group_points_grad_kernel(const float* __restrict__ grad_out,
                                                                const int* __restrict__ idx,
                                                                float* __restrict__ grad_points,
                                                                int C, int N, long long P,
                                                                int c_per_block, long long gbs) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= P) return;
  const int i = idx[(size_t)b * P + t];
  for (int c = c0; c < c1; ++c)
    atomicAdd(grad_points + ((size_t)b * C + c) * N + i, grad_out[(size_t)b * gbs + (size_t)c * P + t]);
}

// group_points backward without global atomics: one workgroup per (batch, channel) owns the column
// grad_points[b,c,:] in LDS, streams grad_out[b,c,:,:] and idx[b,:,:] with 16-byte loads, adds with
// ds_add_f32 and ADDS the column to the caller's (zero-filled) output once.  At config 4 the
// reference's formulation is 1.07e9 scattered global atomics.  Workgroups of one batch element share
// blockIdx % 8, so idx (re-read per channel) is served by one XCD's L2.
__global__ __launch_bounds__(1024) void group_points_grad_lds_kernel(const float* __restrict__ grad_out,
                                                                     const int* __restrict__ idx,
                                                                     float* __restrict__ grad_points,
                                                                     int B, int C, int N, long long P,
                                                                     long long gbs) {
  extern __shared__ __attribute__((aligned(16))) float s_col[];
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / C);
  const int c = y % C;
  if (b >= B) return;
  const int t = threadIdx.x;
  for (int k = t; k < N; k += 1024) s_col[k] = 0.0f;
  __syncthreads();
  const float* __restrict__ go = grad_out + (size_t)b * gbs + (size_t)c * P;
  const int* __restrict__ ib = idx + (size_t)b * P;
  const long long p4 = P >> 2;
  for (long long e = t; e < p4; e += 1024) {
    const pp::f4 g = reinterpret_cast<const pp::f4*>(go)[e];
    const pp::i4 i = reinterpret_cast<const pp::i4*>(ib)[e];
    atomicAdd(&s_col[i.x], g.x);
    atomicAdd(&s_col[i.y], g.y);
    atomicAdd(&s_col[i.z], g.z);
    atomicAdd(&s_col[i.w], g.w);
  }
  for (long long e = (p4 << 2) + t; e < P; e += 1024) atomicAdd(&s_col[ib[e]], go[e]);
  __syncthreads();
  float* __restrict__ gp = grad_points + ((size_t)b * C + c) * N;
  for (int k = t; k < N; k += 1024) gp[k] += s_col[k];  // accumulate: the ABI's contract
}

// The same with a DOUBLE column: on gfx950 ds_add_f64 runs at 3.3 lanes/clk/CU against 0.36 for
// ds_add_f32 (tools/lds_atomic_probe.hip), so widening the accumulator is 9x faster -- and the sum
// is rounded to fp32 once, at the end, instead of at every addend.  A workgroup owns W <= 19456
// destinations (8 W bytes of LDS) of one (batch, channel): the whole column when N fits, otherwise
// one of `nsplit` ranges, each workgroup streaming the column's grad_out and skipping the entries
// of the other ranges.  U 16-byte load pairs are in flight per thread before the first atomic.
template <int U, bool VEC = true>
__global__ __launch_bounds__(1024) void group_points_grad_lds64_kernel(const float* __restrict__ grad_out,
                                                                       const int* __restrict__ idx,
                                                                       float* __restrict__ grad_points,
                                                                       int B, int C, int N, long long P,
                                                                       long long gbs, int nsplit, int W, int overwrite) {
  extern __shared__ __attribute__((aligned(16))) double s_col64[];
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int per_b = C * nsplit;
  const int b = x + 8 * (y / per_b);
  const int r = y % per_b;
  const int c = r / nsplit;
  const int lo = (r - c * nsplit) * W;     // this workgroup's destinations: [lo, lo + w)
  const int w = min(W, N - lo);
  if (b >= B) return;
  const int t = threadIdx.x;
  for (int k = t; k < w; k += 1024) s_col64[k] = 0.0;
  __syncthreads();
  const float* __restrict__ go = grad_out + (size_t)b * gbs + (size_t)c * P;
  const int* __restrict__ ib = idx + (size_t)b * P;
  const long long p4 = P >> 2;
  auto add1 = [&](int i, double v) {
    if ((unsigned)(i - lo) < (unsigned)w) atomicAdd(&s_col64[i - lo], v);
  };
  // Runs of equal consecutive indices (the pad of a ball_query row repeats its first index) are
  // summed in registers and added once, by the run's first element: same-address LDS atomics
  // serialise, and at config 4 a third of all entries are pads.
  auto add4 = [&](const pp::f4& g, const pp::i4& i) {
    const bool e1 = i.y == i.x, e2 = i.z == i.y, e3 = i.w == i.z;
    const double sw = (double)g.w;
    const double sz = (double)g.z + (e3 ? sw : 0.0);
    const double sy = (double)g.y + (e2 ? sz : 0.0);
    const double sx = (double)g.x + (e1 ? sy : 0.0);
    add1(i.x, sx);
    if (!e1) add1(i.y, sy);
    if (!e2) add1(i.z, sz);
    if (!e3) add1(i.w, sw);
  };
  // The same for a wave whose 64 lanes hold 64 CONSECUTIVE quads (the main loop below), with runs merged ACROSS lanes
  // as well (round 5): the pad of a ball_query row is the row's first index repeated to the row's end --
  // at config 4 a third of a row, six quads in six neighbouring lanes that all add to ONE address in the same wave
  // instruction (same-address LDS atomics serialise).  A quad that is a single run and continues the run its left
  // neighbour ends with hands its sum to the left instead of adding it: a segmented suffix sum over the lanes of a row of
  // sixteen (DPP row shifts; 64 entries = a row of nsample = 64), in doubles, so that what is added is what the lanes would
  // have added one by one up to the order of a sum of doubles.
  auto add4_merged = [&](const pp::f4& g, const pp::i4& i) {
    const bool e1 = i.y == i.x, e2 = i.z == i.y, e3 = i.w == i.z;
    const double sw = (double)g.w;
    const double sz = (double)g.z + (e3 ? sw : 0.0);
    const double sy = (double)g.y + (e2 ? sz : 0.0);
    const double sx = (double)g.x + (e1 ? sy : 0.0);
    const bool full = e1 & e2 & e3;
    auto shl = [](int v, auto k_c) {  // lane l <- lane l + k of its row of sixteen; 0 beyond the row
      constexpr int K = decltype(k_c)::value;
      return __builtin_amdgcn_update_dpp(0, v, 0x100 + K, 0xf, 0xf, false);
    };
    auto shl_d = [&](double v, auto k_c) {
      const long long b = __builtin_bit_cast(long long, v);
      const unsigned lo = (unsigned)shl((int)(unsigned)b, k_c), hi = (unsigned)shl((int)(unsigned)(b >> 32), k_c);
      return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
    };
    using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
    using K4 = std::integral_constant<int, 4>; using K8 = std::integral_constant<int, 8>;
    // cont: this quad is one run and continues the run the left neighbour (same row of sixteen) ends with
    const int left_w = __builtin_amdgcn_update_dpp(0, i.w, 0x111, 0xf, 0xf, false);  // row_shr:1
    const bool cont = full && (threadIdx.x & 15) != 0 && i.x == left_w;
    // A = the sums of the quads to the right for as long as they continue this lane's last run
    // (every shift is evaluated by ALL lanes, then selected: inside `f ? shift : 0` it would run under f's mask and read
    //  the lanes that mask switches off -- the chain's last quad -- as zero)
    int f = shl(cont ? 1 : 0, K1{});                    // the right neighbour continues
    const double sx_right = shl_d(sx, K1{});
    double A = f ? sx_right : 0.0;
    {
      const double a2 = shl_d(A, K1{});
      const int f2 = shl(f, K1{});
      A += f ? a2 : 0.0;
      f &= f2;                                          // lanes l+1, l+2 both continue
    }
    {
      const double a4 = shl_d(A, K2{});
      const int f4 = shl(f, K2{});
      A += f ? a4 : 0.0;
      f &= f4;                                          // lanes l+1 .. l+4
    }
    {
      const double a8 = shl_d(A, K4{});
      const int f8 = shl(f, K4{});
      A += f ? a8 : 0.0;
      f &= f8;                                          // lanes l+1 .. l+8
    }
    {
      const double a16 = shl_d(A, K8{});
      A += f ? a16 : 0.0;
    }
    // the lane's LAST run takes A along; a continuing quad adds nothing
    if (!cont) add1(i.x, sx + (full ? A : 0.0));
    if (!e1) add1(i.y, sy + ((e2 & e3) ? A : 0.0));
    if (!e2) add1(i.z, sz + (e3 ? A : 0.0));
    if (!e3) add1(i.w, sw + A);
  };
  if constexpr (VEC) {
    long long e = t;
    // (the bound is the wave's LAST lane's: a wave enters the loop whole or not at all -- add4_merged's row shifts read
    //  the neighbouring lanes -- and the quads of a wave that straddles the end go to the lane-by-lane loop below)
    for (; (e | 63) + 1024 * (U - 1) < p4; e += 1024 * U) {
      pp::f4 g[U];
      pp::i4 i[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // (grad_out is read once: non-temporal, so that the 4 GiB stream does not push the indices -- re-read by
        //  every channel's workgroup -- out of the XCD's L2)
        g[u] = __builtin_nontemporal_load(reinterpret_cast<const pp::f4*>(go) + e + 1024 * u);
        i[u] = reinterpret_cast<const pp::i4*>(ib)[e + 1024 * u];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) add4_merged(g[u], i[u]);  // (every lane is active here: the DPP shifts see whole rows)
    }
    for (; e < p4; e += 1024) add4(reinterpret_cast<const pp::f4*>(go)[e], reinterpret_cast<const pp::i4*>(ib)[e]);
    for (long long q = (p4 << 2) + t; q < P; q += 1024) add1(ib[q], (double)go[q]);
  } else {
    // rows that are not 16-byte aligned (P or the batch stride not a multiple of 4): 4-byte loads, still
    // coalesced, 2U in flight per thread -- slower than the vector form, far from the global-atomic one
    long long e = t;
    for (; e + 1024 * (2 * U - 1) < P; e += 1024 * 2 * U) {
      float g[2 * U];
      int i[2 * U];
#pragma unroll
      for (int u = 0; u < 2 * U; ++u) {
        g[u] = go[e + 1024 * u];
        i[u] = ib[e + 1024 * u];
      }
#pragma unroll
      for (int u = 0; u < 2 * U; ++u) add1(i[u], (double)g[u]);
    }
    for (; e < P; e += 1024) add1(ib[e], (double)go[e]);
  }
  __syncthreads();
  float* __restrict__ gp = grad_points + ((size_t)b * C + c) * N + lo;
  if (overwrite) {  // (uniform) pp_group_points_grad_out_*: every element is written once, nothing is read
    for (int k = t; k < w; k += 1024) gp[k] = (float)s_col64[k];
  } else {
    for (int k = t; k < w; k += 1024) gp[k] += (float)s_col64[k];  // accumulate: the reference ABI's contract
  }
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
                                                       int tiles_per_b, const pp::GridSet* __restrict__ skip) {
  const int b = blockIdx.x / tiles_per_b;
  if (skip && skip[b].pad[0]) return;  // this batch element was handled by the grid search
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

// three_interpolate, LDS-staged form: like group_points v2 -- the channel row points[b,c,:]
// (M floats) is staged in LDS once per 512-thread workgroup (next row prefetched through
// registers), a thread keeps (idx, weight) of four consecutive n for all C channels and streams
// 16-byte stores.  Same canonical fma order as the gather form.
template <int KR>
__global__ __launch_bounds__(512, 4) void three_interpolate_lds_kernel(const float* __restrict__ points,
                                                                       const int* __restrict__ idx,
                                                                       const float* __restrict__ weight,
                                                                       float* __restrict__ out, int B, int C,
                                                                       int M, int N, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float s_row[];
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / chunks);
  const int chunk = y % chunks;
  if (b >= B) return;
  const int t = threadIdx.x;
  const int n0 = chunk * 2048 + t * 4;  // four consecutive outputs per thread (N % 4 == 0)
  const bool active = n0 < N;
  const int nc = active ? n0 : 0;
  int ii[12];
  float ww[12];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const pp::i4 q = reinterpret_cast<const pp::i4*>(idx + ((size_t)b * N + nc) * 3)[e];
    const pp::f4 w = reinterpret_cast<const pp::f4*>(weight + ((size_t)b * N + nc) * 3)[e];
    ii[4 * e] = q.x; ii[4 * e + 1] = q.y; ii[4 * e + 2] = q.z; ii[4 * e + 3] = q.w;
    ww[4 * e] = w.x; ww[4 * e + 1] = w.y; ww[4 * e + 2] = w.z; ww[4 * e + 3] = w.w;
  }
  const int m4 = M >> 2;
  const pp::f4* __restrict__ row = reinterpret_cast<const pp::f4*>(points + (size_t)b * C * M);
  int ee[KR];
  pp::f4 pre[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    ee[k] = min(t + 512 * k, m4 - 1);
    pre[k] = row[ee[k]];
  }
  for (int c = 0; c < C; ++c) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KR; ++k) reinterpret_cast<pp::f4*>(s_row)[ee[k]] = pre[k];
    __syncthreads();
    const pp::f4* __restrict__ nrow = row + (size_t)(c + 1 < C ? c + 1 : c) * m4;
#pragma unroll
    for (int k = 0; k < KR; ++k) pre[k] = nrow[ee[k]];
    pp::f4 r;
    r.x = __builtin_fmaf(ww[2], s_row[ii[2]], __builtin_fmaf(ww[0], s_row[ii[0]], ww[1] * s_row[ii[1]]));
    r.y = __builtin_fmaf(ww[5], s_row[ii[5]], __builtin_fmaf(ww[3], s_row[ii[3]], ww[4] * s_row[ii[4]]));
    r.z = __builtin_fmaf(ww[8], s_row[ii[8]], __builtin_fmaf(ww[6], s_row[ii[6]], ww[7] * s_row[ii[7]]));
    r.w = __builtin_fmaf(ww[11], s_row[ii[11]], __builtin_fmaf(ww[9], s_row[ii[9]], ww[10] * s_row[ii[10]]));
    if (active) *reinterpret_cast<pp::f4*>(out + ((size_t)b * C + c) * N + n0) = r;
  }
}

template <int KR>
void launch_three_interpolate_lds(const float* points, const int* idx, const float* weight, float* out,
                                  int B, int C, int M, int N, hipStream_t s) {
  const int chunks = (N + 2047) / 2048;
  three_interpolate_lds_kernel<KR><<<dim3((unsigned)(8 * ((B + 7) / 8) * chunks)), dim3(512),
                                     (size_t)M * sizeof(float), s>>>(points, idx, weight, out, B, C, M, N, chunks);
}

// three_interpolate, channel-group form: a workgroup stages CG whole rows points[b, c0..c0+CG, :]
// in LDS once (one barrier) and then walks ALL n, four per thread and step: (idx, weight) of the
// four points are read once for the CG channels, 12 LDS gathers per channel, one 16-byte store per
// channel.  The row-at-a-time form above pays two barriers per channel row for 12 gathers; here
// the only serial part is the staging.  Same canonical fma order.
template <int CG, bool VEC>
__global__ __launch_bounds__(1024) void three_interpolate_rows_kernel(const float* __restrict__ points,
                                                                      const int* __restrict__ idx,
                                                                      const float* __restrict__ weight,
                                                                      float* __restrict__ out, int B, int C,
                                                                      int M, int N) {
  extern __shared__ __attribute__((aligned(16))) float s_irows[];  // [CG][M]
  const int groups = (C + CG - 1) / CG;
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / groups);
  if (b >= B) return;
  const int c0 = (y % groups) * CG;
  const int nc = min(CG, C - c0);
  const int t = threadIdx.x;
  for (int k = 0; k < nc; ++k) {
    const float* __restrict__ rowf = points + ((size_t)b * C + c0 + k) * M;
    if constexpr (VEC) {
      const pp::f4* __restrict__ row = reinterpret_cast<const pp::f4*>(rowf);
      for (int e = t; e < (M >> 2); e += 1024) reinterpret_cast<pp::f4*>(s_irows + (size_t)k * M)[e] = row[e];
    } else {
      for (int e = t; e < M; e += 1024) s_irows[(size_t)k * M + e] = rowf[e];
    }
  }
  __syncthreads();
  if constexpr (VEC) {
    for (int n0 = 4 * t; n0 < N; n0 += 4096) {
      int ii[12];
      float ww[12];
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const pp::i4 q = reinterpret_cast<const pp::i4*>(idx + ((size_t)b * N + n0) * 3)[e];
        const pp::f4 w = reinterpret_cast<const pp::f4*>(weight + ((size_t)b * N + n0) * 3)[e];
        ii[4 * e] = q.x; ii[4 * e + 1] = q.y; ii[4 * e + 2] = q.z; ii[4 * e + 3] = q.w;
        ww[4 * e] = w.x; ww[4 * e + 1] = w.y; ww[4 * e + 2] = w.z; ww[4 * e + 3] = w.w;
      }
#pragma unroll
      for (int k = 0; k < CG; ++k)
        if (k < nc) {
          const float* __restrict__ sr = s_irows + (size_t)k * M;
          pp::f4 r;
          r.x = __builtin_fmaf(ww[2], sr[ii[2]], __builtin_fmaf(ww[0], sr[ii[0]], ww[1] * sr[ii[1]]));
          r.y = __builtin_fmaf(ww[5], sr[ii[5]], __builtin_fmaf(ww[3], sr[ii[3]], ww[4] * sr[ii[4]]));
          r.z = __builtin_fmaf(ww[8], sr[ii[8]], __builtin_fmaf(ww[6], sr[ii[6]], ww[7] * sr[ii[7]]));
          r.w = __builtin_fmaf(ww[11], sr[ii[11]], __builtin_fmaf(ww[9], sr[ii[9]], ww[10] * sr[ii[10]]));
          *reinterpret_cast<pp::f4*>(out + ((size_t)b * C + c0 + k) * N + n0) = r;
        }
    }
  } else {  // no alignment assumed: one n per thread and step, 4-byte accesses (all coalesced)
    for (int n = t; n < N; n += 1024) {
      const int* __restrict__ id = idx + ((size_t)b * N + n) * 3;
      const float* __restrict__ w = weight + ((size_t)b * N + n) * 3;
      const int i0 = id[0], i1 = id[1], i2 = id[2];
      const float w0 = w[0], w1 = w[1], w2 = w[2];
#pragma unroll
      for (int k = 0; k < CG; ++k)
        if (k < nc) {
          const float* __restrict__ sr = s_irows + (size_t)k * M;
          out[((size_t)b * C + c0 + k) * N + n] = __builtin_fmaf(w2, sr[i2], __builtin_fmaf(w0, sr[i0], w1 * sr[i1]));
        }
    }
  }
}

template <int CG, bool VEC>
static int launch_three_interpolate_rows(const float* points, const int* idx, const float* weight, float* out,
                                         int B, int C, int M, int N, hipStream_t s) {
  static pp::DeviceFlags ok;
  const hipError_t e = pp::allow_big_lds(three_interpolate_rows_kernel<CG, VEC>, 152 * 1024, ok);
  if (e != hipSuccess) return (int)e;
  const long long wgs = 8LL * ((B + 7) / 8) * ((C + CG - 1) / CG);
  if (wgs > 0x7fffffffLL) return PP_EINVAL;
  three_interpolate_rows_kernel<CG, VEC><<<dim3((unsigned)wgs), dim3(1024), (size_t)CG * M * sizeof(float), s>>>(
      points, idx, weight, out, B, C, M, N);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
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

// three_interpolate backward with LDS columns in double (see group_points_grad_lds64_kernel for why
// double): one workgroup per (batch, group of CG channels) keeps CG columns grad_points[b,c,:] in
// LDS, reads (idx, weight) of a point once for the CG channels and adds the fp32 products
// grad_out*weight (the reference's rounding, interpolate_gpu.cu:137-139) with ds_add_f64.
template <int CG>
__global__ __launch_bounds__(1024) void three_interpolate_grad_lds64_kernel(
    const float* __restrict__ grad_out, const int* __restrict__ idx, const float* __restrict__ weight,
    float* __restrict__ grad_points, int B, int C, int N, int M) {
  extern __shared__ __attribute__((aligned(16))) double s_acc64[];  // [CG][M]
  const int groups = (C + CG - 1) / CG;
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / groups);
  const int c0 = (y % groups) * CG;
  if (b >= B) return;
  const int nc = min(CG, C - c0);
  const int t = threadIdx.x;
  for (int k = t; k < CG * M; k += 1024) s_acc64[k] = 0.0;
  __syncthreads();
  const float* __restrict__ go = grad_out + ((size_t)b * C + c0) * N;
  for (int n = t; n < N; n += 1024) {
    const int* id = idx + ((size_t)b * N + n) * 3;
    const float* w = weight + ((size_t)b * N + n) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    float g[CG];
#pragma unroll
    for (int k = 0; k < CG; ++k) g[k] = go[(size_t)min(k, nc - 1) * N + n];
#pragma unroll
    for (int k = 0; k < CG; ++k)
      if (k < nc) {
        double* col = s_acc64 + (size_t)k * M;
        atomicAdd(col + i0, (double)(g[k] * w0));
        atomicAdd(col + i1, (double)(g[k] * w1));
        atomicAdd(col + i2, (double)(g[k] * w2));
      }
  }
  __syncthreads();
  for (int k = 0; k < nc; ++k) {
    float* __restrict__ gp = grad_points + ((size_t)b * C + c0 + k) * M;
    for (int m = t; m < M; m += 1024) gp[m] += (float)s_acc64[(size_t)k * M + m];  // accumulate: the ABI's contract
  }
}

template <int CG>
static int launch_three_interpolate_grad_lds64(const float* grad_out, const int* idx, const float* weight,
                                               float* grad_points, int B, int C, int N, int M, hipStream_t s) {
  static pp::DeviceFlags ok;
  const hipError_t e = pp::allow_big_lds(three_interpolate_grad_lds64_kernel<CG>, 152 * 1024, ok);
  if (e != hipSuccess) return (int)e;
  const long long wgs = 8LL * ((B + 7) / 8) * ((C + CG - 1) / CG);
  if (wgs > 0x7fffffffLL) return PP_EINVAL;
  three_interpolate_grad_lds64_kernel<CG><<<dim3((unsigned)wgs), dim3(1024), (size_t)CG * M * sizeof(double), s>>>(
      grad_out, idx, weight, grad_points, B, C, N, M);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
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

// 0 = automatic; 1 = force the global-gather kernel (tests and tuning)
static pp::Knob g_gather_variant;
extern "C" void pp_debug_set_gather_variant(int v) { g_gather_variant.set(v); }

extern "C" int pp_gather_forward_f32(const float* points, const int* idx, float* out, int B, int C,
                                     int N, int M, void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || M == 0) return PP_OK;
  if (!points || !idx || !out || N == 0) return PP_EINVAL;
  // LDS-staged rows: the row fits, reading whole rows pays (N <= 16 M); 16-byte variant when rows and
  // index quads are aligned, 4-byte variant otherwise
  if (g_gather_variant != 1 && (size_t)N * sizeof(float) <= 152 * 1024 && (long long)N <= 16LL * M &&
      (long long)B * C >= 512 && M >= 1024) {
    const bool vec = N % 4 == 0 && M % 4 == 0 && (uintptr_t)points % 16 == 0 && (uintptr_t)idx % 16 == 0 &&
                     (uintptr_t)out % 16 == 0 && N <= 16384;
    int cpb = 4;
    while (cpb > 1 && 8LL * ((B + 7) / 8) * ((C + cpb - 1) / cpb) < 1024) cpb /= 2;
    const int groups = (C + cpb - 1) / cpb;
    const long long wgs = 8LL * ((B + 7) / 8) * groups;
    if (wgs <= 0x7fffffffLL) {
      const dim3 grid((unsigned)wgs), block(1024);
      const size_t lds = (size_t)N * sizeof(float);
      hipStream_t s = (hipStream_t)stream;
      const int n4 = N / 4;
      static pp::DeviceFlags ok1, ok2, ok4, oks;
      if (!vec) {
        if (pp::allow_big_lds(gather_fwd_lds_kernel<1, false>, 152 * 1024, oks) != hipSuccess) return PP_EINVAL;
        gather_fwd_lds_kernel<1, false><<<grid, block, lds, s>>>(points, idx, out, B, C, N, M, cpb, groups);
      } else if (n4 <= 1024) {
        if (pp::allow_big_lds(gather_fwd_lds_kernel<1, true>, 64 * 1024, ok1) != hipSuccess) return PP_EINVAL;
        gather_fwd_lds_kernel<1, true><<<grid, block, lds, s>>>(points, idx, out, B, C, N, M, cpb, groups);
      } else if (n4 <= 2048) {
        if (pp::allow_big_lds(gather_fwd_lds_kernel<2, true>, 64 * 1024, ok2) != hipSuccess) return PP_EINVAL;
        gather_fwd_lds_kernel<2, true><<<grid, block, lds, s>>>(points, idx, out, B, C, N, M, cpb, groups);
      } else {
        if (pp::allow_big_lds(gather_fwd_lds_kernel<4, true>, 64 * 1024, ok4) != hipSuccess) return PP_EINVAL;
        gather_fwd_lds_kernel<4, true><<<grid, block, lds, s>>>(points, idx, out, B, C, N, M, cpb, groups);
      }
      PP_RETURN_IF_LAUNCH_FAILED();
      return PP_OK;
    }
  }
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

// 0 = automatic; 1 = force the one-wave-per-64-centres kernel (tests and tuning)
static pp::Knob g_ball_variant;
extern "C" void pp_debug_set_ball_query_variant(int v) { g_ball_variant.set(v); }

static int ball_query_launch(const float* new_xyz, const float* xyz, int* idx, int B, int N, int M,
                             float radius, int nsample, const pp::GridSet* skip, void* stream) {
  if (B < 0 || N < 0 || M < 0 || nsample < 0) return PP_EINVAL;
  if (B == 0 || M == 0 || nsample == 0) return PP_OK;
  if (!new_xyz || !idx || (N > 0 && !xyz)) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float radius2 = radius * radius;  // fp32, as the reference (sampling_cuda.cu:354)
  // split form: 4 waves share 64 centres (one cloud quarter each); needs its lists in LDS and
  // enough points for four quarters
  if (g_ball_variant != 1 && N >= 4 * kBqGroup) {
    const int tiles64 = (M + 63) / 64;
    const long long blocks64 = (long long)B * tiles64;
    const size_t esz = N <= 65536 ? sizeof(unsigned short) : sizeof(unsigned);
    const size_t lds64 = (size_t)4 * 64 * nsample * esz + (size_t)4 * 64 * sizeof(int);
    if (blocks64 <= 0x7fffffffLL && lds64 <= 64 * 1024) {
      if (esz == 2)
        ball_query_split_kernel<unsigned short><<<dim3((unsigned)blocks64), dim3(256), lds64, s>>>(
            new_xyz, xyz, idx, N, M, radius2, nsample, tiles64, skip);
      else
        ball_query_split_kernel<unsigned><<<dim3((unsigned)blocks64), dim3(256), lds64, s>>>(
            new_xyz, xyz, idx, N, M, radius2, nsample, tiles64, skip);
      PP_RETURN_IF_LAUNCH_FAILED();
      return PP_OK;
    }
  }
  const int tiles = (M + 255) / 256;
  const long long blocks = (long long)B * tiles;
  if (blocks > 0x7fffffffLL) return PP_EINVAL;
  const size_t lds = (size_t)4 * 64 * nsample * sizeof(int);
  if (lds <= 64 * 1024) {
    ball_query_kernel<true><<<dim3((unsigned)blocks), dim3(256), lds, s>>>(
        new_xyz, xyz, idx, N, M, radius2, nsample, tiles, skip);
  } else {
    ball_query_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
        new_xyz, xyz, idx, N, M, radius2, nsample, tiles, skip);
  }
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_ball_query_f32(const float* new_xyz, const float* xyz, int* idx, int B, int N,
                                 int M, float radius, int nsample, void* stream) {
  return ball_query_launch(new_xyz, xyz, idx, B, N, M, radius, nsample, nullptr, stream);
}

namespace pp {
int ball_query_scan_unusable(const float* new_xyz, const float* xyz, int* idx, int B, int N, int M,
                             float radius, int nsample, const GridSet* sets, hipStream_t s) {
  return ball_query_launch(new_xyz, xyz, idx, B, N, M, radius, nsample, sets, (void*)s);
}
}  // namespace pp

// 0 = automatic; 1 = force the global-gather kernel; 2/4/8 = force the LDS-staged kernel with that
// many index quads per thread (tests and tuning)
static pp::Knob g_group_variant;
extern "C" void pp_debug_set_group_points_variant(int v) { g_group_variant.set(v); }

extern "C" int pp_group_points_f32(const float* points, const int* idx, float* out, int B, int C,
                                   int N, int npoint, int nsample, void* stream) {
  return pp_group_points_strided_f32(points, idx, out, B, C, N, npoint, nsample,
                                     (long long)C * npoint * nsample, stream);
}

extern "C" int pp_group_points_strided_f32(const float* points, const int* idx, float* out, int B, int C,
                                           int N, int npoint, int nsample, long long out_batch_stride,
                                           void* stream) {
  if (B < 0 || C < 0 || N < 0 || npoint < 0 || nsample < 0) return PP_EINVAL;
  const long long P = (long long)npoint * nsample;
  const long long obs = out_batch_stride;
  if (obs < (long long)C * P) return PP_EINVAL;
  if (B == 0 || C == 0 || P == 0) return PP_OK;
  if (!points || !idx || !out || N == 0) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool vec4 = (P % 4 == 0) && (obs % 4 == 0) && ((uintptr_t)idx % 16 == 0) && ((uintptr_t)out % 16 == 0);
  // LDS-staged form: row fits 64 KiB (two workgroups per CU), rows 16-byte aligned, enough channels
  // to amortise keeping the indices in registers, enough positions to fill the chip
  if (g_group_variant != 1 && vec4 && N % 4 == 0 && (uintptr_t)points % 16 == 0 &&
      (size_t)N * sizeof(float) <= 64 * 1024 && C >= 4 && (long long)B * P >= 256LL * 2048) {
    const long long per_cu = (long long)B * P / 512;  // positions per half-CU
    bool ok = false;
    // the LDS-DMA form wherever the positions fill whole chunks (16, 8 or 4 index quads per thread): measured round 6
    // at or ahead of every other form on every shape it can take (tools/group_shapes_time.py); the output is written
    // with non-temporal stores (4 GiB at config 4, nothing of it is read back from the caches; 516: ordinary stores)
    const int gv = g_group_variant;
    if (gv == 516) ok = launch_group_dma1<16, false>(points, idx, out, B, C, N, P, obs, s);
    if (!ok && (gv == 616 || gv == 0)) ok = launch_group_dma1<16, true>(points, idx, out, B, C, N, P, obs, s);
    if (!ok && (gv == 608 || gv == 0)) ok = launch_group_dma1<8, true>(points, idx, out, B, C, N, P, obs, s);
    if (!ok && (gv == 604 || gv == 0)) ok = launch_group_dma1<4, true>(points, idx, out, B, C, N, P, obs, s);
    if (ok) {
      PP_RETURN_IF_LAUNCH_FAILED();
      return PP_OK;
    }
    // ragged chunks: the VGPR-staged form, 8 / 4 / 2 index quads per thread by the positions a CU gets
    if (g_group_variant == 8 || ((g_group_variant == 0 || g_group_variant > 100) && per_cu >= 512LL * 4 * 8))
      ok = launch_group_lds<8>(points, idx, out, B, C, N, P, obs, s);
    else if (g_group_variant == 4 || (g_group_variant == 0 && per_cu >= 512LL * 4 * 4))
      ok = launch_group_lds<4>(points, idx, out, B, C, N, P, obs, s);
    else
      ok = launch_group_lds<2>(points, idx, out, B, C, N, P, obs, s);
    if (ok) {
      PP_RETURN_IF_LAUNCH_FAILED();
      return PP_OK;
    }
  }
  // unaligned shapes, or rows beyond 64 KiB: the 4-byte LDS-staged form -- from 2^28 output elements (measured round 6:
  // 1.89 against 7.7 ms for the global gather at config 4 less one point and one sample, but 0.27 against 0.21 at a
  // fifth of a GiB and 0.079 against 0.050 at 69 MB: tools/group_shapes_time.py)
  if (g_group_variant != 1 && (size_t)N * sizeof(float) <= 152 * 1024 && C >= 4 &&
      (long long)B * P >= 256LL * 2048 && (g_group_variant != 0 || (long long)B * P * C >= (1LL << 28))) {
    constexpr int V = 16;
    const long long chunks = (P + 1024LL * V - 1) / (1024LL * V);
    const long long blocks = 8LL * ((B + 7) / 8) * chunks;
    if (chunks <= 0x7fffffLL && blocks <= 0x7fffffffLL) {
      static pp::DeviceFlags ok_s;
      const hipError_t e = pp::allow_big_lds(group_points_lds_scalar_kernel<V>, 152 * 1024, ok_s);
      if (e != hipSuccess) return (int)e;
      group_points_lds_scalar_kernel<V><<<dim3((unsigned)blocks), dim3(1024), (size_t)N * sizeof(float), s>>>(
          points, idx, out, B, C, N, P, (int)chunks, obs);
      PP_RETURN_IF_LAUNCH_FAILED();
      return PP_OK;
    }
  }
  const long long threads = vec4 ? P / 4 : P;
  const long long cols = (threads + 255) / 256;
  const int cpb = pick_c_per_block(cols, B, C);
  const long long gy = (C + cpb - 1) / cpb;
  if (!grid_ok(cols, gy, B)) return PP_EINVAL;
  const dim3 grid((unsigned)cols, (unsigned)gy, (unsigned)B);
  if (vec4)
    group_points_kernel<true><<<grid, dim3(256), 0, s>>>(points, idx, out, C, N, P, cpb, obs);
  else
    group_points_kernel<false><<<grid, dim3(256), 0, s>>>(points, idx, out, C, N, P, cpb, obs);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

// 0 = automatic; 1 = force global atomics; 2 = force the LDS-column form (double column when it
// fits); 3 = force the LDS-column form with the fp32 column (tests and tuning)
static pp::Knob g_group_grad_variant;
extern "C" void pp_debug_set_group_points_grad_variant(int v) { g_group_grad_variant.set(v); }

extern "C" int pp_group_points_grad_f32(const float* grad_out, const int* idx, float* grad_points,
                                        int B, int C, int N, int npoint, int nsample,
                                        void* stream) {
  return pp_group_points_grad_strided_f32(grad_out, idx, grad_points, B, C, N, npoint, nsample,
                                          (long long)C * npoint * nsample, stream);
}

static int group_points_grad_launch(const float* grad_out, const int* idx, float* grad_points, int B, int C, int N,
                                    int npoint, int nsample, long long grad_out_batch_stride, void* stream,
                                    bool overwrite);
extern "C" int pp_group_points_grad_strided_f32(const float* grad_out, const int* idx, float* grad_points,
                                                int B, int C, int N, int npoint, int nsample,
                                                long long grad_out_batch_stride, void* stream) {
  return group_points_grad_launch(grad_out, idx, grad_points, B, C, N, npoint, nsample, grad_out_batch_stride, stream,
                                  false);
}
// overwrite: grad_points need not be zero on entry -- the LDS-column form writes every element once (no read of the
// output, no fill in front of it: 2 x 256 MB less traffic at config 4); the other forms zero it themselves first
static int group_points_grad_launch(const float* grad_out, const int* idx, float* grad_points, int B, int C, int N,
                                    int npoint, int nsample, long long grad_out_batch_stride, void* stream,
                                    bool overwrite) {
  if (B < 0 || C < 0 || N < 0 || npoint < 0 || nsample < 0) return PP_EINVAL;
  const long long P = (long long)npoint * nsample;
  const long long gbs = grad_out_batch_stride;
  if (gbs < (long long)C * P) return PP_EINVAL;
  if (B == 0 || C == 0 || P == 0) return PP_OK;
  if (!grad_out || !idx || !grad_points || N == 0) return PP_EINVAL;
  // LDS-column forms: 16-byte aligned streams, enough work per column
  const bool vec = (uintptr_t)grad_out % 16 == 0 && (uintptr_t)idx % 16 == 0 && P % 4 == 0 && gbs % 4 == 0;
  if (g_group_grad_variant != 1 && (vec || g_group_grad_variant != 3) &&
      8LL * ((B + 7) / 8) * C <= 0x7fffffffLL && (g_group_grad_variant >= 2 || P >= 4096)) {
    constexpr int kW64 = 152 * 1024 / 8;  // destinations per workgroup with a double column
    const int nsplit = (N + kW64 - 1) / kW64;
    const long long wgs = 8LL * ((B + 7) / 8) * C * nsplit;
    if (g_group_grad_variant != 3 && nsplit <= 16 && wgs <= 0x7fffffffLL) {
      const int W = nsplit == 1 ? N : kW64;
      static pp::DeviceFlags lds64_ok, lds64s_ok;
      if (vec) {
        const hipError_t e = pp::allow_big_lds(group_points_grad_lds64_kernel<8, true>, 152 * 1024, lds64_ok);
        if (e != hipSuccess) return (int)e;
        group_points_grad_lds64_kernel<8, true><<<dim3((unsigned)wgs), dim3(1024), (size_t)W * sizeof(double),
                                                  (hipStream_t)stream>>>(grad_out, idx, grad_points, B, C, N, P,
                                                                         gbs, nsplit, W, overwrite ? 1 : 0);
      } else {
        const hipError_t e = pp::allow_big_lds(group_points_grad_lds64_kernel<8, false>, 152 * 1024, lds64s_ok);
        if (e != hipSuccess) return (int)e;
        group_points_grad_lds64_kernel<8, false><<<dim3((unsigned)wgs), dim3(1024), (size_t)W * sizeof(double),
                                                   (hipStream_t)stream>>>(grad_out, idx, grad_points, B, C, N, P,
                                                                          gbs, nsplit, W, overwrite ? 1 : 0);
      }
      PP_RETURN_IF_LAUNCH_FAILED();
      return PP_OK;
    }
    if (overwrite) {  // (the forms below accumulate)
      const hipError_t e = hipMemsetAsync(grad_points, 0, (size_t)B * C * N * sizeof(float), (hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
      overwrite = false;
    }
    if (vec && (size_t)N * sizeof(float) <= 160 * 1024) {  // fp32 column (kept for comparison: ds_add_f32 is slow)
      static pp::DeviceFlags lds_ok;
      const hipError_t e = pp::allow_big_lds(group_points_grad_lds_kernel, 160 * 1024, lds_ok);
      if (e != hipSuccess) return (int)e;
      group_points_grad_lds_kernel<<<dim3((unsigned)(8 * ((B + 7) / 8) * C)), dim3(1024),
                                     (size_t)N * sizeof(float), (hipStream_t)stream>>>(
          grad_out, idx, grad_points, B, C, N, P, gbs);
      PP_RETURN_IF_LAUNCH_FAILED();
      return PP_OK;
    }
  }
  if (overwrite) {
    const hipError_t e = hipMemsetAsync(grad_points, 0, (size_t)B * C * N * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  const long long cols = (P + 255) / 256;
  const int cpb = pick_c_per_block(cols, B, C);
  const long long gy = (C + cpb - 1) / cpb;
  if (!grid_ok(cols, gy, B)) return PP_EINVAL;
  group_points_grad_kernel<<<dim3((unsigned)cols, (unsigned)gy, (unsigned)B), dim3(256), 0,
                             (hipStream_t)stream>>>(grad_out, idx, grad_points, C, N, P, cpb, gbs);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

static int three_nn_launch(const float* unknown, const float* known, float* dist2, int* idx, int B, int N,
                           int M, const pp::GridSet* skip, void* stream) {
  if (B < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || N == 0) return PP_OK;
  if (!unknown || !dist2 || !idx || (M > 0 && !known)) return PP_EINVAL;
  const int tiles = (N + 255) / 256;
  const long long blocks = (long long)B * tiles;
  if (blocks > 0x7fffffffLL) return PP_EINVAL;
  three_nn_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
      unknown, known, dist2, idx, N, M, tiles, skip);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_three_nn_f32(const float* unknown, const float* known, float* dist2, int* idx,
                               int B, int N, int M, void* stream) {
  return three_nn_launch(unknown, known, dist2, idx, B, N, M, nullptr, stream);
}

// 0 = automatic; 1 = force the global-gather kernel; 2 = no channel-group form (row-at-a-time LDS
// form where it applies)   (tests and tuning)
static pp::Knob g_interp_variant;
extern "C" void pp_debug_set_three_interpolate_variant(int v) { g_interp_variant.set(v); }

extern "C" int pp_three_interpolate_f32(const float* points, const int* idx, const float* weight,
                                        float* out, int B, int C, int M, int N, void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || N == 0) return PP_OK;
  if (!points || !idx || !weight || !out || M == 0) return PP_EINVAL;
  // channel-group form (rows staged once, §three_interpolate_rows_kernel): as many whole rows per
  // workgroup as keep two workgroups on a CU (4 x 16 KiB at M = 4096), fewer for long rows; the 16-byte
  // variant when everything is aligned, the 4-byte variant otherwise
  hipStream_t s = (hipStream_t)stream;
  const bool vec = M % 4 == 0 && N % 4 == 0 && (uintptr_t)points % 16 == 0 && (uintptr_t)idx % 16 == 0 &&
                   (uintptr_t)weight % 16 == 0 && (uintptr_t)out % 16 == 0;
  const size_t rowb = (size_t)M * sizeof(float);
  if (g_interp_variant != 1 && g_interp_variant != 2 && rowb <= 152 * 1024 && N >= 2048 &&
      8LL * ((B + 7) / 8) * ((C + 3) / 4) >= 256) {
    const int cg = (4 * rowb <= 64 * 1024 && C >= 4) ? 4 : ((2 * rowb <= 152 * 1024 && C >= 2) ? 2 : 1);
    if (vec) {
      if (cg == 4) return launch_three_interpolate_rows<4, true>(points, idx, weight, out, B, C, M, N, s);
      if (cg == 2) return launch_three_interpolate_rows<2, true>(points, idx, weight, out, B, C, M, N, s);
      return launch_three_interpolate_rows<1, true>(points, idx, weight, out, B, C, M, N, s);
    }
    if (cg == 4) return launch_three_interpolate_rows<4, false>(points, idx, weight, out, B, C, M, N, s);
    if (cg == 2) return launch_three_interpolate_rows<2, false>(points, idx, weight, out, B, C, M, N, s);
    return launch_three_interpolate_rows<1, false>(points, idx, weight, out, B, C, M, N, s);
  }
  // row-at-a-time LDS form: 16-byte aligned rows and quads, row within 64 KiB, enough channels to amortise
  if (g_interp_variant != 1 && vec && M <= 16384 && C >= 4 && (long long)B * N >= 64 * 2048) {
    const int m4 = M / 4;
    if (m4 <= 512) launch_three_interpolate_lds<1>(points, idx, weight, out, B, C, M, N, s);
    else if (m4 <= 1024) launch_three_interpolate_lds<2>(points, idx, weight, out, B, C, M, N, s);
    else if (m4 <= 2048) launch_three_interpolate_lds<4>(points, idx, weight, out, B, C, M, N, s);
    else launch_three_interpolate_lds<8>(points, idx, weight, out, B, C, M, N, s);
    PP_RETURN_IF_LAUNCH_FAILED();
    return PP_OK;
  }
  const long long cols = (N + 255) / 256;
  const int cpb = pick_c_per_block(cols, B, C);
  const long long gy = (C + cpb - 1) / cpb;
  if (!grid_ok(cols, gy, B)) return PP_EINVAL;
  three_interpolate_kernel<<<dim3((unsigned)cols, (unsigned)gy, (unsigned)B), dim3(256), 0,
                             (hipStream_t)stream>>>(points, idx, weight, out, C, M, N, cpb);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

// 0 = automatic; 1 = force global atomics; 2 = force the LDS-column form and, in the _ws entry point,
// skip the sorted form; 3 = the _ws entry point prefers the sorted form (tests and tuning)
static pp::Knob g_interp_grad_variant;
extern "C" void pp_debug_set_three_interpolate_grad_variant(int v) { g_interp_grad_variant.set(v); }

extern "C" int pp_three_interpolate_grad_f32(const float* grad_out, const int* idx,
                                             const float* weight, float* grad_points, int B,
                                             int C, int N, int M, void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || N == 0) return PP_OK;
  if (!grad_out || !idx || !weight || !grad_points || M == 0) return PP_EINVAL;
  // LDS columns in double: the column(s) fit, enough points per column to pay for zeroing/flushing it
  const size_t col = (size_t)M * sizeof(double);
  if (g_interp_grad_variant != 1 && g_interp_grad_variant != 3 && col <= 152 * 1024 &&
      (g_interp_grad_variant == 2 || N >= 2048)) {
    hipStream_t s = (hipStream_t)stream;
    if (C >= 4 && 4 * col <= 152 * 1024)
      return launch_three_interpolate_grad_lds64<4>(grad_out, idx, weight, grad_points, B, C, N, M, s);
    if (C >= 2 && 2 * col <= 152 * 1024)
      return launch_three_interpolate_grad_lds64<2>(grad_out, idx, weight, grad_points, B, C, N, M, s);
    return launch_three_interpolate_grad_lds64<1>(grad_out, idx, weight, grad_points, B, C, N, M, s);
  }
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

// ---- scatter-add backwards with a caller-provided workspace: sorted triples instead of atomics ----
// (scatter.hip).  Each falls back to its atomic form when the workspace is absent / too small or
// the problem does not qualify (destinations per batch element > 20480, tiny problems).
static pp::Knob g_scatter_mode;  // 0 = automatic; 1 = never use the sorted form (tests and tuning)
extern "C" void pp_debug_set_scatter_mode(int v) { g_scatter_mode.set(v); }

extern "C" size_t pp_scatter_workspace_bytes(int B, long long triples_per_batch, int destinations,
                                             int per_source, int weighted) {
  if (per_source < 1) return 0;
  return pp::ssa_workspace_bytes(B, triples_per_batch, per_source, destinations, weighted != 0);
}

static bool scatter_ok(int B, int C, long long P, int R, int Nd, int weighted, const void* ws, size_t bytes) {
  if (g_scatter_mode == 1 || !ws || C < 1) return false;
  if ((long long)B * P * C < (1LL << 20)) return false;  // too small to pay for the sort
  const size_t need = pp::ssa_workspace_bytes(B, P, R, Nd, weighted != 0);
  return need != 0 && bytes >= need;
}

extern "C" int pp_group_points_grad_ws_f32(const float* grad_out, const int* idx, float* grad_points, int B,
                                           int C, int N, int npoint, int nsample,
                                           long long grad_out_batch_stride, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  const long long P = (long long)npoint * nsample;
  // With many triples per destination (16 at config 4) the sorted form moves more bytes through each
  // CU than the LDS-column form spends on its slow ds_add_f32 (6.3 vs 5.8 ms there); it wins when the
  // lists are short.
  if (B > 0 && C > 0 && P > 0 && grad_out && idx && grad_points && N > 0 && P <= 4LL * N &&
      grad_out_batch_stride >= (long long)C * P && scatter_ok(B, C, P, 1, N, 0, workspace, workspace_bytes))
    return pp::ssa_run(grad_out, idx, nullptr, grad_points, B, C, P, 1, N, grad_out_batch_stride, workspace,
                       (hipStream_t)stream, false);
  return pp_group_points_grad_strided_f32(grad_out, idx, grad_points, B, C, N, npoint, nsample,
                                          grad_out_batch_stride, stream);
}

// the same with grad_points WRITTEN, not accumulated into: the caller passes uninitialised memory (the Python shim:
// torch.empty instead of torch.zeros)
extern "C" int pp_group_points_grad_out_ws_f32(const float* grad_out, const int* idx, float* grad_points, int B,
                                               int C, int N, int npoint, int nsample,
                                               long long grad_out_batch_stride, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  const long long P = (long long)npoint * nsample;
  if (B < 0 || C < 0 || N < 0 || npoint < 0 || nsample < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || N == 0) return PP_OK;
  if (!grad_points) return PP_EINVAL;
  if (P == 0) return (int)hipMemsetAsync(grad_points, 0, (size_t)B * C * N * sizeof(float), (hipStream_t)stream);
  if (grad_out && idx && P <= 4LL * N && grad_out_batch_stride >= (long long)C * P &&
      scatter_ok(B, C, P, 1, N, 0, workspace, workspace_bytes)) {  // the sorted form accumulates
    const hipError_t e = hipMemsetAsync(grad_points, 0, (size_t)B * C * N * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return pp::ssa_run(grad_out, idx, nullptr, grad_points, B, C, P, 1, N, grad_out_batch_stride, workspace,
                       (hipStream_t)stream, false);
  }
  return group_points_grad_launch(grad_out, idx, grad_points, B, C, N, npoint, nsample, grad_out_batch_stride, stream,
                                  true);
}

extern "C" int pp_gather_backward_ws_f32(const float* grad_out, const int* idx, float* grad_points, int B,
                                         int C, int N, int M, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  if (B > 0 && C > 0 && M > 0 && N > 0 && grad_out && idx && grad_points &&
      scatter_ok(B, C, M, 1, N, 0, workspace, workspace_bytes))
    return pp::ssa_run(grad_out, idx, nullptr, grad_points, B, C, M, 1, N, (long long)C * M, workspace,
                       (hipStream_t)stream, false);
  return pp_gather_backward_f32(grad_out, idx, grad_points, B, C, N, M, stream);
}

extern "C" int pp_three_interpolate_grad_ws_f32(const float* grad_out, const int* idx, const float* weight,
                                                float* grad_points, int B, int C, int N, int M,
                                                void* workspace, size_t workspace_bytes, void* stream) {
  // the LDS-column form (0.18 ms at B=32, C=128, N=16384, M=4096) beats the sorted form (0.55 ms)
  // wherever its column fits
  const bool columns = g_interp_grad_variant != 1 && g_interp_grad_variant != 3 &&
                       (size_t)M * sizeof(double) <= 152 * 1024 && (g_interp_grad_variant == 2 || N >= 2048);
  if (!columns && B > 0 && C > 0 && N > 0 && M > 0 && grad_out && idx && weight && grad_points &&
      scatter_ok(B, C, 3LL * N, 3, M, 1, workspace, workspace_bytes))
    return pp::ssa_run(grad_out, idx, weight, grad_points, B, C, 3LL * N, 3, M, (long long)C * N, workspace,
                       (hipStream_t)stream, false);
  return pp_three_interpolate_grad_f32(grad_out, idx, weight, grad_points, B, C, N, M, stream);
}

// ---- ORDERED scatter-add backwards (deterministic mode) ------------------------------------------------
// The sorted-triples form of scatter.hip with every destination's terms added in ascending source order: no
// atomics on floating-point data, results identical from run to run and identical, bit for bit, to a
// sequential loop in the order of the reference's launch (the CPU oracle).  Selected by the Python / C++ host
// side when torch.are_deterministic_algorithms_enabled().  PP_ENOTSUP when the problem does not fit the form
// (more than 20480 destinations per batch element, workspace too small): there is no deterministic substitute.
static bool ordered_ok(int B, long long P, int R, int Nd, int weighted, const void* ws, size_t bytes) {
  const size_t need = pp::ssa_workspace_bytes(B, P, R, Nd, weighted != 0);
  return ws && need != 0 && bytes >= need;
}

extern "C" int pp_group_points_grad_ordered_f32(const float* grad_out, const int* idx, float* grad_points, int B,
                                                int C, int N, int npoint, int nsample,
                                                long long grad_out_batch_stride, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  if (B < 0 || C < 0 || N < 0 || npoint < 0 || nsample < 0) return PP_EINVAL;
  const long long P = (long long)npoint * nsample;
  if (B == 0 || C == 0 || N == 0 || P == 0) return PP_OK;
  if (!grad_out || !idx || !grad_points || grad_out_batch_stride < (long long)C * P) return PP_EINVAL;
  if (!ordered_ok(B, P, 1, N, 0, workspace, workspace_bytes)) return PP_ENOTSUP;
  return pp::ssa_run(grad_out, idx, nullptr, grad_points, B, C, P, 1, N, grad_out_batch_stride, workspace,
                     (hipStream_t)stream, true);
}

extern "C" int pp_gather_backward_ordered_f32(const float* grad_out, const int* idx, float* grad_points, int B,
                                              int C, int N, int M, void* workspace, size_t workspace_bytes,
                                              void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || N == 0 || M == 0) return PP_OK;
  if (!grad_out || !idx || !grad_points) return PP_EINVAL;
  if (!ordered_ok(B, M, 1, N, 0, workspace, workspace_bytes)) return PP_ENOTSUP;
  return pp::ssa_run(grad_out, idx, nullptr, grad_points, B, C, M, 1, N, (long long)C * M, workspace,
                     (hipStream_t)stream, true);
}

extern "C" int pp_three_interpolate_grad_ordered_f32(const float* grad_out, const int* idx, const float* weight,
                                                     float* grad_points, int B, int C, int N, int M,
                                                     void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return PP_EINVAL;
  if (B == 0 || C == 0 || N == 0 || M == 0) return PP_OK;
  if (!grad_out || !idx || !weight || !grad_points) return PP_EINVAL;
  if (!ordered_ok(B, 3LL * N, 3, M, 1, workspace, workspace_bytes)) return PP_ENOTSUP;
  return pp::ssa_run(grad_out, idx, weight, grad_points, B, C, 3LL * N, 3, M, (long long)C * N, workspace,
                     (hipStream_t)stream, true);
}
