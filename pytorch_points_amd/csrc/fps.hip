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
#include "fps_common.h"

// per-step phase marks of the cluster kernel (tools/fps_probe.hip accumulates clocks; nothing otherwise)
#ifndef PP_FPS_MARK
#define PP_FPS_MARK(n)
#define PP_FPS_MARK_END()
#endif

namespace {

using pp::dist3;
using namespace ppfps;

constexpr int kFpsThreads = 1024;
constexpr int kFpsWaves = kFpsThreads / 64;

// R > 0: temp of this thread's R points in registers.  R == 0: temp stays in global memory
// (any N); slower, only used when N > 1024 * 64.
template <int R>
__global__ __launch_bounds__(kFpsThreads) void fps_block_kernel(const float* __restrict__ xyz,
                                                                float* __restrict__ temp,
                                                                int* __restrict__ idx, int N,
                                                                int npoint, int seed,
                                                                TieOrder order, float* __restrict__ sampled, int cf) {
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
  // (sampled != nullptr: the coordinates of the picks as well -- furthest_point_sample's gather_points, fused: the
  //  coordinates of pick j - 1 are in registers at the top of step j)
  float* __restrict__ smp = sampled ? sampled + (size_t)blockIdx.x * npoint * 3 : nullptr;
  auto put = [&](int j, float x, float y, float z) {
    if (cf) {
      smp[j] = x; smp[(size_t)npoint + j] = y; smp[2 * (size_t)npoint + j] = z;
    } else {
      smp[3 * (size_t)j] = x; smp[3 * (size_t)j + 1] = y; smp[3 * (size_t)j + 2] = z;
    }
  };
  for (int j = 1; j < npoint; ++j) {
    const float ox = p[3 * (size_t)old + 0], oy = p[3 * (size_t)old + 1], oz = p[3 * (size_t)old + 2];
    if (smp && t == 0) put(j - 1, ox, oy, oz);
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
  if (smp && t == 0) put(npoint - 1, p[3 * (size_t)old + 0], p[3 * (size_t)old + 1], p[3 * (size_t)old + 2]);
  if (R > 0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int k = t + kFpsThreads * i;
      if (k < N) tmp[k] = td[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// v2: a CLUSTER of CL workgroups (one per CU) per batch element.  Each workgroup keeps its slice
// of the cloud -- coordinates AND running minima -- in registers for the whole call (R points per
// thread, 512 threads), so a step touches no memory except:
//   * one 8-byte granule per workgroup per step, {float_bits(d2) : 32 | ~tie_rank : 24 | tag : 8},
//     published with an agent-scope relaxed store (sc1, write-through) into a 2-deep ring indexed
//     by step parity, and polled by CL lanes of wave 0 of every workgroup of the cluster with
//     agent-scope relaxed loads until all CL tags equal the step's tag;
//   * the coordinates of the winner, read from the immutable input (scalar load, L2).
// The granule IS the flag (no separate flag, no fence: nothing else is handed over).  Two ring
// slots suffice: a workgroup can publish step j+2 only after it has read every member's step j+1
// granule, which that member publishes only after it has read all of step j.  The ring is reset
// by a hipMemsetAsync node in front of every launch (tag 0xFF never matches: tags are 7 bits).
// (Tried and rejected: publishing the winner's coordinates with the key -- four granules per member
// and 64 polling lanes -- to drop the dependent scalar load of x_old: 1.78 -> 2.50 us per pick.)
// Correctness does not depend on placement or dispatch order; it needs the B*CL workgroups to be
// co-resident, which the launcher guarantees by keeping B*CL <= 256 (one 512-thread workgroup per
// CU).  Every spin is bounded (2 s of s_memrealtime): on timeout the workgroup raises the error
// word at the head of the workspace and leaves.
// ------------------------------------------------------------------------------------------------
constexpr int kClThreads = 512;
constexpr int kClWaves = kClThreads / 64;
constexpr unsigned long long kSpinLimitTicks = 200000000ull;  // 2 s at 100 MHz

typedef __attribute__((address_space(1))) u64 gu64;

struct ClusterGeom {
  int cl;        // workgroups per batch element (power of two, <= 64)
  int slice;     // points per workgroup (multiple of kClThreads)
  int per_xcd8;  // cl * ceil(B / 8): blocks that share blockIdx % 8
};

template <int R>
__global__ __launch_bounds__(kClThreads) void fps_cluster_kernel(
    const float* __restrict__ xyz, float* __restrict__ temp, int* __restrict__ idx, int B, int N,
    int npoint, int seed, TieOrder order, ClusterGeom geo, u64* __restrict__ ring,
    unsigned* __restrict__ err, float* __restrict__ sampled, int cf) {
  __shared__ u64 s_key[2][kClWaves];
  __shared__ int s_old[2];
  // members of one batch element share blockIdx % 8 (one XCD under round-robin dispatch): speed only
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int b = x + 8 * (y / geo.cl);
  const int c = y % geo.cl;
  if (b >= B) return;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  float* __restrict__ tmp = temp + (size_t)b * N;
  int* __restrict__ out = idx + (size_t)b * npoint;
  gu64* bring = (gu64*)(ring + (size_t)b * 2 * geo.cl);
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = pp::wave_id_uniform();
  const int k0 = c * geo.slice;

  float px[R], py[R], pz[R], td[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int k = k0 + t + kClThreads * i;
    const bool ok = k < N && t + kClThreads * i < geo.slice;
    const int kc = ok ? k : 0;
    px[i] = p[3 * (size_t)kc + 0];
    py[i] = p[3 * (size_t)kc + 1];
    pz[i] = p[3 * (size_t)kc + 2];
    // points outside the slice never win: -1 < every real d2 (>= 0); min(d, -1) stays -1
    td[i] = ok ? tmp[kc] : -1.0f;
  }
  int old = seed;
  // The results -- the pick, and its coordinates if asked for -- are stored by lane 0 of WAVE 1 of member 0: wave 0
  // polls the cluster's granules with vector loads, and its wait for a poll (vmcnt counts in order) would wait for the
  // stores of the step before as well; wave 1 issues no vector load inside the loop.
  const bool writer = c == 0 && t == 64;
  if (writer) out[0] = old;
  bool dead = false;
  // (sampled != nullptr: the picks' coordinates as well -- see fps_block_kernel)
  float* __restrict__ smp = (sampled && writer) ? sampled + (size_t)b * npoint * 3 : nullptr;
  auto put = [&](int j, float x, float y, float z) {
    if (cf) {
      smp[j] = x; smp[(size_t)npoint + j] = y; smp[2 * (size_t)npoint + j] = z;
    } else {
      smp[3 * (size_t)j] = x; smp[3 * (size_t)j + 1] = y; smp[3 * (size_t)j + 2] = z;
    }
  };
  for (int j = 1; j < npoint; ++j) {
    PP_FPS_MARK(0);
    const float ox = p[3 * (size_t)old + 0], oy = p[3 * (size_t)old + 1], oz = p[3 * (size_t)old + 2];
    if (smp) put(j - 1, ox, oy, oz);
    const unsigned tag = (((unsigned)j & 63u) << 1) | 1u;  // odd, 7 bits: never 0, never 0xFF
    // A thread's points k0 + t + 512 i have tie ranks that grow with i (T = 512 = the stride: the launcher
    // checks it), so "strictly greater, first wins" over i keeps exactly the point the packed key would:
    // only the best distance and its slot are tracked per point, the key is built once per thread.
    // Padding slots carry -1 (min(d, -1) = -1): they beat the initial -2 but produce the losing key below.
    float bd = -2.0f;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const float d = dist3(px[i], py[i], pz[i], ox, oy, oz);
      const float d2 = __builtin_fminf(d, td[i]);
      td[i] = d2;
      const bool take = d2 > bd;
      bd = take ? d2 : bd;
      bi = take ? i : bi;
    }
    // a workgroup whose slice is empty still publishes a tagged (losing) granule
    u64 best = bd < 0.0f ? (u64)tag
                         : (((u64)__float_as_uint(bd) << 32) |
                            ((u64)(0xFFFFFFu - order.rank(k0 + t + kClThreads * bi)) << 8) | tag);
    PP_FPS_MARK(1);
    best = wave_max_key<6>(best);
    if (lane == 0) s_key[j & 1][wave] = best;
    __syncthreads();
    PP_FPS_MARK(2);
    if (wave == 0) {
      u64 m = s_key[j & 1][lane & (kClWaves - 1)];
      m = wave_max_key<3>(m);  // the eight wave results sit in lanes 0..7: three row steps
      gu64* slot = bring + (size_t)(j & 1) * geo.cl;
      if (lane == 0) __hip_atomic_store(slot + c, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      PP_FPS_MARK(3);
      // poll the cluster's granules of this step
      u64 v = m;
      const bool poller = lane < geo.cl && lane != c;
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        if (poller) v = __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = !poller || (unsigned)(v & 0xFFu) == tag;
        if (__all(ready)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinLimitTicks) {
          dead = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      PP_FPS_MARK(4);
      if (lane >= geo.cl) v = 0ull;
      v = wave_max_key<6>(v);
      const unsigned r = 0xFFFFFFu - (unsigned)((v >> 8) & 0xFFFFFFull);
      if (lane == 0) s_old[j & 1] = dead ? -1 : order.unrank(r);
    }
    __syncthreads();
    PP_FPS_MARK(5);
    old = __builtin_amdgcn_readfirstlane(s_old[j & 1]);
    if (old < 0) {  // timed out: flag, leave defined (zero) indices behind and go (uniform across the workgroup)
      if (t == 0) atomicOr(err, 1u);
      if (c == 0)
        for (int jj = j + t; jj < npoint; jj += kClThreads) {
          out[jj] = 0;
          if (sampled)  // (defined values behind a reported failure: point 0's coordinates, as the indices say)
            for (int a = 0; a < 3; ++a) sampled[(size_t)b * npoint * 3 + (cf ? (size_t)a * npoint + jj : 3 * (size_t)jj + a)] = p[a];
        }
      return;
    }
    if (writer) out[j] = old;
  }
  PP_FPS_MARK_END();
  if (smp) put(npoint - 1, p[3 * (size_t)old + 0], p[3 * (size_t)old + 1], p[3 * (size_t)old + 2]);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int k = k0 + t + kClThreads * i;
    if (k < N && t + kClThreads * i < geo.slice) tmp[k] = td[i];
  }
}

template <int R>
void launch_fps_cluster(const float* xyz, float* temp, int* idx, int B, int N, int npoint, int seed,
                        TieOrder order, ClusterGeom geo, u64* ring, unsigned* err, float* sampled, int cf, hipStream_t s) {
  const int groups8 = (B + 7) / 8;
  fps_cluster_kernel<R><<<dim3(8 * groups8 * geo.cl), dim3(kClThreads), 0, s>>>(
      xyz, temp, idx, B, N, npoint, seed, order, geo, ring, err, sampled, cf);
}

// Workgroups the cluster kernel may count on being resident together: one 512-thread workgroup per CU of the
// device that is current (its real CU count: a partitioned or masked device reports fewer than the chip's
// 256), provided the occupancy query admits the kernel at all.  The cluster's members wait for each other,
// so the launch must never exceed this; kMaxClusterBlocks (the whole chip) only sizes the workspace, which
// must not depend on a device being present.
constexpr int kMaxClusterBlocks = 256;
template <int R>
int resident_cluster_blocks() {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_cluster_kernel<R>, kClThreads, 0) != hipSuccess) return 0;
  if (per_cu < 1) return 0;
  return cus < kMaxClusterBlocks ? cus : kMaxClusterBlocks;  // one per CU, whatever the query allows beyond that
}

// cluster size: as many workgroups per batch element as keeps B*CL <= `blocks` (co-residency), at
// least 512 points each; 0 = do not use the cluster kernel
int pick_cluster(int B, int N, int blocks = kMaxClusterBlocks) {
  if (N > (1 << 24) - 1024) return 0;  // tie rank (< N + 512) must fit 24 bits
  int cl = 1;
  while (cl * 2 <= 64 && (long long)8 * ((B + 7) / 8) * (cl * 2) <= blocks && (N + cl * 2 - 1) / (cl * 2) >= kClThreads)
    cl *= 2;
  if (cl < 2) return 0;
  const int slice = ((N + cl - 1) / cl + kClThreads - 1) / kClThreads * kClThreads;
  if (slice / kClThreads > 32) return 0;  // more than 32 points per thread: registers
  return cl;
}

constexpr size_t kFpsErrBytes = 256;  // error word + padding in front of the ring

template <int R>
void launch_fps(const float* xyz, float* temp, int* idx, int B, int N, int npoint, int seed,
                TieOrder order, float* sampled, int cf, hipStream_t s) {
  fps_block_kernel<R><<<dim3(B), dim3(kFpsThreads), 0, s>>>(xyz, temp, idx, N, npoint, seed, order, sampled, cf);
}

}  // namespace

static pp::Knob g_fps_force_v1;
extern "C" void pp_debug_set_fps_v1(int on) { g_fps_force_v1.set(on); }

// bytes of the cluster kernel's ring (0 = that kernel does not serve these sizes), 256-byte granules
static size_t ring_bytes(int B, int N) {
  const int cl = pick_cluster(B, N);
  if (cl == 0) return 0;
  return ((size_t)8 * ((B + 7) / 8) * 2 * cl * sizeof(u64) + 255) / 256 * 256;
}

// layout: status word (256 B) | ring of the cluster kernel | scratch of the bucketed kernel
extern "C" size_t pp_furthest_sampling_workspace_bytes(int B, int N, int npoint) {
  if (B <= 0 || N <= 0) return 0;
  const size_t ring = ring_bytes(B, N);
  const size_t bucket = ppfps::bucket_applies(B, N, npoint) ? ppfps::bucket_workspace_bytes(B, N) : 0;
  if (ring == 0 && bucket == 0) return 0;
  return kFpsErrBytes + ring + bucket;
}

// 0 = ok; 1 = a cluster wait timed out in some earlier call that used this workspace (that call's
// indices are zeros from the step of the failure on) and the word has not been cleared since.
// Synchronises the stream: a debugging / test aid, not a hot-path call.
extern "C" int pp_furthest_sampling_status(const void* workspace, void* stream) {
  if (!workspace) return 0;
  unsigned v = 0;
  if (hipMemcpyAsync(&v, workspace, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  return (int)v;
}

extern "C" int pp_furthest_sampling_gather_f32(const float* xyz, float* temp, int* idx, float* sampled,
                                               int channels_first, int B, int N, int npoint, int seed_idx,
                                               void* workspace, size_t workspace_bytes, void* stream);

extern "C" int pp_furthest_sampling_f32(const float* xyz, float* temp, int* idx, int B, int N,
                                        int npoint, int seed_idx, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return pp_furthest_sampling_gather_f32(xyz, temp, idx, nullptr, 0, B, N, npoint, seed_idx, workspace, workspace_bytes,
                                         stream);
}

extern "C" int pp_furthest_sampling_gather_f32(const float* xyz, float* temp, int* idx, float* sampled,
                                               int channels_first, int B, int N, int npoint, int seed_idx,
                                               void* workspace, size_t workspace_bytes, void* stream) {
  const int cf = channels_first ? 1 : 0;
  if (B < 0 || N < 0 || npoint < 0) return PP_EINVAL;
  if (B == 0 || npoint <= 0) return PP_OK;  // ref: `if (m <= 0) return;` (sampling_cuda.cu:166)
  if (N == 0 || !xyz || !idx) return PP_EINVAL;
  if (seed_idx < 0 || seed_idx >= N) return PP_EINVAL;  // the reference would read out of bounds
  hipStream_t s = (hipStream_t)stream;
  const int T = pp_opt_n_threads(N);
  TieOrder order;
  order.t_mask = T - 1;
  order.t_shift = __builtin_ctz((unsigned)T);
  order.rows = (N + T - 1) / T;
  if ((long long)T * order.rows > 0xFFFFFFFELL) return PP_EINVAL;
  // The bucketed kernel (fps_bucket.hip) wherever it applies: a step visits the buckets the pick can change instead
  // of every point, and no workgroup waits for another.  Knob: 0 = this choice, 1 = one workgroup per batch element
  // over all points, 2 = the CU cluster over all points, 3 = bucketed.
  const int form = g_fps_force_v1;
  // (beyond 65536 points the minima no longer fit the registers and LDS of one CU: there the CU cluster, where the
  //  batch leaves room for one, is the faster of the two -- B=4, N=262144: 11.3 against 13.7 ms)
  // The cluster this device can actually keep resident (a partitioned device, a CU mask -- the members of a
  // cluster spin on each other, so never more workgroups than stay resident at once; fewer CUs -> a smaller cluster or
  // none).  Worked out BEFORE the choice between the cluster and the bucketed kernel (decided on the nominal
  // cluster, a device that then could not host it fell through to the single-workgroup kernel over all points although
  // the bucketed kernel still applied).  A kernel of ANOTHER stream holding CUs can still delay members: the waits are
  // bounded, the kernel then leaves zeros and raises the workspace's error word (pp_furthest_sampling_status).
  // (the cluster kernel's per-thread tie rule assumes the reference's thread count equals its point stride)
  int cl = (form == 1 || form == 3 || T != kClThreads) ? 0 : pick_cluster(B, N);
  if (cl >= 2 && npoint > 1) {
    auto blocks_for = [](int r) {
      return r <= 1 ? resident_cluster_blocks<1>() : r <= 2 ? resident_cluster_blocks<2>() : r <= 4 ? resident_cluster_blocks<4>()
           : r <= 8 ? resident_cluster_blocks<8>() : r <= 16 ? resident_cluster_blocks<16>() : resident_cluster_blocks<32>();
    };
    auto r_of = [&](int c) { return (((N + c - 1) / c + kClThreads - 1) / kClThreads * kClThreads) / kClThreads; };
    const int blocks = blocks_for(r_of(cl));
    cl = blocks > 0 ? pick_cluster(B, N, blocks) : 0;  // (a smaller cluster means more points per thread)
    if (cl >= 2 && (long long)8 * ((B + 7) / 8) * cl > blocks_for(r_of(cl))) cl = 0;
  }
  const bool cluster_first = form == 0 && N > 65536 && cl >= 2 && npoint > 1 && temp != nullptr;
  if ((form == 0 || form == 3) && !cluster_first && ppfps::bucket_applies(B, N, npoint)) {
    const size_t need = pp_furthest_sampling_workspace_bytes(B, N, npoint);
    if (workspace && workspace_bytes >= need)
      return ppfps::bucket_launch(xyz, temp, idx, B, N, npoint, seed_idx, order,
                                  (char*)workspace + kFpsErrBytes + ring_bytes(B, N), sampled, cf, s);
    if (form == 3) return PP_EINVAL;
  }
  // temp == NULL ("start every point at 1e10 and keep nothing": what furthest_point_sample does with a temp of its own)
  // is served by the bucketed kernel only; the caller then allocates one and calls again
  if (!temp) return PP_ENOTSUP;
  if (cl >= 2 && npoint > 1) {
    const size_t need = kFpsErrBytes + ring_bytes(B, N);
    if (!workspace || workspace_bytes < need) return PP_EINVAL;
    ClusterGeom geo;
    geo.cl = cl;
    geo.slice = ((N + cl - 1) / cl + kClThreads - 1) / kClThreads * kClThreads;
    geo.per_xcd8 = cl * ((B + 7) / 8);
    // reset the ring (tag 0xFF never matches a step tag).  The status word in front of it is STICKY: the
    // caller zeroes it once after allocating the workspace, a timed-out wait sets it, and it stays set until
    // the caller clears it -- so a failure cannot be wiped out by the next call before anybody has looked.
    hipError_t e = hipMemsetAsync((char*)workspace + kFpsErrBytes, 0xFF, ring_bytes(B, N), s);
    if (e != hipSuccess) return (int)e;
    u64* ring = (u64*)((char*)workspace + kFpsErrBytes);
    unsigned* err = (unsigned*)workspace;
    const int r = geo.slice / kClThreads;
    if (r <= 1) launch_fps_cluster<1>(xyz, temp, idx, B, N, npoint, seed_idx, order, geo, ring, err, sampled, cf, s);
    else if (r <= 2) launch_fps_cluster<2>(xyz, temp, idx, B, N, npoint, seed_idx, order, geo, ring, err, sampled, cf, s);
    else if (r <= 4) launch_fps_cluster<4>(xyz, temp, idx, B, N, npoint, seed_idx, order, geo, ring, err, sampled, cf, s);
    else if (r <= 8) launch_fps_cluster<8>(xyz, temp, idx, B, N, npoint, seed_idx, order, geo, ring, err, sampled, cf, s);
    else if (r <= 16) launch_fps_cluster<16>(xyz, temp, idx, B, N, npoint, seed_idx, order, geo, ring, err, sampled, cf, s);
    else launch_fps_cluster<32>(xyz, temp, idx, B, N, npoint, seed_idx, order, geo, ring, err, sampled, cf, s);
    PP_RETURN_IF_LAUNCH_FAILED();
    return PP_OK;
  }
  const int per_thread = (N + kFpsThreads - 1) / kFpsThreads;
  if (per_thread <= 1) launch_fps<1>(xyz, temp, idx, B, N, npoint, seed_idx, order, sampled, cf, s);
  else if (per_thread <= 2) launch_fps<2>(xyz, temp, idx, B, N, npoint, seed_idx, order, sampled, cf, s);
  else if (per_thread <= 4) launch_fps<4>(xyz, temp, idx, B, N, npoint, seed_idx, order, sampled, cf, s);
  else if (per_thread <= 8) launch_fps<8>(xyz, temp, idx, B, N, npoint, seed_idx, order, sampled, cf, s);
  else if (per_thread <= 16) launch_fps<16>(xyz, temp, idx, B, N, npoint, seed_idx, order, sampled, cf, s);
  else if (per_thread <= 32) launch_fps<32>(xyz, temp, idx, B, N, npoint, seed_idx, order, sampled, cf, s);
  else if (per_thread <= 64) launch_fps<64>(xyz, temp, idx, B, N, npoint, seed_idx, order, sampled, cf, s);
  else launch_fps<0>(xyz, temp, idx, B, N, npoint, seed_idx, order, sampled, cf, s);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
