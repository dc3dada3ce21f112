// chamfer_grid.hip -- exact nearest neighbour through a uniform grid (C == 3), with the brute-force
// scan of chamfer.hip as the fallback.  Same outputs, bit for bit, as the brute force: the
// candidates' distances are evaluated with the same canonical arithmetic (pp::chamfer_d3), ties
// are resolved to the lowest original index explicitly, and a query stops expanding only when
// every unexamined point is PROVABLY farther in computed fp32 distance:
//
//   * reference points of one (batch, direction) set are counting-sorted into cubic cells of side
//     h over their bounding box (<= 32^3 cells, counters in LDS);
//   * a query is clamped to the box (the projection q' onto a convex set never increases the
//     distance to points inside it, so bounds derived for q' hold for q) and examines the cells of
//     first the 2x2x2 block of cells nearest to q' (everything else is at least `reach` away: the
//     distance from q to the nearest face of that block with grid beyond it, >= h/2), then the
//     cubes of Chebyshev radius rho = 1 and 2 around its own cell (everything else >= rho*h away);
//   * fp32 evaluation of the canonical formula has relative error <= 6 * 2^-24, and the cell
//     assignment (one subtraction, one multiplication, one truncation) can misplace a point by
//     <= 1e-5 h; both are covered by stopping only if  best < bound^2 * 0.999  (strict).
//     Then no unexamined point can have a computed distance <= best, i.e. none can win or tie;
//   * clouds that are not evenly sampled surfaces (everything below stays inside the two launches):
//       - a cell holding more than kCrowd points (clusters, several scales) carries a grid of its own
//         (grid_common.h: SubGrid); queries whose block touches one search it through that grid;
//       - a wave in which many queries are open after the 2x2x2 block (thin regions) runs the cubes of radius
//         1 and 2 a lane per query, otherwise the whole wave serves them one by one;
//       - a query that cannot stop at rho = 2 (far from the reference cloud: clusters at different places,
//         disjoint clouds, the tail of a Gaussian) joins a group of such queries of its wave that lie close
//         together; the wave stages the cell rows the group can need -- bounded by the best candidate its
//         members know -- through LDS and every lane walks them for its own query (wave_group_search);
//       - labeled searches and non-finite queries keep the older forms of the last step (larger cubes by
//         the whole wave, the occupied cells or the whole cloud a lane per query); so does every query of a
//         set whose grid is useless (non-finite coordinates).  Two launches per forward, no list.
#include <algorithm>
#include <hip/hip_ext.h>
#include <cstdlib>
#include <mutex>

#ifdef PP_BUILD_PROBE
// diagnostic build only (tools/build_phases.py): the 100 MHz clock at the phase boundaries of the build, thread 0 of
// every workgroup of the build launch (256 at config 2)
#include <hip/hip_runtime.h>
__device__ unsigned long long g_bphase[512][16];
#define PP_PHASE(n)                                                                                        \
  do {                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("" ::: "memory");                                                                          \
    if (threadIdx.x == 0 && blockIdx.x < 512) g_bphase[blockIdx.x][n] = __builtin_amdgcn_s_memrealtime();   \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)
extern "C" int pp_debug_read_build_phases(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bphase), sizeof(g_bphase));
}
#endif
#include "grid_common.h"

// waves per SIMD the unlabeled search kernels are compiled for (tools/build_variant_lib.sh -DPP_WAVE_WAVES=4 ...)
#ifndef PP_WAVE_WAVES
#define PP_WAVE_WAVES 6
#endif

#ifdef PP_QUERY_PROBE
// diagnostic build only (tools/query_probe.py): 100 MHz clock at the phase boundaries of a few workgroups, and
// what every wave spent in each phase (10 ns units; the first kQWaves waves of the launch)
constexpr int kQWaves = 1 << 17;
__device__ unsigned long long g_qphase[8][16];
__device__ unsigned g_qwave[kQWaves][10];
__device__ unsigned g_qstart[kQWaves][2];    // a wave's first and last stamp (10 ns units, low 32 bits of the clock)
__device__ unsigned long long g_qgroup[8];  // group search: calls, groups, blind groups, candidates of the row cuts, max of them in one call, rows of the
                                             // boxes, rows listed by pass 1, candidates the whole wave walked
__device__ unsigned g_qgw[kQWaves][16];       // group search, per wave: groups, blind groups, candidates of the row cuts, candidates walked, rows listed, batches
extern "C" int pp_debug_read_query_wave_groups(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qgw), sizeof(g_qgw)); }
__device__ unsigned long long g_qgt[8];      // group search: time (10 ns) in sampling, grouping, row listing, row cuts + spans, candidate fetch + sift, walk
extern "C" int pp_debug_read_query_group_times(void* out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qgt), sizeof(g_qgt));
  if (reset) {
    unsigned long long z[8] = {0};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_qgt), z, sizeof(z));
  }
  return rc;
}
#ifdef PP_QUERY_PROBE_NO_GROUP_STATS  // (the waves' phase stamps only: the group search's own clocks and counters cost it a factor)
#define PP_GT_DECL
#define PP_GT(i)
#else
#define PP_GT_DECL unsigned long long pp_gtp = wall_clock64(), pp_gt[6] = {0, 0, 0, 0, 0, 0}
#define PP_GT(i)                                   \
  do {                                             \
    const unsigned long long pp_n = wall_clock64(); \
    pp_gt[i] += pp_n - pp_gtp;                     \
    pp_gtp = pp_n;                                 \
  } while (0)
#endif
extern "C" int pp_debug_read_query_phases(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qphase), sizeof(g_qphase));
}
extern "C" int pp_debug_read_query_group_stats(void* out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qgroup), sizeof(g_qgroup));
  if (reset) {
    unsigned long long z[8] = {0};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_qgroup), z, sizeof(z));
  }
  return rc;
}
extern "C" int pp_debug_read_query_wave_phases(void* out) {  // kQWaves x 10 unsigned
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qwave), sizeof(g_qwave));
}
extern "C" int pp_debug_read_query_wave_span(void* out) {  // kQWaves x 2 unsigned
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qstart), sizeof(g_qstart));
}
#define PP_QPHASE_DECL unsigned long long pp_prev = wall_clock64()
#define PP_QPHASE(n)                                                                     \
  do {                                                                                   \
    const unsigned long long pp_now = wall_clock64();                                    \
    const unsigned pp_w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                          \
    if ((threadIdx.x & 63) == 0 && pp_w < (unsigned)kQWaves) {                            \
      g_qwave[pp_w][n] = (unsigned)(pp_now - pp_prev);                                   \
      if ((n) == 0) g_qstart[pp_w][0] = (unsigned)pp_prev;                                \
      g_qstart[pp_w][1] = (unsigned)pp_now;                                               \
    }                                                                                    \
    pp_prev = pp_now;                                                                    \
    if (threadIdx.x == 0 && (blockIdx.x & 511) == 0 && (blockIdx.x >> 9) < 8)            \
      g_qphase[blockIdx.x >> 9][n] = pp_now;                                             \
  } while (0)
#else
#define PP_QPHASE_DECL
#define PP_QPHASE(n)
#define PP_GT_DECL
#define PP_GT(i)
#endif

namespace {

using pp::GridSet;
using pp::cell_coord;
using pp::kGridCells;
using pp::kGridMax;
using pp::kBuildThreads;

constexpr float kBoundSlack = 0.999f;
// v_sqrt_f32 (one ulp; sqrtf is the correctly rounded sequence of a dozen instructions): only where the result is
// widened by a slack factor of 1e-5 or more right after -- bounds of searches, never a distance that is returned
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// entries between two sets' cell tables: kGridCells + 1 used, padded to a multiple of four so that every set's table is
// 16-byte aligned (the LDS-sorted build copies a slab's 8192 entries out in 16-byte pieces)
constexpr int kCellStride = kGridCells + 4;

// Workspace layout (bytes), S = 2*B sets, T = B*(N+M) points:
//   [0, 64*S)                      GridSet[S]
//   [.., +4*kCellStride*S)         unsigned cell_start[S][kCellStride]   (kGridCells + 1 used)
//   [.., +16*T)                    float4 sorted[T]   (x, y, z, original index bits)
//   [.., +4*(2*T + 2*S))           unsigned sub_start[...]   second level: cell tables of the crowded cells,
//                                  the table of the cell whose points start at `start` of set s at 2*(set offset + start) + 2*s
//   [.., +32*(T/kCrowd + 2*S))     SubGrid sub_desc[...]     their descriptors (grid_common.h)
//   [.., +16*T)                    float4 sorted2[T]  spare copy the refinement sorts through
//   [.., +4*T, +4*T)               float slab[T], slab2[T]   labels in sorted order + spare (labeled Chamfer only)
// (the second-level arrays are only touched for sets that have crowded cells: never at config 2)
//   [.., +4 * S * kBuildSlabs * 2 * chunks)  int tile_z[S][kBuildSlabs][chunks][2]   chunk table (grid_common.h: kChunk)
//   [.., +4 * S * kLayerWords)     unsigned layers[S][kLayerWords]   layer table + pending counter (grid_common.h)
//   [.., +4*T)                     int pend[T]        unlabeled searches: the queries stage A left (positions in the query
//                                  cloud's sorted order), 64 slots per wave of the stage-A kernel
//   [.., +4*(T/64 + S + 1))        unsigned pend_cnt[...]   how many of a wave's 64 slots are filled
//   [.., +4*S)                     unsigned routed[S]       directions routed to the every-pair kernel in front of the build
//   [.., +128*S)                   unsigned rowbits[S][32]  bit (y + gy z) of a set: its cell row (y, z) holds points; written by
//                                  the stage-A launch's tail workgroups for the group search of the list kernel (round 6)
struct Layout {
  size_t sets, cell_start, sorted, sub_start, sub_desc, sorted2, slab, slab2, tile_z, layers, pend, pend_cnt, routed, rowbits, total;
  int chunks;  // chunk-table entries per set and slab (0: sets too large for the table)
};
__host__ __device__ inline Layout make_layout(int B, int N, int M, bool labeled = false) {
  Layout L;
  const size_t S = (size_t)2 * B, T = (size_t)B * ((size_t)N + M);
  L.sets = 0;
  L.cell_start = L.sets + ((64 * S + 255) / 256) * 256;
  L.sorted = L.cell_start + ((4 * (size_t)kCellStride * S + 255) / 256) * 256;
  L.sub_start = L.sorted + 16 * T;
  L.sub_desc = L.sub_start + ((4 * (2 * T + 2 * S) + 255) / 256) * 256;
  L.sorted2 = L.sub_desc + ((32 * (T / pp::kCrowd + 2 * S) + 255) / 256) * 256;
  L.slab = L.sorted2 + 16 * T;
  L.slab2 = L.slab + (labeled ? 4 * T : 0);
  L.tile_z = ((L.slab2 + (labeled ? 4 * T : 0) + 255) / 256) * 256;
  const int chq = (((N > M ? N : M) + pp::kChunk - 1) / pp::kChunk + 3) & ~3;  // (a tile reads up to four chunks' pairs)
  L.chunks = (!labeled && chq <= pp::kChunkMax) ? chq : 0;
  L.layers = ((L.tile_z + 4 * S * pp::kBuildSlabs * 2 * (size_t)L.chunks + 255) / 256) * 256;
  L.pend = L.layers + (L.chunks ? ((4 * S * pp::kLayerWords + 255) / 256) * 256 : 0);
  L.pend_cnt = L.pend + (L.chunks ? 4 * T : 0);
  L.routed = L.pend_cnt + (L.chunks ? ((4 * (T / 64 + S + 1) + 255) / 256) * 256 : 0);  // [S]: directions routed to the every-pair kernel before the build
  L.rowbits = L.routed + (L.chunks ? ((4 * S + 255) / 256) * 256 : 0);  // [S][32]: which cell rows of a set hold points (bit y + gy z)
  L.total = L.rowbits + (L.chunks ? 128 * S : 0);
  return L;
}
// second-level arrays of set (b, dir): first table entry / first descriptor
__host__ __device__ inline size_t set_sub_start_offset(int b, int dir, int N, int M) {
  const size_t po = (size_t)b * ((size_t)N + M) + (dir ? (size_t)M : 0);  // = set_point_offset
  return 2 * po + 2 * (size_t)(2 * b + dir);
}
__host__ __device__ inline size_t set_sub_desc_offset(int b, int dir, int N, int M) {
  const size_t po = (size_t)b * ((size_t)N + M) + (dir ? (size_t)M : 0);
  return po / pp::kCrowd + 2 * (size_t)(2 * b + dir);
}
// the pending list of the queries of (b, dir) -- they are the points of set (b, dir ^ 1) -- and its per-wave counts
__host__ __device__ inline size_t pend_count_offset(int b, int dir, int N, int M) {
  const size_t po = (size_t)b * ((size_t)N + M) + (dir ? 0 : (size_t)M);  // = set_point_offset(b, dir ^ 1)
  return po / 64 + (size_t)(2 * b + (dir ^ 1));
}
// set s = 2*b + dir; dir 0: queries = cloud 1 (N), references = cloud 2 (M)
__host__ __device__ inline size_t set_point_offset(int b, int dir, int N, int M) {
  return (size_t)b * ((size_t)N + M) + (dir ? (size_t)M : 0);  // references of (b,0) first (M), then (b,1) (N)
}

// Round 6: is a direction one no search can prune?  A wave per direction: eight of its queries against 64 of its reference
// points; where a query sees more than half of the samples within half a percent of its nearest one, every bound of the
// search is beaten by everybody -- a shell against a cluster at its centre, a cluster against a far shell, identical
// points -- and six of eight such queries ROUTE the direction to the every-pair kernel.  A heuristic of speed only: both
// paths give the same bits.  (The same test on samples of the sorted clouds runs in the stage-A launch's tail while the
// host has not seen a routed direction: see grid_stage_a_kernel.)
__device__ __forceinline__ bool direction_unprunable(const float* __restrict__ q, int nq, int qstride,
                                                     const float* __restrict__ r, int nr, int rstride, int lane) {
  if (nr < 64 || nq < 16) return false;
  const float* rp = r + (size_t)((size_t)lane * (size_t)nr / 64) * rstride;
  const float rx = rp[0], ry = rp[1], rz = rp[2];
  int votes = 0;
  for (int i = 0; i < 8; ++i) {
    const float* qp = q + (size_t)((size_t)(2 * i + 1) * (size_t)nq / 16) * qstride;  // (wave-uniform)
    const float d = pp::chamfer_d3(rx, ry, rz, qp[0], qp[1], qp[2]);
    const float dmin = -pp::wave_reduce_dpp<false>(-d);
    votes += __builtin_popcountll(__ballot(d <= dmin * 1.01f)) >= 32 ? 1 : 0;
  }
  return votes >= 6;
}
// ... in front of the build, on the clouds as they are given (the host has seen routed directions lately: the build, the
// stage-A kernel and the list kernel then leave a routed direction alone from their first instruction)
__global__ __launch_bounds__(64) void route_decide_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                          unsigned* __restrict__ routed, int B, int N, int M,
                                                          unsigned* __restrict__ routed_host, unsigned epoch) {
  const int set = blockIdx.x, b = set >> 1, dir = set & 1, lane = threadIdx.x;
  const int nr = dir ? N : M, nq = dir ? M : N;  // dir 0: queries = cloud 1 (N), references = cloud 2 (M)
  const float* r = (dir ? xyz1 : xyz2) + (size_t)b * nr * 3;
  const float* q = (dir ? xyz2 : xyz1) + (size_t)b * nq * 3;
  const bool hopeless = direction_unprunable(q, nq, 3, r, nr, 3, lane);
  if (lane == 0) {
    routed[set] = hopeless ? 1u : 0u;
    if (hopeless && routed_host) *reinterpret_cast<volatile unsigned*>(routed_host) = epoch;
  }
}

// kBuildSlabs workgroups per set: bounding box, cell histogram (LDS), exclusive scan, scatter
// (grid_common.h).
template <bool VEC>
__global__ __launch_bounds__(kBuildThreads) void grid_build_kernel(const float* __restrict__ xyz1,
                                                                   const float* __restrict__ xyz2,
                                                                   unsigned char* __restrict__ ws, int B,
                                                                   int N, int M,
                                                                   const float* __restrict__ label1,
                                                                   const float* __restrict__ label2, int fast,
                                                                   const unsigned* __restrict__ pre_routed) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_cnt[];  // max(grid_build_lds_bytes(kBuildSlabs), grid_build_fast_lds_bytes())
  // a set is built on the XCD that will search it (the search kernel's set -> XCD mapping): its sorted
  // points and cell table are then already in that L2
  const int V = pp::xcd_virtual_block(blockIdx.x, (2 * B * pp::kBuildSlabs + 7) / 8);
  if (V >= 2 * B * pp::kBuildSlabs) return;
  const int set = V / pp::kBuildSlabs, slab = V % pp::kBuildSlabs;
  // (round 6) both directions of the batch element routed to the every-pair kernel before this launch: neither grid
  // is needed (a set is one direction's reference grid and the other's query order)
  if (pre_routed != nullptr && pre_routed[set] != 0u && pre_routed[set ^ 1] != 0u) return;
  const int b = set >> 1, dir = set & 1;
  const int nr = dir ? N : M;
  const float* __restrict__ ref = (dir ? xyz1 : xyz2) + (size_t)b * nr * 3;
  const bool labeled = label1 != nullptr;
  const Layout L = make_layout(B, N, M, labeled);
  const float* __restrict__ lab = labeled ? (dir ? label1 : label2) + (size_t)b * nr : nullptr;
  GridSet* gset = reinterpret_cast<GridSet*>(ws + L.sets) + set;
  unsigned* cstart = reinterpret_cast<unsigned*>(ws + L.cell_start) + (size_t)set * kCellStride;
  pp::f4* sorted = reinterpret_cast<pp::f4*>(ws + L.sorted) + set_point_offset(b, dir, N, M);
  unsigned* sub_start = reinterpret_cast<unsigned*>(ws + L.sub_start) + set_sub_start_offset(b, dir, N, M);
  pp::SubGrid* sub_desc = reinterpret_cast<pp::SubGrid*>(ws + L.sub_desc) + set_sub_desc_offset(b, dir, N, M);
  pp::f4* sorted2 = reinterpret_cast<pp::f4*>(ws + L.sorted2) + set_point_offset(b, dir, N, M);
  int* tz = L.chunks ? reinterpret_cast<int*>(ws + L.tile_z) + (size_t)set * pp::kBuildSlabs * 2 * L.chunks : nullptr;
  unsigned* layers = L.chunks ? reinterpret_cast<unsigned*>(ws + L.layers) + (size_t)set * pp::kLayerWords : nullptr;
  // round 5: unlabeled sets of config 2's class (one register chunk, aligned) are sorted through the LDS, a slab owning
  // whole z-layers (grid_common.h: grid_build_set_fast); what that path declines takes the general one, with its plan
  // forced where the other slabs of the set may have used it
  int how = 1;
  if constexpr (VEC) {
    if (!labeled && fast) how = pp::grid_build_set_fast(ref, nr, gset, cstart, sorted, s_cnt, slab, sub_start, sub_desc, sorted2, tz, L.chunks, layers);
  }
  if (how == 0) return;
  __syncthreads();  // (the general path reuses the LDS the fast one was using)
  pp::grid_build_set_impl<false, VEC, true>(
      ref, nr, gset, cstart, sorted, nullptr, s_cnt, lab,
      labeled ? reinterpret_cast<float*>(ws + L.slab) + set_point_offset(b, dir, N, M) : nullptr, slab, pp::kBuildSlabs,
      sub_start, sub_desc, sorted2, labeled ? reinterpret_cast<float*>(ws + L.slab2) + set_point_offset(b, dir, N, M) : nullptr,
      tz, L.chunks, layers, how == 2 ? reinterpret_cast<const pp::BuildPlan*>(s_cnt + pp::kFastLdsPlan) : nullptr);
}

// Stages B and C (cubes of Chebyshev radius 1 and 2 around the query's cell) for the queries stage A left
// over, one WAVE per query.  The list is short (0.3 % of the queries at config 2), so what
// a query costs is the LENGTH of its chain of dependent loads, not the lane-cycles: with one lane per
// query that chain is 9 (then 25) cell rows walked one after the other (22 us at config 2); here a whole
// wave takes one query (7 us), lane r fetches the range of row r, the rows are laid end to end (prefix sum over the lanes)
// and the wave examines 64 candidates per step -- two dependent load rounds per stage.  Each lane keeps
// the smallest (distance bits, index) key it has seen -- for non-negative non-NaN distances the order
// of the packed key is the order of "d < best || (d == best && id < bidx)", and a NaN distance (bits
// above +inf) is never taken, as in the lane-per-query form -- and one wave-wide minimum ends a stage.
// PIPE = false: one step of 64 candidates per round; a scan of more than kScanInline candidates (rows through a
// crowded cell) is not started -- `skipped` -- and left to the pipelined form.
// PIPE = true (serve_long_scans): four steps per round, their loads issued together: with one dependent load per step
// a scan of thousands of candidates costs its length in memory round trips.
constexpr unsigned kScanInline = 512;
template <bool LAB, bool PIPE>
__device__ __forceinline__ unsigned long long wave_scan_rows(int nrows, unsigned rs, unsigned re,
                                                             const pp::f4* __restrict__ sorted,
                                                             const float* __restrict__ slab, float qx, float qy,
                                                             float qz, float ql, unsigned long long key,
                                                             bool& skipped) {
  const int lane = threadIdx.x & 63;
  const unsigned len = lane < nrows ? re - rs : 0u;
  const unsigned incl = pp::wave_scan_u32_dpp(len);  // (DPP: __shfl_up was five dependent ds_bpermute round trips)
  const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
  const unsigned excl = incl - len;
  const unsigned shift = rs - excl;  // candidate c of row r sits at sorted[c + shift_r]
  if (!PIPE && total > kScanInline) {  // wave-uniform
    skipped = true;
    return key;
  }
  if constexpr (!PIPE) {
    for (unsigned c0 = 0; c0 < total; c0 += 64) {
      const unsigned c = c0 + lane;
      unsigned add = 0;
      for (int r = 0; r < nrows; ++r) {  // the last row whose first candidate is <= c (empty rows are overridden)
        const unsigned ex = (unsigned)__builtin_amdgcn_readlane((int)excl, r);
        const unsigned sh = (unsigned)__builtin_amdgcn_readlane((int)shift, r);
        add = c >= ex ? sh : add;
      }
      if (c < total) {
        const pp::f4 p = sorted[c + add];
        const float d = pp::chamfer_d3(p.x, p.y, p.z, qx, qy, qz);
        const unsigned long long cand =
            ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
        const bool ok = !LAB || slab[c + add] == ql;
        key = (ok && cand < key) ? cand : key;
      }
    }
  } else {
    for (unsigned c0 = 0; c0 < total; c0 += 256) {  // (a step past the end repeats the last candidate: harmless)
      unsigned at[4];
      // (round 5) a step that lies inside ONE row -- all but a few of the steps of a scan through a crowded cell, whose
      // row is thousands of candidates long -- needs no search for its candidates' rows: two ballots find that out and
      // the row (the search is 3 nrows instructions per candidate against 12 for its distance)
      const bool mine = lane < nrows && len > 0u;
      if (__ballot(mine && excl > c0 && excl < c0 + 256u) == 0ull) {  // (wave-uniform)
        const unsigned long long upto = __ballot(mine && excl <= c0);
        const unsigned sh = (unsigned)__builtin_amdgcn_readlane((int)shift, 63 - (int)__builtin_clzll(upto | 1ull));
#pragma unroll
        for (int u = 0; u < 4; ++u) at[u] = min(c0 + (unsigned)(u * 64 + lane), total - 1) + sh;
      } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned c = min(c0 + (unsigned)(u * 64 + lane), total - 1);
        unsigned add = 0;
        for (int r = 0; r < nrows; ++r) {
          const unsigned ex = (unsigned)__builtin_amdgcn_readlane((int)excl, r);
          const unsigned sh = (unsigned)__builtin_amdgcn_readlane((int)shift, r);
          add = c >= ex ? sh : add;
        }
        at[u] = c + add;
      }
      }
      pp::f4 p[4];
      float pl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p[u] = sorted[at[u]];
        if (LAB) pl[u] = slab[at[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = pp::chamfer_d3(p[u].x, p[u].y, p[u].z, qx, qy, qz);
        const unsigned long long cand =
            ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p[u].w);
        const bool ok = !LAB || pl[u] == ql;
        key = (ok && cand < key) ? cand : key;
      }
    }
  }
  return pp::wave_min_u64_dpp(key);  // wave-wide minimum (DPP)
}

// The wide stages for one query, executed by a whole wave (every lane active, all arguments wave-uniform):
// cube of Chebyshev radius 1 around the query's cell, then 2.  Returns whether the query is settled;
// (best, bidx) is the nearest examined candidate ((0, -1) for a labeled query whose label nobody carries).
template <bool LAB, bool PIPE>
__device__ __forceinline__ bool wide_stages_wave(float qx, float qy, float qz, float ql, const GridSet& g,
                                                 const unsigned* __restrict__ cell_start,
                                                 const pp::f4* __restrict__ sorted, const float* __restrict__ slab,
                                                 float& best, int& bidx, bool& skipped) {
  const int lane = threadIdx.x & 63;
  const int cx = cell_coord(qx, g.minx, g.invh, g.gx);
  const int cy = cell_coord(qy, g.miny, g.invh, g.gy);
  const int cz = cell_coord(qz, g.minz, g.invh, g.gz);
  const float fx = (qx - g.minx) * g.invh - (float)cx, fy = (qy - g.miny) * g.invh - (float)cy,
              fz = (qz - g.minz) * g.invh - (float)cz;
  // distance (in cells) from q to the nearest face of the cube of Chebyshev radius rho around its
  // cell that has grid beyond it: rho + f below, rho + 1 - f above (>= rho)
  auto reach_cube = [&](int rho) {
    auto axis = [&](float f, int c, int gdim) {
      const float lo = c - rho >= 1 ? (float)rho + f : __builtin_inff();
      const float hi = c + rho <= gdim - 2 ? (float)(rho + 1) - f : __builtin_inff();
      return fminf(lo, hi);
    };
    return fminf(axis(fx, cx, g.gx), fminf(axis(fy, cy, g.gy), axis(fz, cz, g.gz)));
  };
  unsigned long long key = ((unsigned long long)0x7f800000u << 32) | 0x7fffffffu;  // (+inf, no index)
  bool resolved = false;
  auto stage = [&](const int rho) {  // cube of radius rho (re-examining cells is harmless)
    const int side = 2 * rho + 1;
    const int x0 = max(cx - rho, 0), x1 = min(cx + rho, g.gx - 1);
    // lane r < side*side fetches the range of row (cz - rho + r / side, cy - rho + r % side)
    const int z = cz - rho + lane / side, y = cy - rho + lane % side;
    const bool ok = lane < side * side && z >= 0 && z < g.gz && y >= 0 && y < g.gy;
    const int c = pp::cell_linear(0, min(max(y, 0), g.gy - 1), min(max(z, 0), g.gz - 1), g.gx, g.gy);
    unsigned rs = 0, re = 0;
    if (ok) {
      rs = cell_start[c + x0];
      re = cell_start[c + x1 + 1];
    }
    key = wave_scan_rows<LAB, PIPE>(side * side, rs, re, sorted, slab, qx, qy, qz, ql, key, skipped);
    if (skipped) return;  // (PIPE = false only) a long scan: the whole query goes to serve_long_scans
    const float kbest = __uint_as_float((unsigned)(key >> 32));
    const int kidx = (int)(unsigned)key;
    const bool all = cz - rho <= 0 && cz + rho >= g.gz - 1 && cy - rho <= 0 && cy + rho >= g.gy - 1 &&
                     cx - rho <= 0 && cx + rho >= g.gx - 1;
    const float reach = g.h * reach_cube(rho);
    resolved = all ? (LAB || kidx != 0x7fffffff) : (kbest < reach * reach * kBoundSlack);
  };
  if constexpr (PIPE) {  // the two radii as constants: the row search of the long scans is unrolled
    stage(1);
    if (!resolved && !skipped) stage(2);
  } else {
    // one body for both radii (the short scans live in a leaf function that must stay within the registers a callee
    // need not save: serve_pending)
#pragma nounroll
    for (int rho = 1; rho <= 2 && !resolved && !skipped; ++rho) stage(rho);
  }
  best = __uint_as_float((unsigned)(key >> 32));
  bidx = (int)(unsigned)key;
  if (LAB && resolved && bidx == 0x7fffffff) {  // whole grid examined, nobody carries this label
    best = 0.0f;                                  // (ref nmdistance_cuda.cu:110-113)
    bidx = -1;
  }
  return resolved;
}

// Group k of a lane's stage-A sequence (see grid_query_wave_kernel): four points of the row it falls in,
// read from global memory (waves whose region does not fit their LDS slice).
// Everything per-row arrives BY VALUE: selects between variables captured by reference in a lambda come
// out of hipcc as indexed loads from a pointer table in scratch memory.
template <bool LAB>
__device__ __forceinline__ void stage_a_fetch(unsigned k, unsigned T1, unsigned T2, unsigned T3, unsigned T4,
                                              unsigned adj0, unsigned adj1, unsigned adj2, unsigned adj3,
                                              unsigned last0, unsigned last1, unsigned last2, unsigned last3,
                                              const pp::f4* __restrict__ sorted, const float* __restrict__ slab,
                                              pp::f4 (&p)[4], float (&pl)[4]) {
  const bool a = k < T1, b2 = k < T2, c = k < T3, live = k < T4;
  const unsigned adj = a ? adj0 : (b2 ? adj1 : (c ? adj2 : adj3));
  // a lane that has run out of groups re-reads point 0 of the set (a valid candidate: harmless), so
  // that nothing in the loop is conditional and the compiler can count the loads in flight exactly
  const unsigned last = live ? (a ? last0 : (b2 ? last1 : (c ? last2 : last3))) : 0u;
  const unsigned i = live ? adj + 4 * k : 0u;
  // (32-bit byte offsets from the wave-uniform bases: one VGPR per address instead of two)
  const char* __restrict__ sp = reinterpret_cast<const char*>(sorted);
  const char* __restrict__ lp = reinterpret_cast<const char*>(slab);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const unsigned e = min(i + u, last);
    p[u] = *reinterpret_cast<const pp::f4*>(sp + (e << 4));
    if (LAB) pl[u] = *reinterpret_cast<const float*>(lp + (e << 2));
  }
}

// LDS pointers carry their address space (a generic pointer would make the loads flat)
typedef const pp::f4 __attribute__((address_space(3))) * lds_f4_ptr;
typedef const float __attribute__((address_space(3))) * lds_f_ptr;
typedef pp::f4 __attribute__((address_space(3))) * lds_f4_wptr;
typedef float __attribute__((address_space(3))) * lds_f_wptr;

// helpers of the search kernels' in-kernel fallbacks (no brute-force list, no third launch)
constexpr int kStageLayers = 8;
constexpr int kLaneCubeMaxGroups = 48;  // groups of four points a lane walks per four rows of a cube before it gives up
constexpr int kSubMaxRho = 2;  // widest cube of sub-cells a lane examines inside a crowded cell
constexpr int kGroupBatch = 256;  // candidates per LDS batch of wave_group_search (<= the smallest per-wave slice)
// a blind seed's group: the open queries within this fraction of its candidate's distance (round 3: 1/4 made groups as
// long as a row of the query grid -- 64 consecutive sorted queries run along x -- and the bound of a group is its
// farthest member's: disjoint clouds examined 6000 candidates per query)
constexpr float kBlindGroup = 0.35f;  // (round 6, with the nearest-first parts below: 0.25 -> 0.35 blobs8 0.468 -> 0.455, disjoint +1 %)
// The constants of the stages behind stage A, as round 5's A/B runs settled them (each was a -D switch of a variant
// library then: profiles/r5/near_field_stages_ab.txt; frozen in round 6).  The forms they chose between and that lost --
// a lane per query instead of the pooled ball / cube, no ball after the cubes, a lane per query for labeled searches --
// are gone from the dispatch; lane_ball_search remains for the wave slices below 384 points.
constexpr int kMemberCutMin = 2;    // (round 5: 2, was 8 -- disjoint clouds 0.225 -> 0.203 ms) candidates per member the pieces' cuts must leave in a block of rows for the member-by-member cut
constexpr float kBallRmax = 2.5f;   // cells: the farthest candidate whose ball is walked (a box of at most 6 x 6 rows)
constexpr int kPoolCubeMin = 1;
constexpr int kPoolMin = 1;         // ... pooled over the wave (unlabeled searches)
constexpr int kBallMin = 6;         // lanes with a candidate from which the ball around it is walked a lane per query
constexpr int kSerialFar = 8;
constexpr int kSerialMax = 24;      // open lanes of a wave from which the whole-wave cubes are skipped for the group search
constexpr int kOpenJoin = 16;       // open lanes of a wave from which its pending lanes go to the group search with them
constexpr int kLaneStageMin = 6;    // open lanes of a wave from which the cubes are searched a lane per query

// distance (in cells) from a query at position f inside cell c to the nearer face of its 2-cell block along one
// axis that has grid beyond it (s = -1: the block is cells c-1, c; +1: c, c+1; beyond the grid there is nothing)
__device__ __forceinline__ float block_reach(float f, int s, int c, int gdim) {
  const float lo = s < 0 ? (c >= 1 ? f + 1.0f : __builtin_inff()) : (c >= 1 ? f : __builtin_inff());
  const float hi = s < 0 ? (c + 1 <= gdim - 1 ? 1.0f - f : __builtin_inff())
                         : (c + 1 <= gdim - 1 ? 2.0f - f : __builtin_inff());
  return fminf(lo, hi);
}

// One candidate in the exact (distance, index) order (bitwise operators: no exec-mask branches).
template <bool LAB>
__device__ __forceinline__ void take_candidate(const pp::f4& p, float pl, float qx, float qy, float qz, float ql,
                                               float& best, int& bidx) {
  const float d = pp::chamfer_d3(p.x, p.y, p.z, qx, qy, qz);
  const int id = __float_as_int(p.w);
  const bool take = (!LAB || pl == ql) & ((d < best) | ((d == best) & (id < bidx)));
  best = take ? d : best;
  bidx = take ? id : bidx;
}

// A query the cubes around its cell could not settle (far from the cloud, or in a gap of it): instead of the
// whole cloud it walks the OCCUPIED CELLS (the build's compact list; wave-uniform, scalar loads), one lane per
// query.  Pass 1 finds the cell with the smallest lower bound on the distance (the cell's cube; a cell on the
// grid's boundary also holds the points clamped into it, so its cube is open on that side) and examines its
// points; pass 2 examines every other cell whose bound does not already exceed the best distance.  Exact: every
// point lies in some occupied cell, and a skipped cell cannot hold a point as close as the best one (the bound
// is compared with slack for its own rounding; equal distances are examined, so ties still go to the lowest
// index).  ~14 VALU per occupied cell and pass instead of ~8 per point of the cloud.
// Points [cs, ce) of the sorted cloud for one lane, four loads in flight (a lane walking alone pays the full
// latency of every load it waits for; the last batch repeats the range's last point, which is harmless).
template <bool LAB>
__device__ __forceinline__ void walk_range(unsigned cs, unsigned ce, const pp::f4* __restrict__ sorted,
                                           const float* __restrict__ slab, float qx, float qy, float qz, float ql,
                                           float& best, int& bidx) {
  for (unsigned i = cs; i < ce; i += 4) {
    pp::f4 p[4];
    float pl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned e = min(i + (unsigned)u, ce - 1);
      p[u] = sorted[e];
      if (LAB) pl[u] = slab[e];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) take_candidate<LAB>(p[u], pl[u], qx, qy, qz, ql, best, bidx);
  }
}

// Every lane scans the whole reference cloud (original order, so "strictly smaller" keeps the lowest index)
// for its own query; the reference point is wave-uniform and arrives through scalar loads.  Lanes without
// `want` run along and discard the result.  Same values as nmdist_fwd_c3_kernel for every input.
template <bool LAB>
__device__ __forceinline__ void lane_scan_cloud(const float* __restrict__ ref, const float* __restrict__ rlab,
                                                int nr, float qx, float qy, float qz, float ql, float& best,
                                                int& bidx) {
  best = __builtin_inff();
  bidx = 0;
  int k = 0;
  for (; k + 8 <= nr; k += 8) {
    float rr[24], rl[8];
    const float* __restrict__ rp = ref + 3 * (size_t)k;  // wave-uniform -> s_load
#pragma unroll
    for (int e = 0; e < 24; ++e) rr[e] = rp[e];
    if (LAB) {
#pragma unroll
      for (int e = 0; e < 8; ++e) rl[e] = rlab[k + e];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float d = pp::chamfer_d3(rr[3 * u], rr[3 * u + 1], rr[3 * u + 2], qx, qy, qz);
      if (LAB) d = rl[u] == ql ? d : __builtin_inff();
      const bool lt = d < best;
      best = lt ? d : best;
      bidx = lt ? k + u : bidx;
    }
  }
  for (; k < nr; ++k) {
    float d = pp::chamfer_d3(ref[3 * (size_t)k], ref[3 * (size_t)k + 1], ref[3 * (size_t)k + 2], qx, qy, qz);
    if (LAB) d = rlab[k] == ql ? d : __builtin_inff();
    const bool lt = d < best;
    best = lt ? d : best;
    bidx = lt ? k : bidx;
  }
  if (LAB && !(best < __builtin_inff())) {
    best = 0.0f;
    bidx = -1;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The search kernel.  One lane per query; the queries are walked in the SORTED ORDER OF THEIR OWN CLOUD --
// available for free because each cloud is the other direction's reference set -- so a wave's 64 queries are
// neighbours in space.  Walking the candidates straight from global memory (round 1) made every lane's
// 16-byte load touch its own cache line: 3.0e7 L1 accesses per launch at config 2.  Here the wave first
// copies the part of the sorted reference cloud its queries can reach in stage A into its own slice of LDS
// with coalesced loads and walks it from there (1.4e7 L1 accesses, and no brute-force list / third launch).
// A wave's 64 queries are consecutive in the cell order of their own cloud, so their stage-A blocks touch a handful of
// rows in two to four z-layers of the reference grid (~150-300 points on a surface).  The wave finds the
// layers and the row range in each with DPP reductions, fetches the spans' bounds (two loads per layer),
// copies the spans into its own slice of LDS and walks them from there.  No workgroup barrier anywhere: the
// four waves of a workgroup are independent, 28-32 of them share a CU, and the chain of dependent loads of
// one (query -> span bounds -> span copy) is covered by the others.
// The search inside a crowded cell's own grid (grid_common.h: SubGrid), one lane: first the 2x2x2 block of
// sub-cells nearest to the query (clamped into the sub-grid exactly as a query is clamped into the top-level
// grid), then cubes of Chebyshev radius 1, 2, ... around its sub-cell, until the best distance found so far
// (anywhere) lies below what the examined part guarantees for the rest OF THIS CELL, or the cube covers the
// sub-grid.  Returns that guarantee (+inf when the whole cell has been examined): the cell settles itself, so
// that only the top-level block's own reach is left to decide whether the query is done.
template <bool LAB>
__device__ __forceinline__ float sub_cell_search(const pp::SubGrid sg, const unsigned* __restrict__ tbl,
                                                 const pp::f4* __restrict__ sorted, const float* __restrict__ slab,
                                                 float qx, float qy, float qz, float ql, float& best, int& bidx) {
  const int cx = cell_coord(qx, sg.minx, sg.invh, sg.gx);
  const int cy = cell_coord(qy, sg.miny, sg.invh, sg.gy);
  const int cz = cell_coord(qz, sg.minz, sg.invh, sg.gz);
  const float fx = (qx - sg.minx) * sg.invh - (float)cx, fy = (qy - sg.miny) * sg.invh - (float)cy,
              fz = (qz - sg.minz) * sg.invh - (float)cz;
  auto walk_box = [&](int x0, int x1, int y0, int y1, int z0, int z1) {
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const int base = (z * sg.gy + y) * sg.gx;
        const unsigned rs = tbl[base + x0], re = tbl[base + x1 + 1];
        walk_range<LAB>(rs, re, sorted, slab, qx, qy, qz, ql, best, bidx);
      }
  };
  {
    const int sx = fx < 0.5f ? -1 : 1, sy = fy < 0.5f ? -1 : 1, sz = fz < 0.5f ? -1 : 1;
    walk_box(max(min(cx, cx + sx), 0), min(max(cx, cx + sx), sg.gx - 1), max(min(cy, cy + sy), 0),
             min(max(cy, cy + sy), sg.gy - 1), max(min(cz, cz + sz), 0), min(max(cz, cz + sz), sg.gz - 1));
    const float reach = sg.h * fminf(block_reach(fx, sx, cx, sg.gx), fminf(block_reach(fy, sy, cy, sg.gy), block_reach(fz, sz, cz, sg.gz)));
    if (best < reach * reach * kBoundSlack) return reach;
  }
  for (int rho = 1;; ++rho) {
    const int x0 = max(cx - rho, 0), x1 = min(cx + rho, sg.gx - 1), y0 = max(cy - rho, 0), y1 = min(cy + rho, sg.gy - 1),
              z0 = max(cz - rho, 0), z1 = min(cz + rho, sg.gz - 1);
    walk_box(x0, x1, y0, y1, z0, z1);
    auto axis = [&](float f, int c, int gdim) {
      const float lo = c - rho >= 1 ? (float)rho + f : __builtin_inff();
      const float hi = c + rho <= gdim - 2 ? (float)(rho + 1) - f : __builtin_inff();
      return fminf(lo, hi);
    };
    const float reach = sg.h * fminf(axis(fx, cx, sg.gx), fminf(axis(fy, cy, sg.gy), axis(fz, cz, sg.gz)));
    if (!(reach < __builtin_inff()) || best < reach * reach * kBoundSlack) return reach;
    // a query outside the cell's points (it sits in a neighbouring cell) would widen the cube sub-cell by sub-cell
    // up to the whole sub-grid, one lane's loads at a time: leave it to the stages of the whole wave
    if (rho >= kSubMaxRho) return reach;
  }
}

// Second-level search for one lane whose 2x2x2 block touches crowded cells: the block's cells one by one, an
// ordinary cell point by point, a crowded one through its own grid.  Returns the threshold below which the best
// distance settles the query.  Out of line: it runs for clustered data only, and inlined it costs the search
// kernel 16 VGPRs (a wave per SIMD) on the evenly sampled surfaces that never call it.
// (results BY VALUE: a reference argument of a function that is not inlined lives in scratch memory, and the
// caller then stores to it on its main path)
struct Found {
  float best;
  int bidx;
  float aux;  // lane_cube_search: 1 settled, 0 not, 2 gave up; refined_block_search: the threshold it reached
};
// The whole-wave cubes for the queries whose scans are long (rows through crowded cells), one after the other, with
// the pipelined scan.  Out of line; never called for evenly sampled surfaces.  (The grid's descriptor by POINTER:
// by value it would occupy sixteen of the registers a callee may use without saving them.)
struct OpenMask {
  unsigned lo, hi;      // queries the cube of radius 2 left open
  unsigned llo, lhi;    // serve_pending: queries whose scans are long (for serve_long_scans)
};
template <bool LAB, int W>
__device__ __attribute__((noinline)) OpenMask serve_long_scans(const GridSet* __restrict__ gp,
                                                               const unsigned* __restrict__ cell_start,
                                                               const pp::f4* __restrict__ sorted,
                                                               const float* __restrict__ slab, float* __restrict__ od,
                                                               int* __restrict__ oi, float qx, float qy, float qz,
                                                               float ql, int j, unsigned todo_lo, unsigned todo_hi) {
  const GridSet g = *gp;
  const int lane = threadIdx.x & 63;
  unsigned long long todo = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)todo_hi) << 32) |
                            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)todo_lo);
  unsigned long long open = 0ull;
  while (todo) {
    const int l = (int)__builtin_ctzll(todo);
    todo &= todo - 1;
    const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), l));
    const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), l));
    const float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), l));
    const float wl = LAB ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ql), l)) : 0.0f;
    const int wj = __builtin_amdgcn_readlane(j, l);
    float wbest;
    int widx;
    bool skipped = false;
    // (round 5, measured and removed: for a query that comes with a candidate, the rows of the BALL around it in one exact
    //  scan instead of the cubes of radius 1 and 2 -- two_scales 0.295 -> 0.305 / 0.358 / 0.415 / 0.437 ms for balls of up
    //  to 0.5 / 1 / 1.5 / 2.5 cells: the scan's row lookup is unrolled for the cubes' 9 and 25 rows and a loop for a ball's)
    if (wide_stages_wave<LAB, true>(wx, wy, wz, wl, g, cell_start, sorted, slab, wbest, widx, skipped)) {
      if (lane == 0) {
        od[wj] = wbest;
        oi[wj] = widx;
      }
    } else {
      open |= 1ull << l;
    }
  }
  OpenMask o;
  o.lo = (unsigned)open;
  o.hi = (unsigned)(open >> 32);
  o.llo = o.lhi = 0u;
  return o;
}

// The queries of a wave left for the whole-wave cubes, one after the other (out of line: one call per wave that has
// any -- one in five at config 2 -- so the cubes' code and registers are not the search kernel's).  Short scans are
// done here; a query whose cubes run through crowded cells is reported back for serve_long_scans.  Writes the results
// of the queries it settles; returns the masks of those the cube of radius 2 left open and of the long scans.
template <bool LAB, int W>
__device__ __attribute__((noinline)) OpenMask serve_pending(const GridSet* __restrict__ gp,
                                                            const unsigned* __restrict__ cell_start,
                                                            const pp::f4* __restrict__ sorted,
                                                            const float* __restrict__ slab, float* __restrict__ od,
                                                            int* __restrict__ oi, float qx, float qy, float qz, float ql,
                                                            int j, unsigned pending_lo, unsigned pending_hi) {
  const GridSet g = *gp;
  const int lane = threadIdx.x & 63;
  unsigned long long pending = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)pending_hi) << 32) |
                               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)pending_lo);
  unsigned long long open = 0ull, longscan = 0ull;
  while (pending) {
    const int l = (int)__builtin_ctzll(pending);
    pending &= pending - 1;
    const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), l));
    const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), l));
    const float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), l));
    const float wl = LAB ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ql), l)) : 0.0f;
    const int wj = __builtin_amdgcn_readlane(j, l);
    float wbest;
    int widx;
    bool skipped = false;
    const bool settled = wide_stages_wave<LAB, false>(wx, wy, wz, wl, g, cell_start, sorted, slab, wbest, widx, skipped);
    if (skipped) {
      longscan |= 1ull << l;
    } else if (settled) {
      if (lane == 0) {
        od[wj] = wbest;
        oi[wj] = widx;
      }
    } else {
      open |= 1ull << l;
    }
  }
  // (a LEAF: values that had to live across a call in here would sit in registers the function must save and
  //  restore -- 18 MB of scratch traffic per launch at config 2 when it called serve_long_scans itself)
  OpenMask o;
  o.lo = (unsigned)open;
  o.hi = (unsigned)(open >> 32);
  o.llo = (unsigned)longscan;
  o.lhi = (unsigned)(longscan >> 32);
  return o;
}

template <bool LAB, int W>
__device__ __attribute__((noinline)) Found refined_block_search(
    const GridSet g, const unsigned* __restrict__ cell_start, const pp::f4* __restrict__ sorted,
    const float* __restrict__ slab, const unsigned* __restrict__ sub_start, const pp::SubGrid* __restrict__ sub_desc,
    float qx, float qy, float qz, float ql, int cx, int cy, int cz, int sx, int sy, int sz, float reach) {
  const int x0 = max(min(cx, cx + sx), 0), x1 = min(max(cx, cx + sx), g.gx - 1);
  float bound = reach;
  float bb = __builtin_inff();
  int bi = 0x7fffffff;
  for (int e = 0; e < 8; ++e) {
    const int z = cz + (e >> 2) * sz, y = cy + ((e >> 1) & 1) * sy;
    if (z < 0 || z >= g.gz || y < 0 || y >= g.gy || ((e & 1) && x1 == x0)) continue;
    const int lin = pp::cell_linear((e & 1) ? x1 : x0, y, z, g.gx, g.gy);
    const unsigned cs = cell_start[lin], ce = cell_start[lin + 1];
    if (ce - cs <= (unsigned)pp::kCrowd) {
      walk_range<LAB>(cs, ce, sorted, slab, qx, qy, qz, ql, bb, bi);
    } else {
      bound = fminf(bound, sub_cell_search<LAB>(sub_desc[(cs + pp::kCrowd - 1) / pp::kCrowd], sub_start + 2 * (size_t)cs,
                                                sorted, slab, qx, qy, qz, ql, bb, bi));
    }
  }
  Found o;
  o.best = bb;
  o.bidx = bi;
  o.aux = bound * bound * kBoundSlack;
  return o;
}

// The queries a wave could not settle inside the cube of radius 2 (far from the reference cloud: disjoint clouds,
// clusters at different places, the tail of a Gaussian), served by the WHOLE wave, group by group.  A group is the
// open queries within r of one of them (the seed), r = a quarter of the distance the seed's neighbour can be at (its
// best so far; if it has seen no candidate yet, the nearest of one sample point per cell row).  Every member then
// has a neighbour within U = min(largest best-so-far of the group, (that distance + sqrt(3) r)^2), so what the
// group can need lies in the cell rows within sqrt(U) of its bounding box, and in each row between the cells
// that the rest of the budget allows along x.  Those row pieces are laid end to end and staged through the wave's slice
// of LDS, 256 points at a time; every lane walks every staged point (LDS broadcast) keeping its own exact
// (distance, index) minimum -- brute force restricted to the rows that can matter.  Extra candidates are
// harmless, so no lane is masked.  Finite queries only.  Labeled searches: a candidate counts for a query of the
// same label only, so a seed without a candidate samples points of its own label and takes along queries of its
// label only (a label nobody carries ends in one scan of the whole cloud, which settles every open query).
template <bool LAB, int W>
__device__ __attribute__((noinline)) Found wave_group_search(const GridSet g, const unsigned* __restrict__ cell_start,
                                                             const pp::f4* __restrict__ sorted,
                                                             const float* __restrict__ slab, float qx, float qy,
                                                             float qz, float ql, float best, int bidx, unsigned open_lo,
                                                             unsigned open_hi, lds_f4_wptr lw, lds_f_wptr lwl,
                                                             int row_room, const unsigned* __restrict__ rowbits) {
  const lds_f4_ptr lr = (lds_f4_ptr)lw;
  const lds_f_ptr lrl = (lds_f_ptr)lwl;
  const int lane = threadIdx.x & 63;
  const float inf = __builtin_inff();
  unsigned long long open = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)open_hi) << 32) |
                            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)open_lo);
  auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
  // (a point may sit in the neighbouring cell by the rounding of its cell coordinate, and the faces themselves
  //  are rounded: every gap is shortened by this much)
  // ... or by a few ulps of the coordinates themselves where the cloud sits far from the origin against its extent
  const float slack = fmaxf(1.0e-4f * g.h, 5.0e-7f * (fabsf(g.minx) + fabsf(g.miny) + fabsf(g.minz) +
                                                        (float)(g.gx + g.gy + g.gz) * g.h));
  // distance from the interval [blo, bhi] to the slab of cell c along one axis (rim cells hold the outliers: they
  // extend to infinity)
  // (round 3) ... where the box was trimmed to the bulk of the cloud.  An untrimmed box is the bounding box of all
  // points: its rim cells end where it ends -- between far clouds every nearest face is a rim cell, and open rims
  // made every row of them a candidate (2500 candidates per group between disjoint clouds)
  const bool rim_open = g.pad[2] != 0;
  auto axis_gap = [&](float blo, float bhi, float mn, int c, int gdim) {
    const float lo = (c == 0 && rim_open) ? -inf : mn + (float)c * g.h,
                hi = (c == gdim - 1 && rim_open) ? inf : mn + (float)(c + 1) * g.h;
    return fmaxf(fmaxf(lo - bhi, blo - hi) - slack, 0.0f);
  };
  // Round 6: the set's row bitmap (bit y + gy z: the cell row (y, z) holds points; nullptr: not available, every row
  // counts).  Between clusters a group's box is hundreds of rows of which a handful hold anything: the sampling, the row
  // list and the cuts below take the non-empty ones only.  Lane l holds word l & 31.
  // (used where at most half of the rows hold points: on a filled grid the lookups cost a Gaussian 3 % and skip nothing)
  const unsigned rbw = rowbits != nullptr ? rowbits[lane & 31] : 0xffffffffu;
  const unsigned rb_pc = lane < 32 ? (unsigned)__builtin_popcount(rbw) : 0u;
  const unsigned rb_incl = pp::wave_scan_u32_dpp(rb_pc);
  const int rb_rows = __builtin_amdgcn_readlane((int)rb_incl, 63);
  const bool have_rb = rowbits != nullptr && 2 * rb_rows <= g.gy * g.gz;  // (wave-uniform)
  auto row_has_points = [&](int row) {  // row = y + gy z (any lane, any row < 1024)
    return ((unsigned)__shfl((int)rbw, row >> 5) >> (row & 31)) & 1u;
  };
#ifdef PP_QUERY_PROBE
  unsigned long long pp_ngroups = 0, pp_nblind = 0, pp_ncand = 0, pp_nrows = 0, pp_nwalk = 0, pp_nlist = 0;
#endif
  PP_GT_DECL;
  while (open) {
    const int seed = (int)__builtin_ctzll(open);
    const float sx = rl(qx, seed), sy = rl(qy, seed), sz = rl(qz, seed);
    const float sl = LAB ? rl(ql, seed) : 0.0f;
    float us = rl(best, seed);
    // wave-uniform: the seed has seen no candidate yet, or only one picked up by accident far outside its cubes
    const bool blind = !(us < 16.0f * g.h * g.h);
    bool have_smp = false;  // (blind seeds) the nearest of the sample points, a bound for every member
    float spx = 0.0f, spy = 0.0f, spz = 0.0f;
    if (blind) {
      // one sample per non-empty cell row -- the first point at or after the seed's cell along x, else the row's last
      // point -- kSmp chunks of rows in flight; the nearest sample bounds the seed's neighbour
      const int cxs = cell_coord(sx, g.minx, g.invh, g.gx);
      float u1 = inf;
      float smx = 0.0f, smy = 0.0f, smz = 0.0f;  // this lane's nearest sample
      constexpr int kSmp = 6;  // chunks of 64 rows in flight (676 rows of a 26^3 grid: two rounds of two dependent loads)
      // (the rows by rank among the non-empty ones: the word by bisection over the words' prefix counts, the bit by halving)
      const int nall = have_rb ? rb_rows : g.gy * g.gz;
      const bool rb_select = have_rb;
      auto kth_row = [&](int k) {  // k < nall
        int lo = 0, hi = 31;
#pragma unroll
        for (int it = 0; it < 5; ++it) {
          const int mid = (lo + hi + 1) >> 1;
          const unsigned before = (unsigned)__shfl((int)(rb_incl - rb_pc), mid);  // set bits in the words before `mid`
          const bool ge = before <= (unsigned)k;
          lo = ge ? mid : lo;
          hi = ge ? hi : mid - 1;
        }
        unsigned v = (unsigned)__shfl((int)rbw, lo);
        int rnk = k - (int)(unsigned)__shfl((int)(rb_incl - rb_pc), lo), pos = 0;
#pragma unroll
        for (int wdt = 16; wdt >= 1; wdt >>= 1) {
          const int c = __builtin_popcount(v & ((1u << wdt) - 1u));
          const bool up = rnk >= c;
          rnk -= up ? c : 0;
          v = up ? v >> wdt : v;
          pos += up ? wdt : 0;
        }
        return lo * 32 + pos;
      };
      for (int r0 = 0; r0 < nall; r0 += 64 * kSmp) {
        unsigned rs[kSmp], rm[kSmp], re[kSmp];
#pragma unroll
        for (int u = 0; u < kSmp; ++u) {
          const int rk = r0 + u * 64 + lane;
          const bool ok = rk < nall;
          int r = ok ? rk : 0;
          if (rb_select && r0 + u * 64 < nall) r = kth_row(r);  // (wave-uniform condition: whole chunks beyond the last row skip it)
          const int base = r * g.gx;
          rs[u] = cell_start[base];
          rm[u] = cell_start[base + cxs];
          re[u] = ok ? cell_start[base + g.gx] : rs[u];
        }
        pp::f4 smp[kSmp];
        float sml[kSmp] = {};
#pragma unroll
        for (int u = 0; u < kSmp; ++u) {
          const unsigned at = re[u] > rs[u] ? min(rm[u], re[u] - 1) : 0u;
          smp[u] = sorted[at];
          if (LAB) sml[u] = slab[at];
        }
#pragma unroll
        for (int u = 0; u < kSmp; ++u) {
          const float d = pp::chamfer_d3(smp[u].x, smp[u].y, smp[u].z, sx, sy, sz);
          const bool tk = re[u] > rs[u] && (!LAB || sml[u] == sl) && d < u1;
          u1 = tk ? d : u1;
          smx = tk ? smp[u].x : smx;
          smy = tk ? smp[u].y : smy;
          smz = tk ? smp[u].z : smz;
        }
      }
      const float mine = u1;
      u1 = -pp::wave_reduce_dpp<false>(-u1);  // (min; DPP)
      us = fminf(us, u1 * 1.0001f);
      // the nearest sample itself (a real point of the cloud, of the seed's label): every query that joins the group
      // has a neighbour within ITS OWN distance to it -- a far tighter bound for the members than the triangle
      // inequality through the seed (round 3: disjoint clouds examined ~2000 candidates per query with that one)
      have_smp = u1 < inf;
      if (have_smp) {
        const int wl = (int)__builtin_ctzll(__ballot(mine == u1));
        spx = rl(smx, wl);
        spy = rl(smy, wl);
        spz = rl(smz, wl);
      }
    }
    PP_GT(0);
    // The seed takes along the open queries within r of it, r = a quarter of the distance of its candidate.  Where
    // the seed knew a candidate: at least two cells, and only queries whose own candidate is no more than twice as
    // far (the group's bound is the largest of them).  Where it did not (the bound is the sample's, a crude one):
    // any open query within r -- it has a neighbour within that distance + sqrt(3) r through the seed.
#ifdef PP_QUERY_PROBE
    ++pp_ngroups;
    pp_nblind += blind ? 1 : 0;
#endif
    const float ds = sqrtf(us);
    const float r = blind ? kBlindGroup * ds : fmaxf(2.0f * g.h, 0.25f * ds);
    const float via = ds + 1.7321f * r;
    const bool member = (((open >> lane) & 1ull) != 0ull && (blind ? (!LAB || ql == sl) : best <= 4.0f * us) &&
                         !(fmaxf(fabsf(qx - sx), fmaxf(fabsf(qy - sy), fabsf(qz - sz))) > r)) ||
                        lane == seed;
    open &= ~__ballot(member);
    float v[6] = {member ? -qx : -inf, member ? -qy : -inf, member ? -qz : -inf,
                  member ? qx : -inf,  member ? qy : -inf,  member ? qz : -inf};
    pp::wave_reduce6_dpp<false, 6>(v);
    const float blx = -rl(v[0], 63), bly = -rl(v[1], 63), blz = -rl(v[2], 63);
    const float bhx = rl(v[3], 63), bhy = rl(v[4], 63), bhz = rl(v[5], 63);
    // every member has a neighbour within its own best so far (the seed: within us)
    float ub = member ? (lane == seed ? us : (blind ? fminf(best, via * via) : best)) : 0.0f;
    if (have_smp && member && lane != seed && (!LAB || ql == sl))
      ub = fminf(ub, pp::chamfer_d3(spx, spy, spz, qx, qy, qz) * 1.0001f);
    float ubm = ub * 1.0001f;  // the member's own bound (kept per lane: the rows are cut member by member below)
    const unsigned long long members = __ballot(member);
    // The group along x in kSub pieces, each with its own extent and bound (round 3): 64 consecutive queries of the
    // sorted cloud run along x, and a candidate must lie within SOME member's bound -- cutting a cell row by the whole
    // box and the largest bound made far clouds examine the union of the box's ends' needs (6000 candidates per query
    // between disjoint clouds).  A row's cut along x is the hull of the pieces' cuts.
    constexpr int kSub = 4;
    float sbl[kSub], sbh[kSub], su[kSub];
    const float wx = bhx - blx;
    if (!(wx > 8.0f * g.h)) {  // (wave-uniform) a short group: one piece (the reductions below cost a small group more
                               // than the cut saves: the tail of a Gaussian makes 80000 groups of a few queries)
      sbl[0] = blx;
      sbh[0] = bhx;
      su[0] = pp::wave_reduce_dpp<false>(ub) * 1.0001f;
#pragma unroll
      for (int k = 1; k < kSub; ++k) {
        sbl[k] = blx;
        sbh[k] = bhx;
        su[k] = -1.0f;
      }
    } else {
      const int mypiece = min(kSub - 1, (int)((qx - blx) / wx * (float)kSub));
#pragma unroll
      for (int k = 0; k < kSub; ++k) {
        const bool in = member && mypiece == k;
        float a = in ? -qx : -inf, b2 = in ? qx : -inf, c = in ? ub : -1.0f;
        a = pp::wave_reduce_dpp<false>(a);
        b2 = pp::wave_reduce_dpp<false>(b2);
        c = pp::wave_reduce_dpp<false>(c);
        sbl[k] = -a;
        sbh[k] = b2;
        su[k] = c >= 0.0f ? c * 1.0001f : -1.0f;  // (a piece without members cuts nothing)
      }
    }
    ub = pp::wave_reduce_dpp<false>(ub);
    float U = ub * 1.0001f;
#ifdef PP_QUERY_PROBE
    {
      const unsigned pp_w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
      if (lane == 0 && pp_w < (unsigned)kQWaves && pp_ngroups == 1) {
        g_qgw[pp_w][6] = __float_as_uint(bhx - blx);
        g_qgw[pp_w][7] = __float_as_uint(bhy - bly);
        g_qgw[pp_w][8] = __float_as_uint(bhz - blz);
        g_qgw[pp_w][9] = __float_as_uint(sqrtf(us));
        g_qgw[pp_w][10] = __float_as_uint(sqrtf(U));
        g_qgw[pp_w][11] = (unsigned)__builtin_popcountll(members);
        g_qgw[pp_w][12] = __float_as_uint(g.h);
      }
    }
#endif
    // the rows within sqrt(U) of the box (cell coordinates are monotonic in the coordinate: exact)
    const bool bounded = U < inf;
    const float R = bounded ? fast_sqrt(U) * 1.0001f + slack : 0.0f;
    const int y0 = bounded ? cell_coord(bly - R, g.miny, g.invh, g.gy) : 0;
    const int y1 = bounded ? cell_coord(bhy + R, g.miny, g.invh, g.gy) : g.gy - 1;
    const int z0 = bounded ? cell_coord(blz - R, g.minz, g.invh, g.gz) : 0;
    const int z1 = bounded ? cell_coord(bhz + R, g.minz, g.invh, g.gz) : g.gz - 1;
    const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
    const float inv_ny = 1.0f / (float)ny;
#ifdef PP_QUERY_PROBE
    pp_nrows += (unsigned long long)nrows;
#endif
    // The rows' cut along x by the pieces (the hull of the pieces' cuts; a piece whose cut ends before an untrimmed
    // box begins, or begins behind its end, has nothing in this row -- cell_coord would clamp it into the rim cell)
    const float box_x0 = g.minx - slack, box_x1 = g.minx + (float)g.gx * g.h + slack;
    auto piece_cut = [&](int rr, int cy, int cz, float& xlo, float& xhi) {
      xlo = inf;
      xhi = -inf;
      // (the bitmap's words travel through ds_bpermute: looked up by EVERY lane before the lanes part -- a disabled lane's
      //  word reads as zero -- and for a row inside the table whatever rr is)
      const bool has_points = row_has_points((cy + g.gy * cz) & (pp::kGridMax * pp::kGridMax - 1)) != 0u;
      if (rr >= nrows || !has_points) return;
      const float gy_ = axis_gap(bly, bhy, g.miny, cy, g.gy), gz_ = axis_gap(blz, bhz, g.minz, cz, g.gz);
      const float gyz = __builtin_fmaf(gz_, gz_, gy_ * gy_) * 0.9999f;
      if (!bounded) {
        xlo = -inf;
        xhi = inf;
        return;
      }
#pragma unroll
      for (int k = 0; k < kSub; ++k) {
        const float remk = su[k] - gyz;
        const float rxk = fast_sqrt(fmaxf(remk, 0.0f)) * 1.0001f + slack;
        const float lo = sbl[k] - rxk, hi = sbh[k] + rxk;
        const bool ok = remk >= 0.0f && (rim_open || (hi >= box_x0 && lo <= box_x1));
        xlo = ok ? fminf(xlo, lo) : xlo;
        xhi = ok ? fmaxf(xhi, hi) : xhi;
      }
    };
    PP_GT(1);
    // Pass 1 (round 3): the rows the pieces' cuts leave, packed into a list in the wave's slice behind the candidate
    // batch (<= 1024 rows, two bytes each).  Between far clouds that is a tenth of the rows of the box around the
    // group; the member-by-member cut below is then paid for full blocks of rows that have a chance.
    typedef unsigned short __attribute__((address_space(3))) * lds_u16_ptr;
    const lds_u16_ptr rowq = (lds_u16_ptr)(lw + kGroupBatch);
    int nq = 0;
    // (a box of at most 64 rows is one block anyway: no list; row_room: the two-byte entries the wave's slice holds
    //  behind the candidate batch -- 1056 for slices of 384 points, the default; a smaller slice lists smaller boxes only)
    const bool listed = nrows > 64 && nrows <= row_room;
    if (!listed) nq = nrows;  // every block of rows
    for (int r0 = 0; listed && r0 < nrows; r0 += 64) {  // wave-uniform
      const int rr = r0 + lane;
      // (rr < 2^11, ny < 2^6: the rounded product is the exact quotient)
      const int dz = (int)(((float)rr + 0.5f) * inv_ny);
      float xlo, xhi;
      piece_cut(rr, y0 + (rr - dz * ny), z0 + dz, xlo, xhi);
      const bool pass = xlo <= xhi;
      const unsigned long long bal = __ballot(pass);
      if (pass) rowq[nq + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] =
          (unsigned short)rr;
      nq += (int)__builtin_popcountll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef PP_QUERY_PROBE
    pp_nlist += (unsigned long long)nq;
#endif
    bool dirty = false;
    PP_GT(2);
    for (int r0 = 0; r0 < nq; r0 += 64) {  // wave-uniform
      PP_GT(5);
      if (dirty) {  // (wave-uniform) candidates have been examined since the bound was last taken
        // the bound follows what the members have found in the rows examined so far: the rows still to come are cut
        // by the candidates already seen (a member's own best is always a valid bound for it)
        dirty = false;
        U = fminf(U, pp::wave_reduce_dpp<false>(member ? best : 0.0f) * 1.0001f);
#pragma unroll
        for (int k = 0; k < kSub; ++k) su[k] = fminf(su[k], U);
        ubm = fminf(ubm, best * 1.0001f);
      }
      int rr = r0 + lane < nq ? (listed ? (int)rowq[r0 + lane] : r0 + lane) : nrows;  // (travels with the row's span below)
      int cy, cz;
      auto row_coords = [&]() {
        const int dz = (int)(((float)rr + 0.5f) * inv_ny);
        cz = z0 + dz;
        cy = y0 + (rr - dz * ny);
      };
      row_coords();
      unsigned cs = 0u, len = 0u;
      float xlo, xhi;  // the row's cut along x
      piece_cut(rr, cy, cz, xlo, xhi);
      if (__ballot(xlo <= xhi) == 0ull) {  // wave-uniform: no row of these within the group's bound
        PP_GT(3);
        continue;
      }
      auto load_span = [&]() {
        cs = 0u;
        len = 0u;
        if (rr < nrows && xlo <= xhi) {
          const int x0 = cell_coord(xlo, g.minx, g.invh, g.gx);
          const int x1 = cell_coord(xhi, g.minx, g.invh, g.gx);
          const int base = pp::cell_linear(0, cy, cz, g.gx, g.gy);
          cs = cell_start[base + x0];
          len = cell_start[base + x1 + 1] - cs;
        }
      };
      // Member by member (round 3): the cut above measures from the BOX (its nearest face) with the LARGEST bound of
      // a piece -- between far clouds that is the box's diagonal too generous, thousands of candidates where every
      // member's own ball holds a handful.  A candidate of this row matters only if it lies within SOME member's own
      // bound: the hull of the members' own cuts, intersected with the cut above.
      // (the slack folded into the row's faces: gap = max3(lo' - q, q - hi', 0))
      auto member_cut = [&]() {
        const float ylo = ((cy == 0 && rim_open) ? -inf : g.miny + (float)cy * g.h) - slack,
                    yhi = ((cy == g.gy - 1 && rim_open) ? inf : g.miny + (float)(cy + 1) * g.h) + slack;
        const float zlo = ((cz == 0 && rim_open) ? -inf : g.minz + (float)cz * g.h) - slack,
                    zhi = ((cz == g.gz - 1 && rim_open) ? inf : g.minz + (float)(cz + 1) * g.h) + slack;
        // (round 5, measured and removed: the members taken eight consecutive lanes at a time, as their box and largest
        //  bound -- eight steps a block of rows instead of 64 -- : the boxes' cuts are so much wider than the members' own
        //  that disjoint clouds went 0.22 -> 0.64 ms, blobs8 0.70 -> 0.81)
        float mlo = inf, mhi = -inf;
        for (unsigned long long mm = members; mm; mm &= mm - 1ull) {
          const int m = (int)__builtin_ctzll(mm);
          const float mx = rl(qx, m), my = rl(qy, m), mz = rl(qz, m), mu = rl(ubm, m);
          const float gy_ = fmaxf(fmaxf(ylo - my, my - yhi), 0.0f), gz_ = fmaxf(fmaxf(zlo - mz, mz - zhi), 0.0f);
          const float remm = mu - __builtin_fmaf(gz_, gz_, gy_ * gy_) * 0.9999f;
          // (a member out of reach of the row: a cut of -inf width leaves the hull alone)
          const float rxm = remm >= 0.0f ? fast_sqrt(remm) * 1.0001f + slack : -inf;
          mlo = fminf(mlo, mx - rxm);
          mhi = fmaxf(mhi, mx + rxm);
        }
        xlo = fmaxf(xlo, mlo);
        xhi = fminf(xhi, mhi);
        if (!rim_open && (xhi < box_x0 || xlo > box_x1)) xhi = -inf;  // (no member reaches the box in this row)
      };
      load_span();
      // candidates of the pieces' cuts in these rows
      const unsigned box_total = (unsigned)__builtin_amdgcn_readlane((int)pp::wave_scan_u32_dpp(len), 63);
      // (the member-by-member cut costs the wave ~30 instructions per member: it pays when the cut above left more
      //  candidates than that buys examined by every lane)
      const bool by_member = bounded && members != (1ull << seed);  // (wave-uniform)
      if (by_member && box_total > (unsigned)kMemberCutMin * (unsigned)__builtin_popcountll(members)) {  // (wave-uniform)
        member_cut();
        load_span();
      }
      if (__ballot(len != 0u) == 0ull) {  // wave-uniform: nothing in these rows
        PP_GT(3);
        continue;
      }
      dirty = true;
      unsigned incl = pp::wave_scan_u32_dpp(len);
      unsigned rows_total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
      // Round 6: the rows of a block NEAREST FIRST (by the distance from the seed to the middle point of the row's span)
      // where they hold more than one batch, and a block of more than kSplitMin candidates IN PARTS: the nearest rows
      // that hold two batches first, then the rows still to come are cut again, member by member, by what the members
      // have found -- a group between clusters meets its neighbours' cluster in its first batches, and the other clusters
      // in reach of the first, crude bound (the nearest of one sample per row) are then out of every member's own reach,
      // whole rows of them.  In cell order, cut once, the waves that decide the length of blobs8's launch walked 7000
      // candidates of four clusters, every lane every point.  The order of the candidates does not matter to the result:
      // the minimum is the exact (distance, index) one.
      constexpr unsigned kSplitMin = 4u * (unsigned)kGroupBatch;  // (from 2 / 3 batches on, parts of 1 batch: blobs8 +1..4 %)
      bool ordered = false;
      {
        const unsigned long long nzr = __ballot(len != 0u);
        if (rows_total > (unsigned)kGroupBatch && (nzr & (nzr - 1ull)) != 0ull) {  // (wave-uniform)
          const pp::f4 mp = sorted[cs + (len >> 1)];  // (an empty row reads sorted[0]: its key is not used)
          const unsigned key = (unsigned)__float_as_int(pp::chamfer_d3(mp.x, mp.y, mp.z, sx, sy, sz));  // (bits: a total order)
          unsigned rank = 0u;
          for (unsigned long long m = nzr; m; m &= m - 1ull) {
            const int jn = (int)__builtin_ctzll(m);
            const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)key, jn);
            rank += (kj < key || (kj == key && jn < lane)) ? 1u : 0u;
          }
          const unsigned long long zr = ~nzr;  // the empty rows behind the others, in lane order: a permutation of the lanes
          const unsigned zrank = (unsigned)__builtin_popcountll(nzr) +
                                 __builtin_amdgcn_mbcnt_hi((unsigned)(zr >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)zr, 0u));
          rank = len != 0u ? rank : zrank;
          cs = (unsigned)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)cs);
          len = (unsigned)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)len);
          rr = __builtin_amdgcn_ds_permute((int)(rank << 2), rr);
          incl = pp::wave_scan_u32_dpp(len);
          ordered = true;
        }
      }
      unsigned done_rows = 0u;  // lanes (rows, in their order after the sort) examined so far
      for (;;) {                // wave-uniform: the block's parts
      unsigned upto = 64u;
      if (ordered && by_member && rows_total > kSplitMin)  // the leading rows that hold two batches (incl counts the rows not yet examined)
        upto = (unsigned)__builtin_ctzll(__ballot(incl >= 2u * (unsigned)kGroupBatch)) + 1u;
      const unsigned plen = ((unsigned)lane >= done_rows && (unsigned)lane < upto) ? len : 0u;
      const unsigned pincl = upto == 64u ? incl : pp::wave_scan_u32_dpp(plen);
      const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)pincl, 63);
      const unsigned excl = pincl - plen;
      const unsigned shift = cs - excl;  // candidate c of this lane's row sits at sorted[c + shift]
#ifdef PP_QUERY_PROBE
      pp_ncand += total;
#endif
      PP_GT(3);
      for (unsigned t0 = 0; t0 < total; t0 += kGroupBatch) {  // wave-uniform
        PP_GT(5);
        if (t0 != 0u) {
          // the bound follows the batches too (round 3): a crowded cell is a thousand candidates, and after the first
          // few hundred of them the members know their neighbour to within a little -- the test below then drops most
          // of the rest before the whole wave looks at them
          U = fminf(U, pp::wave_reduce_dpp<false>(member ? best : 0.0f) * 1.0001f);
#pragma unroll
          for (int k = 0; k < kSub; ++k) su[k] = fminf(su[k], U);
        }
        // kGroupBatch candidates into the wave's slice of LDS: a lane per candidate, its row found by bisection over
        // the lanes' offsets (the last lane whose first candidate is <= c; empty rows are passed over because the
        // row after them starts at the same offset)
        pp::f4 pt[kGroupBatch / 64];
        float ptl[kGroupBatch / 64];
#pragma unroll
        for (int u = 0; u < kGroupBatch / 64; ++u) {
          const unsigned c = min(t0 + (unsigned)(u * 64 + lane), total - 1);  // (the tail repeats the last candidate)
          int lo = 0, hi = 63;
#pragma unroll
          for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi + 1) >> 1;
            const bool ge = (unsigned)__shfl((int)excl, mid) <= c;
            lo = ge ? mid : lo;
            hi = ge ? hi : mid - 1;
          }
          const unsigned at = c + (unsigned)__shfl((int)shift, lo);
          pt[u] = sorted[at];
          ptl[u] = LAB ? slab[at] : 0.0f;
        }
        // One lane, one candidate: is it within the bound of the piece of the group it is nearest to?  (The row cuts
        // keep whole cells; a candidate outside every piece's bound is farther from every member than what that member
        // already has -- strictly, with room for the rounding of this test -- and is dropped here, for one lane's ~30
        // operations instead of eight by each of the 64.)  The survivors are packed; the tail up to a multiple of four
        // is filled with points at infinity.
        unsigned cnt = 0u;
        const bool sift = bounded && total - t0 > 64u;  // (wave-uniform; a few candidates are cheaper examined than sifted)
#pragma unroll
        for (int u = 0; u < kGroupBatch / 64; ++u) {
          const pp::f4 c4 = pt[u];
          const float ey = fmaxf(fmaxf(bly - c4.y, c4.y - bhy), 0.0f), ez = fmaxf(fmaxf(blz - c4.z, c4.z - bhz), 0.0f);
          const float eyz = __builtin_fmaf(ez, ez, ey * ey);
          bool keep = !sift;
#pragma unroll
          for (int k = 0; k < kSub; ++k) {
            const float ex = fmaxf(fmaxf(sbl[k] - c4.x, c4.x - sbh[k]), 0.0f);
            keep = keep || (__builtin_fmaf(ex, ex, eyz) * 0.9999f <= su[k]);  // (su < 0: a piece without members)
          }
          keep = keep && t0 + (unsigned)(u * 64 + lane) < total;
          const unsigned long long bal = __ballot(keep);
          const unsigned at = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          if (keep) {
            lw[at] = c4;
            if (LAB) lwl[at] = ptl[u];
          }
          cnt += (unsigned)__builtin_popcountll(bal);
        }
        const unsigned n = (cnt + 3u) & ~3u;
        if (lane < (int)(n - cnt)) {
          pp::f4 far4;
          far4.x = far4.y = far4.z = inf;
          far4.w = __int_as_float(0x7fffffff);
          lw[cnt + lane] = far4;
          if (LAB) lwl[cnt + lane] = ql;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef PP_QUERY_PROBE
        pp_nwalk += n;
#endif
        PP_GT(4);
        if (n == 0u) continue;  // (wave-uniform)
        // Every lane, every candidate (uniform address: LDS broadcast), trimmed for VALU issue like the staged walk of
        // stage A: per group of four only the running minimum (v_min3 + v_min) and the group that last lowered it;
        // the winner's index is recovered from that group afterwards (it is still in the slice).  A distance EQUAL to
        // the running minimum (duplicates, lattices, or the candidate the lane already holds) cannot be ordered that
        // way: the wave then repeats the batch with the exact (distance, index) comparison.
        const float best0 = best;
        const int bidx0 = bidx;
        unsigned gi = 0xffffffffu;
        bool tie = false;
        for (unsigned i = 0; i < n; i += 4) {
          pp::f4 q4[4];
          float l4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            q4[u] = lr[i + u];
            if (LAB) l4[u] = lrl[i + u];
          }
          float d[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            d[u] = pp::chamfer_d3(q4[u].x, q4[u].y, q4[u].z, qx, qy, qz);
            if (LAB) d[u] = l4[u] == ql ? d[u] : inf;
          }
          const float gmin = fminf(pp::min3(d[0], d[1], d[2]), d[3]);
          tie = tie | ((gmin == best) & (gmin < inf));
          const bool lt = gmin < best;
          gi = lt ? i : gi;
          best = lt ? gmin : best;
        }
        if (__any(tie)) {  // exact redo of the batch (rare)
          best = best0;
          bidx = bidx0;
          for (unsigned i = 0; i < n; i += 4) {
            pp::f4 q4[4];
            float l4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              q4[u] = lr[i + u];
              if (LAB) l4[u] = lrl[i + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) take_candidate<LAB>(q4[u], l4[u], qx, qy, qz, ql, best, bidx);
          }
        } else if (gi != 0xffffffffu) {  // the winner is in group gi: lowest index among its minima
          int cand = 0x7fffffff;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const pp::f4 q = lr[gi + u];
            float du = pp::chamfer_d3(q.x, q.y, q.z, qx, qy, qz);
            if (LAB) du = lrl[gi + u] == ql ? du : inf;
            const int id = __float_as_int(q.w);
            cand = ((du == best) & (id < cand)) ? id : cand;
          }
          bidx = cand;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      done_rows = upto;
      if (done_rows >= 64u) break;
      // the rows still to come, cut again by the bounds as they are now
      U = fminf(U, pp::wave_reduce_dpp<false>(member ? best : 0.0f) * 1.0001f);
#pragma unroll
      for (int k = 0; k < kSub; ++k) su[k] = fminf(su[k], U);
      ubm = fminf(ubm, best * 1.0001f);
      row_coords();
      piece_cut(rr, cy, cz, xlo, xhi);
      member_cut();
      load_span();
      len = (unsigned)lane >= done_rows ? len : 0u;
      incl = pp::wave_scan_u32_dpp(len);
      rows_total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
      if (rows_total == 0u) break;
      }
    }
  }
#ifdef PP_QUERY_PROBE
  {
    const unsigned pp_w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (lane == 0 && pp_w < (unsigned)kQWaves) {
      g_qgw[pp_w][0] = (unsigned)pp_ngroups;
      g_qgw[pp_w][1] = (unsigned)pp_nblind;
      g_qgw[pp_w][2] = (unsigned)pp_ncand;
      g_qgw[pp_w][3] = (unsigned)pp_nwalk;
      g_qgw[pp_w][4] = (unsigned)pp_nlist;
      g_qgw[pp_w][5] = (unsigned)__builtin_popcountll(((unsigned long long)open_hi << 32) | open_lo);
    }
  }
#endif
#if defined(PP_QUERY_PROBE) && !defined(PP_QUERY_PROBE_NO_GROUP_STATS)  // (same-address atomics: they distort p7)
  if (lane == 0) {
    atomicAdd(&g_qgroup[0], 1ull);
    atomicAdd(&g_qgroup[1], pp_ngroups);
    atomicAdd(&g_qgroup[2], pp_nblind);
    atomicAdd(&g_qgroup[3], pp_ncand);
    atomicMax(&g_qgroup[4], pp_ncand);
    atomicAdd(&g_qgroup[5], pp_nrows);
    atomicAdd(&g_qgroup[6], pp_nlist);
    atomicAdd(&g_qgroup[7], pp_nwalk);
    PP_GT(5);
    for (int i = 0; i < 6; ++i) atomicAdd(&g_qgt[i], pp_gt[i]);
  }
#endif
  Found o;
  o.best = best;
  o.bidx = bidx;
  o.aux = 0.0f;
  return o;
}

// A cube of Chebyshev radius rho around the query's cell, a LANE per query (waves in which many lanes are open
// after stage A: the thin parts of a cloud, the sparse scale of a two-scale cloud): the cube's (2 rho + 1)^2 rows
// four at a time, each four as one sequence of groups like stage A, candidates from global memory in the exact
// (distance, index) order.  Returns whether the cube settles the lane's query; (best, bidx) carry on.
struct RowSpan {
  unsigned s, e;
};
__device__ __forceinline__ RowSpan cube_row(int r, int nrows, int side, int rho, int cy, int cz, int x0, int x1,
                                            bool active, const GridSet& g, const unsigned* __restrict__ cell_start) {
  const int z = cz - rho + r / side, y = cy - rho + r % side;
  const bool ok = active && r < nrows && z >= 0 && z < g.gz && y >= 0 && y < g.gy;
  const int c = pp::cell_linear(0, min(max(y, 0), g.gy - 1), min(max(z, 0), g.gz - 1), g.gx, g.gy);
  RowSpan o;
  o.s = ok ? cell_start[c + x0] : 0u;
  o.e = ok ? cell_start[c + x1 + 1] : 0u;
  return o;
}

template <bool LAB, int W>
__device__ __attribute__((noinline)) Found lane_cube_search(const GridSet g, const unsigned* __restrict__ cell_start,
                                                            const pp::f4* __restrict__ sorted,
                                                            const float* __restrict__ slab, float qx, float qy, float qz,
                                                            float ql, int rho, bool active, float best_in, int bidx_in) {
  const int cx = cell_coord(qx, g.minx, g.invh, g.gx);
  const int cy = cell_coord(qy, g.miny, g.invh, g.gy);
  const int cz = cell_coord(qz, g.minz, g.invh, g.gz);
  const float fx = (qx - g.minx) * g.invh - (float)cx, fy = (qy - g.miny) * g.invh - (float)cy,
              fz = (qz - g.minz) * g.invh - (float)cz;
  const int x0 = max(cx - rho, 0), x1 = min(cx + rho, g.gx - 1);
  const int side = 2 * rho + 1, nrows = side * side;
  float best = best_in;
  int bidx = bidx_in;
  bool gave_up = false;
  for (int r0 = 0; r0 < nrows; r0 += 4) {  // wave-uniform
    const RowSpan a0 = cube_row(r0, nrows, side, rho, cy, cz, x0, x1, active, g, cell_start);
    const RowSpan a1 = cube_row(r0 + 1, nrows, side, rho, cy, cz, x0, x1, active, g, cell_start);
    const RowSpan a2 = cube_row(r0 + 2, nrows, side, rho, cy, cz, x0, x1, active, g, cell_start);
    const RowSpan a3 = cube_row(r0 + 3, nrows, side, rho, cy, cz, x0, x1, active, g, cell_start);
    unsigned t0 = (a0.e - a0.s + 3) >> 2, t1 = (a1.e - a1.s + 3) >> 2, t2 = (a2.e - a2.s + 3) >> 2,
             t3 = (a3.e - a3.s + 3) >> 2;
    // a lane whose rows run through a crowded cell would keep the whole wave waiting on its loads, one lane's
    // worth at a time: it gives up and is served by the whole wave afterwards
    if (t0 + t1 + t2 + t3 > (unsigned)kLaneCubeMaxGroups) {
      gave_up = true;
      t0 = t1 = t2 = t3 = 0u;
    }
    const unsigned T1 = t0, T2 = T1 + t1, T3 = T2 + t2, T4 = T3 + t3;
    const unsigned adj0 = a0.s, adj1 = a1.s - 4 * T1, adj2 = a2.s - 4 * T2, adj3 = a3.s - 4 * T3;
    const unsigned last0 = a0.e - 1, last1 = a1.e - 1, last2 = a2.e - 1, last3 = a3.e - 1;
    pp::f4 pa[4], pb[4];
    float la[4] = {0.0f, 0.0f, 0.0f, 0.0f}, lb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    stage_a_fetch<LAB>(0, T1, T2, T3, T4, adj0, adj1, adj2, adj3, last0, last1, last2, last3, sorted, slab, pa, la);
    for (unsigned k = 0; __any(k < T4); k += 2) {
      stage_a_fetch<LAB>(k + 1, T1, T2, T3, T4, adj0, adj1, adj2, adj3, last0, last1, last2, last3, sorted, slab, pb, lb);
      if (k < T4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) take_candidate<LAB>(pa[u], la[u], qx, qy, qz, ql, best, bidx);
      }
      stage_a_fetch<LAB>(k + 2, T1, T2, T3, T4, adj0, adj1, adj2, adj3, last0, last1, last2, last3, sorted, slab, pa, la);
      if (k + 1 < T4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) take_candidate<LAB>(pb[u], lb[u], qx, qy, qz, ql, best, bidx);
      }
    }
  }
  auto axis = [&](float f, int c, int gdim) {
    const float lo = c - rho >= 1 ? (float)rho + f : __builtin_inff();
    const float hi = c + rho <= gdim - 2 ? (float)(rho + 1) - f : __builtin_inff();
    return fminf(lo, hi);
  };
  const float reach = g.h * fminf(axis(fx, cx, g.gx), fminf(axis(fy, cy, g.gy), axis(fz, cz, g.gz)));
  const bool all = cz - rho <= 0 && cz + rho >= g.gz - 1 && cy - rho <= 0 && cy + rho >= g.gy - 1 && cx - rho <= 0 &&
                   cx + rho >= g.gx - 1;
  const bool settled = all ? (LAB || bidx != 0x7fffffff) : (best < reach * reach * kBoundSlack);
  Found o;
  o.best = active ? best : best_in;
  o.bidx = active ? bidx : bidx_in;
  if (LAB && active && settled && !gave_up && bidx == 0x7fffffff) {  // whole grid examined, nobody carries this label
    o.best = 0.0f;                                         // (ref nmdistance_cuda.cu:110-113)
    o.bidx = -1;
  }
  o.aux = gave_up ? 2.0f : ((settled && active) ? 1.0f : 0.0f);  // (candidates seen before giving up stay valid)
  return o;
}

// Round 5 -- the BALL around a query that already holds a candidate (its block was not empty, only too small to settle
// it), a lane per query.  The cubes above know nothing of the best distance: radius 1 walks 27 cells, eight of them again,
// and where it does not settle -- the sparse part of a cloud of mixed dimension: the points inside an object beside its
// faces -- radius 2 walks 125, the 27 again.  A candidate at distance d makes everything beyond d irrelevant: only the
// rows (y, z) whose slab lies within d of the query can matter, and in such a row only the cells within what is left of
// d along x.  The rows of the box of cells around the ball, four at a time like the cubes; a row or a cell is passed
// over by the rule that settles a query everywhere else in this file -- (its distance)^2 * kBoundSlack > best, the
// distance to a rim cell measured as if the cell went on for ever outwards (it holds what was clamped into it) -- with
// the best distance as it stands at that moment.  What is left when the rows are through is EXACT: no second stage.
// A lane gives up (and is left to the stages after the cubes, with what it has found) if the box has more than
// kBallMaxRows rows or four rows hold more than kLaneCubeMaxGroups groups.  aux: 1 settled, 2 gave up, 0 not active.
constexpr int kBallMaxRows = 36;
template <bool LAB, int W>
__device__ __attribute__((noinline)) Found lane_ball_search(const GridSet g, const unsigned* __restrict__ cell_start,
                                                            const pp::f4* __restrict__ sorted,
                                                            const float* __restrict__ slab, float qx, float qy, float qz,
                                                            float ql, bool active, float best_in, int bidx_in) {
  const float px = (qx - g.minx) * g.invh, py = (qy - g.miny) * g.invh, pz = (qz - g.minz) * g.invh;  // in cells
  const int cy = cell_coord(qy, g.miny, g.invh, g.gy), cz = cell_coord(qz, g.minz, g.invh, g.gz);
  const float k2 = g.invh * g.invh / kBoundSlack;  // (distance in cells)^2 > best * k2: beyond the ball
  float best = best_in;
  int bidx = bidx_in;
  // the box of rows: what the ball of the FIRST candidate reaches (cell_coord clamps: the rim rows stand for everything
  // beyond them)
  const float R = fast_sqrt(best_in * k2) * 1.00001f;
  const bool box_ok = active && R <= kBallRmax;
  // (from the query's position in cells, as the rows' distances below: (int) saturates, the clamp stands for the rim)
  const int gy1 = g.gy - 1, gz1 = g.gz - 1;
  int y0 = (int)(py - R), y1 = (int)(py + R), z0 = (int)(pz - R), z1 = (int)(pz + R);
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(y0) : "v"(y0), "v"(gy1));
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(y1) : "v"(y1), "v"(gy1));
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(z0) : "v"(z0), "v"(gz1));
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(z1) : "v"(z1), "v"(gz1));
  y0 = min(y0, cy); y1 = max(y1, cy); z0 = min(z0, cz); z1 = max(z1, cz);
  const int ny = y1 - y0 + 1;
  int nrows = ny * (z1 - z0 + 1);
  bool gave_up = active && (!box_ok || nrows > kBallMaxRows);
  if (!active || gave_up) nrows = 0;
  const float inv_ny = 1.0f / (float)ny;
  const int gx1 = g.gx - 1;
  // a row's span of the sorted cloud: empty when the row, or every cell of it, is beyond the ball as it stands
  auto row_span = [&](int r) {
    RowSpan o;
    o.s = 0u;
    o.e = 0u;
    const int zi = (int)(((float)r + 0.5f) * inv_ny);
    const int z = z0 + zi, y = y0 + (r - zi * ny);
    const float dy = y < cy ? py - (float)(y + 1) : (y > cy ? (float)y - py : 0.0f);
    const float dz = z < cz ? pz - (float)(z + 1) : (z > cz ? (float)z - pz : 0.0f);
    const float w2 = best * k2 - (dy * dy + dz * dz);
    if (r < nrows && w2 >= 0.0f) {
      const float w = fast_sqrt(w2) * 1.00001f;
      int x0 = (int)(px - w), x1 = (int)(px + w);  // (saturating conversions; NaN -> 0)
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(x0) : "v"(x0), "v"(gx1));
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(x1) : "v"(x1), "v"(gx1));
      const int c = pp::cell_linear(0, y, z, g.gx, g.gy);
      o.s = cell_start[c + x0];
      o.e = cell_start[c + x1 + 1];
    }
    return o;
  };
  const int rmax = (int)pp::wave_reduce_dpp<false>((float)nrows);
  for (int r0 = 0; r0 < rmax; r0 += 4) {  // wave-uniform
    const RowSpan a0 = row_span(r0), a1 = row_span(r0 + 1), a2 = row_span(r0 + 2), a3 = row_span(r0 + 3);
    unsigned t0 = (a0.e - a0.s + 3) >> 2, t1 = (a1.e - a1.s + 3) >> 2, t2 = (a2.e - a2.s + 3) >> 2,
             t3 = (a3.e - a3.s + 3) >> 2;
    if (t0 + t1 + t2 + t3 > (unsigned)kLaneCubeMaxGroups) {  // rows through a crowded region: the whole wave's work
      gave_up = true;
      nrows = 0;
      t0 = t1 = t2 = t3 = 0u;
    }
    const unsigned T1 = t0, T2 = T1 + t1, T3 = T2 + t2, T4 = T3 + t3;
    const unsigned adj0 = a0.s, adj1 = a1.s - 4 * T1, adj2 = a2.s - 4 * T2, adj3 = a3.s - 4 * T3;
    const unsigned last0 = a0.e - 1, last1 = a1.e - 1, last2 = a2.e - 1, last3 = a3.e - 1;
    pp::f4 pa[4], pb[4];
    float la[4] = {0.0f, 0.0f, 0.0f, 0.0f}, lb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    stage_a_fetch<LAB>(0, T1, T2, T3, T4, adj0, adj1, adj2, adj3, last0, last1, last2, last3, sorted, slab, pa, la);
    for (unsigned k = 0; __any(k < T4); k += 2) {
      stage_a_fetch<LAB>(k + 1, T1, T2, T3, T4, adj0, adj1, adj2, adj3, last0, last1, last2, last3, sorted, slab, pb, lb);
      if (k < T4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) take_candidate<LAB>(pa[u], la[u], qx, qy, qz, ql, best, bidx);
      }
      stage_a_fetch<LAB>(k + 2, T1, T2, T3, T4, adj0, adj1, adj2, adj3, last0, last1, last2, last3, sorted, slab, pa, la);
      if (k + 1 < T4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) take_candidate<LAB>(pb[u], lb[u], qx, qy, qz, ql, best, bidx);
      }
    }
  }
  Found o;
  o.best = active ? best : best_in;
  o.bidx = active ? bidx : bidx_in;
  o.aux = !active ? 0.0f : (gave_up ? 2.0f : 1.0f);
  return o;
}

// Round 5 -- the same balls, POOLED over the wave.  A lane per query wastes the wave where few lanes
// are open (a tenth of a wave's queries in the sparse part of a cloud of mixed dimension) and every step waits for the
// lane with the longest rows; here the open queries' work is laid out flat, twice:
//   * the (query, row) pairs of every open query's box, 64 at a time: a lane works out ONE row's span inside the ball
//     (two table entries: the only dependent trip besides the points themselves) and appends it, if not empty, to a list
//     of pieces in LDS with the number of the piece's first group of four points;
//   * the (piece, group) pairs, 64 at a time: a lane fetches ONE group of four points of some piece, measures them
//     against that piece's query and lowers the query's 64-bit (distance bits, index) key in LDS with ds_min_u64 -- the
//     exact (distance, index) order, whoever comes first.
// Which piece a group belongs to, and which query a row belongs to, is found without a search: the pieces (queries) that
// START inside the current window of 64 groups (rows) mark their first position in a 64-entry window of LDS; a ballot
// of the marks and two popcounts give every lane its piece.  Every step is full width whatever the number of open
// lanes, and the chain is two trips to memory for the whole wave instead of two per four rows of the slowest lane.
// The ball is that of the candidate the query came with (it does not shrink on the way).  A row of more than
// kPoolMaxGroups groups (a crowded region) makes its query give up (aux 2), as in the lane form.
// LDS (the wave's slice, free after stage A): see the offsets below; <= 6208 bytes (CAPW = 384).
constexpr int kPoolItems = 256;       // pieces per flush
constexpr int kPoolMaxGroups = 48;    // groups of a row before its query gives up
typedef unsigned __attribute__((address_space(3))) * lds_u_wptr;
typedef unsigned long long __attribute__((address_space(3))) * lds_u64_wptr;
template <bool LAB, int W>
__device__ __attribute__((noinline)) Found wave_pooled_ball_search(const GridSet g, const unsigned* __restrict__ cell_start,
                                                                   const pp::f4* __restrict__ sorted,
                                                                   const float* __restrict__ slab, float qx, float qy,
                                                                   float qz, float ql, bool active, float best_in,
                                                                   int bidx_in, lds_f4_wptr slice, int rho) {
  // rho > 0 (wave-uniform): not the ball but the CUBE of Chebyshev radius rho around the query's cell, for the lanes that
  // hold no candidate yet (lane_cube_search's job, pooled): every row of the cube, the cells cx - rho .. cx + rho of each;
  // settled (aux 1) if the best found lies below what the cube guarantees, else aux 0 with what was found.
  const int lane = threadIdx.x & 63;
  // ---- the slice, in bytes: queries (x, y, z, best * k2) | keys | boxes | first row of a query | pieces: start, first
  // group, (points << 8 | query) | the window | the mask of the queries that gave up
  const lds_f4_wptr s_q = slice;                                   // 64 x 16
  const lds_u64_wptr s_key = (lds_u64_wptr)(slice + 64);           // 64 x 8
  const lds_u_wptr s_box = (lds_u_wptr)(slice + 96);               // 64 x 4
  const lds_u_wptr s_rowp = s_box + 64;                            // 65 x 4 (+ pad to 68)
  const lds_u_wptr s_start = s_rowp + 68;                          // kPoolItems x 4
  const lds_u_wptr s_first = s_start + kPoolItems;                 // (kPoolItems + 64) x 4: read up to 64 past the list
  const lds_u_wptr s_meta = s_first + kPoolItems + 64;             // kPoolItems x 4
  const lds_u_wptr s_win = s_meta + kPoolItems;                    // 64 x 4
  const lds_u_wptr s_gave = s_win + 64;                            // 2 x 4 (+ 2 pad)
  const lds_f_wptr s_ql = (lds_f_wptr)(s_gave + 4);                // 64 x 4: the queries' labels (labeled searches)
  static_assert(96 * 16 + (64 + 68 + kPoolItems * 3 + 64 + 64 + 4 + 64) * 4 <= (384 + 4) * 16, "the wave's slice");
  auto lds_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  const float k2 = g.invh * g.invh / kBoundSlack;
  const float py = (qy - g.miny) * g.invh, pz = (qz - g.minz) * g.invh;
  const int cy = cell_coord(qy, g.miny, g.invh, g.gy), cz = cell_coord(qz, g.minz, g.invh, g.gz);
  const int gx1 = g.gx - 1, gy1 = g.gy - 1, gz1 = g.gz - 1;
  // ---- the open queries, compacted: query k = the k-th active lane
  const unsigned long long am = __ballot(active);
  const int nq = __builtin_popcountll(am);
  const int k = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
  const float R = fast_sqrt(best_in * k2) * 1.00001f;
  int y0 = (int)(py - R), y1 = (int)(py + R), z0 = (int)(pz - R), z1 = (int)(pz + R);
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(y0) : "v"(y0), "v"(gy1));
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(y1) : "v"(y1), "v"(gy1));
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(z0) : "v"(z0), "v"(gz1));
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(z1) : "v"(z1), "v"(gz1));
  y0 = min(y0, cy); y1 = max(y1, cy); z0 = min(z0, cz); z1 = max(z1, cz);
  if (rho > 0) {
    y0 = max(cy - rho, 0); y1 = min(cy + rho, gy1); z0 = max(cz - rho, 0); z1 = min(cz + rho, gz1);
  }
  const int ny = y1 - y0 + 1;
  const unsigned nrows = active ? (unsigned)(ny * (z1 - z0 + 1)) : 0u;  // (<= 36: R <= kBallRmax, rho <= 2)
  const unsigned rincl = pp::wave_scan_u32_dpp(nrows);
  const unsigned rtotal = (unsigned)__builtin_amdgcn_readlane((int)rincl, 63);
  if (active) {
    s_q[k] = pp::f4{qx, qy, qz, rho > 0 ? __builtin_inff() : best_in * k2};  // (a cube: every row, nothing is cut by a distance)
    s_key[k] = ((unsigned long long)__float_as_uint(best_in) << 32) | (unsigned)bidx_in;
    if (LAB) s_ql[k] = ql;
    s_box[k] = (unsigned)y0 | ((unsigned)z0 << 8) | ((unsigned)ny << 16);
    s_rowp[k] = rincl - nrows;
  }
  if (lane == 0) {
    s_rowp[nq] = rtotal;
    s_gave[0] = 0u;
    s_gave[1] = 0u;
  }
  lds_sync();
  // One window step of the "which one am I in" search: `first` is an ascending list of n first positions (rows of a
  // query, groups of a piece; every entry owns at least one position), `cur` of them start before position c0; returns
  // the index of the entry that holds position c0 + lane and moves cur past the entries that start in this window.
  auto locate = [&](const lds_u_wptr first, int n, unsigned c0, int& cur) {
    s_win[lane] = 0u;
    const int cand = cur + lane;
    const unsigned fpos = first[min(cand, n)];  // (entry n: the total -- beyond every window that is looked at)
    lds_sync();
    if (cand < n && fpos - c0 < 64u) s_win[fpos - c0] = 1u;
    lds_sync();
    const unsigned long long m = __ballot(s_win[lane] != 0u);
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const int idx = cur + __builtin_popcountll(m & le) - 1;
    cur += __builtin_popcountll(m);
    lds_sync();
    return idx;
  };
  int nitems = 0;       // pieces in the list
  unsigned ctotal = 0;  // groups in the list
  // ---- the groups of the listed pieces, 64 at a time
  auto flush = [&]() {
    if (lane == 0) s_first[nitems] = ctotal;
    lds_sync();
    int cur = 0;
    for (unsigned c0 = 0; c0 < ctotal; c0 += 64) {  // wave-uniform
      const int it = max(locate(s_first, nitems, c0, cur), 0);
      const unsigned c = min(c0 + (unsigned)lane, ctotal - 1);  // (a lane past the end repeats the last group: harmless)
      const int itc = c0 + (unsigned)lane < ctotal ? it : nitems - 1;
      const unsigned st = s_start[itc], fs = s_first[itc], meta = s_meta[itc];
      const unsigned npts = meta >> 8, kq = meta & 63u;
      const pp::f4 q = s_q[kq];
      const unsigned o4 = (c - fs) << 2;
      pp::f4 p[4];
      float pl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned at = st + min(o4 + (unsigned)u, npts - 1);
        p[u] = sorted[at];
        if (LAB) pl[u] = slab[at];
      }
      const float wl = LAB ? s_ql[kq] : 0.0f;
      unsigned long long key = ~0ull;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = pp::chamfer_d3(p[u].x, p[u].y, p[u].z, q.x, q.y, q.z);
        const unsigned long long cand = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p[u].w);
        // (a NaN distance -- bits above +inf -- never beats a real candidate; labeled: a candidate of the query's label only)
        key = ((!LAB || pl[u] == wl) && cand < key) ? cand : key;
      }
      __hip_atomic_fetch_min(s_key + kq, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    lds_sync();
    nitems = 0;
    ctotal = 0;
  };
  // ---- the rows of the open queries' boxes, 64 at a time
  int curq = 0;
  for (unsigned j0 = 0; j0 < rtotal; j0 += 64) {  // wave-uniform
    const int kq0 = locate(s_rowp, nq, j0, curq);
    const bool live = j0 + (unsigned)lane < rtotal;
    const int kq = live ? max(kq0, 0) : 0;
    const pp::f4 q = s_q[kq];
    const unsigned box = s_box[kq];
    const int r = (int)(j0 + (unsigned)lane - s_rowp[kq]);
    const int by0 = (int)(box & 255u), bz0 = (int)((box >> 8) & 255u), bny = (int)(box >> 16);
    const int zi = (int)(((float)r + 0.5f) * (1.0f / (float)bny));
    const int z = bz0 + zi, y = by0 + (r - zi * bny);
    const float wpx = (q.x - g.minx) * g.invh, wpy = (q.y - g.miny) * g.invh, wpz = (q.z - g.minz) * g.invh;
    const int wcy = cell_coord(q.y, g.miny, g.invh, g.gy), wcz = cell_coord(q.z, g.minz, g.invh, g.gz);
    const float dy = y < wcy ? wpy - (float)(y + 1) : (y > wcy ? (float)y - wpy : 0.0f);
    const float dz = z < wcz ? wpz - (float)(z + 1) : (z > wcz ? (float)z - wpz : 0.0f);
    const float w2 = q.w - (dy * dy + dz * dz);
    unsigned rs = 0u, re = 0u;
    if (live && w2 >= 0.0f) {
      const float w = fast_sqrt(w2) * 1.00001f;
      int x0 = (int)(wpx - w), x1 = (int)(wpx + w);
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(x0) : "v"(x0), "v"(gx1));
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(x1) : "v"(x1), "v"(gx1));
      if (rho > 0) {
        const int wcx = cell_coord(q.x, g.minx, g.invh, g.gx);
        x0 = max(wcx - rho, 0);
        x1 = min(wcx + rho, gx1);
      }
      const int c = pp::cell_linear(0, y, z, g.gx, g.gy);
      rs = cell_start[c + x0];
      re = cell_start[c + x1 + 1];
    }
    unsigned t = (re - rs + 3u) >> 2;
    if (t > (unsigned)kPoolMaxGroups) {  // a crowded region: the query is the whole wave's (what it has found stays valid)
      atomicOr((unsigned*)(s_gave + (kq >> 5)), 1u << (kq & 31));
      t = 0u;
    }
    const unsigned long long nm = __ballot(t > 0u);
    const int slot = nitems + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(nm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nm, 0u));
    const unsigned tincl = pp::wave_scan_u32_dpp(t);
    if (t > 0u) {
      s_start[slot] = rs;
      s_first[slot] = ctotal + tincl - t;
      s_meta[slot] = ((re - rs) << 8) | (unsigned)kq;
    }
    nitems += __builtin_popcountll(nm);
    ctotal += (unsigned)__builtin_amdgcn_readlane((int)tincl, 63);
    if (nitems > kPoolItems - 64) flush();  // (the next 64 rows would not fit for sure)
  }
  if (nitems > 0) flush();
  lds_sync();
  const unsigned long long key = active ? s_key[k] : 0ull;
  const bool gave = active && ((s_gave[k >> 5] >> (k & 31)) & 1u) != 0u;
  lds_sync();
  Found o;
  o.best = active ? __uint_as_float((unsigned)(key >> 32)) : best_in;
  o.bidx = active ? (int)(unsigned)key : bidx_in;
  o.aux = !active ? 0.0f : (gave ? 2.0f : 1.0f);
  if (rho > 0) {  // (uniform) what the cube guarantees: lane_cube_search's rule
    const int cx = cell_coord(qx, g.minx, g.invh, g.gx);
    const float fx = (qx - g.minx) * g.invh - (float)cx, fy = py - (float)cy, fz = pz - (float)cz;
    auto axis = [&](float f, int c, int gdim) {
      const float lo = c - rho >= 1 ? (float)rho + f : __builtin_inff();
      const float hi = c + rho <= gdim - 2 ? (float)(rho + 1) - f : __builtin_inff();
      return fminf(lo, hi);
    };
    const float reach = g.h * fminf(axis(fx, cx, g.gx), fminf(axis(fy, cy, g.gy), axis(fz, cz, g.gz)));
    const bool all = cz - rho <= 0 && cz + rho >= gz1 && cy - rho <= 0 && cy + rho >= gy1 && cx - rho <= 0 && cx + rho >= gx1;
    const bool settled = all ? (LAB || o.bidx != 0x7fffffff) : (o.best < reach * reach * kBoundSlack);
    if (LAB && active && settled && !gave && o.bidx == 0x7fffffff) {  // whole grid examined, nobody carries this label
      o.best = 0.0f;                                                  // (ref nmdistance_cuda.cu:110-113)
      o.bidx = -1;
    }
    o.aux = !active ? 0.0f : (gave ? 2.0f : (settled ? 1.0f : 0.0f));
  }
  return o;
}

// first staged position of group k of a lane's sequence (by value: see stage_a_fetch)
__device__ __forceinline__ unsigned stage_first(unsigned k, unsigned T1, unsigned T2, unsigned T3, unsigned T4,
                                                unsigned adj0, unsigned adj1, unsigned adj2, unsigned adj3) {
  const unsigned adj = k < T1 ? adj0 : (k < T2 ? adj1 : (k < T3 ? adj2 : adj3));
  return k < T4 ? adj + 4 * k : 0u;
}

// v_min_f32 without the canonicalising v_max the compiler puts in front of fminf (operands are results of
// fma chains; a NaN operand is dropped, as v_min3 does)
__device__ __forceinline__ float min2(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// The search of (up to) 64 queries by one wave, one lane per query: stage A, then whatever stage A leaves (see the
// top of the file).  `jj` = the lane's query as a position in the sorted query cloud (in the cloud itself when that
// cloud has no grid), `valid` = the lane has a query; s_pts_w / s_lab_w = the wave's private slice of LDS (CAPW + 4
// points / labels).  skip_a (wave-uniform): stage A has been run for every one of these queries already (the stage-A
// kernel) and settled none of them -- go straight to the stages after it.
// (phase stamps of search_queries: off in a -DPP_PROBE_STAGE_A_ONLY probe build, which clocks the stage-A kernel alone)
#if defined(PP_QUERY_PROBE) && !defined(PP_PROBE_STAGE_A_ONLY)
#define PP_SPHASE(n) PP_QPHASE(n)
#else
#define PP_SPHASE(n)
#endif
// W: the waves per SIMD the calling kernel is compiled for -- it only makes the out-of-line stages below separate
// functions per kernel, each compiled for its caller's register budget (the list kernel runs at 5: 96 registers, no
// spills in the group search; gaussian -8 %, blobs8 -6 %, disjoint -6 % against 6 waves and 80 registers)
template <bool LAB, int CAPW, int W>
__device__ __forceinline__ void search_queries(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                               float* __restrict__ dist1, int* __restrict__ idx1,
                                               float* __restrict__ dist2, int* __restrict__ idx2,
                                               unsigned char* __restrict__ ws, int B, int N, int M,
                                               const float* __restrict__ label1, const float* __restrict__ label2,
                                               const Layout& L, const int b, const int dir, const int jj,
                                               const bool valid, const bool skip_a, lds_f4_wptr s_pts_w,
                                               lds_f_wptr s_lab_w, const unsigned* __restrict__ rowbits = nullptr) {
  PP_QPHASE_DECL;
  PP_SPHASE(0);
  const int lane = threadIdx.x & 63;
  const int set = 2 * b + dir;
  const int nq = dir ? M : N, nr = dir ? N : M;
  const GridSet g = reinterpret_cast<const GridSet*>(ws + L.sets)[set];
  const GridSet gp = reinterpret_cast<const GridSet*>(ws + L.sets)[set ^ 1];
  const pp::f4* __restrict__ qsorted =
      reinterpret_cast<const pp::f4*>(ws + L.sorted) + set_point_offset(b, dir ^ 1, N, M);
  const bool g_useless = pp::grid_useless(g), gp_useless = pp::grid_useless(gp);
  float* __restrict__ od = (dir ? dist2 : dist1) + (size_t)b * nq;
  int* __restrict__ oi = (dir ? idx2 : idx1) + (size_t)b * nq;
  float qx, qy, qz, ql = 0.0f;
  int j;
  if (!gp_useless) {
    const pp::f4 qq = qsorted[jj];
    qx = qq.x; qy = qq.y; qz = qq.z;
    j = __float_as_int(qq.w);
    if (LAB) ql = (reinterpret_cast<const float*>(ws + L.slab) + set_point_offset(b, dir ^ 1, N, M))[jj];
  } else {
    const float* __restrict__ q = (dir ? xyz2 : xyz1) + ((size_t)b * nq + jj) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
    j = jj;
    if (LAB) ql = (dir ? label2 : label1)[(size_t)b * nq + jj];
  }
  if (g_useless) {  // no grid for this set (non-finite or zero-extent data, crowded cells): every pair
    float best;
    int bidx;
    lane_scan_cloud<LAB>((dir ? xyz1 : xyz2) + (size_t)b * nr * 3, LAB ? (dir ? label1 : label2) + (size_t)b * nr : nullptr,
                         nr, qx, qy, qz, ql, best, bidx);
    if (valid) {
      od[j] = best;
      oi[j] = bidx;
    }
    return;
  }
  const unsigned* __restrict__ cell_start =
      reinterpret_cast<const unsigned*>(ws + L.cell_start) + (size_t)set * kCellStride;
  const pp::f4* __restrict__ sorted =
      reinterpret_cast<const pp::f4*>(ws + L.sorted) + set_point_offset(b, dir, N, M);
  const float* __restrict__ slab =
      LAB ? reinterpret_cast<const float*>(ws + L.slab) + set_point_offset(b, dir, N, M) : nullptr;
  // second level (sets with cells of more than kCrowd points: dense clusters, several scales): see below
  const bool refined_set = pp::grid_refined(g);
  const unsigned* __restrict__ sub_start =
      reinterpret_cast<const unsigned*>(ws + L.sub_start) + set_sub_start_offset(b, dir, N, M);
  const pp::SubGrid* __restrict__ sub_desc =
      reinterpret_cast<const pp::SubGrid*>(ws + L.sub_desc) + set_sub_desc_offset(b, dir, N, M);

  // (+4: a group of four is read from any staged position without clamping; the tail repeats a real point)

  bool deferred = false;
  float best = __builtin_inff();
  int bidx = 0x7fffffff;
  float thr = 0.0f;  // (nothing settles below it: skip_a leaves every query to the stages after stage A)
  // Round 6: a query in the void -- by the set's row bitmap (list kernel, sets with crowded cells; words of ones where it
  // was not made) the nine cell rows (y +- 1, z +- 1) around its cell hold nothing, so neither does its block nor its cube
  // of radius 1: straight to the group search, which is where the cubes sent it after finding them empty, two dependent
  // round trips per four rows later (blobs8: 10 us of a wave's 64 in the cubes, 7 in a stage A over empty rows).  A wave
  // all of whose queries are such skips stage A.  A choice of route only: the group search is complete.
  bool void_around = false;
  if (rowbits != nullptr && pp::grid_refined(g)) {  // (wave-uniform; every lane takes part in the lookups: ds_bpermute)
    const unsigned rbw = rowbits[lane & 31];
    const int vy = cell_coord(qy, g.miny, g.invh, g.gy), vz = cell_coord(qz, g.minz, g.invh, g.gz);
    unsigned any = 0u;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const int y = vy + e % 3 - 1, z = vz + e / 3 - 1;
      const bool in = y >= 0 && y < g.gy && z >= 0 && z < g.gz;
      const int row = in ? y + g.gy * z : 0;
      // (no short-circuit: a lane that sat a lookup out would hand its word out as zero)
      const unsigned w = (unsigned)__shfl((int)rbw, row >> 5);
      any |= in ? (w >> (row & 31)) & 1u : 0u;
    }
    void_around = any == 0u;
  }
  if (!skip_a && !__all(void_around || !valid)) {  // (wave-uniform)
  // Stage A: the 2x2x2 block of cells nearest to q' (own cell + the neighbour on the side of the cell q' lies
  // in, per axis).  A point outside that block is beyond the far face of q''s cell along some axis (>= h/2
  // away) or beyond the neighbour (>= h away).  The block is four rows (y, z) of one or two cells (x0..x1).
  const int cx = cell_coord(qx, g.minx, g.invh, g.gx);
  const int cy = cell_coord(qy, g.miny, g.invh, g.gy);
  const int cz = cell_coord(qz, g.minz, g.invh, g.gz);
  const float fx = (qx - g.minx) * g.invh - (float)cx, fy = (qy - g.miny) * g.invh - (float)cy,
              fz = (qz - g.minz) * g.invh - (float)cz;  // position inside the cell, in cells
  const int sx = fx < 0.5f ? -1 : 1, sy = fy < 0.5f ? -1 : 1, sz = fz < 0.5f ? -1 : 1;
  const int x0 = max(min(cx, cx + sx), 0), x1 = min(max(cx, cx + sx), g.gx - 1);
  const int y0 = max(min(cy, cy + sy), 0), y1 = min(max(cy, cy + sy), g.gy - 1);
  const int z0 = max(min(cz, cz + sz), 0), z1 = min(max(cz, cz + sz), g.gz - 1);
  unsigned rs0, rs1, rs2, rs3, re0, re1, re2, re3;
  bool crowd = false;  // some cell of the block holds more than kCrowd points (it then has a grid of its own)
  // The bounds of the block's four rows (y, z): a row is one or two cells wide, its two bounds are at most two
  // entries apart, so ONE 12-byte load fetches both (the entry after a set's table is the next set's or the
  // sorted cloud: valid memory).  The loads are only ISSUED here; their values are first touched after the wave's
  // region has been worked out (below), whose own loads then travel at the same time instead of one round trip later.
  typedef unsigned u3 __attribute__((ext_vector_type(3)));
  u3 rv0, rv1, rv2, rv3;
  {
    auto row_issue = [&](int a, int bq, u3& v) {
      const int z = cz + a * sz, y = cy + bq * sy;
      const int c = pp::cell_linear(0, min(max(y, 0), g.gy - 1), min(max(z, 0), g.gz - 1), g.gx, g.gy);
      __builtin_memcpy(&v, cell_start + c + x0, sizeof(v));
    };
    row_issue(0, 0, rv0);
    row_issue(0, 1, rv1);
    row_issue(1, 0, rv2);
    row_issue(1, 1, rv3);
  }
  // (named scalars, not arrays: hipcc turns a select between array elements into an indexed load from scratch)
  auto row_finish = [&](int a, int bq, const u3 v, unsigned& s_out, unsigned& e_out) {
    const int z = cz + a * sz, y = cy + bq * sy;
    const bool ok = z >= 0 && z < g.gz && y >= 0 && y < g.gy;
    s_out = ok ? v.x : 0u;
    e_out = ok ? (x1 > x0 ? v.z : v.y) : 0u;
    crowd = crowd | (ok & ((v.y - v.x > (unsigned)pp::kCrowd) | ((x1 > x0) & (v.z - v.y > (unsigned)pp::kCrowd))));
  };
  auto rows_finish = [&]() {
    row_finish(0, 0, rv0, rs0, re0);
    row_finish(0, 1, rv1, rs1, re1);
    row_finish(1, 0, rv2, rs2, re2);
    row_finish(1, 1, rv3, rs3, re3);
  };
  // What the block guarantees for THIS query: along each axis the nearer face of the block that has grid
  // beyond it (beyond the grid there are no points).  Lower face: 1 + f cells away when the block includes
  // cell c-1, f when it starts at c; upper face: 1 - f or 2 - f.  Never below h/2.  The query is settled if
  // its best distance is below reach^2 * kBoundSlack (strict).
  // (computed after the walk, from the query again: the cell coordinates need not live through it)
  // Lanes whose block touches a crowded cell do not take part in the staged walk: their candidates are the
  // sub-cells near the query, found through the crowded cells' own grids further down.  (refined_set is
  // wave-uniform and false for every set of an evenly sampled surface: config 2 pays one scalar branch.)
  PP_SPHASE(1);
  const float ninf = -__builtin_inff();
  auto lane63 = [](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); };
  int Lz, nz;
  bool staged;
  unsigned n_staged = 0, delta = 0, offv = 0;  // lanes 0..7: the layer's (global start - LDS start), LDS start
  // (a function of its two flags, instantiated twice: for sets without crowded cells -- every evenly sampled
  // surface -- both are constants and the region does not depend on the row bounds at all)
  auto region = [&](const bool dfr, const bool nonorm) {
    // ---- the wave's region: its z-layers and the row range in each (cell coordinates are < 2^24: exact as floats)
    {
      float v[6] = {dfr ? ninf : -(float)z0, dfr ? ninf : (float)z1, ninf, ninf, ninf, ninf};
      pp::wave_reduce6_dpp<false, 6>(v);
      Lz = nonorm ? 0 : -(int)lane63(v[0]);
      nz = nonorm ? 1 : (int)lane63(v[1]) - Lz + 1;
    }
    staged = nz <= kStageLayers && !nonorm;
    if (staged) {
      int ya = 1, yb = 0;  // lane l < nz: row range of layer Lz + l
      for (int base = 0; base < nz; base += 3) {  // wave-uniform
        float v[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int z = Lz + base + i;
          const bool m = (z0 == z || z1 == z) && !dfr;
          v[2 * i] = m ? -(float)y0 : ninf;
          v[2 * i + 1] = m ? (float)y1 : ninf;
        }
        pp::wave_reduce6_dpp<false, 6>(v);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float a = lane63(v[2 * i]), bmax = lane63(v[2 * i + 1]);
          if (lane == base + i && bmax >= 0.0f) {
            ya = -(int)a;
            yb = (int)bmax;
          }
        }
      }
      unsigned gs = 0, ge = 0;
      if (lane < nz && ya <= yb) {
        gs = cell_start[pp::cell_linear(0, ya, Lz + lane, g.gx, g.gy)];
        ge = cell_start[pp::cell_linear(0, yb, Lz + lane, g.gx, g.gy) + g.gx];
      }
      const unsigned len = ge - gs;
      const unsigned incl = pp::wave_scan_u32_dpp(len);  // (lanes from nz on hold nothing)
      offv = incl - len;
      delta = gs - offv;
      n_staged = (unsigned)__builtin_amdgcn_readlane((int)incl, kStageLayers - 1);
      staged = n_staged <= (unsigned)CAPW;
    }
  };
  deferred = false;
  if (!refined_set) {  // wave-uniform
    region(false, false);
    rows_finish();
  } else {
    rows_finish();
    deferred = crowd;
    if (deferred) {
      re0 = rs0; re1 = rs1; re2 = rs2; re3 = rs3;
    }
    region(deferred, !__any(!deferred));
  }
  PP_SPHASE(2);
  const lds_f4_ptr lpts = (lds_f4_ptr)s_pts_w;
  const lds_f_ptr llab = (lds_f_ptr)s_lab_w;
  if (staged) {
    // every lane's rows live in its two layers: global position -> LDS position
    const unsigned dA = __shfl(delta, deferred ? 0 : cz - Lz);
    const int zb = cz + sz;
    const unsigned dB = __shfl(delta, (!deferred && zb >= 0 && zb < g.gz) ? zb - Lz : 0);
    rs0 -= dA; re0 -= dA; rs1 -= dA; re1 -= dA;
    rs2 -= dB; re2 -= dB; rs3 -= dB; re3 -= dB;
    // copy layer by layer: everything but the lane offset is wave-uniform (a span's last piece may be partial)
    for (int l = 0; l < nz; ++l) {
      const unsigned o0 = (unsigned)__builtin_amdgcn_readlane((int)offv, l);
      const unsigned dlt = (unsigned)__builtin_amdgcn_readlane((int)delta, l);
      const unsigned end = l + 1 < nz ? (unsigned)__builtin_amdgcn_readlane((int)offv, l + 1) : n_staged;
      for (unsigned p0 = o0; p0 < end; p0 += 4 * 64) {
        pp::f4 v[4];
        float vl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned p = min(p0 + (unsigned)(u * 64 + lane), end - 1);  // (duplicates store the same value)
          v[u] = sorted[p + dlt];
          if (LAB) vl[u] = slab[p + dlt];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned p = min(p0 + (unsigned)(u * 64 + lane), end - 1);
          s_pts_w[p] = v[u];
          if (LAB) s_lab_w[p] = vl[u];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (n_staged > 0) {  // the padding repeats the last point (a real candidate)
      const pp::f4 padv = lpts[n_staged - 1];
      const float padl = LAB ? llab[n_staged - 1] : 0.0f;
      if (lane < 4) {
        s_pts_w[n_staged + lane] = padv;
        if (LAB) s_lab_w[n_staged + lane] = padl;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  PP_SPHASE(3);

  // ---- stage A: the four rows as ONE sequence of groups of four points -----------------------------------
  // A lane's rows hold t_r = ceil(len_r / 4) groups each; group k of the sequence belongs to the row r with
  // T_r <= k < T_{r+1} (T = running sums).  k is wave-uniform, so the wave runs max over lanes of the TOTAL
  // group count instead of the sum over rows of the per-row maxima, and the loads of group k + 1 are in
  // flight while group k is evaluated.
  best = __builtin_inff();
  bidx = 0x7fffffff;
  {
    const unsigned t0 = (re0 - rs0 + 3) >> 2, t1 = (re1 - rs1 + 3) >> 2, t2 = (re2 - rs2 + 3) >> 2,
                   t3 = (re3 - rs3 + 3) >> 2;
    const unsigned T1 = t0, T2 = T1 + t1, T3 = T2 + t2, T4 = T3 + t3;
    const unsigned adj0 = rs0, adj1 = rs1 - 4 * T1, adj2 = rs2 - 4 * T2, adj3 = rs3 - 4 * T3;
    const unsigned last0 = re0 - 1, last1 = re1 - 1, last2 = re2 - 1, last3 = re3 - 1;
    auto examine = [&](const pp::f4 (&p)[4], const float (&pl)[4]) {  // exact (distance, index) order
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = pp::chamfer_d3(p[u].x, p[u].y, p[u].z, qx, qy, qz);
        const int id = __float_as_int(p[u].w);
        const bool take = (!LAB || pl[u] == ql) & ((d < best) | ((d == best) & (id < bidx)));
        best = take ? d : best;
        bidx = take ? id : bidx;
      }
    };
    pp::f4 pa[4], pb[4];
    float la[4] = {0.0f, 0.0f, 0.0f, 0.0f}, lb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (staged) {
      // Staged walk, trimmed for VALU issue (the kernel's roof once the candidates come from LDS).  A group is
      // four CONSECUTIVE staged points from its first position on -- one address, four ds_read_b128 with
      // immediate offsets; running past the end of a row only examines more real points of the set, which is
      // harmless, and the region is padded so that nothing is clamped.  Per group only the running minimum
      // is kept (v_min3 + v_min) and the number of the first group that lowered it; the winner's index is
      // recovered afterwards by re-examining that one group.  A distance EQUAL to the running minimum seen
      // in a later group (an exact tie across groups: duplicated points, lattices) cannot be ordered that
      // way: the wave then repeats the walk with the exact (distance, index) comparison.
      auto first_of = [&](unsigned k) { return stage_first(k, T1, T2, T3, T4, adj0, adj1, adj2, adj3); };
      auto fetch4 = [&](unsigned e0, pp::f4 (&p)[4], float (&pl)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          p[u] = lpts[e0 + u];
          if (LAB) pl[u] = llab[e0 + u];
        }
      };
      if (n_staged > 0) {
        unsigned gk = 0xffffffffu;
        bool tie = false;
        auto track = [&](unsigned k, const pp::f4 (&p)[4], const float (&pl)[4]) {
          float d[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            d[u] = pp::chamfer_d3(p[u].x, p[u].y, p[u].z, qx, qy, qz);
            if (LAB) d[u] = pl[u] == ql ? d[u] : __builtin_inff();
          }
          const float gmin = min2(pp::min3(d[0], d[1], d[2]), d[3]);
          const bool lt = gmin < best;
          tie = tie | ((gmin == best) & (k < T4));
          gk = lt ? k : gk;
          best = lt ? gmin : best;
        };
        fetch4(first_of(0), pa, la);
        for (unsigned k = 0; __any(k < T4); k += 2) {
          fetch4(first_of(k + 1), pb, lb);
          track(k, pa, la);
          fetch4(first_of(k + 2), pa, la);
          track(k + 1, pb, lb);
        }
        if (__any(tie)) {  // exact redo (rare)
          best = __builtin_inff();
          fetch4(first_of(0), pa, la);
          for (unsigned k = 0; __any(k < T4); k += 2) {
            fetch4(first_of(k + 1), pb, lb);
            examine(pa, la);
            fetch4(first_of(k + 2), pa, la);
            examine(pb, lb);
          }
        } else if (gk != 0xffffffffu) {  // the winner is in group gk: lowest index among its minima
          fetch4(first_of(gk), pa, la);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float d = pp::chamfer_d3(pa[u].x, pa[u].y, pa[u].z, qx, qy, qz);
            if (LAB) d = la[u] == ql ? d : __builtin_inff();
            const int id = __float_as_int(pa[u].w);
            const bool take = (d == best) & (id < bidx);
            bidx = take ? id : bidx;
          }
        }
      }
    } else {
      auto fetch = [&](unsigned k, pp::f4 (&p)[4], float (&pl)[4]) {
        stage_a_fetch<LAB>(k, T1, T2, T3, T4, adj0, adj1, adj2, adj3, last0, last1, last2, last3, sorted, slab, p, pl);
      };
      fetch(0, pa, la);
      for (unsigned k = 0; __any(k < T4); k += 2) {
        fetch(k + 1, pb, lb);
        examine(pa, la);
        fetch(k + 2, pa, la);
        examine(pb, lb);
      }
    }
  }
  PP_SPHASE(4);
  float reach;
  int cx2, cy2, cz2, sx2, sy2, sz2;
  {
    float ax = qx, ay = qy, az = qz;
    asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az));  // (opaque copies: nothing below is merged with the values above)
    cx2 = cell_coord(ax, g.minx, g.invh, g.gx);
    cy2 = cell_coord(ay, g.miny, g.invh, g.gy);
    cz2 = cell_coord(az, g.minz, g.invh, g.gz);
    const float fx2 = (ax - g.minx) * g.invh - (float)cx2, fy2 = (ay - g.miny) * g.invh - (float)cy2,
                fz2 = (az - g.minz) * g.invh - (float)cz2;
    sx2 = fx2 < 0.5f ? -1 : 1;
    sy2 = fy2 < 0.5f ? -1 : 1;
    sz2 = fz2 < 0.5f ? -1 : 1;
    reach = g.h * fminf(block_reach(fx2, sx2, cx2, g.gx), fminf(block_reach(fy2, sy2, cy2, g.gy), block_reach(fz2, sz2, cz2, g.gz)));
  }
  thr = reach * reach * kBoundSlack;
  if (refined_set && __any(deferred)) {  // wave-uniform
    // ---- second level: the block's cells one by one; a crowded cell through its own grid ---------------------
    if (deferred) {
      const Found f = refined_block_search<LAB, W>(g, cell_start, sorted, slab, sub_start, sub_desc, qx, qy, qz, ql, cx2, cy2,
                                                cz2, sx2, sy2, sz2, reach);
      best = f.best;
      bidx = f.bidx;
      thr = f.aux;
    }
  }
  }  // (!skip_a)
  PP_SPHASE(5);
  const bool resolved = best < thr;
  if (resolved && valid) {
    od[j] = best;
    oi[j] = bidx;
  }
  // ---- what stage A left.  Many lanes of the wave (thin regions, the sparse scale of a two-scale cloud): the
  // cubes of radius 1 and 2 a lane per query.  (Lanes next to a crowded cell stay with the whole-wave stages:
  // a cube around them holds thousands of points.)
  bool pend = !resolved && valid;
  bool open_lane = false;
  // A query more than three cells outside the grid's box has nothing to find in the cubes around the rim cell it is
  // clamped to (whatever they hold is too far to settle it): straight to the group search (round 3: between disjoint
  // clouds the cubes were 47 us of a wave's life).  A choice of route only: the group search is complete.
  const float out_x = fmaxf(g.minx - qx, qx - (g.minx + (float)g.gx * g.h)), out_y = fmaxf(g.miny - qy, qy - (g.miny + (float)g.gy * g.h)),
              out_z = fmaxf(g.minz - qz, qz - (g.minz + (float)g.gz * g.h));
  const bool far_out = fmaxf(out_x, fmaxf(out_y, out_z)) > 3.0f * g.h;
  if (pend && (far_out || void_around) && !deferred) {
    open_lane = true;
    pend = false;
  }
  // (round 5) lanes that hold a candidate: the ball around it, exact in one stage (wave_pooled_ball_search / lane_ball_search);
  // once for the candidates of the blocks, once more behind the cubes of radius 1 for the lanes those gave their first
  // candidate (an empty block in the sparse part of a cloud; labeled searches, where a block seldom holds the query's label)
  bool ball_left = false;  // the ball gave up (rows through a crowded region): not for the lane cubes either
  auto ball_stage = [&]() {
    // (a candidate further than kBallRmax cells: a box of rows larger than the cubes' -- those lanes stay with the cubes)
    const bool ball = pend && !deferred && !ball_left &&
                      best * (g.invh * g.invh) <= kBallRmax * kBallRmax * kBoundSlack * 0.9999f;
    constexpr bool kPooled = CAPW >= 384;  // (the pooled form's lists need the slice of CAPW = 384)
    if (__builtin_popcountll(__ballot(ball)) >= (kPooled ? kPoolMin : kBallMin)) {
      Found f;
      if constexpr (kPooled)
        f = wave_pooled_ball_search<LAB, W>(g, cell_start, sorted, slab, qx, qy, qz, ql, ball, best, bidx, s_pts_w, 0);
      else
        f = lane_ball_search<LAB, W>(g, cell_start, sorted, slab, qx, qy, qz, ql, ball, best, bidx);
      best = f.best;
      bidx = f.bidx;
      if (f.aux == 1.0f) {
        od[j] = best;
        oi[j] = bidx;
        pend = false;
      }
      ball_left = ball_left || f.aux == 2.0f;
    }
  };
  ball_stage();
  // the cubes of radius 1 and 2 for the lanes without a candidate: pooled over the wave like the balls (from kPoolCubeMin
  // lanes on), or a lane per query (from kLaneStageMin on)
  constexpr bool kPooledCube = CAPW >= 384;
  constexpr int kCubeMin = kPooledCube ? kPoolCubeMin : kLaneStageMin;
  auto cube_stage = [&](int rho, bool mine) {
    if constexpr (kPooledCube)
      return wave_pooled_ball_search<LAB, W>(g, cell_start, sorted, slab, qx, qy, qz, ql, mine, best, bidx, s_pts_w, rho);
    else
      return lane_cube_search<LAB, W>(g, cell_start, sorted, slab, qx, qy, qz, ql, rho, mine, best, bidx);
  };
  if (__builtin_popcountll(__ballot(pend && !deferred && !ball_left)) >= kCubeMin) {
    const bool mine = pend && !deferred && !ball_left;
    Found f = cube_stage(1, mine);
    best = f.best;
    bidx = f.bidx;
    if (f.aux == 1.0f) {
      od[j] = best;
      oi[j] = bidx;
      pend = false;
    }
    // a lane whose cube of radius 1 holds NOTHING (27 empty cells: a query in the void between clusters, beside a
    // cloud) would walk 125 more cells to find them empty too, one row's bounds at a time: to the group search,
    // which finds its neighbour from one sample per row (round 3: 14 of the 20 us the cubes cost a wave of blobs8)
    if (pend && mine && f.aux == 0.0f && !(best < __builtin_inff())) {
      open_lane = true;
      pend = false;
    }
    ball_stage();
    const bool mine2 = pend && !deferred && !ball_left && f.aux != 2.0f;
    if (__builtin_popcountll(__ballot(mine2)) >= kCubeMin) {
      f = cube_stage(2, mine2);
      best = f.best;
      bidx = f.bidx;
      if (f.aux == 1.0f) {
        od[j] = best;
        oi[j] = bidx;
        pend = false;
      }
      open_lane = open_lane || (pend && mine2 && f.aux == 0.0f);  // radius 2 examined in full and not enough
      pend = pend && !open_lane;
    }
  }
  PP_SPHASE(6);
  // ---- few lanes: wide stages by the whole wave; then the whole cloud ---------------------------------------
  unsigned long long pending = __ballot(pend);
  unsigned long long open = __ballot(open_lane);  // lanes whose query the cube of radius 2 could not settle
  // (round 5) ... and from kSerialFar on for the lanes that are pending for want of anything near them -- an empty block
  // and empty cubes, or a candidate too far for the ball -- rather than next to a crowded cell (a block through one, a
  // ball that gave up on its rows): the whole-wave cubes around the former are empty cells walked one query after the
  // other (blobs8 0.65 -> 0.58 ms), around the latter they are the right tool (two_scales doubles without them)
  {
    const unsigned long long farish = pending & ~__ballot(deferred || ball_left);
    if (__builtin_popcountll(farish) >= kSerialFar) {
      open |= farish;
      pending &= ~farish;
    }
  }
  // (round 6) ... or where the group search runs anyway for many lanes: the pending lanes join its groups for next to
  // nothing -- the walk is paid per group, not per member -- where one after the other they cost the wave ~9 us each
  // (blobs8 0.625 -> 0.49 ms; from 1, 4, 8 or 32 open lanes on: the same)
  if (__builtin_popcountll(pending) >= kSerialMax ||
      (pending != 0ull && __builtin_popcountll(open) >= kOpenJoin)) {
    // many (next to crowded cells, typically: every one of them would scan those cells by itself, ~ 0.8 wave
    // instructions per candidate and query against ~ 13 per candidate for all of them in the group search)
    open |= pending;
    pending = 0ull;
  }
  unsigned long long longscan = 0ull;
  if (pending) {  // wave-uniform
    const OpenMask om = serve_pending<LAB, W>(reinterpret_cast<const GridSet*>(ws + L.sets) + set, cell_start, sorted, slab, od,
                                           oi, qx, qy, qz, ql, j, (unsigned)pending, (unsigned)(pending >> 32));
    open |= ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)om.hi) << 32) |
            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)om.lo);
    longscan = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)om.lhi) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)om.llo);
  }
  if (longscan) {  // wave-uniform: cubes through crowded cells (never on an evenly sampled surface), or the list kernel's
    const OpenMask ol = serve_long_scans<LAB, W>(reinterpret_cast<const GridSet*>(ws + L.sets) + set, cell_start, sorted,
                                              slab, od, oi, qx, qy, qz, ql, j, (unsigned)longscan,
                                              (unsigned)(longscan >> 32));
    open |= ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)ol.hi) << 32) |
            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)ol.lo);
  }
  PP_SPHASE(7);
  if (open) {  // wave-uniform: far from everything the cubes hold -- group by group, the whole wave (see above)
    const bool finite = __builtin_isfinite(qx) && __builtin_isfinite(qy) && __builtin_isfinite(qz);
    const unsigned long long todo = open & __ballot(finite);
    if (todo) {
      const Found f = wave_group_search<LAB, W>(g, cell_start, sorted, slab, qx, qy, qz, ql, best, bidx, (unsigned)todo,
                                             (unsigned)(todo >> 32), s_pts_w, s_lab_w, (CAPW + 4 - kGroupBatch) * 8,
                                             refined_set ? rowbits : nullptr);  // (only such sets have a bitmap)
      if ((todo >> lane) & 1ull) {
        const bool none = f.bidx == 0x7fffffff;  // (labeled: nobody carries this label -- ref nmdistance_cuda.cu:110-113)
        od[j] = (LAB && none) ? 0.0f : f.best;
        oi[j] = none ? (LAB ? -1 : 0) : f.bidx;
      }
      open &= ~todo;
    }
  }
  PP_SPHASE(8);
  if (open) {  // non-finite queries: every pair, as the brute force orders them
    float sb;
    int si;
    lane_scan_cloud<LAB>((dir ? xyz1 : xyz2) + (size_t)b * nr * 3, LAB ? (dir ? label1 : label2) + (size_t)b * nr : nullptr,
                         nr, qx, qy, qz, ql, sb, si);
    if ((open >> lane) & 1ull) {
      od[j] = sb;
      oi[j] = si;
    }
  }
  PP_SPHASE(9);
}

// The whole search in one kernel (labeled searches; unlabeled ones when there is no stage-A kernel): 256-thread
// workgroups of four independent waves, workgroup `tile` of (b, dir) takes queries [256 tile, 256 tile + 256).
template <bool LAB, int CAPW>
__global__ __launch_bounds__(256, LAB ? (CAPW > 384 ? 3 : 4) : (CAPW > 384 ? 4 : PP_WAVE_WAVES)) void grid_query_wave_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, float* __restrict__ dist1, int* __restrict__ idx1,
    float* __restrict__ dist2, int* __restrict__ idx2, unsigned char* __restrict__ ws, int B, int N, int M, int tiles1,
    int tiles2, int total, int per_xcd, const float* __restrict__ label1, const float* __restrict__ label2) {
  const int V = pp::xcd_virtual_block(blockIdx.x, per_xcd);
  if (V >= total) return;
  const int per_b = tiles1 + tiles2;
  const int b = V / per_b;
  const int r = V - b * per_b;
  const int dir = r >= tiles1 ? 1 : 0;
  const int tile = dir ? r - tiles1 : r;
  const int nq = dir ? M : N;
  const int t = threadIdx.x;
  const Layout L = make_layout(B, N, M, LAB);
  const bool valid = tile * 256 + t < nq;
  const int jj = valid ? tile * 256 + t : nq - 1;
  // (+4: a group of four is read from any staged position without clamping; the tail repeats a real point)
  __shared__ pp::f4 s_pts[4][CAPW + 4];
  __shared__ float s_lab[4][LAB ? CAPW + 4 : 1];
  const int wave = pp::wave_id_uniform();
  search_queries<LAB, CAPW, (LAB ? (CAPW > 384 ? 3 : 4) : (CAPW > 384 ? 4 : PP_WAVE_WAVES))>(xyz1, xyz2, dist1, idx1, dist2, idx2, ws, B, N, M, label1, label2, L, b, dir, jj, valid, false,
                            (lds_f4_wptr)(&s_pts[wave][0]), (lds_f_wptr)(&s_lab[wave][0]));
}

// Round 3, unlabeled searches: what the stage-A kernel (below) left.  The pending list of a direction is 64 slots per
// wave of that kernel plus a count per wave (no atomics there); here every WAVE works by itself: the direction's total
// first (one scalar load -- on an evenly sampled surface 0.3 % of the queries are left, ~50 per direction, and most
// waves leave at once), then, per piece of the list it takes, a prefix sum of the per-wave counts (in its own slice of
// LDS, before the slice is used for points) that turns entry numbers into slots.  A direction's waves_per_set waves
// share its list: few entries are spread one or two to a wave, so that each is served by a whole wave at once (what
// costs there is a query's chain of dependent loads, not lanes); a long list -- clouds far from each other -- fills the
// waves, 64 entries each, in the order of the sorted query cloud (neighbours stay together), piece after piece.
// An entry carries bit 30 when stage A has been run for it (and failed): a wave of such entries skips stage A.
constexpr int kPendTried = 1 << 30;
#ifndef PP_LIST_WAVES
#define PP_LIST_WAVES 5  // waves per SIMD the kernel and its out-of-line stages are compiled for (96 registers)
#endif
#ifndef PP_LIST_WG_WAVES
#define PP_LIST_WG_WAVES 2  // waves per workgroup of the list kernel (independent waves; round 6: 4 -> 2 -- a workgroup's LDS and wave
                            // slots are free when its LAST wave ends, and a long wave held three finished ones': blobs8 0.458 ->
                            // 0.440 ms, 1: the same with the sphere's empty launch +1 %, 8 / 16: 0.49 / 0.62)
#endif
constexpr int kListWgWaves = PP_LIST_WG_WAVES;
#ifndef PP_LIST_RIM_UNITS
#define PP_LIST_RIM_UNITS 4
#endif
constexpr int kListRimUnits = PP_LIST_RIM_UNITS;  // workgroups from either end of a set's tiles that the launch starts first
template <int CAPW>
__global__ __launch_bounds__(64 * kListWgWaves, PP_LIST_WAVES) void grid_query_list_kernel(const float* __restrict__ xyz1,
                                                                  const float* __restrict__ xyz2,
                                                                  float* __restrict__ dist1, int* __restrict__ idx1,
                                                                  float* __restrict__ dist2, int* __restrict__ idx2,
                                                                  unsigned char* __restrict__ ws, int B, int N, int M,
                                                                  int waves_per_set, const Layout L,
                                                                  const unsigned* __restrict__ pre_routed,
                                                                  const unsigned* __restrict__ rowbits) {
  static_assert((CAPW + 4) * 16 >= 1024 * 4, "the prefix sums of up to 1024 counts use the wave's slice");
  __shared__ pp::f4 s_pts[kListWgWaves][CAPW + 4];
  const int wave = pp::wave_id_uniform();
  const int lane = threadIdx.x & 63;
  // a direction's list is served on the XCD that built its grids and wrote the list (the build's and the stage-A kernel's
  // set -> XCD mapping: contiguous ranges of sets per XCD): its tables and points are in that L2 -- served from another
  // XCD every one of a query's dependent loads went to the memory side (24 -> 1x us for the whole launch at config 2).
  // Speed only: whatever the placement, the results are the same.
  // Round 5: the workgroups of an XCD take its sets' RIM tiles first -- the first and the last kListRimUnits workgroups'
  // worth of every set, i.e. the lowest and the highest z-layers of the sorted query cloud, both ends inwards, set after
  // set -- and the rest in plain order afterwards.  Where a direction is searched in whole here (clouds stage A cannot
  // serve), the sparse ends of a cloud hold the queries far from everything: a Gaussian's rim waves live 110-150 us
  // against a mean of 33, and in plain order the last set's started when everybody else was done (the kernel lasted
  // 125 us of work + 110 us of tail, tools/query_probe.py's timeline).  Only the order of the launch changes: every
  // (set, wave) is still served exactly once.
  int gw;
  {
    const int per_xcd = (int)(gridDim.x >> 3);  // (the grid is a multiple of 8)
    const int xcd = (int)blockIdx.x & 7, u = (int)blockIdx.x >> 3;  // this workgroup: number u of its XCD
    const int units = waves_per_set / kListWgWaves;                 // workgroups per set
    const int nsets = units > 0 ? per_xcd / units : 0;              // whole sets per XCD
    const bool regular = waves_per_set % kListWgWaves == 0 && nsets > 0 && nsets * units == per_xcd && units > 4 * kListRimUnits;
    int v = u;  // the XCD's workgroup in plain order (a set's tiles in order, set after set)
    if (regular) {
      const int nrim = 2 * kListRimUnits * nsets;
      if (u < nrim) {
        const int r = u / nsets, j = u - r * nsets;  // rim rank (0: first unit, 1: last, 2: second, 3: last but one, ...), set
        v = j * units + ((r & 1) ? units - 1 - (r >> 1) : (r >> 1));
      } else {
        const int w = u - nrim, inner = units - 2 * kListRimUnits;
        const int j = w / inner;
        v = j * units + kListRimUnits + (w - j * inner);
      }
    }
    gw = (xcd * per_xcd + v) * kListWgWaves + wave;
  }
  const int set = gw / waves_per_set, wi = gw - set * waves_per_set;
  // (round 5, measured and removed: every rim workgroup launched 2 / 4 / 8 times, each copy serving a part of its waves'
  //  lanes -- a Gaussian's first waves live the whole launch -- made every cloud slower: gaussian 0.210 -> 0.212 / 0.221 /
  //  0.231, blobs8 0.70 -> 0.71 / 0.75 / 0.78: a rim wave's life is not in proportion to its open queries)
  // (round 3, tried: a set's waves from its two ends inwards -- the rim waves are the expensive ones -- changed nothing;
  //  the XCD's sets interleaved as well spread the expensive waves over the launch -- gaussian 0.265 -> 0.243 ms, blobs8
  //  0.83 -> 0.78 -- but cost the clouds whose neighbouring waves share cells their cache hits: two_scales 0.316 -> 0.335,
  //  shapenet_like 0.21 -> 0.23.  Plain order.)
  if (set >= 2 * B) return;
  // (round 6) a direction routed to the every-pair kernel in front of the build: nothing of it is here (its tables may
  // not even have been built)
  if (pre_routed != nullptr && pre_routed[set] != 0u) return;
  const int b = set >> 1, dir = set & 1;
  const int nq = dir ? M : N;
  const int nwq = (nq + 63) / 64;  // waves of the stage-A kernel in this direction (<= 1024)
  const unsigned* __restrict__ pcnt = reinterpret_cast<const unsigned*>(ws + L.pend_cnt) + pend_count_offset(b, dir, N, M);
  const int* __restrict__ plist = reinterpret_cast<const int*>(ws + L.pend) + set_point_offset(b, dir ^ 1, N, M);
  unsigned __attribute__((address_space(3)))* pref = (unsigned __attribute__((address_space(3)))*)(&s_pts[wave][0]);
  // Exclusive prefix sums of the per-wave counts into the wave's slice (lane l takes counts cpl l .. cpl l + cpl - 1,
  // cpl = 4, 8 or 16 by the number of waves); returns the direction's total.  Redone for every piece: the search
  // overwrites the slice.
  // (loads unconditional, indices clamped, the count of loads a compile-time constant: a load inside an `if` is waited
  //  for on the spot, one round trip each)
  const int cpl = nwq <= 256 ? 4 : (nwq <= 512 ? 8 : 16);
  auto scan_n = [&](auto cpl_c) -> unsigned {
    constexpr int CPL = decltype(cpl_c)::value;
    unsigned c[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) c[i] = pcnt[min(CPL * lane + i, nwq - 1)];
    unsigned mine = 0;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      c[i] = CPL * lane + i < nwq ? c[i] : 0u;
      mine += c[i];
    }
    const unsigned incl = pp::wave_scan_u32_dpp(mine);
    unsigned run = incl - mine;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      pref[CPL * lane + i] = run;
      run += c[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
  };
  auto scan_counts = [&]() -> unsigned {
    if (cpl == 4) return scan_n(std::integral_constant<int, 4>{});
    if (cpl == 8) return scan_n(std::integral_constant<int, 8>{});
    return scan_n(std::integral_constant<int, 16>{});
  };
  // the direction's total first (one scalar load; the stage-A kernel adds it up per tile): on an evenly sampled surface
  // nothing, or next to nothing, is left and the wave leaves here
  const unsigned total = (reinterpret_cast<const unsigned*>(ws + L.layers) + (size_t)set * pp::kLayerWords)[pp::kLayerPending];
  if (total == 0u) return;
  const unsigned per_wave = min(64u, max(1u, (total + (unsigned)waves_per_set - 1) / (unsigned)waves_per_set));
  // ONE piece of per_wave entries per wave (the launch has a wave for every 64 queries of a direction, so a list of
  // every query still fits; no loop: state that lives across the search costs this kernel registers it does not have)
  // (round 5, measured and removed: a persistent launch -- 4 / 5 / 8 workgroups per CU, each wave on to the piece of
  //  virtual workgroup w + k gridDim.x: the hardware's dispatch of the next waiting workgroup to whichever CU has room
  //  balances the long pieces better than a fixed stride: gaussian 0.198 -> 0.264 / 0.232 / 0.213 ms, blobs8 0.65 ->
  //  0.81 / 0.71 / 0.73, and the sphere's 7 us did not move; 97 spilled scalar registers on top)
  const bool everything = total == (unsigned)nq;  // the stage-A kernel served nothing of this direction (sets with
                                                  // crowded cells, degenerate sets): entry e is query e, no table needed
  {
    const unsigned first = (unsigned)wi * per_wave;
    if (first >= total) return;  // (wave-uniform)
    const unsigned e = first + (unsigned)lane;
    const bool valid = (unsigned)lane < per_wave && e < total;
    int entry = (int)(valid ? e : total - 1);  // (everything pending: entry e is query e)
    bool skip_a = false;
    if (!everything) {
      scan_counts();
      const unsigned ec = valid ? e : min(first, total - 1);
      int lo = 0, hi = nwq - 1;  // the last wave whose first entry is <= ec (waves without entries are passed over)
#pragma unroll
      for (int it = 0; it < 10; ++it) {
        const int mid = (lo + hi + 1) >> 1;
        const bool ge = pref[mid] <= ec;
        lo = ge ? mid : lo;
        hi = ge ? hi : mid - 1;
      }
      entry = plist[lo * 64 + (int)(ec - pref[lo])];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();  // (the slice is the search's from here on)
      skip_a = __all(!valid || (entry & kPendTried) != 0);
    }
    search_queries<false, CAPW, PP_LIST_WAVES>(xyz1, xyz2, dist1, idx1, dist2, idx2, ws, B, N, M, nullptr, nullptr, L, b, dir,
                                entry & ~kPendTried, valid, skip_a, (lds_f4_wptr)(&s_pts[wave][0]), (lds_f_wptr) nullptr,
                                rowbits != nullptr ? rowbits + (size_t)set * 32 : nullptr);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Round 3: stage A as a kernel of its own (unlabeled searches; the kernel above, in LIST mode, then serves what it
// leaves).  Why: inside one kernel the rare long tails -- a wave with leftover queries runs the whole-wave cubes for
// microseconds -- held back the LDS of their whole workgroup, and the tails' registers (80) capped the occupancy of
// the 99.7 % of the work that never needs them.  Here a workgroup is a TILE of TQ consecutive queries of the sorted
// query cloud:
//   * the z-layers of the reference grid its queries' 2x2x2 blocks can touch come from the QUERY cloud's chunk table
//     (grid_common.h: kChunk; two scalar loads per build slab and chunk -- no reduction over the tile, nothing waits
//     for the queries themselves); those layers are ONE contiguous piece of the sorted reference cloud (cells are
//     z-major), copied into ONE LDS image of at most CAP points shared by the tile's waves, while the lanes' own
//     queries and the bounds of their blocks' rows are still on their way; one workgroup barrier;
//   * every lane walks the four rows of its block in the image -- one sequence of groups of four consecutive points,
//     only the running minimum (v_min3) and the byte position of the group that last lowered it tracked; the winner's
//     index is recovered from that group afterwards, an exact tie across groups repeats the walk in the exact
//     (distance, index) order -- and is settled if its best distance is below what the block guarantees (reach);
//   * settled queries store their result; the others are written to the wave's 64 slots of the pending list with the
//     wave's count (no atomics).  A tile that cannot be served this way (sets with crowded cells, degenerate sets,
//     images beyond CAP, no chunk table) leaves ALL its queries pending: same results either way.
// Same candidates, same arithmetic (pp::chamfer_d3), same tie rule as the kernel above.
__device__ __forceinline__ unsigned lean_group_pos(unsigned k, unsigned T1, unsigned T2, unsigned T3, unsigned a0,
                                                   unsigned a1, unsigned a2, unsigned a3, unsigned endb) {
  // byte position in the image of group k of a lane's sequence (everything by value: see stage_a_fetch)
  const unsigned a = k < T1 ? a0 : (k < T2 ? a1 : (k < T3 ? a2 : a3));
  return min(a + (k << 6), endb);
}
typedef const char __attribute__((address_space(3))) * lds_c_ptr;

// (phase stamps of the stage-A kernel: off in a -DPP_PROBE_LIST_ONLY probe build, which clocks the list kernel alone)
#if defined(PP_QUERY_PROBE) && !defined(PP_PROBE_LIST_ONLY)
#define PP_APHASE(n) PP_QPHASE(n)
#else
#define PP_APHASE(n)
#endif
// What a tile's front needs from memory, ordered in ONE round trip and a tile AHEAD (the kernel is persistent: a
// workgroup walks tile i while this is on its way for tile i + 1): lanes 0..47 of `meta` = the reference grid's
// descriptor (16 words), the query grid's (16), the tile's chunk-table entries (<= 16 words for TQ <= 512; 32 at 1024:
// meta2); `lay` = the reference grid's layer table (lane z: first point of layer z); `qq` = the lane's query.
template <int TQ>
struct StageAFront {
  int live;             // the virtual tile exists (the grid is padded to a multiple of eight)
  int b, dir, tile, jj;
  bool valid;
  unsigned meta, meta2, lay;
  pp::f4 qq;
};
template <int TQ>
__device__ __forceinline__ void stage_a_issue(StageAFront<TQ>& f, int it, int per_xcd, int total, int tiles1, int tiles2,
                                              int N, int M, const unsigned char* __restrict__ ws, const Layout& L, int t,
                                              int lane) {
  const int V = pp::xcd_virtual_block(it, per_xcd);
  f.live = V < total ? 1 : 0;
  const int Vc = f.live ? V : 0;  // (a padding tile loads tile 0's front and does nothing with it)
  const int per_b = tiles1 + tiles2;
  f.b = Vc / per_b;
  const int r = Vc - f.b * per_b;
  f.dir = r >= tiles1 ? 1 : 0;
  f.tile = f.dir ? r - tiles1 : r;
  const int nq = f.dir ? M : N;
  f.valid = f.live && f.tile * TQ + t < nq;
  f.jj = f.valid ? f.tile * TQ + t : nq - 1;
  const int set = 2 * f.b + f.dir;
  constexpr int kTz = 8 * (TQ / pp::kChunk);  // chunk-table words of a tile
  static_assert(pp::kBuildSlabs == 4 && kTz <= 32, "eight words per chunk");
  const unsigned* __restrict__ gsets = reinterpret_cast<const unsigned*>(ws + L.sets);
  const unsigned* __restrict__ tz = reinterpret_cast<const unsigned*>(ws + L.tile_z) +
                                    ((size_t)(set ^ 1) * L.chunks + (size_t)f.tile * (TQ / pp::kChunk)) * 8;
  const unsigned* __restrict__ src = lane < 16 ? gsets + 16 * (size_t)set + lane
                                               : (lane < 32 ? gsets + 16 * (size_t)(set ^ 1) + (lane - 16)
                                                            : tz + min(lane - 32, kTz - 1));
  f.meta = *src;
  f.meta2 = kTz > 16 ? tz[min(lane, kTz - 1)] : 0u;
  f.lay = (reinterpret_cast<const unsigned*>(ws + L.layers) + (size_t)set * pp::kLayerWords)[lane < pp::kLayerWords ? lane : 0];
  const pp::f4* __restrict__ qsorted = reinterpret_cast<const pp::f4*>(ws + L.sorted) + set_point_offset(f.b, f.dir ^ 1, N, M);
  // (an asm load: a plain one the compiler sinks to its first use, a round trip later; it is invisible to the compiler's
  //  counting of loads in flight, so an explicit s_waitcnt vmcnt(0) precedes every use -- stage_a_kernel's loop top)
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(f.qq) : "v"((unsigned)f.jj << 4), "s"(qsorted) : "memory");
}

template <int TQ, int CAP, int WPE, bool PERSIST>
__global__ __launch_bounds__(TQ, WPE) void grid_stage_a_kernel(float* __restrict__ dist1, int* __restrict__ idx1,
                                                              float* __restrict__ dist2, int* __restrict__ idx2,
                                                              unsigned char* __restrict__ ws, int B, int N, int M,
                                                              int tiles1, int tiles2, int total, int per_xcd,
                                                              const Layout L, unsigned* __restrict__ routed_host,
                                                              unsigned epoch, const unsigned* __restrict__ pre_routed) {
  static_assert(TQ % pp::kChunk == 0 && (CAP + 63) / 64 * 64 + 4 <= 4096 + 4, "");
  // PERSISTENT: the launch is a few workgroups per CU (a multiple of eight, so that a workgroup's tiles stay on its
  // XCD under the round-robin placement -- speed only); workgroup w takes the virtual tiles w, w + gridDim.x, ...
  const int nvt = per_xcd * 8;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = pp::wave_id_uniform();
  // (the per-direction workgroups come FIRST in the launch -- 2 B of them rounded up to a multiple of eight, so that the
  //  tiles keep their XCDs: behind the tiles they started last and a launch whose tiles all decline waited 2-3 us for them)
  const int head = PERSIST ? 0 : (2 * B + 7) / 8 * 8;
  const int bid = (int)blockIdx.x - head;  // the tile's workgroup
  if (!PERSIST && bid < 0) {
    // Round 6: 2 B workgroups behind the tiles' (dispatched last, into the launch's tail), one per direction: is this
    // a direction no search can prune (direction_unprunable, on samples of the sorted clouds)?  Then the host is told
    // (routed_host): from its next calls on the decision is taken in front of the build (route_decide_kernel) and the
    // every-pair kernel serves the routed directions.  This call's search serves them itself.
    const int set = (int)blockIdx.x;
    if (set >= 2 * B) return;
    const GridSet* gs = reinterpret_cast<const GridSet*>(ws + L.sets);
    // ... and which cell rows of the set's grid hold points at all (rowbits): between clusters the group search's boxes
    // are hundreds of rows of which a handful are not empty (blobs8: 236 rows listed per group, ~6 with points)
    if (pre_routed == nullptr || pre_routed[set] == 0u) {  // (a routed direction's tables may not have been built)
      const GridSet g = gs[set];
      const int gx = min(max(g.gx, 1), pp::kGridMax);
      const int nall = g.useless ? 0 : min(max(g.gy, 1), pp::kGridMax) * min(max(g.gz, 1), pp::kGridMax);
      const unsigned* __restrict__ cs = reinterpret_cast<const unsigned*>(ws + L.cell_start) + (size_t)set * kCellStride;
      unsigned* __restrict__ rb = reinterpret_cast<unsigned*>(ws + L.rowbits) + (size_t)set * 32;
      // (only for a set with crowded cells -- clusters, several scales: the clouds whose grids are mostly empty rows.  The
      //  lookups touch every line of the cell table, 4.5 MB over the 64 sets of config 2, and an evenly sampled surface
      //  never gets to the group search: its words say "every row", which the search reads as "no bitmap")
      const bool wanted = pp::grid_refined(g);
      for (int r0 = wave * 64; r0 < pp::kGridMax * pp::kGridMax; r0 += TQ) {  // (wave-uniform)
        const int r = r0 + lane;
        const bool ne = !wanted || (r < nall && cs[(r + 1) * gx] != cs[r * gx]);
        const unsigned long long bal = __ballot(ne);
        if (lane == 0) {
          rb[r0 >> 5] = (unsigned)bal;
          rb[(r0 >> 5) + 1] = (unsigned)(bal >> 32);
        }
      }
    }
    if (wave != 0 || routed_host == nullptr) return;
    const int b = set >> 1, dir = set & 1;
    const int r_useless = gs[set].useless, q_useless = gs[set ^ 1].useless;
    const int nr = dir ? N : M, nq = dir ? M : N;
    bool hopeless = r_useless != 0;  // (a degenerate reference set -- identical points: no grid -- is routed as it is)
    if (!r_useless && !q_useless) {
      const float* rs = reinterpret_cast<const float*>(reinterpret_cast<const pp::f4*>(ws + L.sorted) + set_point_offset(b, dir, N, M));
      const float* qs = reinterpret_cast<const float*>(reinterpret_cast<const pp::f4*>(ws + L.sorted) + set_point_offset(b, dir ^ 1, N, M));
      hopeless = direction_unprunable(qs, nq, 4, rs, nr, 4, lane);
    }
    if (lane == 0) {
      (reinterpret_cast<unsigned*>(ws + L.layers) + (size_t)set * pp::kLayerWords)[pp::kLayerRouted] = hopeless ? 1u : 0u;
      if (hopeless && routed_host) *reinterpret_cast<volatile unsigned*>(routed_host) = epoch;
    }
    return;
  }
  constexpr int kW = TQ / 64;
  constexpr int kQueue = 64;  // leftovers of a tile served from the image (more than that stay for the list kernel)
  __shared__ pp::f4 s_img[(CAP + 63) / 64 * 64 + 4];  // (whole pieces of 64 points, then the padding)
  __shared__ pp::f4 s_queue[kQueue];
  __shared__ unsigned s_qres[kQueue];
  __shared__ unsigned s_qn;
  __shared__ unsigned s_tot;  // queries of the tile left pending
  PP_QPHASE_DECL;
  StageAFront<TQ> nx;
  stage_a_issue<TQ>(nx, bid, per_xcd, total, tiles1, tiles2, N, M, ws, L, t, lane);
  // A tile's results are stored at the top of the NEXT iteration, behind that iteration's wait for its front: the
  // wait below must be vmcnt(0) (the asm load), and scattered 4-byte stores issued just before it would make every
  // tile wait for them to retire.
  float sv_d = 0.0f;
  int sv_i = 0, sv_off = 0, sv_dir = 0;
  bool sv_ok = false;
  // (!PERSIST: a workgroup per tile, the loop is one pass and the compiler knows it: no state lives across it)
  for (int it = bid; it < (PERSIST ? nvt : bid + 1); it += (PERSIST ? (int)gridDim.x : 1)) {  // workgroup-uniform
  PP_APHASE(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's front has landed (the asm load of the query too)
  if (sv_ok) {
    (sv_dir ? dist2 : dist1)[sv_off] = sv_d;
    (sv_dir ? idx2 : idx1)[sv_off] = sv_i;
  }
  sv_ok = false;
  if (t == 0) {  // the tile's queue of leftovers and its count of pending queries: ordered before their first use by the
    s_qn = 0u;   // barrier behind the image's arrival -- or by the one a tile whose image does not fit meets instead
    s_tot = 0u;
  }
  const StageAFront<TQ> f = nx;
  if (PERSIST && it + (int)gridDim.x < nvt)  // the next tile's front goes out now and travels while this tile is walked
    stage_a_issue<TQ>(nx, it + (int)gridDim.x, per_xcd, total, tiles1, tiles2, N, M, ws, L, t, lane);
  const int b = f.b, dir = f.dir, tile = f.tile, jj = f.jj;
  const bool valid = f.valid;
  const int nq = dir ? M : N;
  const int set = 2 * b + dir;
  if (pre_routed != nullptr && f.live && pre_routed[set] != 0u) {  // (uniform over the direction) served by the every-pair kernel
    if constexpr (PERSIST) {
      __syncthreads();
      continue;
    } else {
      return;
    }
  }
  const pp::f4 qq = f.qq;
  const unsigned lay = f.lay;
  auto meta = [&](int l) { return __builtin_amdgcn_readlane((int)f.meta, l); };
  int kmin = 0x7fffffff, kmax = (int)0x80000000;  // the tile's lowest / highest z (pp::zkey)
  {
    constexpr int kTz = 8 * (TQ / pp::kChunk);
#pragma unroll
    for (int w = 0; w < kTz; w += 2) {
      const int lo = kTz > 16 ? __builtin_amdgcn_readlane((int)f.meta2, w) : meta(32 + w);
      const int hi = kTz > 16 ? __builtin_amdgcn_readlane((int)f.meta2, w + 1) : meta(32 + w + 1);
      kmin = min(kmin, lo);
      kmax = max(kmax, hi);
    }
  }
  int* __restrict__ plist = reinterpret_cast<int*>(ws + L.pend) + set_point_offset(b, dir ^ 1, N, M);
  unsigned* __restrict__ pcnt = reinterpret_cast<unsigned*>(ws + L.pend_cnt) + pend_count_offset(b, dir, N, M);
  bool pend = valid;  // (until settled)
  int tried = 0;      // kPendTried once stage A has really been run for the tile
  GridSet g;          // the reference grid (lanes 0..15 of meta: see the static_assert on GridSet's layout)
  static_assert(sizeof(GridSet) == 64 && offsetof(GridSet, gx) == 20 && offsetof(GridSet, useless) == 32 &&
                    offsetof(GridSet, pad) == 36 && offsetof(GridSet, crowd) == 48, "the words read below");
  g.minx = __int_as_float(meta(0)); g.miny = __int_as_float(meta(1)); g.minz = __int_as_float(meta(2));
  g.h = __int_as_float(meta(3)); g.invh = __int_as_float(meta(4));
  g.gx = meta(5); g.gy = meta(6); g.gz = meta(7);
  // (uniform over the set) grids without crowded cells on both sides (crowd: 1 useless, 2 second-level grids), no
  // degenerate set, and the query cloud's chunk table exists; no short-circuits: nothing here is worth a branch
  const bool grids_ok = ((meta(8) | meta(12) | meta(13) | meta(14) | meta(15) | meta(16 + 8) | meta(16 + 12) | meta(16 + 13) |
                          meta(16 + 14) | meta(16 + 15) | (meta(16 + 10) ^ 1)) == 0);
  // ... and the images of this direction's tiles are likely to fit: a tile of TQ queries spans about TQ gz / nq layers
  // of the reference grid, its image those and three more (one straddled, one either side), each at most as full as
  // the grid's fullest layer.  A volume-filling cloud (coarser grid, fuller layers), a plane (one layer) or a Gaussian
  // (its core) fails this, and its tiles would find their images too large one by one.
  // (the fullest layer counts, not the average: the core of a Gaussian, a face of a box)
  unsigned lmax;
  {
    const unsigned nxt = (unsigned)__shfl_down((int)lay, 1);
    const float sz = lane < g.gz ? (float)(nxt - lay) : 0.0f;  // (layer populations are < 2^24: exact)
    lmax = (unsigned)pp::wave_reduce_dpp<false>(sz);
  }
  const bool fits = (long long)(3 + (TQ * g.gz + nq - 1) / nq) * (long long)lmax <= (long long)CAP;
  if (f.live && !(grids_ok && fits)) {
    // (uniform over the direction) nothing of this direction is served here: its total is set to "every query" by the
    // first tile (no list is written: the list kernel then takes entry e to be query e) and the tile is done -- a
    // cloud stage A cannot serve costs this launch little more than its dispatch
    // (round 5, measured and removed: these tiles running stage A in the wave-private form here -- search_queries cut
    //  behind stage A, out of line, the open queries to the pending list so that the list kernel finds them compacted:
    //  bit-identical, but this kernel then took 124 us on a filled cube and 167 on a Gaussian -- three workgroups per CU
    //  by its image's LDS, 416 bytes of call frames -- against the 60 us stage A costs inside the list kernel, and the
    //  list kernel gained 10-40 us: cube 0.129 -> 0.199 ms, gaussian 0.226 -> 0.374)
    if (tile == 0 && t == 0)
      (reinterpret_cast<unsigned*>(ws + L.layers) + (size_t)set * pp::kLayerWords)[pp::kLayerPending] = (unsigned)nq;
    if constexpr (PERSIST) {
      __syncthreads();
      continue;
    } else {
      return;
    }
  }
  const bool lean_ok = f.live != 0;
  if (lean_ok) {
    const float qx = qq.x, qy = qq.y, qz = qq.z;
    const unsigned* __restrict__ cell_start =
        reinterpret_cast<const unsigned*>(ws + L.cell_start) + (size_t)set * kCellStride;
    const pp::f4* __restrict__ sorted = reinterpret_cast<const pp::f4*>(ws + L.sorted) + set_point_offset(b, dir, N, M);
    const float inf = __builtin_inff();
    const int gx1 = __builtin_amdgcn_readfirstlane(g.gx - 1), gy1 = __builtin_amdgcn_readfirstlane(g.gy - 1),
              gz1 = __builtin_amdgcn_readfirstlane(g.gz - 1);
    // the tile's z range (world coordinates) -> the layers of the reference grid its blocks can touch: a block holds
    // the query's layer and one neighbour, and pp::cell_coord is monotone in z
    const int Lz = max(__builtin_amdgcn_readfirstlane(cell_coord(pp::zkey_inv(kmin), g.minz, g.invh, g.gz)) - 1, 0);
    const int Hz = min(__builtin_amdgcn_readfirstlane(cell_coord(pp::zkey_inv(kmax), g.minz, g.invh, g.gz)) + 1, gz1);
    const unsigned tb0 = (unsigned)__builtin_amdgcn_readlane((int)lay, Lz);
    const unsigned ns = (unsigned)__builtin_amdgcn_readlane((int)lay, Hz + 1) - tb0;
    if (!(ns > 0u && ns <= (unsigned)CAP)) __syncthreads();  // (the counters above are zero before anyone adds to them)
    if (ns > 0u && ns <= (unsigned)CAP) {  // workgroup-uniform
      // the image: [tb0, tb0 + ns) of the sorted cloud, by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write),
      // in pieces of 64 points, wave w the pieces w, w + kW, ...; ordered here, before anything waits.  A piece's lanes
      // beyond the image's end re-read its last point (the slots behind the end are the padding's, written below).
      const pp::f4* __restrict__ src = sorted + tb0;
      for (unsigned pc = (unsigned)wave; pc * 64u < ns; pc += (unsigned)kW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + min(pc * 64u + (unsigned)lane, ns - 1)),
                                         (__attribute__((address_space(3))) void*)(&s_img[pc * 64u]), 16, 0, 0);
      const float px = (qx - g.minx) * g.invh, py = (qy - g.miny) * g.invh, pz = (qz - g.minz) * g.invh;  // in cells
      int cx, cy, cz;  // the query's cell (pp::cell_coord's arithmetic)
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(cx) : "v"((int)px), "s"(gx1));
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(cy) : "v"((int)py), "s"(gy1));
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(cz) : "v"((int)pz), "s"(gz1));
      // the 2x2x2 block: cells l, l + 1 per axis (the neighbour on the side of the cell the query lies in), clamped
      const int lx = px - (float)cx < 0.5f ? cx - 1 : cx, ly = py - (float)cy < 0.5f ? cy - 1 : cy,
                lz = pz - (float)cz < 0.5f ? cz - 1 : cz;
      const int x0 = max(lx, 0), x1 = min(lx + 1, gx1), y0 = max(ly, 0), y1 = min(ly + 1, gy1), z0 = max(lz, 0),
                z1 = min(lz + 1, gz1);
      // What the block guarantees: along each axis the distance to the nearer face of the block that has grid beyond
      // it (lower face at coordinate l, cells below it exist iff l >= 1; upper face at l + 2, cells above iff
      // l + 2 <= cells - 1); beyond the grid there is nothing, the rim cells hold what was clamped into them.
      auto face = [&](float p, int l, int g1) {
        const float lo = l >= 1 ? p - (float)l : inf;
        const float hi = l + 1 < g1 ? (float)(l + 2) - p : inf;
        return fminf(lo, hi);
      };
      const float reach = g.h * fminf(face(px, lx, gx1), fminf(face(py, ly, gy1), face(pz, lz, gz1)));
      // the bounds of the block's four rows (y, z): a row is one or two cells wide, its bounds at most two table
      // entries apart: one 12-byte load each
      typedef unsigned u3 __attribute__((ext_vector_type(3)));
      u3 r00, r01, r10, r11;
      {
        const unsigned gx4 = (unsigned)g.gx << 2, x04 = (unsigned)x0 << 2;
        auto row_off = [&](int z, int y) {  // byte offset of entry (x0, y, z): every factor fits 24 bits
          unsigned o;
          asm("v_mad_u32_u24 %0, %1, %2, %3\n\tv_mad_u32_u24 %0, %0, %4, %5" : "=&v"(o) : "v"(z), "s"(g.gy), "v"(y), "s"(gx4), "v"(x04));
          return o;
        };
        const char* __restrict__ tb = reinterpret_cast<const char*>(cell_start);
        __builtin_memcpy(&r00, tb + row_off(z0, y0), 12);
        __builtin_memcpy(&r01, tb + row_off(z0, y1), 12);
        __builtin_memcpy(&r10, tb + row_off(z1, y0), 12);
        __builtin_memcpy(&r11, tb + row_off(z1, y1), 12);
      }
      PP_APHASE(1);
      // rows -> byte positions in the image
      const bool wide = x1 > x0;
      const bool va0 = lz >= 0, va1 = lz < gz1, vb0 = ly >= 0, vb1 = ly < gy1;  // the row's layer / line exists
      const unsigned s0 = r00.x, e0 = (va0 & vb0) ? (wide ? r00.z : r00.y) : s0;
      const unsigned s1 = r01.x, e1 = (va0 & vb1) ? (wide ? r01.z : r01.y) : s1;
      const unsigned s2 = r10.x, e2 = (va1 & vb0) ? (wide ? r10.z : r10.y) : s2;
      const unsigned s3 = r11.x, e3 = (va1 & vb1) ? (wide ? r11.z : r11.y) : s3;
      const unsigned t0 = (e0 - s0 + 3) >> 2, t1 = (e1 - s1 + 3) >> 2, t2 = (e2 - s2 + 3) >> 2, t3 = (e3 - s3 + 3) >> 2;
      const unsigned T1 = t0, T2 = T1 + t1, T3 = T2 + t2, T4 = T3 + t3;
      // group k of the lane's sequence starts at byte a_r + 64 k of the image, r the row k falls in
      const unsigned a0 = (s0 - tb0) << 4, a1 = ((s1 - tb0) << 4) - (T1 << 6), a2 = ((s2 - tb0) << 4) - (T2 << 6),
                     a3 = ((s3 - tb0) << 4) - (T3 << 6);
      const unsigned endb = ns << 4;  // the padding
      const int kmax = (int)pp::wave_reduce_dpp<false>((float)T4);
#ifdef PP_QUERY_PROBE_WALK  // (diagnostic: how full the walk's iterations are -- groups of the lanes against 64 x the wave's longest)
      {
        const float tsum = pp::wave_reduce_dpp<true>((float)T4);
        if (lane == 63) {
          atomicAdd(&g_qgroup[0], (unsigned long long)tsum);
          atomicAdd(&g_qgroup[1], 64ull * (unsigned long long)kmax);
          atomicAdd(&g_qgroup[2], 1ull);
        }
      }
#endif
      PP_APHASE(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the image have landed
      // the padding: four points that can never be taken (their distance is NaN); lanes whose rows are finished, and
      // groups that run past the end of the image, land here.  Written by the wave that owns the piece the image ends
      // in, after that piece has landed (its lanes beyond the end wrote the slots first).
      if (wave == (int)((ns >> 6) % (unsigned)kW) && lane < 4) {
        const float qn = __builtin_nanf("");
        const pp::f4 nanp = {qn, qn, qn, __int_as_float(0x7fffffff)};
        s_img[ns + lane] = nanp;
      }
      __syncthreads();
      PP_APHASE(3);
      const lds_c_ptr lb = (lds_c_ptr)(&s_img[0]);
      auto pos_of = [=](unsigned k) { return lean_group_pos(k, T1, T2, T3, a0, a1, a2, a3, endb); };
      pp::f4 pa[4], pb[4];
      auto fetch4 = [&](unsigned pos, pp::f4 (&p)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = *(lds_f4_ptr)(lb + pos + 16 * u);
      };
      float best = inf;
      int bidx = 0x7fffffff;
      unsigned gpos = endb;           // byte position of the group that holds the winner
      unsigned long long tie = 0ull;  // lanes that met a distance equal to their running minimum in a later group
      auto track = [&](unsigned pos, const pp::f4 (&p)[4]) {
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = pp::chamfer_d3(p[u].x, p[u].y, p[u].z, qx, qy, qz);
        const float gmin = min2(pp::min3(d[0], d[1], d[2]), d[3]);
        const bool lt = gmin < best;
        tie |= __ballot(gmin == best);
        gpos = lt ? pos : gpos;
        best = lt ? gmin : best;
      };
      unsigned pcur = pos_of(0), pnext;
      fetch4(pcur, pa);
      for (int k = 0; k < kmax; k += 2) {
        pnext = pos_of(k + 1);
        fetch4(pnext, pb);
        track(pcur, pa);
        pcur = pos_of(k + 2);
        fetch4(pcur, pa);
        track(pnext, pb);
      }
      if (tie) {  // an exact tie across groups (duplicated points, lattices): the walk again in the exact order
        best = inf;
        auto examine = [&](const pp::f4 (&p)[4]) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float d = pp::chamfer_d3(p[u].x, p[u].y, p[u].z, qx, qy, qz);
            const int id = __float_as_int(p[u].w);
            const bool take = (d < best) | ((d == best) & (id < bidx));
            best = take ? d : best;
            bidx = take ? id : bidx;
          }
        };
        fetch4(pos_of(0), pa);
        for (int k = 0; k < kmax; k += 2) {
          fetch4(pos_of(k + 1), pb);
          examine(pa);
          fetch4(pos_of(k + 2), pa);
          examine(pb);
        }
      } else {  // the winner is in the group at gpos: lowest index among its minima
        fetch4(gpos, pa);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = pp::chamfer_d3(pa[u].x, pa[u].y, pa[u].z, qx, qy, qz);
          const int id = __float_as_int(pa[u].w);
          bidx = ((d == best) & (id < bidx)) ? id : bidx;
        }
      }
      PP_APHASE(4);
      tried = kPendTried;
      if (valid && best < reach * reach * kBoundSlack) {  // settled (strict; a NaN bound settles nothing)
        sv_ok = true;  // (stored at the top of the next iteration, or after the loop)
        sv_d = best;
        sv_i = bidx;
        sv_off = b * nq + __float_as_int(qq.w);  // (B * (N + M) < 2^31: grid_applicable)
        pend = false;
      }
      sv_dir = dir;
      // ---- what the block left (about one query in 150 on an evenly sampled surface): the cube of Chebyshev radius 1
      // around the query's cell, FROM THE IMAGE (it holds the layers cz - 1 .. cz + 1 of every query of the tile), by
      // a whole wave per query: the tile's leftovers are gathered in a queue in LDS, wave w takes entries w, w + kW,
      // ...; lane r < 9 fetches the bounds of row r of the cube (the only trip to memory), the rows are laid end to
      // end and every lane examines its share of the candidates in the exact (distance bits, index) order.  Settled
      // if the best distance lies below what the cube guarantees.  Served here these queries cost a microsecond of
      // the tile's time; left to the list kernel each costs that launch a chain of eight dependent round trips.
      int qslot = -1;
      if (pend) {
        qslot = (int)atomicAdd(&s_qn, 1u);
        if (qslot < kQueue) s_queue[qslot] = pp::f4{qx, qy, qz, qq.w};
      }
      __syncthreads();
      const int nqueue = min((int)s_qn, kQueue);
      for (int e = wave; e < nqueue; e += kW) {  // wave-uniform
        const pp::f4 w = s_queue[e];
        const int wcx = cell_coord(w.x, g.minx, g.invh, g.gx), wcy = cell_coord(w.y, g.miny, g.invh, g.gy),
                  wcz = cell_coord(w.z, g.minz, g.invh, g.gz);
        const int wx0 = max(wcx - 1, 0), wx1 = min(wcx + 1, gx1);
        const int rz = wcz - 1 + lane / 3, ry = wcy - 1 + lane % 3;
        const bool rok = lane < 9 && rz >= 0 && rz <= gz1 && ry >= 0 && ry <= gy1;
        unsigned rs = 0, re = 0;
        if (rok) {
          const int c = pp::cell_linear(0, ry, rz, g.gx, g.gy);
          rs = cell_start[c + wx0];
          re = cell_start[c + wx1 + 1];
        }
        // (every row of the cube lies in the image by construction; a row that does not -- never, but the image's
        //  extent is an estimate made from the chunk table -- leaves the query to the list kernel)
        const bool inside = __all(!rok || (rs >= tb0 && re <= tb0 + ns));
        const unsigned len = rok ? re - rs : 0u;
        const unsigned incl = pp::wave_scan_u32_dpp(len);  // (DPP: the shuffles were a chain of ds_bpermute round trips in a
                                                            //  tile's tail; lanes 9.. hold no row)
        const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 15);
        const unsigned excl = incl - len, shift = (rs - tb0) - excl;  // candidate k of row r: image[k + shift_r]
        unsigned long long key = ((unsigned long long)0x7f800000u << 32) | 0x7fffffffu;  // (+inf, no index)
        for (unsigned k0 = 0; k0 < tot; k0 += 64) {
          const unsigned k = k0 + (unsigned)lane;
          unsigned add = 0;
#pragma unroll
          for (int r = 0; r < 9; ++r) {  // the last row whose first candidate is <= k (empty rows are overridden)
            const unsigned ex = (unsigned)__builtin_amdgcn_readlane((int)excl, r);
            const unsigned sh = (unsigned)__builtin_amdgcn_readlane((int)shift, r);
            add = k >= ex ? sh : add;
          }
          if (inside && k < tot) {
            const pp::f4 p = s_img[k + add];
            const float d = pp::chamfer_d3(p.x, p.y, p.z, w.x, w.y, w.z);
            const unsigned long long cand = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
            key = cand < key ? cand : key;  // (a NaN distance -- bits above +inf -- is never taken)
          }
        }
        key = pp::wave_min_u64_dpp(key);
        const float kbest = __uint_as_float((unsigned)(key >> 32));
        const int kidx = (int)(unsigned)key;
        const float wfx = (w.x - g.minx) * g.invh - (float)wcx, wfy = (w.y - g.miny) * g.invh - (float)wcy,
                    wfz = (w.z - g.minz) * g.invh - (float)wcz;
        auto axis1 = [&](float ff, int cc, int g1) {  // distance (cells) to the nearer face of the cube with grid beyond it
          const float lo = cc - 1 >= 1 ? 1.0f + ff : inf;
          const float hi = cc + 1 < g1 ? 2.0f - ff : inf;
          return fminf(lo, hi);
        };
        const float reach1 = g.h * fminf(axis1(wfx, wcx, gx1), fminf(axis1(wfy, wcy, gy1), axis1(wfz, wcz, gz1)));
        const bool settled = inside && kidx != 0x7fffffff && kbest < reach1 * reach1 * kBoundSlack;
        if (lane == 0) {
          s_qres[e] = settled ? 1u : 0u;
          if (settled) {
            const int wj = __float_as_int(w.w);
            (dir ? dist2 : dist1)[(size_t)b * nq + wj] = kbest;
            (dir ? idx2 : idx1)[(size_t)b * nq + wj] = kidx;
          }
        }
      }
      __syncthreads();
      if (pend && qslot < kQueue && s_qres[qslot] != 0u) pend = false;
    }
  }
  // what is left goes to the wave's slots of the pending list (in lane order: the order of the sorted cloud)
  const unsigned long long pm = __ballot(pend);
  const int wq = tile * (TQ / 64) + wave;  // this wave among the waves of the direction
  if (f.live && wq * 64 < nq) {
    if (pend) plist[wq * 64 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u))] = jj | tried;
    if (lane == 0) {
      pcnt[wq] = (unsigned)__builtin_popcountll(pm);
      if (pm) atomicAdd(&s_tot, (unsigned)__builtin_popcountll(pm));
    }
  }
  PP_APHASE(5);
  __syncthreads();  // every wave has left the image: the next tile's may be written
  // the direction's total (zeroed by the build), ONE global atomic per tile that leaves anything (a wave-by-wave
  // count -- 256 adds to one word per direction -- cost the kernel 5 us): lets the list kernel's waves leave at once
  if (t == 0 && s_tot != 0u)
    atomicAdd(reinterpret_cast<unsigned*>(ws + L.layers) + (size_t)set * pp::kLayerWords + pp::kLayerPending, s_tot);
  }  // (tiles)
  if (sv_ok) {
    (sv_dir ? dist2 : dist1)[sv_off] = sv_d;
    (sv_dir ? idx2 : idx1)[sv_off] = sv_i;
  }
}

}  // namespace

// 0 = automatic (grid when a workspace is given and the problem is large enough to pay for its two
// launches); 1 = brute force; 2 = grid wherever it is structurally possible (tests)
static pp::Knob g_grid_mode;
extern "C" void pp_debug_set_nmdistance_search(int v) { g_grid_mode.set(v); }
// LDS points per wave of the whole-search kernel (grid_query_wave_kernel): 0 = default (384); 320 / 512 for comparison
static pp::Knob g_stage_cap;
extern "C" void pp_debug_set_nmdistance_stage_cap(int v) { g_stage_cap.set(v); }
// unlabeled searches: queries per workgroup of the stage-A kernel: 0 = default (512); 256, 512, 1024; -1 = no stage-A
// kernel (round 2's two launches: the whole-search kernel serves every query)
static pp::Knob g_tile;
extern "C" void pp_debug_set_nmdistance_tile(int v) { g_tile.set(v); }
// the build of config 2's class: 0 = the LDS-sorted path where it applies (default), 1 = the general path always
static pp::Knob g_build_fast;
extern "C" void pp_debug_set_nmdistance_build(int v) { g_build_fast.set(v); }

// Per-kernel timing of the grid forward (bench.py's roofline of the dominant kernel): when switched on, HIP
// events are recorded on the launch stream before the build, between the two kernels and after the search;
// pp_debug_nmdistance_kernel_ms waits for the last one and reports the two durations of the most recent
// forward.  One set of events per process (a measurement aid for one stream at a time, not a product feature).
static pp::Knob g_time_kernels;
static std::mutex g_ev_mutex;
static hipEvent_t g_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // before the build, after it, after stage A, after the search
static bool g_ev_valid = false;
static bool g_ev_two_stage = false;
extern "C" void pp_debug_set_nmdistance_kernel_timing(int on) { g_time_kernels.set(on); }
extern "C" int pp_debug_nmdistance_kernel_ms(float* build_ms, float* search_ms) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  if (!g_ev_valid || !build_ms || !search_ms) return PP_EINVAL;
  hipError_t e = hipEventSynchronize(g_ev[3]);
  if (e == hipSuccess) e = hipEventElapsedTime(build_ms, g_ev[0], g_ev[1]);
  if (e == hipSuccess) e = hipEventElapsedTime(search_ms, g_ev[1], g_ev[3]);
  return (int)e;
}
// the search's two launches by themselves (unlabeled searches: the stage-A kernel, then the kernel that serves what it
// left); stage_a_ms = 0 when the most recent forward had no stage-A kernel
extern "C" int pp_debug_nmdistance_kernel_ms3(float* build_ms, float* stage_a_ms, float* rest_ms) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  if (!g_ev_valid || !build_ms || !stage_a_ms || !rest_ms) return PP_EINVAL;
  hipError_t e = hipEventSynchronize(g_ev[3]);
  if (e == hipSuccess) e = hipEventElapsedTime(build_ms, g_ev[0], g_ev[1]);
  *stage_a_ms = 0.0f;
  if (e == hipSuccess && g_ev_two_stage) e = hipEventElapsedTime(stage_a_ms, g_ev[1], g_ev[2]);
  if (e == hipSuccess) e = hipEventElapsedTime(rest_ms, g_ev[g_ev_two_stage ? 2 : 1], g_ev[3]);
  return (int)e;
}
// The build's and the stage-A kernel's OWN durations: with the timing knob on those two launches go through
// hipExtLaunchKernelGGL, whose start / stop events are stamped by the kernel's own begin and end -- what rocprofv3
// reports for the dispatch, free of the cost of an event recorded on the stream (bench.py's roofline.kernel_ms).
static hipEvent_t g_evk[4] = {nullptr, nullptr, nullptr, nullptr};  // build start / stop, stage A start / stop
static bool g_evk_valid = false;
static bool own_events() {
  if (!g_evk[0])
    for (int k = 0; k < 4; ++k)
      if (hipEventCreate(&g_evk[k]) != hipSuccess) return false;
  return true;
}
extern "C" int pp_debug_nmdistance_kernel_own_ms(float* build_ms, float* stage_a_ms) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  if (!g_evk_valid || !build_ms || !stage_a_ms) return PP_EINVAL;
  hipError_t e = hipEventSynchronize(g_evk[3]);
  if (e == hipSuccess) e = hipEventElapsedTime(build_ms, g_evk[0], g_evk[1]);
  if (e == hipSuccess) e = hipEventElapsedTime(stage_a_ms, g_evk[2], g_evk[3]);
  return (int)e;
}
static void record_timing_event(int i, hipStream_t s, bool two_stage = false) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  if (!g_ev[0])
    for (int k = 0; k < 4; ++k)
      if (hipEventCreate(&g_ev[k]) != hipSuccess) return;
  if (hipEventRecord(g_ev[i], s) == hipSuccess && i == 3) {
    g_ev_valid = true;
    g_ev_two_stage = two_stage;
  }
}

namespace pp {
int nmdist_forward_routed(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2, int* idx2, int B,
                          int N, int M, const unsigned* routed, int stride, hipStream_t s);  // chamfer.hip
}
// Routing to the every-pair kernel (round 6).  Whether a direction is routed is decided on the device, by the stage-A
// launch, call by call; whether the every-pair launch that serves routed directions FOLLOWS the list kernel is the
// host's decision, and the host only knows what earlier calls found: a word of pinned host memory per device that a
// stage-A launch sets to its call's number when it routes a direction.  A call issues the extra launch (a few
// microseconds when nothing is routed: its workgroups leave at once) while that word names one of the last
// kRouteMemory calls; without it the list kernel serves routed directions itself -- slower, the same bits.  So a stream
// of ordinary clouds never pays for the launch, and a stream of adversarial ones pays the slow path once.
constexpr unsigned kRouteMemory = 64;
static std::atomic<unsigned> g_route_epoch{0};
struct RouteWord {
  unsigned* host;    // pinned, mapped: the host reads it, a launch writes through `dev`
  unsigned* dev;
};
static std::atomic<RouteWord*> g_route_word[64];  // per device; allocated once, never freed
static const RouteWord* route_word(int dev) {
  if (dev < 0 || dev >= 64) return nullptr;
  RouteWord* w = g_route_word[dev].load(std::memory_order_acquire);
  if (w) return w;
  unsigned* host = nullptr;
  unsigned* devp = nullptr;
  if (hipHostMalloc((void**)&host, 64, hipHostMallocMapped) != hipSuccess || !host) return nullptr;
  *host = 0u;
  if (hipHostGetDevicePointer((void**)&devp, host, 0) != hipSuccess || !devp) {
    (void)hipHostFree(host);
    return nullptr;
  }
  RouteWord* fresh = new RouteWord{host, devp};
  RouteWord* expected = nullptr;
  if (!g_route_word[dev].compare_exchange_strong(expected, fresh, std::memory_order_acq_rel)) {
    (void)hipHostFree(host);
    delete fresh;
    return expected;
  }
  return fresh;
}
static pp::Knob g_rowbits_mode;  // 0: the group search skips the empty cell rows by the set's row bitmap; 1: it does not (A/B, tests)
extern "C" void pp_debug_set_nmdistance_row_bitmap(int v) { g_rowbits_mode.set(v); }
static pp::Knob g_route_mode;  // 0 automatic, 1 never route (the list kernel serves everything), 2 always issue the every-pair launch
extern "C" void pp_debug_set_nmdistance_routing(int mode) { g_route_mode.set(mode); }

// queries the stage-A kernel of the most recent unlabeled forward on this workspace left to the list kernel, per
// direction (2 B numbers; synchronises the device): what fraction of the search ran outside stage A
extern "C" int pp_debug_nmdistance_pending(const void* workspace, int B, int N, int M, unsigned* totals) {
  if (!workspace || !totals || B <= 0) return PP_EINVAL;
  const Layout L = make_layout(B, N, M, false);
  if (L.chunks == 0) return PP_EINVAL;
  hipError_t e = hipDeviceSynchronize();
  for (int s = 0; s < 2 * B && e == hipSuccess; ++s)  // the direction's total: added up per tile by the stage-A kernel
    e = hipMemcpy(totals + s, (const unsigned char*)workspace + L.layers + 4 * ((size_t)s * pp::kLayerWords + pp::kLayerPending), 4,
                  hipMemcpyDeviceToHost);
  return (int)e;
}

static bool grid_applicable(int B, int N, int M, int C) {
  if (!(C == 3 && B > 0 && N >= 2048 && M >= 2048 && (long long)B * ((long long)N + M) < (1LL << 31) - 1)) return false;
  // the search costs ~40 us whatever the size; the brute force evaluates ~9e6 pairs per microsecond once it
  // fills the chip and cannot fill it with a few large clouds (tools/threshold_probe.py: B=8, N=M=4096: 39
  // vs 51 us; B=4, 4096: 39 vs 37; B=32, 2048: 47 vs 48; B=1, 8192: 38 vs 62)
  return g_grid_mode == 2 || (long long)B * N * M >= 125000000LL || (N >= 8192 && M >= 8192);
}

extern "C" size_t pp_nmdistance_forward_workspace_bytes(int B, int N, int M, int C) {
  if (!grid_applicable(B, N, M, C)) return 0;
  return make_layout(B, N, M).total;
}

extern "C" size_t pp_labeled_nmdistance_forward_workspace_bytes(int B, int N, int M, int C) {
  if (!grid_applicable(B, N, M, C)) return 0;
  return make_layout(B, N, M, true).total;
}

// build -> search (stage A from LDS, wide stages and whole-cloud scans for what it leaves): two launches
template <bool LAB>
static int grid_forward(const float* xyz1, const float* xyz2, const float* label1, const float* label2,
                        float* dist1, int* idx1, float* dist2, int* idx2, int B, int N, int M,
                        unsigned char* ws, hipStream_t s) {
  hipError_t e;
  static pp::DeviceFlags lds_ok, lds_ok_vec;
  const size_t lds = std::max(pp::grid_build_lds_bytes(pp::kBuildSlabs), pp::grid_build_fast_lds_bytes());
  const bool vec = pp::clouds_vec_aligned(xyz1, N, B) && pp::clouds_vec_aligned(xyz2, M, B);
  e = vec ? pp::allow_big_lds(grid_build_kernel<true>, (int)lds, lds_ok_vec)
          : pp::allow_big_lds(grid_build_kernel<false>, (int)lds, lds_ok);
  if (e != hipSuccess) return (int)e;
  const bool timing = g_time_kernels != 0;
  if (timing) record_timing_event(0, s);
  static const int tile_env = [] {
    const char* e = getenv("PP_NMDISTANCE_TILE");
    return e ? atoi(e) : 0;
  }();
  const int tile = g_tile != 0 ? (int)g_tile : tile_env;
  const Layout lay = make_layout(B, N, M, LAB);
  const bool two_stage = !LAB && tile != -1 && lay.chunks > 0;
  // routing (see route_word above): while the host has not seen a routed direction lately, the stage-A launch's tail
  // tests every direction and tells the host; once it has, the test runs in front of the build (route_decide_kernel)
  // and the every-pair kernel serves the routed directions behind the list kernel
  unsigned* route_dev = nullptr;  // the device's view of the pinned word
  unsigned epoch = 0u;
  bool routing = false;
  if (two_stage && g_route_mode != 1) {
    int dev = 0;
    const RouteWord* rw = hipGetDevice(&dev) == hipSuccess ? route_word(dev) : nullptr;
    if (rw) {
      route_dev = rw->dev;
      epoch = g_route_epoch.fetch_add(1u, std::memory_order_relaxed) + 1u;
      if (epoch == 0u) epoch = g_route_epoch.fetch_add(1u, std::memory_order_relaxed) + 1u;  // (0 = "never")
      const unsigned seen = *reinterpret_cast<volatile unsigned*>(rw->host);
      routing = g_route_mode == 2 || (seen != 0u && epoch - seen <= kRouteMemory);
    }
  }
  const unsigned* pre_routed = nullptr;
  if (routing) {
    route_decide_kernel<<<dim3(2 * B), dim3(64), 0, s>>>(xyz1, xyz2, reinterpret_cast<unsigned*>(ws + lay.routed), B, N, M,
                                                         route_dev, epoch);
    PP_RETURN_IF_LAUNCH_FAILED();
    pre_routed = reinterpret_cast<const unsigned*>(ws + lay.routed);
    route_dev = nullptr;  // (the stage-A launch's tail has nothing to decide)
  }
  bool own = false;  // (timing: the two kernels' own begin / end stamps beside the stream events)
  if (g_time_kernels == 2) {  // (2: these instead of the stream events' figures, which the stamped launches would distort)
    std::lock_guard<std::mutex> lock(g_ev_mutex);
    own = own_events();
    g_evk_valid = false;
  }
  if (own)
    hipExtLaunchKernelGGL((vec ? grid_build_kernel<true> : grid_build_kernel<false>), dim3(8 * ((2 * B * pp::kBuildSlabs + 7) / 8)),
                          dim3(kBuildThreads), lds, s, g_evk[0], g_evk[1], 0, xyz1, xyz2, ws, B, N, M,
                          LAB ? label1 : nullptr, LAB ? label2 : nullptr, g_build_fast != 1 ? 1 : 0, pre_routed);
  else
    (vec ? grid_build_kernel<true> : grid_build_kernel<false>)<<<dim3(8 * ((2 * B * pp::kBuildSlabs + 7) / 8)), dim3(kBuildThreads), lds, s>>>(
        xyz1, xyz2, ws, B, N, M, LAB ? label1 : nullptr, LAB ? label2 : nullptr, g_build_fast != 1 ? 1 : 0, pre_routed);
  PP_RETURN_IF_LAUNCH_FAILED();
  if (timing) record_timing_event(1, s);
  const int tiles1 = (N + 255) / 256, tiles2 = (M + 255) / 256;
  const long long blocks = (long long)B * (tiles1 + tiles2);
  if (blocks > 0x7fffffffLL) return PP_EINVAL;
  const int per_xcd = (int)((blocks + 7) / 8);
  // (PP_NMDISTANCE_TILE: the debug knob's value from the environment, read once -- benchmarks of the forms in processes
  //  that do not call the knob)
  if (two_stage) {  // stage A by tiles, then the whole-search kernel over what it left (LIST)
    const int tq = tile == 256 || tile == 1024 || tile == 513 ? tile : 512;
    const int tqq = tq == 513 ? 512 : tq;
    const int ta1 = (N + tqq - 1) / tqq, ta2 = (M + tqq - 1) / tqq;
    const long long ablocks = (long long)B * (ta1 + ta2);
    const int aper = (int)((ablocks + 7) / 8);
    // persistent: as many workgroups as stay resident (CUs x workgroups per CU by the image's size), a multiple of
    // eight, at most one per tile; each walks the tiles w, w + grid, ... with the next tile's front loads in flight
    static std::atomic<int> cus{0};
    int ncu = cus.load(std::memory_order_relaxed);
    if (ncu == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          ncu <= 0)
        ncu = 256;
      cus.store(ncu, std::memory_order_relaxed);
    }
#define PP_LAUNCH_A(TQ_, CAP_, PER_CU_, WPE_)                                                                          \
  do {                                                                                                          \
    long long g_ = (long long)ncu * (PER_CU_);                                                                  \
    g_ = (g_ < (long long)aper * 8 ? g_ : (long long)aper * 8);                                                 \
    g_ = (g_ + 7) / 8 * 8;                                                                                      \
    if ((PER_CU_) >= 1024) g_ += (2 * B + 7) / 8 * 8;  /* a workgroup per direction in front of the tiles': routing test, row bitmap */ \
    if (own) {                                                                                                  \
      hipExtLaunchKernelGGL((grid_stage_a_kernel<TQ_, CAP_, WPE_, ((PER_CU_) < 1024)>), dim3((unsigned)g_), dim3(TQ_), 0, s, \
                            g_evk[2], g_evk[3], 0, dist1, idx1, dist2, idx2, ws, B, N, M, ta1, ta2, (int)ablocks, aper, lay, \
                            route_dev, epoch, pre_routed);                                                        \
      std::lock_guard<std::mutex> lock(g_ev_mutex);                                                              \
      g_evk_valid = true;                                                                                        \
    } else                                                                                                       \
      grid_stage_a_kernel<TQ_, CAP_, WPE_, ((PER_CU_) < 1024)><<<dim3((unsigned)g_), dim3(TQ_), 0, s>>>(dist1, idx1, dist2, idx2, ws, B, N, M, ta1, \
                                                                           ta2, (int)ablocks, aper, lay, route_dev, epoch, pre_routed); \
  } while (0)
    switch (tq) {
      // (workgroups per CU: 1 << 20 = a workgroup per tile, not persistent -- measured as fast at config 2 (the front of
      //  a tile is hidden by the other workgroups of the CU either way) and free of the loop's register pressure;
      //  513: the persistent form, two workgroups per CU at 96 registers, kept for comparison)
      case 256: PP_LAUNCH_A(256, 2044, 1 << 20, 4); break;
      case 1024: PP_LAUNCH_A(1024, 4032, 1 << 20, 8); break;
      case 513: PP_LAUNCH_A(512, 3068, 2, 5); break;
      default: PP_LAUNCH_A(512, 3260, 1 << 20, 6); break;
    }
#undef PP_LAUNCH_A
    PP_RETURN_IF_LAUNCH_FAILED();
    if (timing) record_timing_event(2, s);
  }
#define PP_LAUNCH_W(CAP_)                                                                                  \
  grid_query_wave_kernel<LAB, CAP_><<<dim3((unsigned)(per_xcd * 8)), dim3(256), 0, s>>>(                       \
      xyz1, xyz2, dist1, idx1, dist2, idx2, ws, B, N, M, tiles1, tiles2, (int)blocks, per_xcd, label1, label2)
  if (two_stage) {
    // what stage A left: a wave for every 64 queries of a direction (as many as the whole-search kernel has, for the
    // clouds stage A cannot serve), every wave by itself
    const int sets = 2 * B;
    const int wps = ((N > M ? N : M) + 63) / 64;  // a wave for every 64 queries of a direction: as the whole-search kernel
    const long long lwaves = (long long)sets * wps;
    // (the row bitmaps are written by the tail workgroups of the stage-A launch: the non-persistent forms)
    const unsigned* rowbits = (tile != 513 && g_rowbits_mode != 1) ? reinterpret_cast<const unsigned*>(ws + lay.rowbits) : nullptr;
    grid_query_list_kernel<384><<<dim3((unsigned)(((lwaves + kListWgWaves - 1) / kListWgWaves + 7) / 8 * 8)), dim3(64 * kListWgWaves), 0, s>>>(xyz1, xyz2, dist1, idx1, dist2, idx2,
                                                                                      ws, B, N, M, wps, lay, pre_routed, rowbits);
    if (routing) {
      PP_RETURN_IF_LAUNCH_FAILED();
      const int rc = pp::nmdist_forward_routed(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, pre_routed, 1, s);
      if (rc != PP_OK) return rc;
    }
  } else {
    switch (g_stage_cap) {
      case 320: PP_LAUNCH_W(320); break;
      case 512: PP_LAUNCH_W(512); break;
      default: PP_LAUNCH_W(384); break;
    }
  }
#undef PP_LAUNCH_W
  PP_RETURN_IF_LAUNCH_FAILED();
  if (timing) record_timing_event(3, s, two_stage);
  return PP_OK;
}

extern "C" int pp_nmdistance_forward_ws_f32(const float* xyz1, const float* xyz2, float* dist1,
                                            int* idx1, float* dist2, int* idx2, int B, int N, int M,
                                            int C, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  const size_t need = pp_nmdistance_forward_workspace_bytes(B, N, M, C);
  if (g_grid_mode == 1 || need == 0 || !workspace || workspace_bytes < need)
    return pp_nmdistance_forward_f32(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, stream);
  if (!xyz1 || !xyz2 || !dist1 || !idx1 || !dist2 || !idx2) return PP_EINVAL;
  return grid_forward<false>(xyz1, xyz2, nullptr, nullptr, dist1, idx1, dist2, idx2, B, N, M,
                             (unsigned char*)workspace, (hipStream_t)stream);
}

extern "C" int pp_labeled_nmdistance_forward_ws_f32(const float* xyz1, const float* xyz2, const float* label1,
                                                    const float* label2, float* dist1, int* idx1, float* dist2,
                                                    int* idx2, int B, int N, int M, int C, void* workspace,
                                                    size_t workspace_bytes, void* stream) {
  const size_t need = pp_labeled_nmdistance_forward_workspace_bytes(B, N, M, C);
  if (g_grid_mode == 1 || need == 0 || !workspace || workspace_bytes < need)
    return pp_labeled_nmdistance_forward_f32(xyz1, xyz2, label1, label2, dist1, idx1, dist2, idx2, B, N, M, C,
                                             stream);
  if (!xyz1 || !xyz2 || !label1 || !label2 || !dist1 || !idx1 || !dist2 || !idx2) return PP_EINVAL;
  return grid_forward<true>(xyz1, xyz2, label1, label2, dist1, idx1, dist2, idx2, B, N, M,
                            (unsigned char*)workspace, (hipStream_t)stream);
}
