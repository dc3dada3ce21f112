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
#include <mutex>

#include "grid_common.h"

#ifdef PP_QUERY_PROBE
// diagnostic build only (tools/query_probe.py): 100 MHz clock at the phase boundaries of a few workgroups, and
// what every wave spent in each phase (10 ns units; the first kQWaves waves of the launch)
constexpr int kQWaves = 1 << 17;
__device__ unsigned long long g_qphase[8][16];
__device__ unsigned g_qwave[kQWaves][10];
__device__ unsigned long long g_qgroup[8];  // group search: calls, groups, blind groups, candidates staged, max candidates of one call, rows visited
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
#define PP_QPHASE_DECL unsigned long long pp_prev = wall_clock64()
#define PP_QPHASE(n)                                                                     \
  do {                                                                                   \
    const unsigned long long pp_now = wall_clock64();                                    \
    const unsigned pp_w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                          \
    if ((threadIdx.x & 63) == 0 && pp_w < (unsigned)kQWaves) g_qwave[pp_w][n] = (unsigned)(pp_now - pp_prev); \
    pp_prev = pp_now;                                                                    \
    if (threadIdx.x == 0 && (blockIdx.x & 511) == 0 && (blockIdx.x >> 9) < 8)            \
      g_qphase[blockIdx.x >> 9][n] = pp_now;                                             \
  } while (0)
#else
#define PP_QPHASE_DECL
#define PP_QPHASE(n)
#endif

namespace {

using pp::GridSet;
using pp::cell_coord;
using pp::kGridCells;
using pp::kGridMax;
using pp::kBuildThreads;

constexpr float kBoundSlack = 0.999f;

// Workspace layout (bytes), S = 2*B sets, T = B*(N+M) points:
//   [0, 64*S)                      GridSet[S]
//   [.., +4*(kGridCells+1)*S)      unsigned cell_start[S][kGridCells+1]
//   [.., +16*T)                    float4 sorted[T]   (x, y, z, original index bits)
//   [.., +4*(2*T + 2*S))           unsigned sub_start[...]   second level: cell tables of the crowded cells,
//                                  the table of the cell whose points start at `start` of set s at 2*(set offset + start) + 2*s
//   [.., +32*(T/kCrowd + 2*S))     SubGrid sub_desc[...]     their descriptors (grid_common.h)
//   [.., +16*T)                    float4 sorted2[T]  spare copy the refinement sorts through
//   [.., +4*T, +4*T)               float slab[T], slab2[T]   labels in sorted order + spare (labeled Chamfer only)
// (the second-level arrays are only touched for sets that have crowded cells: never at config 2)
//   [.., +4 * S * kBuildSlabs * 2 * chunks)  int tile_z[S][kBuildSlabs][chunks][2]   chunk table (grid_common.h: kChunk)
struct Layout {
  size_t sets, cell_start, sorted, sub_start, sub_desc, sorted2, slab, slab2, tile_z, total;
  int chunks;  // chunk-table entries per set and slab (0: sets too large for the table)
};
__host__ __device__ inline Layout make_layout(int B, int N, int M, bool labeled = false) {
  Layout L;
  const size_t S = (size_t)2 * B, T = (size_t)B * ((size_t)N + M);
  L.sets = 0;
  L.cell_start = L.sets + ((64 * S + 255) / 256) * 256;
  L.sorted = L.cell_start + ((4 * (size_t)(kGridCells + 1) * S + 255) / 256) * 256;
  L.sub_start = L.sorted + 16 * T;
  L.sub_desc = L.sub_start + ((4 * (2 * T + 2 * S) + 255) / 256) * 256;
  L.sorted2 = L.sub_desc + ((32 * (T / pp::kCrowd + 2 * S) + 255) / 256) * 256;
  L.slab = L.sorted2 + 16 * T;
  L.slab2 = L.slab + (labeled ? 4 * T : 0);
  L.tile_z = ((L.slab2 + (labeled ? 4 * T : 0) + 255) / 256) * 256;
  const int chq = ((N > M ? N : M) + pp::kChunk - 1) / pp::kChunk;
  L.chunks = chq <= pp::kChunkMax ? chq : 0;
  L.total = L.tile_z + 4 * S * pp::kBuildSlabs * 2 * (size_t)L.chunks;
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
// set s = 2*b + dir; dir 0: queries = cloud 1 (N), references = cloud 2 (M)
__host__ __device__ inline size_t set_point_offset(int b, int dir, int N, int M) {
  return (size_t)b * ((size_t)N + M) + (dir ? (size_t)M : 0);  // references of (b,0) first (M), then (b,1) (N)
}

// kBuildSlabs workgroups per set: bounding box, cell histogram (LDS), exclusive scan, scatter
// (grid_common.h).
template <bool VEC>
__global__ __launch_bounds__(kBuildThreads) void grid_build_kernel(const float* __restrict__ xyz1,
                                                                   const float* __restrict__ xyz2,
                                                                   unsigned char* __restrict__ ws, int B,
                                                                   int N, int M,
                                                                   const float* __restrict__ label1,
                                                                   const float* __restrict__ label2) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_cnt[];  // pp::grid_build_lds_bytes(kBuildSlabs)
  // a set is built on the XCD that will search it (the search kernel's set -> XCD mapping): its sorted
  // points and cell table are then already in that L2
  const int V = pp::xcd_virtual_block(blockIdx.x, (2 * B * pp::kBuildSlabs + 7) / 8);
  if (V >= 2 * B * pp::kBuildSlabs) return;
  const int set = V / pp::kBuildSlabs, slab = V % pp::kBuildSlabs;
  const int b = set >> 1, dir = set & 1;
  const int nr = dir ? N : M;
  const float* __restrict__ ref = (dir ? xyz1 : xyz2) + (size_t)b * nr * 3;
  const bool labeled = label1 != nullptr;
  const Layout L = make_layout(B, N, M, labeled);
  const float* __restrict__ lab = labeled ? (dir ? label1 : label2) + (size_t)b * nr : nullptr;
  pp::grid_build_set_refined<VEC>(
      ref, nr, reinterpret_cast<GridSet*>(ws + L.sets) + set,
      reinterpret_cast<unsigned*>(ws + L.cell_start) + (size_t)set * (kGridCells + 1),
      reinterpret_cast<pp::f4*>(ws + L.sorted) + set_point_offset(b, dir, N, M), s_cnt, lab,
      labeled ? reinterpret_cast<float*>(ws + L.slab) + set_point_offset(b, dir, N, M) : nullptr, slab, pp::kBuildSlabs,
      reinterpret_cast<unsigned*>(ws + L.sub_start) + set_sub_start_offset(b, dir, N, M),
      reinterpret_cast<pp::SubGrid*>(ws + L.sub_desc) + set_sub_desc_offset(b, dir, N, M),
      reinterpret_cast<pp::f4*>(ws + L.sorted2) + set_point_offset(b, dir, N, M),
      labeled ? reinterpret_cast<float*>(ws + L.slab2) + set_point_offset(b, dir, N, M) : nullptr,
      L.chunks ? reinterpret_cast<int*>(ws + L.tile_z) + (size_t)set * pp::kBuildSlabs * 2 * L.chunks : nullptr, L.chunks);
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
  unsigned incl = len;  // nrows <= 32: five steps
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const unsigned o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 31);
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
  // wave-wide minimum
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)key, off), hi = __shfl_xor((unsigned)(key >> 32), off);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    key = o < key ? o : key;
  }
  return key;
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
constexpr int kSerialMax = 24;  // open lanes of a wave from which the whole-wave cubes are skipped for the group search
constexpr int kLaneStageMin = 6;  // open lanes of a wave from which the cubes are searched a lane per query

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
template <bool LAB>
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
template <bool LAB>
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

template <bool LAB>
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
template <bool LAB>
__device__ __attribute__((noinline)) Found wave_group_search(const GridSet g, const unsigned* __restrict__ cell_start,
                                                             const pp::f4* __restrict__ sorted,
                                                             const float* __restrict__ slab, float qx, float qy,
                                                             float qz, float ql, float best, int bidx, unsigned open_lo,
                                                             unsigned open_hi, lds_f4_wptr lw, lds_f_wptr lwl) {
  const lds_f4_ptr lr = (lds_f4_ptr)lw;
  const lds_f_ptr lrl = (lds_f_ptr)lwl;
  const int lane = threadIdx.x & 63;
  const float inf = __builtin_inff();
  unsigned long long open = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)open_hi) << 32) |
                            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)open_lo);
  auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
  // (a point may sit in the neighbouring cell by the rounding of its cell coordinate, and the faces themselves
  //  are rounded: every gap is shortened by this much)
  const float slack = 1.0e-4f * g.h;
  // distance from the interval [blo, bhi] to the slab of cell c along one axis (rim cells hold the outliers: they
  // extend to infinity)
  auto axis_gap = [&](float blo, float bhi, float mn, int c, int gdim) {
    const float lo = c == 0 ? -inf : mn + (float)c * g.h, hi = c == gdim - 1 ? inf : mn + (float)(c + 1) * g.h;
    return fmaxf(fmaxf(lo - bhi, blo - hi) - slack, 0.0f);
  };
#ifdef PP_QUERY_PROBE
  unsigned long long pp_ngroups = 0, pp_nblind = 0, pp_ncand = 0, pp_nrows = 0;
#endif
  while (open) {
    const int seed = (int)__builtin_ctzll(open);
    const float sx = rl(qx, seed), sy = rl(qy, seed), sz = rl(qz, seed);
    const float sl = LAB ? rl(ql, seed) : 0.0f;
    float us = rl(best, seed);
    // wave-uniform: the seed has seen no candidate yet, or only one picked up by accident far outside its cubes
    const bool blind = !(us < 16.0f * g.h * g.h);
    if (blind) {
      // one sample per non-empty cell row -- the first point at or after the seed's cell along x, else the row's last
      // point -- four chunks of rows in flight; the nearest sample bounds the seed's neighbour
      const int nall = g.gy * g.gz;
      const int cxs = cell_coord(sx, g.minx, g.invh, g.gx);
      float u1 = inf;
      for (int r0 = 0; r0 < nall; r0 += 256) {
        unsigned rs[4], rm[4], re[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + u * 64 + lane;
          const bool ok = r < nall;
          const int base = (ok ? r : 0) * g.gx;
          rs[u] = cell_start[base];
          rm[u] = cell_start[base + cxs];
          re[u] = ok ? cell_start[base + g.gx] : rs[u];
        }
        pp::f4 smp[4];
        float sml[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned at = re[u] > rs[u] ? min(rm[u], re[u] - 1) : 0u;
          smp[u] = sorted[at];
          if (LAB) sml[u] = slab[at];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = pp::chamfer_d3(smp[u].x, smp[u].y, smp[u].z, sx, sy, sz);
          u1 = (re[u] > rs[u] && (!LAB || sml[u] == sl) && d < u1) ? d : u1;
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) u1 = fminf(u1, __shfl_xor(u1, off));
      us = fminf(us, u1 * 1.0001f);
    }
    // The seed takes along the open queries within r of it, r = a quarter of the distance of its candidate.  Where
    // the seed knew a candidate: at least two cells, and only queries whose own candidate is no more than twice as
    // far (the group's bound is the largest of them).  Where it did not (the bound is the sample's, a crude one):
    // any open query within r -- it has a neighbour within that distance + sqrt(3) r through the seed.
#ifdef PP_QUERY_PROBE
    ++pp_ngroups;
    pp_nblind += blind ? 1 : 0;
#endif
    const float ds = sqrtf(us);
    const float r = blind ? 0.25f * ds : fmaxf(2.0f * g.h, 0.25f * ds);
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
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ub = fmaxf(ub, __shfl_xor(ub, off));
    const float U = ub * 1.0001f;
    // the rows within sqrt(U) of the box (cell coordinates are monotonic in the coordinate: exact)
    const bool bounded = U < inf;
    const float R = bounded ? sqrtf(U) * 1.0001f + slack : 0.0f;
    const int y0 = bounded ? cell_coord(bly - R, g.miny, g.invh, g.gy) : 0;
    const int y1 = bounded ? cell_coord(bhy + R, g.miny, g.invh, g.gy) : g.gy - 1;
    const int z0 = bounded ? cell_coord(blz - R, g.minz, g.invh, g.gz) : 0;
    const int z1 = bounded ? cell_coord(bhz + R, g.minz, g.invh, g.gz) : g.gz - 1;
    const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
    const float inv_ny = 1.0f / (float)ny;
#ifdef PP_QUERY_PROBE
    pp_nrows += (unsigned long long)nrows;
#endif
    for (int r0 = 0; r0 < nrows; r0 += 64) {  // wave-uniform
      const int rr = r0 + lane;
      // (rr < 2^11, ny < 2^6: the rounded product is the exact quotient)
      const int dz = (int)(((float)rr + 0.5f) * inv_ny);
      const int cz = z0 + dz, cy = y0 + (rr - dz * ny);
      unsigned cs = 0u, len = 0u;
      if (rr < nrows) {
        const float gy_ = axis_gap(bly, bhy, g.miny, cy, g.gy), gz_ = axis_gap(blz, bhz, g.minz, cz, g.gz);
        const float rem = U - __builtin_fmaf(gz_, gz_, gy_ * gy_) * 0.9999f;  // budget left along x (+inf if unbounded)
        if (rem >= 0.0f) {
          const float rx = bounded ? sqrtf(rem) * 1.0001f + slack : 0.0f;
          const int x0 = bounded ? cell_coord(blx - rx, g.minx, g.invh, g.gx) : 0;
          const int x1 = bounded ? cell_coord(bhx + rx, g.minx, g.invh, g.gx) : g.gx - 1;
          const int base = pp::cell_linear(0, cy, cz, g.gx, g.gy);
          cs = cell_start[base + x0];
          len = cell_start[base + x1 + 1] - cs;
        }
      }
      if (__ballot(len != 0u) == 0ull) continue;  // wave-uniform: nothing in these rows
      unsigned incl = len;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
      const unsigned excl = incl - len;
      const unsigned shift = cs - excl;  // candidate c of this lane's row sits at sorted[c + shift]
#ifdef PP_QUERY_PROBE
      pp_ncand += total;
#endif
      for (unsigned t0 = 0; t0 < total; t0 += kGroupBatch) {  // wave-uniform
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
#pragma unroll
        for (int u = 0; u < kGroupBatch / 64; ++u) {
          lw[u * 64 + lane] = pt[u];
          if (LAB) lwl[u * 64 + lane] = ptl[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned n = min((unsigned)kGroupBatch, (total - t0 + 3u) & ~3u);
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
    }
  }
#ifdef PP_QUERY_PROBE
  if (lane == 0) {
    atomicAdd(&g_qgroup[0], 1ull);
    atomicAdd(&g_qgroup[1], pp_ngroups);
    atomicAdd(&g_qgroup[2], pp_nblind);
    atomicAdd(&g_qgroup[3], pp_ncand);
    atomicMax(&g_qgroup[4], pp_ncand);
    atomicAdd(&g_qgroup[5], pp_nrows);
    atomicMax(&g_qgroup[6], pp_ngroups);
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

template <bool LAB>
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

// byte position in the tile's image of group k of a lane's sequence (lean front; everything by value)
__device__ __forceinline__ unsigned lean_group_pos(unsigned k, unsigned T1, unsigned T2, unsigned T3, unsigned a0,
                                                   unsigned a1, unsigned a2, unsigned a3, unsigned endb) {
  const unsigned a = k < T1 ? a0 : (k < T2 ? a1 : (k < T3 ? a2 : a3));
  return min(a + (k << 6), endb);
}

// Two wave-wide max reductions at once, in place, through DPP (see pp::wave_reduce6_dpp; two interleaved chains need
// one more wait state between dependent DPP operations).  ROWS: only inside every row of 16 lanes (lane 15 of a row
// holds that row's result); otherwise lane 63 holds the wave's.  Every lane must be active.
#define PP_DPP2_STEP(CTRL) \
  "v_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
template <bool ROWS>
__device__ __forceinline__ void wave_max2_dpp(float& a, float& b) {
  if constexpr (ROWS)
    asm volatile("s_nop 1\n\t" PP_DPP2_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") PP_DPP2_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                     PP_DPP2_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") PP_DPP2_STEP("row_shr:8 row_mask:0xf bank_mask:0xf") "s_nop 0"
                 : "+v"(a), "+v"(b));
  else
    asm volatile("s_nop 1\n\t" PP_DPP2_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") PP_DPP2_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                     PP_DPP2_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") PP_DPP2_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                         PP_DPP2_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") PP_DPP2_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0"
                 : "+v"(a), "+v"(b));
}
typedef const char __attribute__((address_space(3))) * lds_c_ptr;

// TQ = 0: the form above (256-thread workgroups of four independent waves, a wave-private region of CAPW points).
// TQ > 0 (round 3, the default): a workgroup is a TILE of TQ consecutive queries; it stages the WHOLE z-layers of
// the reference grid that its queries' blocks touch -- one contiguous piece of the sorted cloud, found with one
// min / max reduction and two scalar loads instead of a per-wave region of row ranges per layer -- into ONE LDS
// image of CAPW points shared by its waves (two workgroup barriers).  Everything after the copy is the same code.
template <bool LAB, int CAPW, int TQ>
__global__ __launch_bounds__(TQ ? TQ : 256, LAB && TQ ? (TQ == 768 ? 3 : 4) : 6) void grid_query_wave_kernel(const float* __restrict__ xyz1,
                                                              const float* __restrict__ xyz2,
                                                              float* __restrict__ dist1, int* __restrict__ idx1,
                                                              float* __restrict__ dist2, int* __restrict__ idx2,
                                                              unsigned char* __restrict__ ws, int B, int N, int M,
                                                              int tiles1, int tiles2, int total, int per_xcd,
                                                              const float* __restrict__ label1,
                                                              const float* __restrict__ label2) {
  const int V = pp::xcd_virtual_block(blockIdx.x, per_xcd);
  if (V >= total) return;
  PP_QPHASE_DECL;
  PP_QPHASE(0);
  constexpr int kT = TQ ? TQ : 256;  // threads = queries per workgroup
  constexpr int kW = kT / 64;
  const int per_b = tiles1 + tiles2;
  const int b = V / per_b;
  const int r = V - b * per_b;
  const int dir = r >= tiles1 ? 1 : 0;
  const int tile = dir ? r - tiles1 : r;
  const int nq = dir ? M : N, nr = dir ? N : M;
  const int t = threadIdx.x, lane = t & 63;
  const bool valid = tile * kT + t < nq;
  const int jj = valid ? tile * kT + t : nq - 1;
  const int set = 2 * b + dir;
  const Layout L = make_layout(B, N, M, LAB);
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
      reinterpret_cast<const unsigned*>(ws + L.cell_start) + (size_t)set * (kGridCells + 1);
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
  constexpr int kSlices = TQ ? 1 : 4;
  __shared__ pp::f4 s_pts[kSlices][CAPW + 4];
  __shared__ float s_lab[kSlices][LAB ? CAPW + 4 : 1];
  __shared__ float s_zr[TQ ? 2 * kW : 1];  // tile mode: every wave's (-lowest, highest) layer
  __shared__ unsigned s_left;              // tile mode: waves that have finished with the image (see the group search)
  if (TQ != 0 && t == 0) s_left = 0u;      // (ordered before every use by the first workgroup barrier)
  static_assert(!TQ || CAPW + 4 >= kW * kGroupBatch, "the group search takes a slice of the image per wave");
  const int wave = pp::wave_id_uniform();
  const int slice = TQ ? 0 : wave;

  bool deferred = false;
  float best = __builtin_inff();
  int bidx = 0x7fffffff;
  float thr = 0.0f;
  bool lean_done = false;
  if constexpr (TQ != 0 && !LAB) {
    // (uniform over the set) no crowded cells on either side, and the query cloud's chunk table exists: every evenly
    // sampled cloud of up to 65536 points
    if (!refined_set && !gp_useless && !pp::grid_refined(gp) && gp.pad[1] == 1 && L.chunks > 0) {
      // ---- the lean front (round 3): the same search as the general front below, for the case that decides the
      // benchmark -- a tile of an unlabeled set without second-level grids -- written for VALU issue and for a short
      // chain of dependent loads: the layers the tile can touch come from the QUERY cloud's chunk table (two scalar
      // loads per build slab and chunk: no reduction over the tile, no barrier before the image is ordered), the image
      // is copied while the lanes' own queries and row bounds are still on their way, and ONE barrier separates the
      // copy from the walk; one chain of cell arithmetic (no second pass for the reach), 32-bit offsets from
      // wave-uniform bases, a walk that tracks the byte position of the winning group.
      const float inf = __builtin_inff();
      const int gx1 = __builtin_amdgcn_readfirstlane(g.gx - 1), gy1 = __builtin_amdgcn_readfirstlane(g.gy - 1),
                gz1 = __builtin_amdgcn_readfirstlane(g.gz - 1);
      // the tile's z range (world coordinates) -> the layers of the reference grid its blocks can touch: a block holds
      // the query's layer and one neighbour, and pp::cell_coord is monotone in z
      int Lz, Hz;
      {
        const int* __restrict__ tzq = reinterpret_cast<const int*>(ws + L.tile_z) + (size_t)(set ^ 1) * pp::kBuildSlabs * 2 * L.chunks;
        const int c0 = tile * (kT / pp::kChunk), cend = min(c0 + kT / pp::kChunk, (nq + pp::kChunk - 1) / pp::kChunk);
        int kmin = 0x7fffffff, kmax = (int)0x80000000;
        for (int sl = 0; sl < pp::kBuildSlabs; ++sl)
          for (int c = c0; c < cend; ++c) {  // (uniform addresses: scalar loads)
            kmin = min(kmin, tzq[(sl * L.chunks + c) * 2]);
            kmax = max(kmax, tzq[(sl * L.chunks + c) * 2 + 1]);
          }
        Lz = max(__builtin_amdgcn_readfirstlane(cell_coord(pp::zkey_inv(kmin), g.minz, g.invh, g.gz)) - 1, 0);
        Hz = min(__builtin_amdgcn_readfirstlane(cell_coord(pp::zkey_inv(kmax), g.minz, g.invh, g.gz)) + 1, gz1);
      }
      const unsigned layer = (unsigned)g.gx * (unsigned)g.gy;
      const unsigned tb0 = cell_start[__builtin_amdgcn_readfirstlane((int)((unsigned)Lz * layer))];
      const unsigned ns = cell_start[__builtin_amdgcn_readfirstlane((int)((unsigned)(Hz + 1) * layer))] - tb0;
      if (ns > 0u && ns <= (unsigned)CAPW) {  // workgroup-uniform (else: the general front, which walks global memory)
        // the image: [tb0, tb0 + ns) of the sorted cloud; the first pieces are ordered here, before anything waits
        const pp::f4* __restrict__ src = sorted + tb0;
        pp::f4 cpy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cpy[u] = src[min((unsigned)(u * kT + t), ns - 1)];  // (duplicates store the same value)
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
        // the bounds of the block's four rows (y, z): one 12-byte load each (see the general front)
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
        PP_QPHASE(1);
        {
#pragma unroll
          for (int u = 0; u < 4; ++u) (&s_pts[0][0])[min((unsigned)(u * kT + t), ns - 1)] = cpy[u];
          for (unsigned p0 = 4 * kT; p0 < ns; p0 += 4 * kT) {  // (images of more than 4 kT points: none at CAPW <= 4 kT)
            pp::f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[min(p0 + (unsigned)(u * kT + t), ns - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) (&s_pts[0][0])[min(p0 + (unsigned)(u * kT + t), ns - 1)] = v[u];
          }
          // the padding: four points that can never be taken (their distance is NaN); lanes whose rows are finished, and
          // groups that run past the end of the image, land here
          if (t < 4) {
            const float qn = __builtin_nanf("");
            const pp::f4 nanp = {qn, qn, qn, __int_as_float(0x7fffffff)};
            (&s_pts[0][0])[ns + t] = nanp;
          }
        }
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
        PP_QPHASE(2);
        __syncthreads();
        PP_QPHASE(3);
        const lds_c_ptr lb = (lds_c_ptr)(&s_pts[0][0]);
        // (by value, through a function: selects between variables a lambda captures by reference come out of hipcc as
        //  indexed loads from a pointer table in scratch memory)
        auto pos_of = [=](unsigned k) { return lean_group_pos(k, T1, T2, T3, a0, a1, a2, a3, endb); };
        pp::f4 pa[4], pb[4];
        auto fetch4 = [&](unsigned pos, pp::f4 (&p)[4]) {
#pragma unroll
          for (int u = 0; u < 4; ++u) p[u] = *(lds_f4_ptr)(lb + pos + 16 * u);
        };
        unsigned gpos = endb;  // byte position of the group that holds the winner
        unsigned long long tie = 0ull;  // lanes that saw a distance equal to their running minimum in a later group
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
        thr = reach * reach * kBoundSlack;
        lean_done = true;
        PP_QPHASE(4);
      }
    }
  }
  if (!lean_done) {  // ---- the general front: labeled searches, sets with crowded cells, the wave-private form, tiles that do not fit
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
  PP_QPHASE(1);
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
      unsigned incl = len;
#pragma unroll
      for (int off = 1; off < kStageLayers; off <<= 1) {
        const unsigned o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      offv = incl - len;
      delta = gs - offv;
      n_staged = (unsigned)__builtin_amdgcn_readlane((int)incl, kStageLayers - 1);
      staged = n_staged <= (unsigned)CAPW;
    }
  };
  // Tile mode: the layers the workgroup's blocks touch, [Lz, Lz + nz), are one contiguous piece of the sorted cloud
  // (cells are z-major): `tbase` = its first point.  Every lane's rows lie inside it by construction.
  unsigned tbase = 0;
  auto region_tile = [&](const bool dfr) {
    float v[6] = {dfr ? ninf : -(float)z0, dfr ? ninf : (float)z1, ninf, ninf, ninf, ninf};
    pp::wave_reduce6_dpp<false, 6>(v);
    if (lane == 63) {
      s_zr[2 * wave] = v[0];
      s_zr[2 * wave + 1] = v[1];
    }
    __syncthreads();
    float w[6] = {lane < kW ? s_zr[2 * (lane < kW ? lane : 0)] : ninf, lane < kW ? s_zr[2 * (lane < kW ? lane : 0) + 1] : ninf,
                  ninf, ninf, ninf, ninf};
    pp::wave_reduce6_dpp<false, 4>(w);  // kW <= 16 values in lanes 0..15: lane 15 holds the result
    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w[0]), 15));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w[1]), 15));
    staged = hi >= 0.0f;  // (no lane takes part: nothing to stage)
    Lz = staged ? -(int)lo : 0;
    nz = staged ? (int)hi - Lz + 1 : 0;
    if (staged) {
      const int c0 = __builtin_amdgcn_readfirstlane(Lz * g.gy * g.gx);
      const int c1 = __builtin_amdgcn_readfirstlane((Lz + nz) * g.gy * g.gx);
      tbase = cell_start[c0];
      n_staged = cell_start[c1] - tbase;
      staged = n_staged <= (unsigned)CAPW;
    }
  };
  deferred = false;
  if (!refined_set) {  // wave-uniform
    if constexpr (TQ != 0) region_tile(false); else region(false, false);
    rows_finish();
  } else {
    rows_finish();
    deferred = crowd;
    if (deferred) {
      re0 = rs0; re1 = rs1; re2 = rs2; re3 = rs3;
    }
    if constexpr (TQ != 0) region_tile(deferred); else region(deferred, !__any(!deferred));
  }
  PP_QPHASE(2);
  const lds_f4_ptr lpts = (lds_f4_ptr)(&s_pts[slice][0]);
  const lds_f_ptr llab = (lds_f_ptr)(&s_lab[slice][0]);
  if constexpr (TQ != 0) {
    if (staged) {  // workgroup-uniform
      rs0 -= tbase; re0 -= tbase; rs1 -= tbase; re1 -= tbase;
      rs2 -= tbase; re2 -= tbase; rs3 -= tbase; re3 -= tbase;
      for (unsigned p0 = 0; p0 < n_staged; p0 += 4 * kT) {
        pp::f4 v[4];
        float vl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned p = min(p0 + (unsigned)(u * kT + t), n_staged - 1);  // (duplicates store the same value)
          v[u] = sorted[p + tbase];
          if (LAB) vl[u] = slab[p + tbase];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned p = min(p0 + (unsigned)(u * kT + t), n_staged - 1);
          (&s_pts[0][0])[p] = v[u];
          if (LAB) (&s_lab[0][0])[p] = vl[u];
        }
      }
      if (n_staged > 0 && t < 4) {  // the padding repeats the last point (a real candidate)
        (&s_pts[0][0])[n_staged + t] = sorted[tbase + n_staged - 1];
        if (LAB) (&s_lab[0][0])[n_staged + t] = slab[tbase + n_staged - 1];
      }
    }
    __syncthreads();
  } else if (staged) {
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
          (&s_pts[wave][0])[p] = v[u];
          if (LAB) (&s_lab[wave][0])[p] = vl[u];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (n_staged > 0) {  // the padding repeats the last point (a real candidate)
      const pp::f4 padv = lpts[n_staged - 1];
      const float padl = LAB ? llab[n_staged - 1] : 0.0f;
      if (lane < 4) {
        (&s_pts[wave][0])[n_staged + lane] = padv;
        if (LAB) (&s_lab[wave][0])[n_staged + lane] = padl;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  PP_QPHASE(3);

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
  PP_QPHASE(4);
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
      const Found f = refined_block_search<LAB>(g, cell_start, sorted, slab, sub_start, sub_desc, qx, qy, qz, ql, cx2, cy2,
                                                cz2, sx2, sy2, sz2, reach);
      best = f.best;
      bidx = f.bidx;
      thr = f.aux;
    }
  }
  }  // (the general front)
  if constexpr (TQ != 0) {  // this wave reads the image no more
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&s_left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  PP_QPHASE(5);
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
  if (__builtin_popcountll(__ballot(pend && !deferred)) >= kLaneStageMin) {
    const bool mine = pend && !deferred;
    Found f = lane_cube_search<LAB>(g, cell_start, sorted, slab, qx, qy, qz, ql, 1, mine, best, bidx);
    best = f.best;
    bidx = f.bidx;
    if (f.aux == 1.0f) {
      od[j] = best;
      oi[j] = bidx;
      pend = false;
    }
    const bool mine2 = pend && !deferred && f.aux != 2.0f;
    if (__builtin_popcountll(__ballot(mine2)) >= kLaneStageMin) {
      f = lane_cube_search<LAB>(g, cell_start, sorted, slab, qx, qy, qz, ql, 2, mine2, best, bidx);
      best = f.best;
      bidx = f.bidx;
      if (f.aux == 1.0f) {
        od[j] = best;
        oi[j] = bidx;
        pend = false;
      }
      open_lane = pend && mine2 && f.aux == 0.0f;  // radius 2 examined in full and not enough
      pend = pend && !open_lane;
    }
  }
  PP_QPHASE(6);
  // ---- few lanes: wide stages by the whole wave; then the whole cloud ---------------------------------------
  unsigned long long pending = __ballot(pend);
  unsigned long long open = __ballot(open_lane);  // lanes whose query the cube of radius 2 could not settle
  if (__builtin_popcountll(pending) >= kSerialMax) {
    // many (next to crowded cells, typically: every one of them would scan those cells by itself, ~ 0.8 wave
    // instructions per candidate and query against ~ 13 per candidate for all of them in the group search)
    open |= pending;
    pending = 0ull;
  }
  if (pending) {  // wave-uniform
    const OpenMask om = serve_pending<LAB>(reinterpret_cast<const GridSet*>(ws + L.sets) + set, cell_start, sorted, slab, od,
                                           oi, qx, qy, qz, ql, j, (unsigned)pending, (unsigned)(pending >> 32));
    open |= ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)om.hi) << 32) |
            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)om.lo);
    const unsigned long long longscan = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)om.lhi) << 32) |
                                        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)om.llo);
    if (longscan) {  // wave-uniform: cubes through crowded cells (never on an evenly sampled surface)
      const OpenMask ol = serve_long_scans<LAB>(reinterpret_cast<const GridSet*>(ws + L.sets) + set, cell_start, sorted,
                                                slab, od, oi, qx, qy, qz, ql, j, (unsigned)longscan,
                                                (unsigned)(longscan >> 32));
      open |= ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)ol.hi) << 32) |
              (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)ol.lo);
    }
  }
  PP_QPHASE(7);
  if (open) {  // wave-uniform: far from everything the cubes hold -- group by group, the whole wave (see above)
    if constexpr (TQ != 0) {
      // the group search stages its candidates through a slice of the image: wait until every wave of the workgroup has
      // finished its walk (a count in LDS, not a barrier: waves with nothing left to do must not wait for the slow ones)
      while (__hip_atomic_load(&s_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)kW) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    const bool finite = __builtin_isfinite(qx) && __builtin_isfinite(qy) && __builtin_isfinite(qz);
    const unsigned long long todo = open & __ballot(finite);
    if (todo) {
      const Found f = wave_group_search<LAB>(g, cell_start, sorted, slab, qx, qy, qz, ql, best, bidx, (unsigned)todo,
                                             (unsigned)(todo >> 32),
                                             (lds_f4_wptr)(TQ ? &s_pts[0][wave * kGroupBatch] : &s_pts[slice][0]),
                                             (lds_f_wptr)(TQ ? &s_lab[0][LAB ? wave * kGroupBatch : 0] : &s_lab[slice][0]));
      if ((todo >> lane) & 1ull) {
        const bool none = f.bidx == 0x7fffffff;  // (labeled: nobody carries this label -- ref nmdistance_cuda.cu:110-113)
        od[j] = (LAB && none) ? 0.0f : f.best;
        oi[j] = none ? (LAB ? -1 : 0) : f.bidx;
      }
      open &= ~todo;
    }
  }
  PP_QPHASE(8);
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
  PP_QPHASE(9);
}

}  // namespace

// 0 = automatic (grid when a workspace is given and the problem is large enough to pay for its two
// launches); 1 = brute force; 2 = grid wherever it is structurally possible (tests)
static pp::Knob g_grid_mode;
extern "C" void pp_debug_set_nmdistance_search(int v) { g_grid_mode.set(v); }
// LDS points per wave of the wave-private form of the search kernel (320 / 384 / 512); selecting one also selects that
// form.  0 = default: the tile form below.
static pp::Knob g_stage_cap;
extern "C" void pp_debug_set_nmdistance_stage_cap(int v) { g_stage_cap.set(v); }
// queries per workgroup of the tile form: 0 = default (512); 256, 512, 768; -1 = the wave-private form (CAPW 384)
static pp::Knob g_tile;
extern "C" void pp_debug_set_nmdistance_tile(int v) { g_tile.set(v); }

// Per-kernel timing of the grid forward (bench.py's roofline of the dominant kernel): when switched on, HIP
// events are recorded on the launch stream before the build, between the two kernels and after the search;
// pp_debug_nmdistance_kernel_ms waits for the last one and reports the two durations of the most recent
// forward.  One set of events per process (a measurement aid for one stream at a time, not a product feature).
static pp::Knob g_time_kernels;
static std::mutex g_ev_mutex;
static hipEvent_t g_ev[3] = {nullptr, nullptr, nullptr};
static bool g_ev_valid = false;
extern "C" void pp_debug_set_nmdistance_kernel_timing(int on) { g_time_kernels.set(on); }
extern "C" int pp_debug_nmdistance_kernel_ms(float* build_ms, float* search_ms) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  if (!g_ev_valid || !build_ms || !search_ms) return PP_EINVAL;
  hipError_t e = hipEventSynchronize(g_ev[2]);
  if (e == hipSuccess) e = hipEventElapsedTime(build_ms, g_ev[0], g_ev[1]);
  if (e == hipSuccess) e = hipEventElapsedTime(search_ms, g_ev[1], g_ev[2]);
  return (int)e;
}
static void record_timing_event(int i, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  if (!g_ev[0])
    for (int k = 0; k < 3; ++k)
      if (hipEventCreate(&g_ev[k]) != hipSuccess) return;
  if (hipEventRecord(g_ev[i], s) == hipSuccess && i == 2) g_ev_valid = true;
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
  const size_t lds = pp::grid_build_lds_bytes(pp::kBuildSlabs);
  const bool vec = pp::clouds_vec_aligned(xyz1, N, B) && pp::clouds_vec_aligned(xyz2, M, B);
  e = vec ? pp::allow_big_lds(grid_build_kernel<true>, (int)lds, lds_ok_vec)
          : pp::allow_big_lds(grid_build_kernel<false>, (int)lds, lds_ok);
  if (e != hipSuccess) return (int)e;
  const bool timing = g_time_kernels != 0;
  if (timing) record_timing_event(0, s);
  (vec ? grid_build_kernel<true> : grid_build_kernel<false>)<<<dim3(8 * ((2 * B * pp::kBuildSlabs + 7) / 8)), dim3(kBuildThreads), lds, s>>>(
      xyz1, xyz2, ws, B, N, M, LAB ? label1 : nullptr, LAB ? label2 : nullptr);
  PP_RETURN_IF_LAUNCH_FAILED();
  if (timing) record_timing_event(1, s);
  const int cap = g_stage_cap, tile = g_tile;
  const bool wave_form = cap == 320 || cap == 384 || cap == 512 || tile == -1;
  const int tq = wave_form ? 256 : (tile == 256 || tile == 768 ? tile : 512);
  const int tiles1 = (N + tq - 1) / tq, tiles2 = (M + tq - 1) / tq;
  const long long blocks = (long long)B * (tiles1 + tiles2);
  if (blocks > 0x7fffffffLL) return PP_EINVAL;
  const int per_xcd = (int)((blocks + 7) / 8);
#define PP_LAUNCH_W(CAP_, TQ_)                                                                             \
  grid_query_wave_kernel<LAB, CAP_, TQ_><<<dim3((unsigned)(per_xcd * 8)), dim3(TQ_ ? TQ_ : 256), 0, s>>>(      \
      xyz1, xyz2, dist1, idx1, dist2, idx2, ws, B, N, M, tiles1, tiles2, (int)blocks, per_xcd, label1, label2)
  if (wave_form) {
    switch (cap) {
      case 320: PP_LAUNCH_W(320, 0); break;
      case 512: PP_LAUNCH_W(512, 0); break;
      default: PP_LAUNCH_W(384, 0); break;
    }
  } else {
    switch (tq) {
      case 256: PP_LAUNCH_W(1532, 256); break;
      case 768: PP_LAUNCH_W(3068, 768); break;
      default: PP_LAUNCH_W(3068, 512); break;
    }
  }
#undef PP_LAUNCH_W
  PP_RETURN_IF_LAUNCH_FAILED();
  if (timing) record_timing_event(2, s);
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
