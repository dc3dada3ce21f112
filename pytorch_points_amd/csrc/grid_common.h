// grid_common.h -- the uniform grid shared by the exact searches (chamfer_grid.hip, ball_grid.hip):
// cubic cells of side h over a cloud's bounding box (<= 32^3), points counting-sorted by cell.
#pragma once
#include "pp_common.h"

// phase marks of the build (tools/build_probe.hip defines PP_PHASE to record a clock; nothing otherwise)
#ifndef PP_PHASE
#define PP_PHASE(n)
#endif

namespace pp {

constexpr int kGridMax = 32;                                  // cells per axis
constexpr int kGridCells = kGridMax * kGridMax * kGridMax;    // LDS counters: 128 KiB
constexpr int kBuildThreads = 1024;

constexpr int kBuildSlabs = 4;  // workgroups that share the build of one set (each owns a range of cells)

// Chunk table (Chamfer's tile search): for every kChunk consecutive points of the sorted cloud, the lowest and the
// highest z coordinate among them, as order-preserving integers (zkey), one (min, max) pair per chunk AND per build
// slab -- a slab records the points it scatters; pairs of chunks it does not touch stay (+inf, -inf) -- so that the
// slabs need not meet: tile_z[(chunk * slabs + slab) * 2 + {0, 1}] (a tile's chunks are one contiguous piece: one wide
// scalar load), every one of the `chunks` pairs the caller allocated written.  Lets a tile of queries find the z-layers of the
// OTHER cloud's grid it can touch from two scalar loads, before its queries have arrived.
// Layer table (same consumer): layers[z] = position in the sorted cloud of the first point of z-layer z (z = 0 .. gz,
// layers[gz] = the number of points) -- the cell table's entries z * gy * gx, gathered where one vector load fetches
// them all; layers[kLayerPending] = a counter the build zeroes (the stage-A kernel counts the queries it leaves there).
constexpr int kLayerWords = 40;
constexpr int kLayerPending = 36;
constexpr int kLayerCursor = 37;  // (zeroed by the build too) pieces of the pending list handed out so far: list kernel
constexpr int kLayerRouted = 38;  // (zeroed by the build) 1: the stage-A kernel has routed the direction to the every-pair kernel (round 6)
constexpr int kChunk = 256;
constexpr int kChunkMax = 256;  // chunks per set the build can track (sets of up to 65536 points)
__host__ __device__ inline int zkey(float z) {  // monotone in z for every non-NaN float
#if defined(__HIP_DEVICE_COMPILE__)
  const int o = __float_as_int(z);
#else
  int o;
  __builtin_memcpy(&o, &z, 4);
#endif
  return o ^ ((o >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float zkey_inv(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

struct GridSet {  // one per (batch, direction); written by the build kernel
  float minx, miny, minz, h, invh;
  int gx, gy, gz;
  int useless;             // 1: degenerate data (non-finite / zero extent): no grid at all
  int pad[3];              // pad[0]: free for the caller (ball_query / three_nn: "the grid path serves this set");
                           // pad[1]: 1 = the set's chunk table (tile_z) has been written
                           // pad[2]: 1 = the box was trimmed (outliers): points lie outside it, in the rim cells
  int crowd[kBuildSlabs];  // crowd[s] = 1: slab s found a cell too crowded to be of use; 2: it refined its crowded
                           // cells into sub-grids (REFINE builds: see SubGrid)
};
static_assert(sizeof(GridSet) == 64, "");

// the grid is not worth using (or does not exist): send the set's queries to the brute force
__device__ __forceinline__ bool grid_useless(const GridSet& g) {
  return g.useless != 0 || g.crowd[0] == 1 || g.crowd[1] == 1 || g.crowd[2] == 1 || g.crowd[3] == 1;
}
// some cells of the grid hold more than kCrowd points and carry a second-level grid
__device__ __forceinline__ bool grid_refined(const GridSet& g) {
  return g.crowd[0] == 2 || g.crowd[1] == 2 || g.crowd[2] == 2 || g.crowd[3] == 2;
}

// Second level (REFINE builds, chamfer_grid.hip): a cell holding more than kCrowd points -- dense clusters,
// clouds with several scales, the core of a Gaussian -- is sorted once more, in place, into a grid of its own
// over the bounding box of ITS points, sized by its population (about four points per sub-cell if the points
// filled the box; at most kSubMaxCells cells).  A cell is "crowded" exactly when it holds more than kCrowd
// points: the searches read that off the cell table they load anyway.  The sub-grid of the crowded cell whose
// points start at `start` in the sorted cloud is described by desc[(start + kCrowd - 1) / kCrowd] (a range of
// more than kCrowd consecutive positions contains exactly one such slot first) and its table of cell starts
// (absolute positions in the sorted cloud, cells + 1 entries) begins at sub_start[2 * start].
constexpr int kCrowd = 128;
constexpr int kSubMaxAxis = 26;
constexpr int kSubMaxCells = 16384;
constexpr int kSubWaveCells = 1536;  // sub-cells of a crowded cell refined by one wave (its slice of the build's LDS)
constexpr int kBuildSlabCrowdMax = 1024;  // crowded cells one slab can refine
struct SubGrid {
  float minx, miny, minz, h, invh;
  int gx, gy, gz;
};
static_assert(sizeof(SubGrid) == 32, "");
__host__ __device__ inline size_t sub_desc_slots(size_t npoints) { return npoints / kCrowd + 2; }

__device__ __forceinline__ int cell_coord(float p, float mn, float invh, int g) {
  const float f = (p - mn) * invh;
  const int c = (int)f;  // v_cvt_i32_f32: saturates, NaN -> 0; negative and huge inputs end up clamped below
  int r;                 // clamp to [0, g - 1] (g >= 1) in one operation instead of max + min
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(c), "v"(g - 1));
  return r;
}

// linear index of cell (cx, cy, cz) in the z-major order; every factor and the result fit 24 bits
// (<= 32 cells per axis), so the multiply-adds are the full-rate v_mad_u32_u24, not the quarter-rate
// 32-bit multiplies
__device__ __forceinline__ int cell_linear(int cx, int cy, int cz, int gx, int gy) {
  int r;  // (spelled out: hipcc turns __umul24(a, b) + c into the 64-bit v_mad_u64_u32)
  asm("v_mad_u32_u24 %0, %1, %2, %3\n\tv_mad_u32_u24 %0, %0, %4, %5"
      : "=&v"(r)
      : "v"(cz), "v"(gy), "v"(cy), "v"(gx), "v"(cx));
  return r;
}

// dynamic LDS of a build workgroup: one bank-skewed counter per cell of its slab, then the cell index of
// every point the workgroup holds in registers (16 per thread), computed once and reused by the passes
__host__ __device__ inline int grid_build_counter_words(int nslab) {
  const int per = (kGridCells + nslab - 1) / nslab + 1;
  return (per + per / 32 + 1 + 3) & ~3;
}
__host__ __device__ inline size_t grid_build_lds_bytes(int nslab) {
  return ((size_t)grid_build_counter_words(nslab) + (size_t)kBuildThreads * 16) * sizeof(unsigned);
}

// 15-bit Morton code of a cell (5 bits per axis)
__device__ __forceinline__ int morton3(int x, int y, int z) {
  auto spread = [](unsigned v) {
    v = (v | (v << 8)) & 0x0000f00fu;
    v = (v | (v << 4)) & 0x000c30c3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
  };
  return (int)(spread((unsigned)x) | (spread((unsigned)y) << 1) | (spread((unsigned)z) << 2));
}

// Body of a build kernel: one 1024-thread workgroup sorts the `nr` points at `ref` into the grid.
// Dynamic LDS: pp::grid_build_lds_bytes(nslab): one bank-skewed unsigned counter per cell of a slab.
// Writes *gs, cell_start[0..ncell] (if non-null), sorted[0..nr) = (x, y, z, original index) and, if
// non-null, inv[k] = position of original point k in `sorted`; if `sorted_payload` is non-null,
// sorted_payload[pos] = payload[k] (a per-point float, e.g. a label, in the sorted order).
// MORTON = false: cells in z-major linear order (a row of cells along x is contiguous in `sorted`);
// a degenerate set (non-finite or zero extent) is marked useless and `sorted` is left unwritten.
// MORTON = true: cells in Morton order -- `sorted` is then just a spatially coherent permutation of
// the points (always written, whatever the data), for callers that walk the points in that order.
// The build of one set is shared by `nslab` <= kBuildSlabs workgroups: slab s owns the cells
// [s*ncell/nslab, (s+1)*ncell/nslab) -- it reads the whole cloud (L2), counts and scatters the points
// of its own cells only, and learns where its range starts in `sorted` by counting the points of the
// cells below.  Nothing is exchanged between the workgroups; the LDS counters shrink by nslab.
// The second level of a REFINE build (see grid_build_set_impl): the crowded cells of one slab, listed in
// s_clist[ncrowd] as (start, count), each sorted in place into its own grid.  NOT inlined: the build kernel of an
// evenly sampled cloud never gets here, and inlined this code costs it registers (148 bytes of scratch per lane).
__device__ __attribute__((noinline)) void grid_refine_cells(f4* __restrict__ sorted, f4* __restrict__ sorted2,
                                                            float* __restrict__ sorted_payload,
                                                            float* __restrict__ payload2,
                                                            unsigned* __restrict__ sub_start,
                                                            SubGrid* __restrict__ sub_desc, unsigned* s_cnt,
                                                            unsigned* s_part, float* s_box,
                                                            const unsigned (*s_clist)[2], unsigned ncrowd) {
  const int t = threadIdx.x;
  {
    {
      // The whole dynamic LDS is free now: one counter per sub-cell.  Small crowded cells (the usual kind: tens
      // to hundreds per set in clustered data) are refined by ONE WAVE each, sixteen at a time, with no workgroup
      // barrier (a wave's slice of the LDS holds kSubWaveCells counters); a cell too large for that is refined by
      // the whole workgroup.  Either way: streamed, register-light passes over the cell's points (they sit in
      // L2) -- bounding box, count, scatter into the spare copy `sorted2`, copy back.
      const int lane = t & 63, wave = t >> 6;
      constexpr unsigned kWaveMaxPoints = 2048;
      auto sub_geometry = [&](const float (&bmn)[6], unsigned n, int max_cells) {
        SubGrid sg;
        sg.minx = -bmn[0]; sg.miny = -bmn[1]; sg.minz = -bmn[2];
        const float sex = bmn[3] - sg.minx, sey = bmn[4] - sg.miny, sez = bmn[5] - sg.minz;
        const float semax = fmaxf(sex, fmaxf(sey, sez));
        int g2 = (int)cbrtf((float)n);  // ~ one point per sub-cell if the points filled their box (they never do)
        g2 = g2 < 2 ? 2 : (g2 > kSubMaxAxis - 1 ? kSubMaxAxis - 1 : g2);
        sg.h = semax / (float)g2;
        if (!(sg.h > 0.0f) || !__builtin_isfinite(sg.h)) sg.h = 1.0f;  // identical points: one sub-cell
        sg.invh = 1.0f / sg.h;
        auto cells2 = [&](float e) {
          const int c = (int)(e * sg.invh) + 1;
          return c < 1 ? 1 : (c > kSubMaxAxis ? kSubMaxAxis : c);
        };
        sg.gx = cells2(sex); sg.gy = cells2(sey); sg.gz = cells2(sez);
        // (cells + 1 table entries must fit sub_start[2 start .. 2 (start + n)), the counters their LDS slice)
        while ((long long)sg.gx * sg.gy * sg.gz > min((long long)max_cells, 2LL * n - 1)) {
          if (sg.gx >= sg.gy && sg.gx >= sg.gz) --sg.gx; else if (sg.gy >= sg.gz) --sg.gy; else --sg.gz;
        }
        return sg;
      };
      // The box the sub-grid is laid over: the bounding box of the cell's points, unless a few stragglers stretch it
      // (a dense blob plus a handful of points of the sparse scale in the same cell) -- then the box within 3.5
      // sigma of the mean, the stragglers landing in the rim sub-cells (cell_coord clamps; every bound of the
      // searches is stated through that monotone coordinate, exactly as at the top level).  sm: sums of (p - c) and
      // (p - c)^2 per axis, c the centre of the bounding box.
      auto trim_box = [&](float (&bmn)[6], const float (&sm)[6], unsigned n) {
        const float inv_n = 1.0f / (float)n;
        float lo[3] = {-bmn[0], -bmn[1], -bmn[2]}, hi[3] = {bmn[3], bmn[4], bmn[5]};
        const float ext0 = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
        float tl[3], th[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float c = 0.5f * (lo[a] + hi[a]);
          const float m = sm[a] * inv_n;
          const float var = fmaxf(sm[3 + a] * inv_n - m * m, 0.0f);
          const float sd = sqrtf(var);
          tl[a] = fmaxf(lo[a], c + m - 3.5f * sd);
          th[a] = fminf(hi[a], c + m + 3.5f * sd);
        }
        const float ext1 = fmaxf(th[0] - tl[0], fmaxf(th[1] - tl[1], th[2] - tl[2]));
        if (ext1 > 0.0f && ext1 < 0.75f * ext0) {  // (NaN moments compare false: the bounding box stays)
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            bmn[a] = -tl[a];
            bmn[3 + a] = th[a];
          }
        }
      };
      auto cell2 = [](const SubGrid& sg, const f4& p) {
        return cell_linear(cell_coord(p.x, sg.minx, sg.invh, sg.gx), cell_coord(p.y, sg.miny, sg.invh, sg.gy),
                           cell_coord(p.z, sg.minz, sg.invh, sg.gz), sg.gx, sg.gy);
      };
      auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      };
      // ---- one wave per small crowded cell ----
      {
        unsigned* s_sub = s_cnt + (size_t)wave * kSubWaveCells;
        for (unsigned ci = wave; ci < ncrowd; ci += kBuildThreads / 64) {  // wave-uniform
          const unsigned start = s_clist[ci][0], n = s_clist[ci][1];
          if (n > kWaveMaxPoints) continue;
          // (every pass: eight loads of a lane in flight, then the work -- a plain loop waits for each load where it is
          //  used, 4 to 32 round trips per pass: 49 of the 70 us of a clustered cloud's build)
          auto for_points = [&](const f4* __restrict__ src, auto&& fn) {
            for (unsigned k0 = 0; k0 < n; k0 += 512u) {
              f4 q8[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) q8[u] = src[start + min(k0 + 64u * (unsigned)u + (unsigned)lane, n - 1u)];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const unsigned k = k0 + 64u * (unsigned)u + (unsigned)lane;
                if (k < n) fn(q8[u], k);
              }
            }
          };
          float bmn[6] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff(),
                          -__builtin_inff()};  // max of (-x, -y, -z, x, y, z)
          for_points(sorted, [&](const f4& q, unsigned) {
            bmn[0] = fmaxf(bmn[0], -q.x); bmn[1] = fmaxf(bmn[1], -q.y); bmn[2] = fmaxf(bmn[2], -q.z);
            bmn[3] = fmaxf(bmn[3], q.x); bmn[4] = fmaxf(bmn[4], q.y); bmn[5] = fmaxf(bmn[5], q.z);
          });
          wave_reduce6_dpp<false, 6>(bmn);
#pragma unroll
          for (int e = 0; e < 6; ++e) bmn[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bmn[e]), 63));
          {
            const float c0 = 0.5f * (bmn[3] - bmn[0]), c1 = 0.5f * (bmn[4] - bmn[1]), c2 = 0.5f * (bmn[5] - bmn[2]);
            float sm[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            for_points(sorted, [&](const f4& q, unsigned) {
              const float dx = q.x - c0, dy = q.y - c1, dz = q.z - c2;
              sm[0] += dx; sm[1] += dy; sm[2] += dz;
              sm[3] = __builtin_fmaf(dx, dx, sm[3]); sm[4] = __builtin_fmaf(dy, dy, sm[4]); sm[5] = __builtin_fmaf(dz, dz, sm[5]);
            });
            wave_reduce6_dpp<true, 6>(sm);
#pragma unroll
            for (int e = 0; e < 6; ++e) sm[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm[e]), 63));
            trim_box(bmn, sm, n);
          }
          const SubGrid sg = sub_geometry(bmn, n, kSubWaveCells);
          const int total2 = sg.gx * sg.gy * sg.gz;
          for (int c = lane; c < total2; c += 64) s_sub[c] = 0u;
          wave_sync();
          for_points(sorted, [&](const f4& q, unsigned) { atomicAdd(&s_sub[cell2(sg, q)], 1u); });
          wave_sync();
          {  // exclusive scan: lane l owns a contiguous run of counters
            const int per2 = (total2 + 63) / 64;
            const int d0 = min(total2, lane * per2), d1 = min(total2, d0 + per2);
            unsigned sum2 = 0;
            for (int c = d0; c < d1; ++c) sum2 += s_sub[c];
            unsigned incl2 = sum2;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
              const unsigned o = __shfl_up(incl2, off);
              if (lane >= off) incl2 += o;
            }
            unsigned run2 = start + incl2 - sum2;
            unsigned* __restrict__ tbl = sub_start + 2 * (size_t)start;
            for (int c = d0; c < d1; ++c) {
              const unsigned v = s_sub[c];
              s_sub[c] = run2;  // sub-cell start (absolute position): table entry and scatter cursor
              tbl[c] = run2;
              run2 += v;
            }
            if (lane == 0) tbl[total2] = start + n;
          }
          wave_sync();
          for_points(sorted, [&](const f4& q, unsigned k) {
            const unsigned pos = atomicAdd(&s_sub[cell2(sg, q)], 1u);
            sorted2[pos] = q;
            if (sorted_payload) payload2[pos] = sorted_payload[start + k];
          });
          wave_sync();
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          for_points(sorted2, [&](const f4& q, unsigned k) {  // back into place (coalesced)
            sorted[start + k] = q;
            if (sorted_payload) sorted_payload[start + k] = payload2[start + k];
          });
          if (lane == 0) sub_desc[(start + kCrowd - 1) / kCrowd] = sg;
        }
      }
      __syncthreads();
      // ---- the whole workgroup for a large crowded cell ----
      unsigned* s_sub = s_cnt;
      for (unsigned ci = 0; ci < ncrowd; ++ci) {
        const unsigned start = s_clist[ci][0], n = s_clist[ci][1];
        if (n <= kWaveMaxPoints) continue;
        // every pass over the cell's points: eight loads of a thread in flight, then the work (a plain loop waits for
        // each load where it is used: eight round trips per pass, five passes -- 40 us of the 65 a cell of 8192 points
        // cost the two-scale cloud's build)
        auto for_points = [&](const f4* __restrict__ src, auto&& fn) {
          for (unsigned k0 = 0; k0 < n; k0 += 8u * kBuildThreads) {
            f4 q8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q8[u] = src[start + min(k0 + (unsigned)u * kBuildThreads + (unsigned)t, n - 1u)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const unsigned k = k0 + (unsigned)u * kBuildThreads + (unsigned)t;
              if (k < n) fn(q8[u], k);
            }
          }
        };
        float bmn[6] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff(),
                        -__builtin_inff()};
        for_points(sorted, [&](const f4& q, unsigned) {
          bmn[0] = fmaxf(bmn[0], -q.x); bmn[1] = fmaxf(bmn[1], -q.y); bmn[2] = fmaxf(bmn[2], -q.z);
          bmn[3] = fmaxf(bmn[3], q.x); bmn[4] = fmaxf(bmn[4], q.y); bmn[5] = fmaxf(bmn[5], q.z);
        });
        wave_reduce6_dpp<false, 6>(bmn);
        __syncthreads();  // s_box / s_sub of the previous cell are no longer read
        if ((t & 63) == 63)
#pragma unroll
          for (int e = 0; e < 6; ++e) s_box[(t >> 6) * 16 + e] = bmn[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 6; ++e) bmn[e] = s_box[(t & 15) * 16 + e];
        wave_reduce6_dpp<false, 4>(bmn);
#pragma unroll
        for (int e = 0; e < 6; ++e) bmn[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bmn[e]), 15));
        {
          const float c0 = 0.5f * (bmn[3] - bmn[0]), c1 = 0.5f * (bmn[4] - bmn[1]), c2 = 0.5f * (bmn[5] - bmn[2]);
          float sm[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
          for_points(sorted, [&](const f4& q, unsigned) {
            const float dx = q.x - c0, dy = q.y - c1, dz = q.z - c2;
            sm[0] += dx; sm[1] += dy; sm[2] += dz;
            sm[3] = __builtin_fmaf(dx, dx, sm[3]); sm[4] = __builtin_fmaf(dy, dy, sm[4]); sm[5] = __builtin_fmaf(dz, dz, sm[5]);
          });
          wave_reduce6_dpp<true, 6>(sm);
          __syncthreads();  // s_box: the bounding-box partials have been read
          if ((t & 63) == 63)
#pragma unroll
            for (int e = 0; e < 6; ++e) s_box[(t >> 6) * 16 + e] = sm[e];
          __syncthreads();
#pragma unroll
          for (int e = 0; e < 6; ++e) sm[e] = s_box[(t & 15) * 16 + e];
          wave_reduce6_dpp<true, 4>(sm);
#pragma unroll
          for (int e = 0; e < 6; ++e) sm[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm[e]), 15));
          trim_box(bmn, sm, n);
        }
        const SubGrid sg = sub_geometry(bmn, n, kSubMaxCells);
        const int total2 = sg.gx * sg.gy * sg.gz;
        for (int c = t; c < total2; c += kBuildThreads) s_sub[c] = 0u;
        __syncthreads();
        for_points(sorted, [&](const f4& q, unsigned) { atomicAdd(&s_sub[cell2(sg, q)], 1u); });
        __syncthreads();
        {  // exclusive scan of the total2 counters (contiguous runs per thread, as above)
          const int per2 = (total2 + kBuildThreads - 1) / kBuildThreads;
          const int d0 = min(total2, t * per2), d1 = min(total2, d0 + per2);
          unsigned sum2 = 0;
          for (int c = d0; c < d1; ++c) sum2 += s_sub[c];
          unsigned incl2 = sum2;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(incl2, off);
            if ((t & 63) >= off) incl2 += o;
          }
          if ((t & 63) == 63) s_part[t >> 6] = incl2;
          __syncthreads();
          unsigned run2 = start + incl2 - sum2;
          for (int w = 0; w < (t >> 6); ++w) run2 += s_part[w];
          unsigned* __restrict__ tbl = sub_start + 2 * (size_t)start;
          for (int c = d0; c < d1; ++c) {
            const unsigned v = s_sub[c];
            s_sub[c] = run2;
            tbl[c] = run2;
            run2 += v;
          }
          if (t == 0) tbl[total2] = start + n;
        }
        __syncthreads();
        for_points(sorted, [&](const f4& q, unsigned k) {
          const unsigned pos = atomicAdd(&s_sub[cell2(sg, q)], 1u);
          sorted2[pos] = q;
          if (sorted_payload) payload2[pos] = sorted_payload[start + k];
        });
        __threadfence_block();
        __syncthreads();
        for_points(sorted2, [&](const f4& q, unsigned k) {  // back into place (coalesced)
          sorted[start + k] = q;
          if (sorted_payload) sorted_payload[start + k] = payload2[start + k];
        });
        if (t == 0) sub_desc[(start + kCrowd - 1) / kCrowd] = sg;
      }
    }
  }
}

// What the planning phases of a build decide for a set: the box, the resolution, and the range of cells one slab owns.
// grid_build_set_fast (below) hands it to grid_build_set_impl when a slab it cannot sort through the LDS has to take the
// general path: that path then skips its own planning (`forced`) -- every slab of a set must sort into the SAME grid.
struct BuildPlan {
  float mnx, mny, mnz, h, invh;
  int gx, gy, gz;
  int cell_lo, cell_hi;  // this slab's cells
  int trimmed;
};

// Round 6 (the plain builds: ball_query, three_nn, knn_points): a point with a non-finite coordinate (or one whose
// square overflows) can never be anybody's neighbour -- its distance is NaN or +inf, never < anything, and where a
// search runs out of finite candidates it ends with the whole grid, rim cells included -- so it must not cost the
// batch element its grid (one NaN in one cloud: knn K = 8 0.18 -> 1.95 ms, the whole element every-pair).  The box
// and the moments once more over the finite points only; the others keep whatever cell cell_coord gives them
// (NaN -> 0, +-inf -> the rim).  The Chamfer build (REFINE) keeps the old rule: there the reference's "first point
// unconditionally" makes a non-finite point at index 0 selectable, which only the every-pair order reproduces.
// OUT OF LINE, the points read again from memory: inlined -- the same arithmetic on the points the builds hold in
// registers -- it cost every build 350 bytes of scratch per thread on the path that never takes it (bq_build_kernel
// 19.8 -> 24.7 us).  Called by every thread of the workgroup; leaves in s_box: [0..2] -min, [3..5] max, [8..13] the
// sums of x, y, z, x^2, y^2, z^2, [14] the number of finite points (the same values in every workgroup of the set: the
// slabs' plans must agree, and they do -- same data, same order).  The caller reads them and meets a barrier.
__device__ __attribute__((noinline)) void plain_finite_plan(const float* __restrict__ ref, int nr, float* s_box) {
  const int t = threadIdx.x;
  const float inf = __builtin_inff();
  float v[6] = {-inf, -inf, -inf, -inf, -inf, -inf};  // -min, max
  float w[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  float cnt = 0.0f;
  for (int i = t; i < nr; i += kBuildThreads) {
    const float x = ref[3 * (size_t)i], y = ref[3 * (size_t)i + 1], z = ref[3 * (size_t)i + 2];
    if (__builtin_isfinite(x * x + y * y + z * z)) {
      v[0] = fmaxf(v[0], -x); v[1] = fmaxf(v[1], -y); v[2] = fmaxf(v[2], -z);
      v[3] = fmaxf(v[3], x);  v[4] = fmaxf(v[4], y);  v[5] = fmaxf(v[5], z);
      cnt += 1.0f;
      w[0] += x; w[1] += y; w[2] += z;
      w[3] += x * x; w[4] += y * y; w[5] += z * z;
    }
  }
  wave_reduce6_dpp<false, 6>(v);
  wave_reduce6_dpp<true, 6>(w);
  cnt = wave_reduce_dpp<true>(cnt);
  __syncthreads();  // s_box was read by every thread before the call
  if ((t & 63) == 63) {
#pragma unroll
    for (int e = 0; e < 6; ++e) s_box[(t >> 6) * 16 + e] = v[e];
#pragma unroll
    for (int e = 0; e < 6; ++e) s_box[(t >> 6) * 16 + 8 + e] = w[e];
    s_box[(t >> 6) * 16 + 14] = cnt;
  }
  __syncthreads();
  if (t < 64) {  // one wave folds the sixteen partials, in a fixed order
    float fv[6], fw[6], fc = 0.0f;
#pragma unroll
    for (int e = 0; e < 6; ++e) { fv[e] = -inf; fw[e] = 0.0f; }
    for (int k = 0; k < kBuildThreads / 64; ++k) {
#pragma unroll
      for (int e = 0; e < 6; ++e) { fv[e] = fmaxf(fv[e], s_box[k * 16 + e]); fw[e] += s_box[k * 16 + 8 + e]; }
      fc += s_box[k * 16 + 14];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();  // (every lane has read the partials before lane 0 overwrites the first row)
    if (t == 0) {
#pragma unroll
      for (int e = 0; e < 6; ++e) { s_box[e] = fv[e]; s_box[8 + e] = fw[e]; }
      s_box[14] = fc;
    }
  }
  __syncthreads();
}

template <bool MORTON, bool VEC, bool REFINE>
__device__ __forceinline__ void grid_build_set_impl(const float* __restrict__ ref, int nr, GridSet* gs,
                                               unsigned* __restrict__ cell_start, f4* __restrict__ sorted,
                                               int* __restrict__ inv, unsigned* s_cnt,
                                               const float* __restrict__ payload,
                                               float* __restrict__ sorted_payload, int slab,
                                               int nslab, unsigned* __restrict__ sub_start,
                                               SubGrid* __restrict__ sub_desc, f4* __restrict__ sorted2,
                                               float* __restrict__ payload2, int* __restrict__ tile_z = nullptr,
                                               int tz_chunks = 0, unsigned* __restrict__ layers = nullptr,
                                               const BuildPlan* forced = nullptr) {
  // (`forced` points into LDS -- grid_build_set_fast leaves the plan there, behind everything this function uses of the
  //  dynamic LDS -- and is read where it is used.  As a local of the calling kernel the plan lived in SCRATCH memory, its
  //  address taken: the fast path's eleven stores to it were 11 MB of write traffic per build at config 2; by value it
  //  cost the kernel 80 spilled registers)
  __shared__ unsigned s_part[kBuildThreads];
  __shared__ int s_tz[REFINE ? 2 * kChunkMax : 2];  // chunk table of this slab: (min, max) z keys
  const int nchunkq = (nr + kChunk - 1) / kChunk;
  const bool track_z = REFINE && tile_z != nullptr && nchunkq <= tz_chunks && tz_chunks <= kChunkMax;  // (uniform)
  __shared__ unsigned s_crowd;
  __shared__ float s_box[(kBuildThreads / 64) * 16];
  static_assert(kBuildThreads / 64 == 16, "the bounding-box reduction assumes 16 waves");
  const int t = threadIdx.x;
  if (t == 0) s_crowd = 0u;
  if constexpr (REFINE) {
    if (track_z && t < 2 * kChunkMax) s_tz[t] = (t & 1) ? zkey(-__builtin_inff()) : zkey(__builtin_inff());
  }

  // A thread keeps KP points in registers (one chunk = 1024*KP points; a single chunk covers
  // 16384 points, so the three passes read the cloud from memory once).  Loads are unconditional
  // (index clamped) so that all KP are in flight together.
  constexpr int KP = 16;
  const int nchunks = (nr + kBuildThreads * KP - 1) / (kBuildThreads * KP);
  float px[KP], py[KP], pz[KP];
  // Point i of thread t in the chunk at `base`.  16-byte aligned clouds are read as float4 (a thread
  // takes 4 consecutive points = 48 bytes, three fully coalesced loads) -- 3x fewer cache-line
  // requests than three 4-byte loads at a 12-byte lane stride; the mapping only has to be the same
  // in every pass.
  constexpr bool vec = VEC;  // the cloud is 16-byte aligned
  auto kidx = [&](int base, int i) { return base + (i >> 2) * (4 * kBuildThreads) + 4 * t + (i & 3); };
  auto load_chunk = [&](int base) {
    if (vec) {
#pragma unroll
      for (int gq = 0; gq < KP / 4; ++gq) {
        const int p0 = base + gq * (4 * kBuildThreads) + 4 * t;
        if (p0 + 3 < nr) {
          const f4* __restrict__ src = reinterpret_cast<const f4*>(ref + 3 * (size_t)p0);
          const f4 a = src[0], b = src[1], c = src[2];
          px[4 * gq] = a.x; py[4 * gq] = a.y; pz[4 * gq] = a.z;
          px[4 * gq + 1] = a.w; py[4 * gq + 1] = b.x; pz[4 * gq + 1] = b.y;
          px[4 * gq + 2] = b.z; py[4 * gq + 2] = b.w; pz[4 * gq + 2] = c.x;
          px[4 * gq + 3] = c.y; py[4 * gq + 3] = c.z; pz[4 * gq + 3] = c.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = min(p0 + j, nr - 1);
            px[4 * gq + j] = ref[3 * (size_t)k];
            py[4 * gq + j] = ref[3 * (size_t)k + 1];
            pz[4 * gq + j] = ref[3 * (size_t)k + 2];
          }
        }
      }
    } else {
      // A cloud that starts at any 4-byte boundary (odd point counts in a batch, views): the same four consecutive
      // points per thread, read as ALIGNED 16-byte vectors -- the 48 bytes begin `sh` dwords into the first of four
      // (three when sh == 0) -- and shifted into place; `sh` is uniform over the workgroup (12 t dwords per thread
      // keep the phase).  No 12-byte loads at a 12-byte lane stride (which the compiler made dwordx3 tuples of and
      // spilled: 168-216 bytes of scratch per lane), the same cache-line traffic as the aligned form.
      // The vectors stay inside the 16-byte granules that hold the cloud's first and last bytes: no page is touched
      // that the cloud does not touch.
      const uintptr_t a0 = reinterpret_cast<uintptr_t>(ref);
#pragma unroll
      for (int gq = 0; gq < KP / 4; ++gq) {
        const int p0 = base + gq * (4 * kBuildThreads) + 4 * t;
        if (p0 + 3 < nr) {
          const uintptr_t a = a0 + (uintptr_t)12 * (uintptr_t)p0;
          const int sh = (int)((a0 + (uintptr_t)12 * (uintptr_t)(base + gq * (4 * kBuildThreads))) >> 2) & 3;  // uniform
          const f4* __restrict__ src = reinterpret_cast<const f4*>(a & ~(uintptr_t)15);
          const f4 v0 = src[0], v1 = src[1], v2 = src[2];
          f4 v3 = v2;
          if (sh != 0) v3 = src[3];  // (uniform)
          const float w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
          float o[12];
#pragma unroll
          for (int e = 0; e < 12; ++e) o[e] = sh == 0 ? w[e] : (sh == 1 ? w[e + 1] : (sh == 2 ? w[e + 2] : w[e + 3]));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            px[4 * gq + j] = o[3 * j];
            py[4 * gq + j] = o[3 * j + 1];
            pz[4 * gq + j] = o[3 * j + 2];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = min(p0 + j, nr - 1);
            px[4 * gq + j] = ref[3 * (size_t)k];
            py[4 * gq + j] = ref[3 * (size_t)k + 1];
            pz[4 * gq + j] = ref[3 * (size_t)k + 2];
          }
        }
      }
    }
  };
  PP_PHASE(0);
  load_chunk(0);
  // bounding box (the clamped duplicates do not change it) + first and second moments (over the real
  // points only), for the outlier test below and for the finiteness test
  float mnx = __builtin_inff(), mny = mnx, mnz = mnx, mxx = -mnx, mxy = -mnx, mxz = -mnx;
  float sm[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // sum x, y, z; sum x^2, y^2, z^2
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch > 0) load_chunk(ch * kBuildThreads * KP);
    // (the moments only steer the outlier heuristic -- every slab computes the same ones -- so they may use
    // fused multiply-adds; a full chunk, the usual case, needs no per-point "is this a real point" factor:
    // 12 instead of 23 VALU operations per point in a kernel that is bound by VALU issue)
    if ((ch + 1) * kBuildThreads * KP <= nr) {
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        const float x = px[i], y = py[i], z = pz[i];
        mnx = fminf(mnx, x); mny = fminf(mny, y); mnz = fminf(mnz, z);
        mxx = fmaxf(mxx, x); mxy = fmaxf(mxy, y); mxz = fmaxf(mxz, z);
        sm[0] += x; sm[1] += y; sm[2] += z;
        sm[3] = __builtin_fmaf(x, x, sm[3]); sm[4] = __builtin_fmaf(y, y, sm[4]); sm[5] = __builtin_fmaf(z, z, sm[5]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        const float x = px[i], y = py[i], z = pz[i];
        mnx = fminf(mnx, x); mny = fminf(mny, y); mnz = fminf(mnz, z);
        mxx = fmaxf(mxx, x); mxy = fmaxf(mxy, y); mxz = fmaxf(mxz, z);
        const float live = kidx(ch * kBuildThreads * KP, i) < nr ? 1.0f : 0.0f;
        sm[0] += live * x; sm[1] += live * y; sm[2] += live * z;
        sm[3] += live * x * x; sm[4] += live * y * y; sm[5] += live * z * z;
      }
    }
  }
  PP_PHASE(1);
  bool any_bad;
  if (forced) {  // (uniform) the planning was done by the caller: same box on every slab of the set
    mnx = forced->mnx; mny = forced->mny; mnz = forced->mnz;
    any_bad = false;
  } else {  // six max-reductions (-min, max) and six sums with one barrier; a non-finite coordinate (or one whose
     // square overflows: treated alike, the set goes to the brute force) shows in the sums of squares
    float v[6] = {-mnx, -mny, -mnz, mxx, mxy, mxz};
    wave_reduce6_dpp<false, 6>(v);
    wave_reduce6_dpp<true, 6>(sm);
    if ((t & 63) == 63) {
#pragma unroll
      for (int e = 0; e < 6; ++e) s_box[(t >> 6) * 16 + e] = v[e];
#pragma unroll
      for (int e = 0; e < 6; ++e) s_box[(t >> 6) * 16 + 8 + e] = sm[e];
    }
    __syncthreads();
    // 16 waves: lane l of every row of 16 reads wave l's partial; four row steps finish the job in lane 15
#pragma unroll
    for (int e = 0; e < 6; ++e) v[e] = s_box[(t & 15) * 16 + e];
#pragma unroll
    for (int e = 0; e < 6; ++e) sm[e] = s_box[(t & 15) * 16 + 8 + e];
    wave_reduce6_dpp<false, 4>(v);
    wave_reduce6_dpp<true, 4>(sm);
#pragma unroll
    for (int e = 0; e < 6; ++e) v[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[e]), 15));
#pragma unroll
    for (int e = 0; e < 6; ++e) sm[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm[e]), 15));
    mnx = -v[0]; mny = -v[1]; mnz = -v[2]; mxx = v[3]; mxy = v[4]; mxz = v[5];
    any_bad = !__builtin_isfinite(sm[3] + sm[4] + sm[5]);
  }
  // Round 6 (the plain builds: ball_query, three_nn, knn_points): a point with a non-finite coordinate (or one whose
  // square overflows) can never be anybody's neighbour -- its distance is NaN or +inf, never < anything, and where a
  // search runs out of finite candidates it ends with the whole grid, rim cells included -- so it must not cost the
  // batch element its grid (one NaN in one cloud: knn K = 8 0.18 -> 1.95 ms, the whole element every-pair).  The
  // box and the moments once more over the finite points only; the others keep whatever cell cell_coord gives them
  // (NaN -> 0, +-inf -> the rim).  The Chamfer build (REFINE) keeps the old rule: there the reference's "first point
  // unconditionally" makes a non-finite point at index 0 selectable, which only the every-pair order reproduces.
  float n_live = (float)nr;
  if (!REFINE && !forced && any_bad) {  // (uniform; out of line: see plain_finite_plan)
    plain_finite_plan(ref, nr, s_box);
    // (wave-uniform values, as the ones they replace: through readfirstlane)
    auto uni = [&](int e) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_box[e]))); };
    mnx = -uni(0); mny = -uni(1); mnz = -uni(2); mxx = uni(3); mxy = uni(4); mxz = uni(5);
#pragma unroll
    for (int e = 0; e < 6; ++e) sm[e] = uni(8 + e);
    n_live = uni(14);
    any_bad = !(n_live > 0.0f) || !__builtin_isfinite(sm[3] + sm[4] + sm[5]);
    __syncthreads();  // (s_box is written again below)
    if (nchunks > 1) load_chunk(0);  // (the passes below reload their chunks themselves, from the first)
  }
  PP_PHASE(2);
  // Outliers: a few points far from the bulk would stretch the box until the bulk sits in a handful of
  // cells.  Any box is valid -- cell_coord clamps, the points outside simply land in the boundary cells
  // and every bound is stated in terms of the (monotone) cell coordinate -- so when the box reaches
  // beyond 6 sigma of the mean on some side, it is replaced by the box of the points within 4 sigma on
  // every axis.  Uniform over the workgroup; clouds without outliers skip the second pass.
  bool trimmed = forced ? forced->trimmed != 0 : false;  // the box does not hold every point (the outliers sit in the rim cells)
  if (!any_bad && !forced) {
    const float inv_n = 1.0f / n_live;
    const float mean[3] = {sm[0] * inv_n, sm[1] * inv_n, sm[2] * inv_n};
    float sig[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) sig[a] = sqrtf(fmaxf(sm[3 + a] * inv_n - mean[a] * mean[a], 0.0f));
    const bool stretched = mxx - mean[0] > 6.0f * sig[0] || mean[0] - mnx > 6.0f * sig[0] ||
                           mxy - mean[1] > 6.0f * sig[1] || mean[1] - mny > 6.0f * sig[1] ||
                           mxz - mean[2] > 6.0f * sig[2] || mean[2] - mnz > 6.0f * sig[2];
    if (stretched) {
      float w[6] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff(),
                    -__builtin_inff()};  // max of (-x, -y, -z, x, y, z) over the inliers
      for (int ch = 0; ch < nchunks; ++ch) {
        if (nchunks > 1) load_chunk(ch * kBuildThreads * KP);
#pragma unroll
        for (int i = 0; i < KP; ++i) {
          const float x = px[i], y = py[i], z = pz[i];
          const bool in = fabsf(x - mean[0]) <= 4.0f * sig[0] && fabsf(y - mean[1]) <= 4.0f * sig[1] &&
                          fabsf(z - mean[2]) <= 4.0f * sig[2];
          if (in) {
            w[0] = fmaxf(w[0], -x); w[1] = fmaxf(w[1], -y); w[2] = fmaxf(w[2], -z);
            w[3] = fmaxf(w[3], x); w[4] = fmaxf(w[4], y); w[5] = fmaxf(w[5], z);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) w[e] = wave_reduce_dpp<false>(w[e]);
      __syncthreads();  // s_box is read above by every thread
      if ((t & 63) == 0)
#pragma unroll
        for (int e = 0; e < 6; ++e) s_box[(t >> 6) * 16 + e] = w[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 6; ++e) w[e] = wave_reduce_dpp<false>(s_box[(t & 15) * 16 + e]);
      if (w[3] > -w[0] || w[4] > -w[1] || w[5] > -w[2]) {  // the trimmed set has an extent: use its box
        mnx = -w[0]; mny = -w[1]; mnz = -w[2]; mxx = w[3]; mxy = w[4]; mxz = w[5];
        trimmed = true;
      }
      if (nchunks > 1) load_chunk(0);  // (the passes below reload their chunks themselves)
    }
  }
  PP_PHASE(3);
  const float ex = mxx - mnx, ey = mxy - mny, ez = mxz - mnz;
  const float emax = fmaxf(ex, fmaxf(ey, ez));
  // first guess: ~2 points per cell if the cloud filled its box; cubic cells of side h
  int g0 = (int)ceilf(cbrtf(2.0f * (float)nr));
  g0 = g0 < 1 ? 1 : (g0 > kGridMax ? kGridMax : g0);
  const bool degenerate = !forced && (any_bad || !(emax > 0.0f) || !__builtin_isfinite(emax));
  float h, invh;
  int gx, gy, gz;
  auto set_resolution = [&](int g) {
    h = emax / (float)g;
    if (!(h > 0.0f) || !__builtin_isfinite(h)) h = 1.0f;
    invh = 1.0f / h;
    auto cells = [&](float e) {
      int c = (int)(e * invh) + 1;  // c*h > e: the box maximum lies inside the last cell
      return c < 1 ? 1 : (c > kGridMax ? kGridMax : c);
    };
    gx = degenerate ? 1 : cells(ex);
    gy = degenerate ? 1 : cells(ey);
    gz = degenerate ? 1 : cells(ez);
  };
  set_resolution(g0);
  if (forced) {
    h = forced->h; invh = forced->invh; gx = forced->gx; gy = forced->gy; gz = forced->gz;
  }
  int* s_cid = reinterpret_cast<int*>(s_cnt + grid_build_counter_words(nslab));  // [KP][kBuildThreads]
  // The searches want ~4-5 points per OCCUPIED cell (then the first, smallest stage answers ~98 % of the
  // queries): the first guess is right for a surface in a cubic box, too fine for a volume (1.3 points per
  // occupied cell for a uniformly filled cube).  Measure the occupancy with a bitmap of cells (every
  // workgroup of the set sees all the points, so all of them take the same decision) and coarsen while
  // it is below 2.5 points per occupied cell.
  auto sk = [](int c) { return c + (c >> 5); };
  unsigned below = 0;    // points in the cells of lower slabs
  bool counted = false;  // the occupancy pass has also counted the points of this slab's cells
  if (!MORTON && !degenerate) {
    // One pass per round does both jobs: every point marks its cell in the bitmap (the occupancy is a
    // property of the whole set) and, if the cell belongs to this slab, bumps the cell's counter -- the
    // cell index is computed once.  A round that ends in "coarsen" (rare for surfaces) recounts.
    __shared__ unsigned s_occ[kGridCells / 32];
    __shared__ unsigned s_nocc;
    constexpr int kRounds = 4;
    bool stale = false;  // the resolution changed after the last count
    for (int round = 0; round < kRounds; ++round) {
      stale = false;
      const int nc = gx * gy * gz;
      const int lo = forced ? forced->cell_lo : (int)((long long)nc * slab / nslab);
      const int nl = forced ? forced->cell_hi - forced->cell_lo : (int)((long long)nc * (slab + 1) / nslab) - lo;
      const int nwords = (nc + 31) / 32;
      for (int wd = t; wd < nwords; wd += kBuildThreads) s_occ[wd] = 0;
      for (int c = t; c < nl; c += kBuildThreads) s_cnt[sk(c)] = 0;
      if (t == 0) s_nocc = 0;
      __syncthreads();
      below = 0;
      for (int ch = 0; ch < nchunks; ++ch) {
        if (nchunks > 1) load_chunk(ch * kBuildThreads * KP);
        auto mark = [&](int i) {
          const int c = cell_linear(cell_coord(px[i], mnx, invh, gx), cell_coord(py[i], mny, invh, gy),
                                    cell_coord(pz[i], mnz, invh, gz), gx, gy);
          atomicOr(&s_occ[c >> 5], 1u << (c & 31));
          const int cl = c - lo;
          below += cl < 0 ? 1u : 0u;
          const bool mine = (unsigned)cl < (unsigned)nl;
          const int slot = sk(cl);
          // kept for the scatter when the cloud is a single chunk (the last round's value): the counter of
          // the point's cell if the cell is this slab's, else -1 -- the scatter then recomputes nothing
          s_cid[i * kBuildThreads + t] = mine ? slot : -1;
          if (mine) atomicAdd(&s_cnt[slot], 1u);
        };
        if ((ch + 1) * kBuildThreads * KP <= nr) {  // uniform: a full chunk needs no per-point bounds test
          // A thread's four consecutive points at a time.  Where every lane of the wave finds its four in ONE cell
          // (a dense part of the cloud that is also contiguous in memory: the dense scale of a two-scale cloud put
          // 8192 points into one counter, 8192 same-address LDS atomics in this pass and again in the scatter: 27 us
          // each) the counter is bumped once per thread, or once per wave if all its lanes agree.
#pragma unroll
          for (int g4 = 0; g4 < KP / 4; ++g4) {
            int c4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int i = 4 * g4 + u;
              c4[u] = cell_linear(cell_coord(px[i], mnx, invh, gx), cell_coord(py[i], mny, invh, gy),
                                  cell_coord(pz[i], mnz, invh, gz), gx, gy);
            }
            const bool same4 = (c4[0] == c4[1]) & (c4[1] == c4[2]) & (c4[2] == c4[3]);
            if (__all(same4)) {  // (wave-uniform, rare)
              const int c = c4[0];
              atomicOr(&s_occ[c >> 5], 1u << (c & 31));
              const int cl = c - lo;
              below += cl < 0 ? 4u : 0u;
              const bool mine = (unsigned)cl < (unsigned)nl;
              const int slot = sk(cl);
#pragma unroll
              for (int u = 0; u < 4; ++u) s_cid[(4 * g4 + u) * kBuildThreads + t] = mine ? slot : -1;
              const int lead = __builtin_amdgcn_readfirstlane(c);
              if (__all(c == lead)) {
                if ((t & 63) == 0 && mine) atomicAdd(&s_cnt[slot], 256u);
              } else if (mine) {
                atomicAdd(&s_cnt[slot], 4u);
              }
            } else {
#pragma unroll
              for (int u = 0; u < 4; ++u) mark(4 * g4 + u);
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < KP; ++i)
            if (kidx(ch * kBuildThreads * KP, i) < nr) mark(i);
        }
      }
      __syncthreads();
      unsigned mine = 0;
      for (int wd = t; wd < nwords; wd += kBuildThreads) mine += __builtin_popcount(s_occ[wd]);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
      if ((t & 63) == 0 && mine) atomicAdd(&s_nocc, mine);
      __syncthreads();
      const unsigned nocc = s_nocc;
      __syncthreads();
      const int gmax = max(gx, max(gy, gz));
      if (forced || (float)nr >= 2.5f * (float)nocc || gmax <= 4) break;
      // coarsen by ~1/sqrt(2) per round: x2.8 points per cell for a volume, x2 for a surface
      set_resolution(max(4, (int)((float)gmax * 0.7071f)));
      stale = true;
    }
    // (a cloud still too fine after the last round is counted again below, at the resolution it ended with)
    counted = !stale;
  }
  PP_PHASE(4);
  const int ncell = MORTON ? kGridCells : gx * gy * gz;

  const int cell_lo = forced ? forced->cell_lo : (int)((long long)ncell * slab / nslab);
  const int cell_hi = forced ? forced->cell_hi : (int)((long long)ncell * (slab + 1) / nslab);
  const int nloc = cell_hi - cell_lo;  // this slab's cells: local index = cell - cell_lo
  auto cell_of = [&](float x, float y, float z) {
    const int cx = cell_coord(x, mnx, invh, gx), cy = cell_coord(y, mny, invh, gy), cz = cell_coord(z, mnz, invh, gz);
    return MORTON ? morton3(cx, cy, cz) : cell_linear(cx, cy, cz, gx, gy);
  };
  PP_PHASE(5);
  const bool place = MORTON || !degenerate;
  if (!counted) {  // Morton mode (and degenerate sets, which only need empty counters)
    for (int c = t; c < nloc; c += kBuildThreads) s_cnt[sk(c)] = 0;
    below = 0;
    __syncthreads();
    if (place)
      for (int ch = 0; ch < nchunks; ++ch) {
        if (nchunks > 1) load_chunk(ch * kBuildThreads * KP);
#pragma unroll
        for (int i = 0; i < KP; ++i)
          if (kidx(ch * kBuildThreads * KP, i) < nr) {
            const int cg = cell_of(px[i], py[i], pz[i]);
            if (nchunks == 1) s_cid[i * kBuildThreads + t] = cg;  // for the scatter
            const int c = cg - cell_lo;
            if (c < 0) ++below;
            else if (c < nloc) atomicAdd(&s_cnt[sk(c)], 1u);
          }
      }
    __syncthreads();
  }
  PP_PHASE(6);
  // exclusive scan: each thread owns a contiguous run of cells
  const int per = (nloc + kBuildThreads - 1) / kBuildThreads;
  const int c0 = min(nloc, t * per), c1 = min(nloc, c0 + per);
  unsigned sum = 0, mx = 0;
  for (int c = c0; c < c1; ++c) {
    sum += s_cnt[sk(c)];
    mx = max(mx, s_cnt[sk(c)]);
  }
  // a cell holding so many points (> 256 + N/32) that walking it lane by lane costs more than the
  // brute-force kernel's share of the cloud (flag cleared before the first barrier, read after the last).
  // REFINE builds give every cell above kCrowd points a grid of its own instead (below).
  if (!REFINE && (float)mx > 256.0f + (float)nr * (1.0f / 32.0f)) s_crowd = 1u;
  if (REFINE && place && sub_start && sub_desc && mx > (unsigned)kCrowd) atomicOr(&s_crowd, 2u);  // (read after barriers)
  // exclusive scan of the 1024 per-thread sums: inclusive scan inside each wave (shuffles), then
  // the 16 wave totals; the points below this slab (summed the same way) are the starting offset
  unsigned incl = sum, bsum = below;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off);
    if ((t & 63) >= off) incl += o;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bsum += __shfl_xor(bsum, off);
  if ((t & 63) == 63) s_part[t >> 6] = incl;
  if ((t & 63) == 0) s_part[64 + (t >> 6)] = bsum;
  __syncthreads();
  unsigned wave_base = 0;
  for (int w = 0; w < kBuildThreads / 64; ++w) {
    wave_base += s_part[64 + w];
    if (w < (t >> 6)) wave_base += s_part[w];
  }
  unsigned run = wave_base + incl - sum;
  for (int c = c0; c < c1; ++c) {
    const unsigned v = s_cnt[sk(c)];
    s_cnt[sk(c)] = run;  // cell start; becomes the scatter cursor below
    run += v;
  }
  __syncthreads();
  PP_PHASE(7);
  if (cell_start) {
    // coalesced copy out (storing from the scan loop above, 32-byte pieces per lane, measured 1.9 us slower)
    for (int c = t; c < nloc; c += kBuildThreads) cell_start[cell_lo + c] = s_cnt[sk(c)];
    if (t == 0 && slab == nslab - 1) cell_start[ncell] = degenerate ? 0u : (unsigned)nr;
  }
  if constexpr (REFINE) {
    if (layers != nullptr) {  // the layer table: the entries of this slab's cells (read before the scatter moves the cursors)
      if (t < gz) {
        const int c = t * gy * gx - cell_lo;
        if (c >= 0 && c < nloc) layers[t] = s_cnt[sk(c)];
      }
      if (t == gz && slab == nslab - 1) layers[gz] = degenerate ? 0u : (unsigned)nr;
      if ((t == kLayerPending || t == kLayerCursor || t == kLayerRouted) && slab == 0) layers[t] = 0u;
    }
  }
  __syncthreads();
  PP_PHASE(8);
  if (place)
    for (int ch = 0; ch < nchunks; ++ch) {
      if (nchunks > 1) load_chunk(ch * kBuildThreads * KP);
      // four cursor atomics at a time (independent, in flight together), then their stores
#pragma unroll
      for (int i0 = 0; i0 < KP; i0 += 4) {
        int c[4];
        unsigned pos[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (counted && nchunks == 1) {  // (uniform) the occupancy pass left the counter's index, or -1
            c[i] = s_cid[(i0 + i) * kBuildThreads + t];  // (entries of points beyond nr were never written
            if (kidx(ch * kBuildThreads * KP, i0 + i) >= nr) c[i] = -1;  //  in that pass: mask them here)
          } else {
            const int cc = (nchunks == 1 ? s_cid[(i0 + i) * kBuildThreads + t]
                                         : cell_of(px[i0 + i], py[i0 + i], pz[i0 + i])) - cell_lo;
            c[i] = (kidx(ch * kBuildThreads * KP, i0 + i) < nr && cc >= 0 && cc < nloc) ? sk(cc) : -1;
          }
        }
        // (the same shortcut as in the count: one cursor bump per thread, or per wave, where four consecutive points
        //  share a cell in every lane)
        const bool same4s = (c[0] >= 0) & (c[0] == c[1]) & (c[1] == c[2]) & (c[2] == c[3]);
        if (__all(same4s)) {  // (wave-uniform, rare)
          const int lead = __builtin_amdgcn_readfirstlane(c[0]);
          unsigned base;
          if (__all(c[0] == lead)) {
            base = 0u;
            if ((t & 63) == 0) base = atomicAdd(&s_cnt[lead], 256u);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + 4u * (unsigned)(t & 63);
          } else {
            base = atomicAdd(&s_cnt[c[0]], 4u);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) pos[i] = base + (unsigned)i;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) pos[i] = c[i] >= 0 ? atomicAdd(&s_cnt[c[i]], 1u) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (c[i] >= 0) {
            const int k = kidx(ch * kBuildThreads * KP, i0 + i);
            f4 v = {px[i0 + i], py[i0 + i], pz[i0 + i], __int_as_float(k)};
            sorted[pos[i]] = v;
            if constexpr (REFINE) {
              if (track_z) {
                const int zk = zkey(pz[i0 + i]);
                atomicMin(&s_tz[2 * (pos[i] / kChunk)], zk);
                atomicMax(&s_tz[2 * (pos[i] / kChunk) + 1], zk);
              }
            }
            if (inv) inv[k] = (int)pos[i];
            if (sorted_payload) sorted_payload[pos[i]] = payload[k];  // one float per point, carried along
          }
      }
    }
  if constexpr (REFINE) {
    if (track_z) {  // (the refinement below permutes points inside a cell only: a cell's points keep their chunk's range
                    //  up to the cell's own extent -- it re-sorts a cell in place, so a chunk may receive points of the
                    //  same cell from a neighbouring chunk; sets with crowded cells do not use the table, see the search)
      __syncthreads();
      for (int c = t; c < 2 * tz_chunks; c += kBuildThreads) tile_z[((size_t)(c >> 1) * nslab + slab) * 2 + (c & 1)] = s_tz[c];
    }
  }
  PP_PHASE(9);
  bool refined = false;
  if constexpr (REFINE && !MORTON) {
    // ---- second level: every cell of this slab with more than kCrowd points gets its own grid --------------
    __shared__ unsigned s_ncrowd;
    __shared__ unsigned s_clist[kBuildSlabCrowdMax][2];  // (start, count) of this slab's crowded cells
    __syncthreads();
    if ((s_crowd & 2u) != 0u) {  // uniform: some cell of this slab is crowded (never at config 2: nothing below runs)
      if (t == 0) s_ncrowd = 0u;
      __threadfence_block();  // the scatter's stores are read back below
      __syncthreads();
      {
        unsigned prev = c0 < c1 ? cell_start[cell_lo + c0] : 0u;  // start of this thread's first cell (table written above)
        for (int c = c0; c < c1; ++c) {
          const unsigned end = s_cnt[sk(c)];  // the cursor stands at the cell's end after the scatter
          const unsigned cnt = end - prev;
          if (cnt > (unsigned)kCrowd) {
            const unsigned at = atomicAdd(&s_ncrowd, 1u);
            if (at < (unsigned)kBuildSlabCrowdMax) {
              s_clist[at][0] = prev;
              s_clist[at][1] = cnt;
            } else {
              atomicOr(&s_crowd, 1u);  // more crowded cells than the list holds: the set goes to the brute force
            }
          }
          prev = end;
        }
      }
      __syncthreads();
      const unsigned ncrowd = min(s_ncrowd, (unsigned)kBuildSlabCrowdMax);
      refined = ncrowd > 0;
      grid_refine_cells(sorted, sorted2, sorted_payload, payload2, sub_start, sub_desc, s_cnt, s_part, s_box, s_clist,
                        ncrowd);
    }
  }
  if (t == 0) {
    gs->crowd[slab] = degenerate ? 0 : ((s_crowd & 1u) != 0u ? 1 : (refined ? 2 : 0));
    if (slab == 0) {
      gs->minx = mnx; gs->miny = mny; gs->minz = mnz; gs->h = h; gs->invh = invh;
      gs->gx = gx; gs->gy = gy; gs->gz = gz;
      gs->useless = degenerate ? 1 : 0;
      for (int i = 0; i < 2; ++i) gs->pad[i] = 0;
      gs->pad[2] = trimmed ? 1 : 0;
      if (REFINE && track_z && place) gs->pad[1] = 1;
      for (int i = nslab; i < kBuildSlabs; ++i) gs->crowd[i] = 0;
    }
  }
  PP_PHASE(10);
}

// inclusive prefix sum over the 64 lanes of a wave through DPP (row shifts inside the rows of 16, then the row
// broadcasts): VALU-rate, where __shfl_up is six dependent ds_bpermute round trips.  Every lane must be active.
__device__ __forceinline__ unsigned wave_scan_u32_dpp(unsigned v) {
#define PP_SCAN_STEP(CTRL, ROWS) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false)
  PP_SCAN_STEP(0x111, 0xf);  // row_shr:1
  PP_SCAN_STEP(0x112, 0xf);  // row_shr:2
  PP_SCAN_STEP(0x114, 0xf);  // row_shr:4
  PP_SCAN_STEP(0x118, 0xf);  // row_shr:8
  PP_SCAN_STEP(0x142, 0xa);  // row_bcast:15 -> rows 1, 3
  PP_SCAN_STEP(0x143, 0xc);  // row_bcast:31 -> rows 2, 3
#undef PP_SCAN_STEP
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// grid_build_set_fast (round 5) -- the REFINE build of a set of at most 16384 points (one register chunk, 16-byte aligned,
// a multiple of four points: BASELINE config 2's class), the same outputs as grid_build_set_impl<false, true, true>.
// The general build's workgroup does everything to all of the set's points although it owns a quarter of the cells: cell
// index, bitmap mark, counter slot and "is it mine" for sixteen points per thread, then again sixteen masked cursor
// atomics in the scatter; it is bound by VALU issue.  Here only the planning looks at every point
// (box, moments, one cell index + one bitmap mark per point: the decisions must be the set's, not the slab's); then
//   * a slab owns whole z-LAYERS of the grid (cells are z-major: still one contiguous range of cells and of `sorted`);
//   * the points of its layers are COMPACTED into an LDS list (wave ballots, one cursor add per wave) as 16-byte records
//     (x, y, z, index | local cell << 16), and everything after that touches a quarter of the set: one returning counter
//     add per own point (its rank inside its cell), the scan of the slab's counters, then position = cell start + rank --
//     no second round of atomics -- written into the LDS list's place IN SORTED ORDER and copied out with full 16-byte
//     coalesced stores (the general path scatters 16-byte pieces straight to memory); the chunk table comes from that
//     copy-out, a wave reduction per 64 sorted points instead of two LDS atomics per point;
//   * crowded cells (> kCrowd points) are refined afterwards by the general build's grid_refine_cells.
// What it cannot do is left to the general path, with this set's plan forced on it (BuildPlan) so that every slab
// still sorts into the same grid: a slab whose layers hold more than kFastCap points (the middle of a Gaussian), more
// crowded cells than its list holds.
// Flat grids (gz < kFastMinLayers) and degenerate sets are not started at all (every slab agrees: uniform data).
// Returns 0 = done, 1 = not applicable (general path, unforced), 2 = general path with `plan` forced.
#ifndef PP_BUILD_SPILL
#define PP_BUILD_SPILL PP_BUILD_STAGE_MASKED  // a slab beyond the list's capacity scatters from its registers (0: general path)
#endif
#ifndef PP_BUILD_JUMP
#define PP_BUILD_JUMP 1  // a coarsening step sized by the occupancy (a volume needs more than the surface's step)
#endif
#ifndef PP_BUILD_BALANCE
#define PP_BUILD_BALANCE 1  // the slabs' layers dealt by a histogram where the cloud is not spread evenly along z
#endif
#ifndef PP_BUILD_STAGE_MASKED
#define PP_BUILD_STAGE_MASKED 1  // the staging writes under the lanes' own mask (a quarter of the lanes hold a point of the slab:
#endif                         // 0.3 us less than every lane writing, the others to dump slots)
constexpr int kFastCap = 6144;       // points per pass (16-byte records in LDS: 96 KiB, the refinement's counters afterwards)
constexpr int kFastCells = 8192;     // cells per slab: ceil(32 / 4) layers of 32 x 32
constexpr int kFastMinLayers = 8;
constexpr int kFastCrowdMax = 256;   // crowded cells one slab lists (more: general path)
constexpr int kFastMaxPoints = kBuildThreads * 16;
static_assert((size_t)kFastCap * 16 >= (size_t)(kBuildThreads / 64) * kSubWaveCells * 4, "grid_refine_cells' counters live in the list");
// dynamic LDS (words): counters | list (bitmap of the planning rounds inside it) | chunk table | box partials | scan partials,
// cursor words | crowded-cell list
constexpr int kFastLdsCnt = 0;
constexpr int kFastLdsList = kFastLdsCnt + kFastCells + 64;
constexpr int kFastLdsTz = kFastLdsList + (kFastCap + 64) * 4;  // (+ 64 records nobody reads: the lanes without a point of the slab)
constexpr int kFastLdsBox = kFastLdsTz + 2 * kChunkMax;
constexpr int kFastLdsPart = kFastLdsBox + (kBuildThreads / 64) * 16;
constexpr int kFastLdsClist = kFastLdsPart + 64;
constexpr int kFastLdsPlan = kFastLdsClist + 2 * kFastCrowdMax;  // a BuildPlan for the general path (return value 2)
constexpr int kFastLdsWords = kFastLdsPlan + 16;
static_assert(sizeof(BuildPlan) <= 16 * 4, "");
__host__ __device__ inline size_t grid_build_fast_lds_bytes() { return (size_t)kFastLdsWords * 4; }

template <bool REFINE = true>
__device__ __forceinline__ int grid_build_set_fast(const float* __restrict__ ref, int nr, GridSet* gs,
                                                   unsigned* __restrict__ cell_start, f4* __restrict__ sorted,
                                                   unsigned* lds, int slab, unsigned* __restrict__ sub_start,
                                                   SubGrid* __restrict__ sub_desc, f4* __restrict__ sorted2,
                                                   int* __restrict__ tile_z, int tz_chunks,
                                                   unsigned* __restrict__ layers) {
  constexpr int KP = 16, nslab = kBuildSlabs;
  const int nchunkq = (nr + kChunk - 1) / kChunk;
  // (REFINE = false: the plain build of ball_query / three_nn / knn_points -- no second level (a cell beyond 256 + nr / 32
  //  points marks the set useless, as the general path does), no chunk and layer tables)
  if (nr > kFastMaxPoints || nr < 4 * kBuildThreads || (nr & 3) != 0 || cell_start == nullptr) return 1;  // (uniform)
  if (REFINE && (tile_z == nullptr || layers == nullptr || sub_start == nullptr || sub_desc == nullptr ||
                 nchunkq > tz_chunks || tz_chunks > kChunkMax))
    return 1;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  unsigned* s_cnt = lds + kFastLdsCnt;
  f4* s_list = reinterpret_cast<f4*>(lds + kFastLdsList);
  unsigned* s_occ = lds + kFastLdsList;  // (planning rounds only: before the list is used)
  int* s_tz = reinterpret_cast<int*>(lds + kFastLdsTz);
  float* s_box = reinterpret_cast<float*>(lds + kFastLdsBox);
  unsigned* s_part = lds + kFastLdsPart;  // [0, 16): wave totals of the scan; [32, 40): cursor, below, ncrowd, nocc, flags
  unsigned(*s_clist)[2] = reinterpret_cast<unsigned(*)[2]>(lds + kFastLdsClist);
  unsigned& s_below = s_part[33];
  unsigned& s_ncrowd = s_part[34];

  PP_PHASE(0);
  // ---- the set's points: four consecutive ones per thread and group of 4096, as 16-byte loads (the general path's map)
  float px[KP], py[KP], pz[KP];
  bool live[KP / 4];
#pragma unroll
  for (int gq = 0; gq < KP / 4; ++gq) {
    const int p0 = gq * (4 * kBuildThreads) + 4 * t;
    live[gq] = p0 < nr;  // (nr is a multiple of four: a group is whole or absent)
    const f4* __restrict__ src = reinterpret_cast<const f4*>(ref + 3 * (size_t)(live[gq] ? p0 : 0));
    const f4 a = src[0], b = src[1], c = src[2];
    px[4 * gq] = a.x; py[4 * gq] = a.y; pz[4 * gq] = a.z;
    px[4 * gq + 1] = a.w; py[4 * gq + 1] = b.x; pz[4 * gq + 1] = b.y;
    px[4 * gq + 2] = b.z; py[4 * gq + 2] = b.w; pz[4 * gq + 2] = c.x;
    px[4 * gq + 3] = c.y; py[4 * gq + 3] = c.z; pz[4 * gq + 3] = c.w;
  }
  auto kidx = [&](int i) { return (i >> 2) * (4 * kBuildThreads) + 4 * t + (i & 3); };
  if (REFINE && t < 2 * kChunkMax) s_tz[t] = (t & 1) ? zkey(-__builtin_inff()) : zkey(__builtin_inff());
  if (t == 0) {
    s_ncrowd = 0u;
    s_below = 0u;
  }
  // the bitmap and every counter a slab can have, zeroed while the points are on their way (the first round of the count
  // then starts without a zeroing phase and its barrier: the box's barrier lies in between)
  s_occ[t] = 0u;
  for (int c = 4 * t; c < kFastCells; c += 4 * kBuildThreads) *reinterpret_cast<uint4*>(&s_cnt[c]) = make_uint4(0u, 0u, 0u, 0u);
  // ---- planning 1: bounding box and moments (the general path's arithmetic; dead groups repeat point 0: no effect on the
  // box, and they are kept out of the sums)
  float mnx = __builtin_inff(), mny = mnx, mnz = mnx, mxx = -mnx, mxy = -mnx, mxz = -mnx;
  float sm[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    const float x = px[i], y = py[i], z = pz[i];
    mnx = fminf(mnx, x); mny = fminf(mny, y); mnz = fminf(mnz, z);
    mxx = fmaxf(mxx, x); mxy = fmaxf(mxy, y); mxz = fmaxf(mxz, z);
    if (live[i >> 2]) {
      sm[0] += x; sm[1] += y; sm[2] += z;
      sm[3] = __builtin_fmaf(x, x, sm[3]); sm[4] = __builtin_fmaf(y, y, sm[4]); sm[5] = __builtin_fmaf(z, z, sm[5]);
    }
  }
  PP_PHASE(1);
  bool any_bad;
  {
    float v[6] = {-mnx, -mny, -mnz, mxx, mxy, mxz};
    wave_reduce6_dpp<false, 6>(v);
    wave_reduce6_dpp<true, 6>(sm);
    if (lane == 63) {
#pragma unroll
      for (int e = 0; e < 6; ++e) s_box[wave * 16 + e] = v[e];
#pragma unroll
      for (int e = 0; e < 6; ++e) s_box[wave * 16 + 8 + e] = sm[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 6; ++e) v[e] = s_box[(t & 15) * 16 + e];
#pragma unroll
    for (int e = 0; e < 6; ++e) sm[e] = s_box[(t & 15) * 16 + 8 + e];
    wave_reduce6_dpp<false, 4>(v);
    wave_reduce6_dpp<true, 4>(sm);
#pragma unroll
    for (int e = 0; e < 6; ++e) v[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[e]), 15));
#pragma unroll
    for (int e = 0; e < 6; ++e) sm[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm[e]), 15));
    mnx = -v[0]; mny = -v[1]; mnz = -v[2]; mxx = v[3]; mxy = v[4]; mxz = v[5];
    any_bad = !__builtin_isfinite(sm[3] + sm[4] + sm[5]);
  }
  // (plain builds) non-finite points are left out of the plan instead of costing the set its grid: the general path's
  // rule and arithmetic (grid_build_set_impl, round 6)
  float n_live = (float)nr;
  if (!REFINE && any_bad) {  // (uniform; out of line: see plain_finite_plan)
    plain_finite_plan(ref, nr, s_box);
    // (wave-uniform values, as the ones they replace: through readfirstlane)
    auto uni = [&](int e) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_box[e]))); };
    mnx = -uni(0); mny = -uni(1); mnz = -uni(2); mxx = uni(3); mxy = uni(4); mxz = uni(5);
#pragma unroll
    for (int e = 0; e < 6; ++e) sm[e] = uni(8 + e);
    n_live = uni(14);
    any_bad = !(n_live > 0.0f) || !__builtin_isfinite(sm[3] + sm[4] + sm[5]);
    __syncthreads();  // (s_box is written again below)
  }
  PP_PHASE(2);
  bool trimmed = false;
  float zmean = 0.0f, zsig = 0.0f;  // (for the slabs' balance below)
  if (!any_bad) {  // outliers: the box of the points within 4 sigma when the bounding box reaches beyond 6 (general path)
    const float inv_n = 1.0f / n_live;
    const float mean[3] = {sm[0] * inv_n, sm[1] * inv_n, sm[2] * inv_n};
    float sig[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) sig[a] = sqrtf(fmaxf(sm[3 + a] * inv_n - mean[a] * mean[a], 0.0f));
    zmean = mean[2];
    zsig = sig[2];
    const bool stretched = mxx - mean[0] > 6.0f * sig[0] || mean[0] - mnx > 6.0f * sig[0] ||
                           mxy - mean[1] > 6.0f * sig[1] || mean[1] - mny > 6.0f * sig[1] ||
                           mxz - mean[2] > 6.0f * sig[2] || mean[2] - mnz > 6.0f * sig[2];
    if (stretched) {
      float w[6] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff(),
                    -__builtin_inff()};
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        const float x = px[i], y = py[i], z = pz[i];
        const bool in = fabsf(x - mean[0]) <= 4.0f * sig[0] && fabsf(y - mean[1]) <= 4.0f * sig[1] &&
                        fabsf(z - mean[2]) <= 4.0f * sig[2];
        if (in) {
          w[0] = fmaxf(w[0], -x); w[1] = fmaxf(w[1], -y); w[2] = fmaxf(w[2], -z);
          w[3] = fmaxf(w[3], x); w[4] = fmaxf(w[4], y); w[5] = fmaxf(w[5], z);
        }
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) w[e] = wave_reduce_dpp<false>(w[e]);
      __syncthreads();
      if (lane == 0)
#pragma unroll
        for (int e = 0; e < 6; ++e) s_box[wave * 16 + e] = w[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 6; ++e) w[e] = wave_reduce_dpp<false>(s_box[(t & 15) * 16 + e]);
      if (w[3] > -w[0] || w[4] > -w[1] || w[5] > -w[2]) {
        mnx = -w[0]; mny = -w[1]; mnz = -w[2]; mxx = w[3]; mxy = w[4]; mxz = w[5];
        trimmed = true;
      }
    }
  }
  PP_PHASE(3);
  const float ex = mxx - mnx, ey = mxy - mny, ez = mxz - mnz;
  const float emax = fmaxf(ex, fmaxf(ey, ez));
  if (any_bad || !(emax > 0.0f) || !__builtin_isfinite(emax)) return 1;  // degenerate: the general path marks it (uniform)
  int g0 = 1;  // the general path's first guess, ceil(cbrt(2 nr)) clamped to [1, 32], on the scalar unit (cbrtf: ~200 dependent
  while (g0 < kGridMax && g0 * g0 * g0 < 2 * nr) ++g0;  // vector instructions in every lane of a phase that is one latency chain)
  float h, invh;
  int gx, gy, gz;
  auto set_resolution = [&](int g) {
    h = emax / (float)g;
    if (!(h > 0.0f) || !__builtin_isfinite(h)) h = 1.0f;
    invh = 1.0f / h;
    auto cells = [&](float e) {
      int c = (int)(e * invh) + 1;
      return c < 1 ? 1 : (c > kGridMax ? kGridMax : c);
    };
    gx = cells(ex); gy = cells(ey); gz = cells(ez);
  };
  set_resolution(g0);
  // ---- planning 2 + count: the general path's rounds (about 2.5 points per occupied cell), one pass per round: a point's
  // cell index, its mark in the bitmap of the SET's cells and -- if the cell lies in this slab's layers -- one returning
  // add on the cell's counter: the point's rank inside its cell, kept in a register with the counter's slot.  A round
  // that ends in "coarsen" (never for a surface) counts again.
  unsigned key[KP];  // slot | rank << 16 of the points of this slab's layers, ~0u for the others
  int gxy = 0, lo = 0, ncs = 0;
  auto count_pass = [&](bool mark) {
    float ox = mnx, oy = mny, oz = mnz;  // (hidden from the optimiser: it would keep 48 differences p - origin alive)
    asm volatile("" : "+s"(ox), "+s"(oy), "+s"(oz));
    unsigned nbelow = 0;
#pragma unroll
    for (int gq = 0; gq < KP / 4; ++gq) {
      int c4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = 4 * gq + u;
        c4[u] = cell_linear(cell_coord(px[i], ox, invh, gx), cell_coord(py[i], oy, invh, gy),
                            cell_coord(pz[i], oz, invh, gz), gx, gy);
      }
      const bool alive = live[gq];
      // (a thread's four consecutive points in one cell in EVERY lane, and that cell the same over the wave -- a dense
      //  blob that is contiguous in memory -- would be 256 serial same-address atomics: one add of 256 instead)
      const bool same4 = alive & (c4[0] == c4[1]) & (c4[1] == c4[2]) & (c4[2] == c4[3]);
      const int lead = __builtin_amdgcn_readfirstlane(c4[0]);
      if (__all(same4 && c4[0] == lead)) {  // (wave-uniform, rare)
        const unsigned slot = (unsigned)(lead - lo);
        if (mark && lane == 0) atomicOr(&s_occ[lead >> 5], 1u << (lead & 31));
        if (lead < lo) nbelow += 256u;
        unsigned b0 = 0u;
        const bool mine = slot < (unsigned)ncs;
        if (mine && lane == 0) b0 = atomicAdd(&s_cnt[slot], 256u);
        b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)b0) + 4u * (unsigned)lane;
#pragma unroll
        for (int u = 0; u < 4; ++u) key[4 * gq + u] = mine ? (slot | ((b0 + (unsigned)u) << 16)) : ~0u;
      } else {
        // (no branch per point: a lane whose point is not the slab's -- or is one of the repeats beyond nr -- adds to a
        //  counter nobody reads, s_cnt[kFastCells + lane]; a repeat marks the cell of point 0, which is occupied)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c4[u];
          if (mark) atomicOr(&s_occ[c >> 5], 1u << (c & 31));
          const unsigned slot = (unsigned)(c - lo);
          const bool mine = alive & (slot < (unsigned)ncs);
          nbelow += (unsigned)__builtin_popcountll(__ballot(alive & (c < lo)));
          const unsigned r = atomicAdd(&s_cnt[mine ? slot : (unsigned)(kFastCells + lane)], 1u);
          key[4 * gq + u] = mine ? (slot | (r << 16)) : ~0u;
        }
      }
    }
    if (lane == 0 && nbelow) atomicAdd(&s_below, nbelow);
  };
  unsigned running = 0, own = 0;
#if PP_BUILD_BALANCE
  const bool balance_z = fabsf(zsig - 0.2887f * ez) > 0.03f * ez || fabsf(zmean - (mnz + 0.5f * ez)) > 0.06f * ez;
#endif
  {
    constexpr int kRounds = 4;
    for (int round = 0;; ++round) {
      // (after the last round's "coarsen" the cloud is counted at the resolution it ended with, whatever its occupancy)
      if (gz < kFastMinLayers) return 1;  // a flat grid: slabs of whole layers would leave workgroups idle (uniform)
      gxy = gx * gy;
      int zl = gz * slab / nslab, zh = gz * (slab + 1) / nslab;  // this slab's layers: an equal share ...
#if PP_BUILD_BALANCE
      // ... unless the cloud is not spread evenly along z (its moments say so: an even spread has sigma = extent /
      // sqrt(12) about the middle -- every surface of revolution about z, every filled box; a Gaussian, an object
      // with a face in its lowest layer have not): then the layers are dealt by a HISTOGRAM of a quarter of the points
      // (one LDS add per thread and group), a quarter of the cloud to each slab as nearly as whole layers allow -- an
      // equal share of layers gave the middle slabs of a Gaussian, or the slab with the face, more points than the LDS
      // list holds, and that slab took the general path (35-40 us for the launch instead of 25).  Every slab of the set
      // computes the same moments and the same histogram from the same points: they agree without meeting.
      if (balance_z) {  // (uniform)
        unsigned* s_hist = reinterpret_cast<unsigned*>(s_box);  // (the box partials are done with)
        if (t < 32) s_hist[t] = 0u;
        __syncthreads();
#pragma unroll
        for (int gq = 0; gq < KP / 4; ++gq)
          if (live[gq]) atomicAdd(&s_hist[cell_coord(pz[4 * gq], mnz, invh, gz)], 1u);
        __syncthreads();
        const unsigned cum = wave_scan_u32_dpp(lane < gz ? s_hist[lane] : 0u);  // (gz <= 32)
        const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)cum, 63);
        int zb[nslab + 1];
        zb[0] = 0;
        zb[nslab] = gz;
#pragma unroll
        for (int k = 1; k < nslab; ++k)  // the layers that lie wholly within the first k quarters; a layer at least per slab
          zb[k] = min(max((int)__builtin_popcountll(__ballot(lane < gz && (unsigned)nslab * cum <= (unsigned)k * tot)), zb[k - 1] + 1),
                      gz - (nslab - k));
        bool fits = true;  // (a slab's counters hold kFastCells cells)
#pragma unroll
        for (int k = 0; k < nslab; ++k) fits = fits && (zb[k + 1] - zb[k]) * gxy <= kFastCells;
        if (fits) {
#pragma unroll
          for (int k = 0; k < nslab; ++k)
            if (k == slab) {
              zl = zb[k];
              zh = zb[k + 1];
            }
        }
      }
#endif
      lo = zl * gxy;
      ncs = (zh - zl) * gxy;
      const int nwords = (gxy * gz + 31) / 32;
      if (round > 0) {  // (the first round finds the bitmap and the counters zeroed at the kernel's start, behind the box's barrier)
        if (t < nwords) s_occ[t] = 0u;  // (<= 1024 words)
        for (int c = 4 * t; c < ((ncs + 7) & ~7); c += 4 * kBuildThreads)  // (to a multiple of eight: the scan reads whole octets)
          *reinterpret_cast<uint4*>(&s_cnt[c]) = make_uint4(0u, 0u, 0u, 0u);
        if (t == 0) s_below = 0u;
        __syncthreads();
      }
      PP_PHASE(4);
      count_pass(round < kRounds);
      __syncthreads();  // the bitmap and the counters are complete
      PP_PHASE(5);
      // ---- one barrier serves both the occupancy of the round (popcounts of the bitmap) and the exclusive scan of the
      // slab's counters (eight consecutive cells per thread): a wave leaves its two totals in s_part
      unsigned v8[8];
      const int c0 = 8 * t;
      if (c0 < ncs) {
        const uint4 a = *reinterpret_cast<const uint4*>(&s_cnt[c0]);
        const uint4 b = *reinterpret_cast<const uint4*>(&s_cnt[c0 + 4]);
        v8[0] = a.x; v8[1] = a.y; v8[2] = a.z; v8[3] = a.w; v8[4] = b.x; v8[5] = b.y; v8[6] = b.z; v8[7] = b.w;
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = 0u;
      }
      unsigned sum = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v8[u];
      const unsigned incl = wave_scan_u32_dpp(sum);
      const unsigned occ = wave_scan_u32_dpp(round < kRounds && t < nwords ? (unsigned)__builtin_popcount(s_occ[t]) : 0u);
      if (lane == 63) {
        s_part[wave] = incl;
        s_part[16 + wave] = occ;
      }
      __syncthreads();
      // the sixteen wave totals: every wave scans them again in its first row of lanes and picks its own base
      const unsigned tot16 = wave_scan_u32_dpp(s_part[lane & 15]);
      const unsigned occ16 = wave_scan_u32_dpp(s_part[16 + (lane & 15)]);
      const unsigned nocc = (unsigned)__builtin_amdgcn_readlane((int)occ16, 15);
      const int gmax = max(gx, max(gy, gz));
      if (round < kRounds && !((float)nr >= 2.5f * (float)nocc || gmax <= 4)) {  // too fine: coarsen and count again
        int gn = max(4, (int)((float)gmax * 0.7071f));  // (half the cells of a surface: the general path's step)
#if PP_BUILD_JUMP
        // ... further where the occupancy says that step will not do: with nr / nocc = r the cells hold about
        // lambda = 2 (r - 1) points each in the mean (r = lambda / (1 - exp(-lambda)) to first order: an overestimate),
        // 2.5 points per occupied cell need lambda = 2.23, and were the cloud a filled VOLUME -- the case that shrinks
        // slowest -- the cells would have to be lambda / 2.23 as many: g' = g cbrt(lambda / 2.23).  A filled cube went
        // 32 -> 22 -> 15 (two more passes over the points, 4.9 points per cell at the end); it now goes 32 -> 19 (2.4).
        // A step never finer than the general path's, so a surface or an object's faces coarsen as before.
        {
          const float lam = 2.0f * ((float)nr / (float)nocc - 1.0f);
          const float want = (float)gmax * (float)gmax * (float)gmax * lam * (1.0f / 2.6f);  // (2.6: a margin, so that the next count passes)
          while (gn > 4 && (float)gn * (float)gn * (float)gn > want) --gn;
        }
#endif
        set_resolution(gn);
        __syncthreads();  // (s_part, the bitmap and the counters are rewritten)
        continue;
      }
      const unsigned wbase = wave == 0 ? 0u : (unsigned)__builtin_amdgcn_readlane((int)tot16, (wave - 1) & 15);
      own = (unsigned)__builtin_amdgcn_readlane((int)tot16, 15);
      running = s_below;  // points of the slabs below: where this slab's first cell starts
      if (t == 0) {  // the plan, for the general path should this slab need it (LDS, behind everything that path uses)
        BuildPlan* plan = reinterpret_cast<BuildPlan*>(lds + kFastLdsPlan);
        plan->mnx = mnx; plan->mny = mny; plan->mnz = mnz; plan->h = h; plan->invh = invh;
        plan->gx = gx; plan->gy = gy; plan->gz = gz; plan->cell_lo = lo; plan->cell_hi = lo + ncs; plan->trimmed = trimmed ? 1 : 0;
      }
#if !PP_BUILD_SPILL
      if (own > (unsigned)kFastCap) return 2;  // more points than the list holds: the general path sorts this slab (uniform)
#endif
      unsigned run = running + wbase + incl - sum;
      if (c0 < ncs) {
        unsigned st[8], mx8 = 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          st[u] = run;
          run += v8[u];
          mx8 = max(mx8, v8[u]);
        }
        if (!REFINE) {  // (no second level: a cell this full makes the grid useless, the general path's rule)
          if ((float)mx8 > 256.0f + (float)nr * (1.0f / 32.0f)) atomicOr(&s_ncrowd, 1u);
        } else if (mx8 > (unsigned)kCrowd) {  // (rare) crowded cells: listed for the refinement
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (v8[u] > (unsigned)kCrowd) {
              const unsigned at = atomicAdd(&s_ncrowd, 1u);
              if (at < (unsigned)kFastCrowdMax) {
                s_clist[at][0] = st[u];
                s_clist[at][1] = v8[u];
              }
            }
        }
        *reinterpret_cast<uint4*>(&s_cnt[c0]) = make_uint4(st[0], st[1], st[2], st[3]);
        *reinterpret_cast<uint4*>(&s_cnt[c0 + 4]) = make_uint4(st[4], st[5], st[6], st[7]);
      }
      break;
    }
  }
  const int ncell = gxy * gz;
  const int zl = lo / gxy, zh = (lo + ncs) / gxy;  // this slab's layers (as the last round dealt them)
  __syncthreads();
  PP_PHASE(6);
  if (((lo | ncs) & 3) == 0 && (reinterpret_cast<uintptr_t>(cell_start) & 15) == 0) {  // (uniform) 16-byte pieces
    for (int c = 4 * t; c < ncs; c += 4 * kBuildThreads)
      *reinterpret_cast<uint4*>(&cell_start[lo + c]) = *reinterpret_cast<const uint4*>(&s_cnt[c]);
  } else {
    for (int c = t; c < ncs; c += kBuildThreads) cell_start[lo + c] = s_cnt[c];  // (coalesced)
  }
  if (REFINE && t < zh - zl) layers[zl + t] = s_cnt[t * gxy];
  // ---- position = cell start + rank: the slab's points into the list in sorted order
  // (no branch per point: the other lanes' records go to 64 slots behind the list)
  // (round 5) a slab with more points than the list holds -- the middle slabs of a Gaussian at 32 layers, which the
  // histogram cannot deal more evenly: a slab's counters hold eight such layers -- no longer leaves for the general path
  // (a second sort of the slab from scratch: 15-20 us more for the launch): its points go from the registers STRAIGHT to
  // their places (scattered 16-byte stores, the general path's way) and its chunk table through two LDS atomics per point.
  const bool spill = PP_BUILD_SPILL && own > (unsigned)kFastCap;  // (uniform)
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    const bool mine = key[i] != ~0u;
#if PP_BUILD_STAGE_MASKED
    if (mine) {
      const unsigned pos = s_cnt[key[i] & 0xffffu] + (key[i] >> 16) - running;
      f4 r = {px[i], py[i], pz[i], __int_as_float(kidx(i))};
      if (spill) {
        sorted[pos + running] = r;
        if (REFINE) {
          const unsigned ch = (pos + running) / (unsigned)kChunk;
          atomicMin(&s_tz[2 * ch], zkey(pz[i]));
          atomicMax(&s_tz[2 * ch + 1], zkey(pz[i]));
        }
      } else {
        s_list[pos] = r;
      }
    }
#else
    const unsigned start = s_cnt[mine ? (key[i] & 0xffffu) : 0u];
    const unsigned pos = mine ? start + (key[i] >> 16) - running : (unsigned)(kFastCap + lane);
    f4 r = {px[i], py[i], pz[i], __int_as_float(kidx(i))};
    s_list[pos] = r;
#endif
  }
  __syncthreads();
  PP_PHASE(7);
  // ---- copy-out: 16-byte coalesced stores, a wave per chunk of kChunk sorted positions; the chunk's pair of the chunk
  // table from the same pass (one DPP reduction per chunk; zkey is monotone: the keys of the extreme z are the extreme keys)
  if (own > 0u && !spill) {
    const unsigned ch_first = running / (unsigned)kChunk, ch_last = (running + own - 1u) / (unsigned)kChunk;
    for (unsigned ch = ch_first + (unsigned)wave; ch <= ch_last; ch += (unsigned)(kBuildThreads / 64)) {  // a wave per chunk
      float zmax = -__builtin_inff(), nzmin = -__builtin_inff();  // max of z, max of -z
      const unsigned p0 = ch * (unsigned)kChunk;
      if (p0 >= running && p0 + (unsigned)kChunk <= running + own) {  // (uniform) the whole chunk is this slab's
#pragma unroll
        for (int q = 0; q < kChunk / 64; ++q) {
          const unsigned pos = p0 + (unsigned)(q * 64 + lane);
          const f4 r = s_list[pos - running];
          sorted[pos] = r;
          zmax = fmaxf(zmax, r.z);
          nzmin = fmaxf(nzmin, -r.z);
        }
      } else {
#pragma unroll
        for (int q = 0; q < kChunk / 64; ++q) {
          const unsigned pos = p0 + (unsigned)(q * 64 + lane);
          const unsigned j = pos - running;  // (wraps below the range: not `have`)
          const bool have = j < own;
          const f4 r = s_list[have ? j : 0u];
          if (have) {
            sorted[pos] = r;
            zmax = fmaxf(zmax, r.z);
            nzmin = fmaxf(nzmin, -r.z);
          }
        }
      }
      if (!REFINE) continue;  // (no chunk table)
      zmax = wave_reduce_dpp<false>(zmax);
      nzmin = wave_reduce_dpp<false>(nzmin);
      if (lane == 0) {  // (this wave is the only writer of the slab's pair for the chunk)
        s_tz[2 * ch] = zkey(-nzmin);
        s_tz[2 * ch + 1] = zkey(zmax);
      }
    }
  }
  __syncthreads();  // (the epilogue reads the chunk table; the refinement reuses the counters and the list)
  PP_PHASE(8);
  PP_PHASE(9);
  const unsigned ncrowd = s_ncrowd;
  if (REFINE && ncrowd > (unsigned)kFastCrowdMax) return 2;  // more crowded cells than the list holds: the general path (uniform)
  if (t == 0 && slab == nslab - 1) {
    cell_start[ncell] = (unsigned)nr;
    if (REFINE) layers[gz] = (unsigned)nr;
  }
  bool refined = false;
  if constexpr (REFINE) {
    if ((t == kLayerPending || t == kLayerCursor || t == kLayerRouted) && slab == 0) layers[t] = 0u;
    for (int c = t; c < 2 * tz_chunks; c += kBuildThreads) tile_z[((size_t)(c >> 1) * nslab + slab) * 2 + (c & 1)] = s_tz[c];
    if (ncrowd > 0) {  // (uniform) second level: the general build's refinement of this slab's crowded cells, in place
      __threadfence_block();  // the copy-out's stores are read back
      __syncthreads();
      refined = true;
      grid_refine_cells(sorted, sorted2, nullptr, nullptr, sub_start, sub_desc, reinterpret_cast<unsigned*>(s_list), s_cnt,
                        s_box, s_clist, ncrowd);
    }
  }
  if (t == 0) {
    gs->crowd[slab] = REFINE ? (refined ? 2 : 0) : (ncrowd != 0u ? 1 : 0);
    if (slab == 0) {
      gs->minx = mnx; gs->miny = mny; gs->minz = mnz; gs->h = h; gs->invh = invh;
      gs->gx = gx; gs->gy = gy; gs->gz = gz;
      gs->useless = 0;
      gs->pad[0] = 0;
      gs->pad[1] = REFINE ? 1 : 0;
      gs->pad[2] = trimmed ? 1 : 0;
    }
  }
  PP_PHASE(10);
  return 0;
}

// VEC: the cloud is 16-byte aligned (read as float4).  A compile-time choice of the calling kernel: with both
// load paths inlined into one kernel the register allocator spills (17 MB of scratch traffic per build at
// config 2); the host picks the kernel with clouds_vec_aligned().
template <bool MORTON, bool VEC>
__device__ __forceinline__ void grid_build_set(const float* __restrict__ ref, int nr, GridSet* gs,
                                               unsigned* __restrict__ cell_start, f4* __restrict__ sorted,
                                               int* __restrict__ inv, unsigned* s_cnt,
                                               const float* __restrict__ payload = nullptr,
                                               float* __restrict__ sorted_payload = nullptr, int slab = 0,
                                               int nslab = 1) {
  grid_build_set_impl<MORTON, VEC, false>(ref, nr, gs, cell_start, sorted, inv, s_cnt, payload, sorted_payload, slab,
                                          nslab, nullptr, nullptr, nullptr, nullptr);
}

// The plain (z-major, no second level) build as the searches of ball_query / three_nn / knn_points call it: the
// LDS-sorted path where it applies, else the general one (with the fast path's plan forced where a slab declined).
// Dynamic LDS: max(grid_build_lds_bytes(nslab), grid_build_fast_lds_bytes()).
template <bool VEC>
__device__ __forceinline__ void grid_build_set_plain(const float* __restrict__ ref, int nr, GridSet* gs,
                                                     unsigned* __restrict__ cell_start, f4* __restrict__ sorted,
                                                     unsigned* s_cnt, int slab, int nslab) {
  int how = 1;
#ifndef PP_PLAIN_GENERAL  // (A/B builds: -DPP_PLAIN_GENERAL keeps the general path)
  if constexpr (VEC) {
    if (nslab == kBuildSlabs)
      how = grid_build_set_fast<false>(ref, nr, gs, cell_start, sorted, s_cnt, slab, nullptr, nullptr, nullptr, nullptr, 0,
                                       nullptr);
  }
#endif
  if (how == 0) return;
  __syncthreads();  // (the general path reuses the LDS the fast one was using; the plan is in place)
  grid_build_set_impl<false, VEC, false>(ref, nr, gs, cell_start, sorted, nullptr, s_cnt, nullptr, nullptr, slab, nslab, nullptr,
                                         nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                                         how == 2 ? reinterpret_cast<const BuildPlan*>(s_cnt + kFastLdsPlan) : nullptr);
}

// the same with the second level: crowded cells refined into sub-grids (sub_start: 2 * nr + 2 entries of this
// set, sub_desc: sub_desc_slots(nr) descriptors of this set; sorted2 / payload2: spare copies as large as
// sorted / sorted_payload, addressed like them)
template <bool VEC>
__device__ __forceinline__ void grid_build_set_refined(const float* __restrict__ ref, int nr, GridSet* gs,
                                                       unsigned* __restrict__ cell_start, f4* __restrict__ sorted,
                                                       unsigned* s_cnt, const float* __restrict__ payload,
                                                       float* __restrict__ sorted_payload, int slab, int nslab,
                                                       unsigned* __restrict__ sub_start,
                                                       SubGrid* __restrict__ sub_desc, f4* __restrict__ sorted2,
                                                       float* __restrict__ payload2, int* __restrict__ tile_z = nullptr,
                                                       int tz_chunks = 0, unsigned* __restrict__ layers = nullptr) {
  grid_build_set_impl<false, VEC, true>(ref, nr, gs, cell_start, sorted, nullptr, s_cnt, payload, sorted_payload, slab,
                                        nslab, sub_start, sub_desc, sorted2, payload2, tile_z, tz_chunks, layers);
}

// every batch element's cloud (base + b * n * 3 floats) is 16-byte aligned
inline bool clouds_vec_aligned(const void* a, int na, int batch) {
  return (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (batch <= 1 || na % 4 == 0);
}

}  // namespace pp
