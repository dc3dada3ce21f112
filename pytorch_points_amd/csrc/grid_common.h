// grid_common.h -- the uniform grid shared by the exact searches (chamfer_grid.hip, ball_grid.hip):
// cubic cells of side h over a cloud's bounding box (<= 32^3), points counting-sorted by cell.
#pragma once
#include "pp_common.h"

namespace pp {

constexpr int kGridMax = 32;                                  // cells per axis
constexpr int kGridCells = kGridMax * kGridMax * kGridMax;    // LDS counters: 128 KiB
constexpr int kBuildThreads = 1024;

struct GridSet {  // one per (batch, direction); written by the build kernel
  float minx, miny, minz, h, invh;
  int gx, gy, gz;
  int useless;  // 1: send every query of this set to the brute force
  int pad[7];
};
static_assert(sizeof(GridSet) == 64, "");

__device__ __forceinline__ int cell_coord(float p, float mn, float invh, int g) {
  const float f = (p - mn) * invh;
  int c = (int)f;  // truncation; negative and NaN inputs end up clamped below
  c = c < 0 ? 0 : c;
  return c > g - 1 ? g - 1 : c;
}


__device__ __forceinline__ float block_reduce(float v, bool take_max, float* s_red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = take_max ? fmaxf(v, o) : fminf(v, o);
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  float r = s_red[0];
  for (int w = 1; w < kBuildThreads / 64; ++w) r = take_max ? fmaxf(r, s_red[w]) : fminf(r, s_red[w]);
  return r;
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
// Dynamic LDS: (kGridCells + kGridCells / 32) unsigned counters, bank-skewed.
// Writes *gs, cell_start[0..ncell] (if non-null), sorted[0..nr) = (x, y, z, original index) and, if
// non-null, inv[k] = position of original point k in `sorted`; if `sorted_payload` is non-null,
// sorted_payload[pos] = payload[k] (a per-point float, e.g. a label, in the sorted order).
// MORTON = false: cells in z-major linear order (a row of cells along x is contiguous in `sorted`);
// a degenerate set (non-finite or zero extent) is marked useless and `sorted` is left unwritten.
// MORTON = true: cells in Morton order -- `sorted` is then just a spatially coherent permutation of
// the points (always written, whatever the data), for callers that walk the points in that order.
template <bool MORTON = false>
__device__ __forceinline__ void grid_build_set(const float* __restrict__ ref, int nr, GridSet* gs,
                                               unsigned* __restrict__ cell_start, f4* __restrict__ sorted,
                                               int* __restrict__ inv, unsigned* s_cnt,
                                               const float* __restrict__ payload = nullptr,
                                               float* __restrict__ sorted_payload = nullptr) {
  __shared__ float s_red[kBuildThreads / 64];
  __shared__ unsigned s_part[kBuildThreads];
  __shared__ int s_bad;
  __shared__ float s_box[(kBuildThreads / 64) * 6];
  const int t = threadIdx.x;

  // A thread keeps KP points in registers (one chunk = 1024*KP points; a single chunk covers
  // 16384 points, so the three passes read the cloud from memory once).  Loads are unconditional
  // (index clamped) so that all KP are in flight together.
  constexpr int KP = 16;
  const int nchunks = (nr + kBuildThreads * KP - 1) / (kBuildThreads * KP);
  float px[KP], py[KP], pz[KP];
  auto load_chunk = [&](int base) {
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      int k = base + t + kBuildThreads * i;
      k = k < nr ? k : nr - 1;
      px[i] = ref[3 * (size_t)k];
      py[i] = ref[3 * (size_t)k + 1];
      pz[i] = ref[3 * (size_t)k + 2];
    }
  };
  load_chunk(0);
  // bounding box + finiteness (the clamped duplicates do not change either)
  float mnx = __builtin_inff(), mny = mnx, mnz = mnx, mxx = -mnx, mxy = -mnx, mxz = -mnx;
  bool bad = false;
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch > 0) load_chunk(ch * kBuildThreads * KP);
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const float x = px[i], y = py[i], z = pz[i];
      bad |= !(__builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z));
      mnx = fminf(mnx, x); mny = fminf(mny, y); mnz = fminf(mnz, z);
      mxx = fmaxf(mxx, x); mxy = fmaxf(mxy, y); mxz = fmaxf(mxz, z);
    }
  }
  if (t == 0) s_bad = 0;
  __syncthreads();
  if (bad) s_bad = 1;
  {  // six reductions with one barrier pair: max of (-min) and max
    float v[6] = {-mnx, -mny, -mnz, mxx, mxy, mxz};
#pragma unroll
    for (int e = 0; e < 6; ++e)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v[e] = fmaxf(v[e], __shfl_xor(v[e], off));
    if ((t & 63) == 0)
#pragma unroll
      for (int e = 0; e < 6; ++e) s_box[(t >> 6) * 6 + e] = v[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      float r = s_box[e];
      for (int w = 1; w < kBuildThreads / 64; ++w) r = fmaxf(r, s_box[w * 6 + e]);
      v[e] = r;
    }
    mnx = -v[0]; mny = -v[1]; mnz = -v[2]; mxx = v[3]; mxy = v[4]; mxz = v[5];
  }
  const float ex = mxx - mnx, ey = mxy - mny, ez = mxz - mnz;
  const float emax = fmaxf(ex, fmaxf(ey, ez));
  // ~2 points per cell if the cloud filled its box; cubic cells of side h
  int g0 = (int)ceilf(cbrtf(2.0f * (float)nr));
  g0 = g0 < 1 ? 1 : (g0 > kGridMax ? kGridMax : g0);
  const bool degenerate = s_bad || !(emax > 0.0f) || !__builtin_isfinite(emax);
  float h = emax / (float)g0;
  if (!(h > 0.0f) || !__builtin_isfinite(h)) h = 1.0f;
  const float invh = 1.0f / h;
  auto cells = [&](float e) {
    int g = (int)(e * invh) + 1;  // g*h > e: the box maximum lies inside the last cell
    return g < 1 ? 1 : (g > kGridMax ? kGridMax : g);
  };
  const int gx = degenerate ? 1 : cells(ex), gy = degenerate ? 1 : cells(ey), gz = degenerate ? 1 : cells(ez);
  const int ncell = MORTON ? kGridCells : gx * gy * gz;

  auto sk = [](int c) { return c + (c >> 5); };
  for (int c = t; c < ncell; c += kBuildThreads) s_cnt[sk(c)] = 0;
  __syncthreads();
  auto cell_of = [&](float x, float y, float z) {
    const int cx = cell_coord(x, mnx, invh, gx), cy = cell_coord(y, mny, invh, gy), cz = cell_coord(z, mnz, invh, gz);
    return MORTON ? morton3(cx, cy, cz) : (cz * gy + cy) * gx + cx;
  };
  const bool place = MORTON || !degenerate;
  if (place)
    for (int ch = 0; ch < nchunks; ++ch) {
      if (nchunks > 1) load_chunk(ch * kBuildThreads * KP);
#pragma unroll
      for (int i = 0; i < KP; ++i)
        if (ch * kBuildThreads * KP + t + kBuildThreads * i < nr) atomicAdd(&s_cnt[sk(cell_of(px[i], py[i], pz[i]))], 1u);
    }
  __syncthreads();
  // exclusive scan: each thread owns a contiguous run of cells
  const int per = (ncell + kBuildThreads - 1) / kBuildThreads;
  const int c0 = t * per, c1 = min(ncell, c0 + per);
  unsigned sum = 0, mx = 0;
  for (int c = c0; c < c1; ++c) {
    sum += s_cnt[sk(c)];
    mx = max(mx, s_cnt[sk(c)]);
  }
  const float fmx = block_reduce((float)mx, true, s_red);
  // exclusive scan of the 1024 per-thread sums: inclusive scan inside each wave (shuffles), then
  // the 16 wave totals
  unsigned incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off);
    if ((t & 63) >= off) incl += o;
  }
  if ((t & 63) == 63) s_part[t >> 6] = incl;
  __syncthreads();
  unsigned wave_base = 0;
  for (int w = 0; w < (t >> 6); ++w) wave_base += s_part[w];
  unsigned run = wave_base + incl - sum;
  for (int c = c0; c < c1; ++c) {
    const unsigned v = s_cnt[sk(c)];
    s_cnt[sk(c)] = run;  // cell start; becomes the scatter cursor below
    run += v;
  }
  __syncthreads();
  if (cell_start) {
    for (int c = t; c < ncell; c += kBuildThreads) cell_start[c] = s_cnt[sk(c)];  // coalesced copy out
    if (t == 0) cell_start[ncell] = degenerate ? 0u : (unsigned)nr;
  }
  __syncthreads();
  if (place)
    for (int ch = 0; ch < nchunks; ++ch) {
      if (nchunks > 1) load_chunk(ch * kBuildThreads * KP);
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        const int k = ch * kBuildThreads * KP + t + kBuildThreads * i;
        if (k < nr) {
          const unsigned pos = atomicAdd(&s_cnt[sk(cell_of(px[i], py[i], pz[i]))], 1u);
          f4 v = {px[i], py[i], pz[i], __int_as_float(k)};
          sorted[pos] = v;
          if (inv) inv[k] = (int)pos;  // coalesced: k is thread-strided
          if (sorted_payload) sorted_payload[pos] = payload[k];  // one float per point, carried along
        }
      }
    }
  if (t == 0) {
    GridSet g;
    g.minx = mnx; g.miny = mny; g.minz = mnz; g.h = h; g.invh = invh;
    g.gx = gx; g.gy = gy; g.gz = gz;
    // useless: degenerate, or one cell holds so many points that scanning it approaches a brute force
    g.useless = (degenerate || fmx > 64.0f + 0.25f * (float)nr) ? 1 : 0;
    for (int i = 0; i < 7; ++i) g.pad[i] = 0;
    *gs = g;
  }
}

}  // namespace pp
