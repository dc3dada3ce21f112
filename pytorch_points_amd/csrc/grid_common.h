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

struct GridSet {  // one per (batch, direction); written by the build kernel
  float minx, miny, minz, h, invh;
  int gx, gy, gz;
  int useless;             // 1: degenerate data (non-finite / zero extent): no grid at all
  int pad[3];              // pad[0]: free for the caller (ball_query / three_nn: "the grid path serves this set")
  int crowd[kBuildSlabs];  // crowd[s] = 1: slab s found a cell holding a large share of the points
};
static_assert(sizeof(GridSet) == 64, "");

// the grid is not worth using (or does not exist): send the set's queries to the brute force
__device__ __forceinline__ bool grid_useless(const GridSet& g) {
  return (g.useless | g.crowd[0] | g.crowd[1] | g.crowd[2] | g.crowd[3]) != 0;
}

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
template <bool MORTON, bool VEC>
__device__ __forceinline__ void grid_build_set_impl(const float* __restrict__ ref, int nr, GridSet* gs,
                                               unsigned* __restrict__ cell_start, f4* __restrict__ sorted,
                                               int* __restrict__ inv, unsigned* s_cnt,
                                               const float* __restrict__ payload,
                                               float* __restrict__ sorted_payload, int slab,
                                               int nslab) {
  __shared__ unsigned s_part[kBuildThreads];
  __shared__ unsigned s_crowd;
  __shared__ float s_box[(kBuildThreads / 64) * 16];
  static_assert(kBuildThreads / 64 == 16, "the bounding-box reduction assumes 16 waves");
  const int t = threadIdx.x;
  if (t == 0) s_crowd = 0u;

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
  auto kidx = [&](int base, int i) {
    return vec ? base + (i >> 2) * (4 * kBuildThreads) + 4 * t + (i & 3) : base + t + kBuildThreads * i;
  };
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
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        int k = base + t + kBuildThreads * i;
        k = k < nr ? k : nr - 1;
        px[i] = ref[3 * (size_t)k];
        py[i] = ref[3 * (size_t)k + 1];
        pz[i] = ref[3 * (size_t)k + 2];
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
  {  // six max-reductions (-min, max) and six sums with one barrier; a non-finite coordinate (or one whose
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
  PP_PHASE(2);
  // Outliers: a few points far from the bulk would stretch the box until the bulk sits in a handful of
  // cells.  Any box is valid -- cell_coord clamps, the points outside simply land in the boundary cells
  // and every bound is stated in terms of the (monotone) cell coordinate -- so when the box reaches
  // beyond 6 sigma of the mean on some side, it is replaced by the box of the points within 4 sigma on
  // every axis.  Uniform over the workgroup; clouds without outliers skip the second pass.
  if (!any_bad) {
    const float inv_n = 1.0f / (float)nr;
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
  const bool degenerate = any_bad || !(emax > 0.0f) || !__builtin_isfinite(emax);
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
    for (int round = 0; round < 4; ++round) {
      const int nc = gx * gy * gz;
      const int lo = (int)((long long)nc * slab / nslab), nl = (int)((long long)nc * (slab + 1) / nslab) - lo;
      for (int wd = t; wd < (nc + 31) / 32; wd += kBuildThreads) s_occ[wd] = 0;
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
#pragma unroll
          for (int i = 0; i < KP; ++i) mark(i);
        } else {
#pragma unroll
          for (int i = 0; i < KP; ++i)
            if (kidx(ch * kBuildThreads * KP, i) < nr) mark(i);
        }
      }
      __syncthreads();
      unsigned mine = 0;
      for (int wd = t; wd < (nc + 31) / 32; wd += kBuildThreads) mine += __builtin_popcount(s_occ[wd]);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
      if ((t & 63) == 0 && mine) atomicAdd(&s_nocc, mine);
      __syncthreads();
      const unsigned nocc = s_nocc;
      __syncthreads();
      const int gmax = max(gx, max(gy, gz));
      if ((float)nr >= 2.5f * (float)nocc || gmax <= 4) break;
      // coarsen by ~1/sqrt(2) per round: x2.8 points per cell for a volume, x2 for a surface
      set_resolution(max(4, (int)((float)gmax * 0.7071f)));
    }
    counted = true;
  }
  PP_PHASE(4);
  const int ncell = MORTON ? kGridCells : gx * gy * gz;

  const int cell_lo = (int)((long long)ncell * slab / nslab), cell_hi = (int)((long long)ncell * (slab + 1) / nslab);
  const int nloc = cell_hi - cell_lo;  // this slab's cells: local index = cell - cell_lo
  auto cell_of = [&](float x, float y, float z) {
    const int cx = cell_coord(x, mnx, invh, gx), cy = cell_coord(y, mny, invh, gy), cz = cell_coord(z, mnz, invh, gz);
    return MORTON ? morton3(cx, cy, cz) : cell_linear(cx, cy, cz, gx, gy);
  };
  PP_PHASE(5);
  const bool place = MORTON || !degenerate;
  if (!counted) {  // Morton mode (and degenerate sets, which only need empty counters)
    for (int c = t; c < nloc; c += kBuildThreads) s_cnt[sk(c)] = 0;
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
  // brute-force kernel's share of the cloud (flag cleared before the first barrier, read after the last)
  if ((float)mx > 256.0f + (float)nr * (1.0f / 32.0f)) s_crowd = 1u;
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
#pragma unroll
        for (int i = 0; i < 4; ++i) pos[i] = c[i] >= 0 ? atomicAdd(&s_cnt[c[i]], 1u) : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (c[i] >= 0) {
            const int k = kidx(ch * kBuildThreads * KP, i0 + i);
            f4 v = {px[i0 + i], py[i0 + i], pz[i0 + i], __int_as_float(k)};
            sorted[pos[i]] = v;
            if (inv) inv[k] = (int)pos[i];
            if (sorted_payload) sorted_payload[pos[i]] = payload[k];  // one float per point, carried along
          }
      }
    }
  PP_PHASE(9);
  if (t == 0) {
    gs->crowd[slab] = (!degenerate && s_crowd != 0u) ? 1 : 0;
    if (slab == 0) {
      gs->minx = mnx; gs->miny = mny; gs->minz = mnz; gs->h = h; gs->invh = invh;
      gs->gx = gx; gs->gy = gy; gs->gz = gz;
      gs->useless = degenerate ? 1 : 0;
      for (int i = 0; i < 3; ++i) gs->pad[i] = 0;
      for (int i = nslab; i < kBuildSlabs; ++i) gs->crowd[i] = 0;
    }
  }
  PP_PHASE(10);
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
  grid_build_set_impl<MORTON, VEC>(ref, nr, gs, cell_start, sorted, inv, s_cnt, payload, sorted_payload, slab, nslab);
}

// every batch element's cloud (base + b * n * 3 floats) is 16-byte aligned
inline bool clouds_vec_aligned(const void* a, int na, int batch) {
  return (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (batch <= 1 || na % 4 == 0);
}

}  // namespace pp
