// chamfer_slab.hip -- nndistance forward for evenly sampled clouds of BASELINE config 2's size class, as ONE kernel
// that sorts and searches inside the LDS (round 4; replaces the reference's NmDistanceKernel x 2,
// _ext/nmdistance_cuda.cu:7-49,118-140, with the same outputs bit for bit).
//
// The three-launch search of chamfer_grid.hip (build -> stage A by tiles -> list kernel) writes both clouds to HBM in
// sorted order and reads them back: at config 2 it moves 3.4x the algorithmic bytes and spends most of its 61 us in
// launch ramps, barrier-separated phases and dependent round trips, not in arithmetic.  Here a batch element is cut
// into kSlabs slabs of grid layers along z, one 1024-thread workgroup each.  A workgroup
//   1. reads both clouds of its batch element (L2) for their common bounding box -> a 32^3 grid of cubic cells;
//   2. reads them again and counts, per cloud, the points of ITS layers plus one layer either side (the halo) per cell
//      (16-bit counters packed in pairs, LDS atomics);
//   3. scans the counters and reads the clouds a third time, scattering those points -- (x, y, z, original index) --
//      into the LDS in cell order: two sorted images of ~3000 points, never written to memory;
//   4. answers the queries of its own layers (both directions) from those images with the searches of
//      chamfer_grid.hip's stage A: the 2x2x2 block of cells nearest to the query, settled if the best candidate lies
//      strictly within what the block guarantees (`reach`, the same expression and the same 0.999 slack); else the
//      cube of Chebyshev radius 1 (it lies inside the halo by construction); else -- a few queries in a million on a
//      surface -- an every-pair scan of the other cloud by the whole wave.  Candidates are compared in the exact
//      (distance, original index) order with pp::chamfer_d3, so the result is the brute force's, bit for bit.
// Nothing is exchanged between workgroups.  A slab whose images do not fit, whose cloud is degenerate or which meets
// too many unsettled queries (a cloud this form is not made for: volumes, clusters, far clouds) DECLINES: every
// workgroup leaves a word saying "served" or "declined" (written unconditionally, so the words need no initialisation),
// and the launches that follow -- chamfer_grid.hip's build and whole-search kernels -- skip the batch elements all of
// whose slabs were served and redo the others in full.  A declined element therefore costs this kernel's first passes
// (a few microseconds) on top of the older path; a served one costs nothing there but two early-exit launches.
#include "grid_common.h"

// phase probe (tools/build_variant_lib.sh with SRC=chamfer_slab -DPP_SLAB_STOP=n): leave after phase n, results unwritten
#ifndef PP_SLAB_STOP
#define PP_SLAB_STOP 0
#endif
#define PP_SLAB_PHASE_END(n)                  \
  if (PP_SLAB_STOP == (n)) {                  \
    if (threadIdx.x == 0) *my_state = kServed; \
    return;                                   \
  }

namespace ppslab {

using pp::f4;
using pp::cell_coord;

constexpr int kSlabs = pp::kSlabKernelSlabs;   // workgroups per batch element
constexpr int kG = 32;                         // cells per axis
constexpr int kOwnLayers = kG / kSlabs;        // 4
constexpr int kLocLayers = kOwnLayers + 2;     // + a halo layer either side
constexpr int kLocCells = kG * kG * kLocLayers;  // 6144 cells per cloud
constexpr int kCap = 3584;                     // points of one cloud a slab can hold (own layers + halo)
constexpr int kPad = 4;                        // points that can never be taken, behind an image's end
constexpr int kTabWords = 3076;                // 16-bit entries 0 .. kLocCells (a sentinel) and one spare, in pairs; 16-byte multiple
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kQueue = 128;                    // queries of a workgroup their 2x2x2 block may leave unsettled
constexpr int kMaxLeft = 16;                   // ... and the cube of radius 1 after it (every-pair scans by the workgroup)
constexpr unsigned kServed = pp::kSlabServed, kDeclined = pp::kSlabDeclined;
constexpr float kBoundSlack = 0.999f;          // (chamfer_grid.hip: the same)
static_assert(kTabWords * 2 >= kLocCells + 3 && kTabWords % 4 == 0 && kLocCells / 2 == 6 * (kThreads / 2), "");

// dynamic LDS: the two images, the two cell tables, the queue
constexpr size_t kImgBytes = (size_t)(kCap + kPad) * sizeof(f4);
constexpr size_t kOffTab = 2 * kImgBytes;
constexpr size_t kOffQueue = kOffTab + (size_t)2 * kTabWords * sizeof(unsigned);
constexpr size_t kLdsBytes = kOffQueue + (size_t)kQueue * sizeof(f4);

typedef const f4 __attribute__((address_space(3))) * lds_f4_ptr;
typedef const char __attribute__((address_space(3))) * lds_c_ptr;

__device__ __forceinline__ unsigned tab_get(const unsigned* tab, int c) {  // entry c of a packed table
  const unsigned w = tab[c >> 1];
  return (c & 1) ? (w >> 16) : (w & 0xFFFFu);
}
// entries c, c + 1, c + 2
__device__ __forceinline__ void tab_get3(const unsigned* tab, int c, unsigned& e0, unsigned& e1, unsigned& e2) {
  const unsigned w0 = tab[c >> 1], w1 = tab[(c >> 1) + 1];
  const bool odd = c & 1;
  const unsigned lo = odd ? ((w0 >> 16) | (w1 << 16)) : w0;
  e0 = lo & 0xFFFFu;
  e1 = lo >> 16;
  e2 = odd ? (w1 >> 16) : (w1 & 0xFFFFu);
}
// byte position in the image of group k of a lane's sequence (chamfer_grid.hip: lean_group_pos; everything by value)
__device__ __forceinline__ unsigned group_pos(unsigned k, unsigned T1, unsigned T2, unsigned T3, unsigned a0, unsigned a1,
                                              unsigned a2, unsigned a3, unsigned endb) {
  const unsigned a = k < T1 ? a0 : (k < T2 ? a1 : (k < T3 ? a2 : a3));
  return min(a + (k << 6), endb);
}
__device__ __forceinline__ float min2(float a, float b) {  // (chamfer_grid.hip: v_min_f32 without canonicalisation)
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// every point of the two clouds (k < N: point k of cloud 1, else point k - N of cloud 2).  A lane takes FOUR consecutive
// points as three 16-byte loads (48 bytes; the clouds are 16-byte aligned and N, M multiples of four: the host checks):
// a 12-byte load per lane costs the texture unit three passes of a 768-byte span each (measured: 8 us per pass over the
// two clouds of config 2, against 2.7 for this form); kGroupBatch groups in flight per lane.
constexpr int kGroupBatch = 4;
template <typename F>
__device__ __forceinline__ void for_all_points(const float* __restrict__ c1, const float* __restrict__ c2, int N, int M,
                                               int t, F&& f) {
  const f4* __restrict__ v1 = reinterpret_cast<const f4*>(c1);
  const f4* __restrict__ v2 = reinterpret_cast<const f4*>(c2);
  const int G1 = N >> 2, G = G1 + (M >> 2);
  for (int g0 = 0; g0 < G; g0 += kGroupBatch * kThreads) {
    f4 a[kGroupBatch][3];
#pragma unroll
    for (int u = 0; u < kGroupBatch; ++u) {
      const int g = min(g0 + u * kThreads + t, G - 1);
      const f4* __restrict__ src = g < G1 ? v1 + 3 * (size_t)g : v2 + 3 * (size_t)(g - G1);
      a[u][0] = src[0];
      a[u][1] = src[1];
      a[u][2] = src[2];
    }
#pragma unroll
    for (int u = 0; u < kGroupBatch; ++u) {
      const int g = g0 + u * kThreads + t;
      if (g < G) {
        const int k = g < G1 ? 4 * g : N + 4 * (g - G1);
        f(k, a[u][0].x, a[u][0].y, a[u][0].z);
        f(k + 1, a[u][0].w, a[u][1].x, a[u][1].y);
        f(k + 2, a[u][1].z, a[u][1].w, a[u][2].x);
        f(k + 3, a[u][2].y, a[u][2].z, a[u][2].w);
      }
    }
  }
}

__global__ __launch_bounds__(kThreads) void chamfer_slab_kernel(const float* __restrict__ xyz1,
                                                                const float* __restrict__ xyz2,
                                                                float* __restrict__ dist1, int* __restrict__ idx1,
                                                                float* __restrict__ dist2, int* __restrict__ idx2,
                                                                unsigned* __restrict__ state, int B, int N, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  f4* s_pts = reinterpret_cast<f4*>(s_raw);                      // [2][kCap + kPad]
  unsigned* s_tab = reinterpret_cast<unsigned*>(s_raw + kOffTab);  // [2][kTabWords]
  f4* s_queue = reinterpret_cast<f4*>(s_raw + kOffQueue);        // [kQueue]: x, y, z, original index | direction << 30
  __shared__ float s_box[kWaves][6];
  __shared__ unsigned s_sel[2];  // points in the image of cloud 0 / 1
  __shared__ unsigned s_wsum[kWaves];
  __shared__ unsigned s_qn, s_ln;
  __shared__ unsigned s_left[kMaxLeft];
  __shared__ unsigned long long s_bf[kMaxLeft];

  // the slabs of a batch element share an XCD (its clouds stay in that L2): speed only
  const int per_xcd = (B * kSlabs + 7) / 8;
  const int V = pp::xcd_virtual_block((int)blockIdx.x, per_xcd);
  if (V >= B * kSlabs) return;
  const int b = V / kSlabs, slab = V - b * kSlabs;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = pp::wave_id_uniform();
  const float* __restrict__ c1 = xyz1 + (size_t)b * N * 3;
  const float* __restrict__ c2 = xyz2 + (size_t)b * M * 3;
  unsigned* __restrict__ my_state = state + (size_t)b * kSlabs + slab;
  auto decline = [&]() {
    if (t == 0) *my_state = kDeclined;
  };
  PP_SLAB_PHASE_END(9)

  // ---------------------------------------------------------------- 1. common bounding box
  {
    float v[6] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};  // -lo, hi
    float bad = 0.0f;
    for_all_points(c1, c2, N, M, t, [&](int, float x, float y, float z) {
      v[0] = fmaxf(v[0], -x); v[1] = fmaxf(v[1], -y); v[2] = fmaxf(v[2], -z);
      v[3] = fmaxf(v[3], x);  v[4] = fmaxf(v[4], y);  v[5] = fmaxf(v[5], z);
      bad += (x - x) + (y - y) + (z - z);  // 0 for finite coordinates, NaN otherwise
    });
    if (!(bad == 0.0f)) v[3] = INFINITY;  // a non-finite coordinate anywhere: the extent below becomes unusable
    pp::wave_reduce6_dpp<false, 6>(v);
    if (lane == 63)
      for (int a = 0; a < 6; ++a) s_box[wave][a] = v[a];
  }
  if (t == 0) {
    s_qn = 0u;
    s_ln = 0u;
  }
  if (t < kMaxLeft) s_bf[t] = ~0ull;
  for (int i = t; i < 2 * kTabWords; i += kThreads) s_tab[i] = 0u;
  __syncthreads();
  float bv[6];
  for (int a = 0; a < 6; ++a) {
    float m = s_box[0][a];
    for (int w = 1; w < kWaves; ++w) m = fmaxf(m, s_box[w][a]);
    bv[a] = m;
  }
  const float minx = -bv[0], miny = -bv[1], minz = -bv[2];
  const float ext = fmaxf(bv[3] + bv[0], fmaxf(bv[4] + bv[1], bv[5] + bv[2]));
  if (!(ext > 0.0f && ext < INFINITY)) {  // degenerate or non-finite: not for this kernel (uniform over the workgroup)
    decline();
    return;
  }
  PP_SLAB_PHASE_END(1)
  const float h = ext * (1.0f / (float)kG) * 1.0001f, invh = 1.0f / h;
  const int zbase = slab * kOwnLayers - 1;  // the layer that is local layer 0 (the lower halo; -1 for slab 0)
  auto local_cell = [&](float x, float y, float z, bool& mine) -> int {
    const int cz = cell_coord(z, minz, invh, kG);
    const int lz = cz - zbase;
    mine = lz >= 0 && lz < kLocLayers;
    return pp::cell_linear(cell_coord(x, minx, invh, kG), cell_coord(y, miny, invh, kG), lz, kG, kG);
  };

  // ---------------------------------------------------------------- 2. count (per cloud: 16-bit counters, two to a word)
  for_all_points(c1, c2, N, M, t, [&](int k, float x, float y, float z) {
    const int cl = k < N ? 0 : 1;
    bool mine;
    const int c = local_cell(x, y, z, mine);
    if (mine) atomicAdd(&s_tab[cl * kTabWords + (c >> 1)], (c & 1) ? 0x10000u : 1u);
  });
  __syncthreads();
  PP_SLAB_PHASE_END(2)
  // ---------------------------------------------------------------- 3. scan: entry c := END of cell c; then scatter
  // (thread t: twelve cells of cloud t / 512; a cloud's points number < 65536: no entry overflows)
  {
    const int cl = t >> 9, tt = t & 511;
    unsigned* __restrict__ tab = s_tab + cl * kTabWords + tt * 6;
    unsigned w[6];
    unsigned sum = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      w[i] = tab[i];
      sum += (w[i] & 0xFFFFu) + (w[i] >> 16);
    }
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (int ww = (wave < 8 ? 0 : 8); ww < wave; ++ww) run += s_wsum[ww];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const unsigned lo = run + (w[i] & 0xFFFFu), hi = lo + (w[i] >> 16);
      tab[i] = lo | (hi << 16);
      run = hi;
    }
    if (tt == 511) {
      tab[6] = run;  // the sentinel (entry kLocCells): the image's size
      s_sel[cl] = run;
    }
  }
  __syncthreads();
  const unsigned ns0 = s_sel[0], ns1 = s_sel[1];
  if (ns0 > (unsigned)kCap || ns1 > (unsigned)kCap) {  // the images do not fit (uniform)
    decline();
    return;
  }
  if (t < 2 * kPad) {  // the padding: points whose distance to anything is NaN
    const float qn = __builtin_nanf("");
    const f4 nanp = {qn, qn, qn, __int_as_float(0x7fffffff)};
    s_pts[(t >> 2) * (kCap + kPad) + ((t >> 2) ? ns1 : ns0) + (t & 3)] = nanp;
  }
  // a point takes the slot below its cell's END and lowers it: afterwards entry c is the START of cell c
  for_all_points(c1, c2, N, M, t, [&](int k, float x, float y, float z) {
    const int cl = k < N ? 0 : 1;
    bool mine;
    const int c = local_cell(x, y, z, mine);
    if (mine) {
      const unsigned old = atomicSub(&s_tab[cl * kTabWords + (c >> 1)], (c & 1) ? 0x10000u : 1u);
      const unsigned pos = ((c & 1) ? (old >> 16) : (old & 0xFFFFu)) - 1u;
      const f4 rec = {x, y, z, __int_as_float(cl ? k - N : k)};
      s_pts[cl * (kCap + kPad) + pos] = rec;
    }
  });
  __syncthreads();
  PP_SLAB_PHASE_END(3)

  // ---------------------------------------------------------------- 4. the 2x2x2 blocks, both directions
  // (chamfer_grid.hip's stage A: the same walk -- groups of four consecutive points of the block's four rows, the
  //  running minimum and the group that last lowered it, the exact order recovered afterwards -- over an image that
  //  was never in memory)
  const float inf = INFINITY;
  constexpr int g1 = kG - 1;
  for (int dir = 0; dir < 2; ++dir) {
    if (dir == 1) { PP_SLAB_PHASE_END(4) }
    const f4* __restrict__ qpts = s_pts + dir * (kCap + kPad);
    const unsigned* __restrict__ qtab = s_tab + dir * kTabWords;
    const unsigned* __restrict__ rtab = s_tab + (dir ^ 1) * kTabWords;
    const lds_c_ptr lb = (lds_c_ptr)(s_raw + (dir ^ 1) * kImgBytes);
    const unsigned endb = (dir ? ns0 : ns1) << 4;
    const int nq = dir ? M : N;
    float* __restrict__ od = (dir ? dist2 : dist1) + (size_t)b * nq;
    int* __restrict__ oi = (dir ? idx2 : idx1) + (size_t)b * nq;
    // the queries of the own layers: local layers 1 .. kOwnLayers, one contiguous piece of the image
    const unsigned q0 = tab_get(qtab, kG * kG), q1 = tab_get(qtab, kG * kG * (kOwnLayers + 1));
    for (unsigned base = q0; base < q1; base += kThreads) {  // (uniform)
      const bool valid = base + (unsigned)t < q1;
      const f4 q = qpts[valid ? base + (unsigned)t : q1 - 1u];
      const float qx = q.x, qy = q.y, qz = q.z;
      const float px = (qx - minx) * invh, py = (qy - miny) * invh, pz = (qz - minz) * invh;  // in cells
      const int cx = cell_coord(qx, minx, invh, kG), cy = cell_coord(qy, miny, invh, kG), cz = cell_coord(qz, minz, invh, kG);
      // the 2x2x2 block: cells l, l + 1 per axis (the neighbour on the side of the cell the query lies in), clamped
      const int lx = px - (float)cx < 0.5f ? cx - 1 : cx, ly = py - (float)cy < 0.5f ? cy - 1 : cy,
                lz = pz - (float)cz < 0.5f ? cz - 1 : cz;
      const int x0 = max(lx, 0), x1 = min(lx + 1, g1), y0 = max(ly, 0), y1 = min(ly + 1, g1), z0 = max(lz, 0),
                z1 = min(lz + 1, g1);
      auto face = [&](float p, int l) {  // (stage A: the same expression)
        const float lo = l >= 1 ? p - (float)l : inf;
        const float hi = l + 1 < g1 ? (float)(l + 2) - p : inf;
        return fminf(lo, hi);
      };
      const float reach = h * fminf(face(px, lx), fminf(face(py, ly), face(pz, lz)));
      const bool wide = x1 > x0;
      const bool va0 = lz >= 0, va1 = lz < g1, vb0 = ly >= 0, vb1 = ly < g1;  // the row's layer / line exists
      unsigned s0, s1, s2, s3, m, e, e0, e1, e2, e3;
      tab_get3(rtab, ((z0 - zbase) * kG + y0) * kG + x0, s0, m, e); e0 = (va0 & vb0) ? (wide ? e : m) : s0;
      tab_get3(rtab, ((z0 - zbase) * kG + y1) * kG + x0, s1, m, e); e1 = (va0 & vb1) ? (wide ? e : m) : s1;
      tab_get3(rtab, ((z1 - zbase) * kG + y0) * kG + x0, s2, m, e); e2 = (va1 & vb0) ? (wide ? e : m) : s2;
      tab_get3(rtab, ((z1 - zbase) * kG + y1) * kG + x0, s3, m, e); e3 = (va1 & vb1) ? (wide ? e : m) : s3;
      const unsigned t0 = (e0 - s0 + 3) >> 2, t1 = (e1 - s1 + 3) >> 2, t2 = (e2 - s2 + 3) >> 2, t3 = (e3 - s3 + 3) >> 2;
      const unsigned T1 = t0, T2 = T1 + t1, T3 = T2 + t2, T4 = T3 + t3;
      // group k of the lane's sequence starts at byte a_r + 64 k of the image, r the row k falls in
      const unsigned a0 = s0 << 4, a1 = (s1 << 4) - (T1 << 6), a2 = (s2 << 4) - (T2 << 6), a3 = (s3 << 4) - (T3 << 6);
      const int kmax = (int)pp::wave_reduce_dpp<false>((float)T4);
      auto pos_of = [=](unsigned k) { return group_pos(k, T1, T2, T3, a0, a1, a2, a3, endb); };
      f4 pa[4], pb[4];
      auto fetch4 = [&](unsigned pos, f4 (&p)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = *(lds_f4_ptr)(lb + pos + 16 * u);
      };
      float best = inf;
      int bidx = 0x7fffffff;
      unsigned gpos = endb;           // byte position of the group that holds the winner
      unsigned long long tie = 0ull;  // lanes that met a distance equal to their running minimum in a later group
      auto track = [&](unsigned pos, const f4 (&p)[4]) {
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
        auto examine = [&](const f4 (&p)[4]) {
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
      if (valid) {
        if (best < reach * reach * kBoundSlack) {  // settled (strict; a NaN bound settles nothing)
          const int j = __float_as_int(q.w);
          od[j] = best;
          oi[j] = bidx;
        } else {
          const unsigned slot = atomicAdd(&s_qn, 1u);
          if (slot < (unsigned)kQueue) s_queue[slot] = f4{qx, qy, qz, __int_as_float(__float_as_int(q.w) | (dir << 30))};
        }
      }
    }
  }
  __syncthreads();
  PP_SLAB_PHASE_END(5)
  // ---------------------------------------------------------------- 5. what the blocks left: the cube of radius 1, a wave per query
  // (about one query in 150 on an evenly sampled surface; the cube lies inside the image by construction)
  const unsigned nqueue = s_qn;
  if (nqueue > (unsigned)kQueue) {  // not a cloud for this kernel (uniform): the batch element is redone by the launches that follow
    decline();
    return;
  }
  for (unsigned en = (unsigned)wave; en < nqueue; en += kWaves) {  // wave-uniform
    const f4 w = s_queue[en];
    const int dir = (__float_as_int(w.w) >> 30) & 1;
    const f4* __restrict__ rpts = s_pts + (dir ^ 1) * (kCap + kPad);
    const unsigned* __restrict__ rtab = s_tab + (dir ^ 1) * kTabWords;
    const int wcx = cell_coord(w.x, minx, invh, kG), wcy = cell_coord(w.y, miny, invh, kG), wcz = cell_coord(w.z, minz, invh, kG);
    const int wx0 = max(wcx - 1, 0), wx1 = min(wcx + 1, g1);
    const int rz = wcz - 1 + lane / 3, ry = wcy - 1 + lane % 3;
    const bool rok = lane < 9 && rz >= 0 && rz <= g1 && ry >= 0 && ry <= g1;
    unsigned rs = 0, re = 0;
    if (rok) {
      const int c = ((rz - zbase) * kG + ry) * kG;
      rs = tab_get(rtab, c + wx0);
      re = tab_get(rtab, c + wx1 + 1);
    }
    const unsigned len = re - rs;
    unsigned incl = len;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const unsigned o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 15);
    const unsigned excl = incl - len, shift = rs - excl;  // candidate k of row r: image[k + shift_r]
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
      if (k < tot) {
        const f4 p = rpts[k + add];
        const float d = pp::chamfer_d3(p.x, p.y, p.z, w.x, w.y, w.z);
        const unsigned long long cand = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
        key = cand < key ? cand : key;  // (a NaN distance -- bits above +inf -- is never taken)
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)key, off), hi = __shfl_xor((unsigned)(key >> 32), off);
      const unsigned long long o = ((unsigned long long)hi << 32) | lo;
      key = o < key ? o : key;
    }
    const float kbest = __uint_as_float((unsigned)(key >> 32));
    const int kidx = (int)(unsigned)key;
    const float wfx = (w.x - minx) * invh - (float)wcx, wfy = (w.y - miny) * invh - (float)wcy,
                wfz = (w.z - minz) * invh - (float)wcz;
    auto axis1 = [&](float ff, int cc) {  // distance (cells) to the nearer face of the cube with grid beyond it
      const float lo = cc - 1 >= 1 ? 1.0f + ff : inf;
      const float hi = cc + 1 < g1 ? 2.0f - ff : inf;
      return fminf(lo, hi);
    };
    const float reach1 = h * fminf(axis1(wfx, wcx), fminf(axis1(wfy, wcy), axis1(wfz, wcz)));
    const bool settled = kidx != 0x7fffffff && kbest < reach1 * reach1 * kBoundSlack;
    if (lane == 0) {
      if (settled) {
        const int wj = __float_as_int(w.w) & 0x3fffffff;
        (dir ? dist2 : dist1)[(size_t)b * (dir ? M : N) + wj] = kbest;
        (dir ? idx2 : idx1)[(size_t)b * (dir ? M : N) + wj] = kidx;
      } else {
        const unsigned slot = atomicAdd(&s_ln, 1u);
        if (slot < (unsigned)kMaxLeft) s_left[slot] = en;
      }
    }
  }
  __syncthreads();
  // ---------------------------------------------------------------- 6. what is left: every pair, by the whole workgroup
  // (a few queries in a million on a surface: isolated points, the ends of an open sheet)
  const unsigned nleft = s_ln;
  if (nleft > (unsigned)kMaxLeft) {
    decline();
    return;
  }
  for (unsigned i = 0; i < nleft; ++i) {  // (uniform)
    const f4 w = s_queue[s_left[i]];
    const int dir = (__float_as_int(w.w) >> 30) & 1;
    const f4* __restrict__ rv = reinterpret_cast<const f4*>(dir ? c1 : c2);
    const int G = (dir ? N : M) >> 2;
    unsigned long long key = ~0ull;
    auto cand = [&](int id, float x, float y, float z) {
      const float d = pp::chamfer_d3(x, y, z, w.x, w.y, w.z);
      const unsigned long long c = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id;
      key = (d == d && c < key) ? c : key;  // (never a NaN distance)
    };
    for (int g0 = 0; g0 < G; g0 += kGroupBatch * kThreads) {
      f4 a[kGroupBatch][3];
#pragma unroll
      for (int u = 0; u < kGroupBatch; ++u) {
        const f4* __restrict__ src = rv + 3 * (size_t)min(g0 + u * kThreads + t, G - 1);
        a[u][0] = src[0];
        a[u][1] = src[1];
        a[u][2] = src[2];
      }
#pragma unroll
      for (int u = 0; u < kGroupBatch; ++u) {
        const int g = g0 + u * kThreads + t;
        if (g < G) {
          cand(4 * g, a[u][0].x, a[u][0].y, a[u][0].z);
          cand(4 * g + 1, a[u][0].w, a[u][1].x, a[u][1].y);
          cand(4 * g + 2, a[u][1].z, a[u][1].w, a[u][2].x);
          cand(4 * g + 3, a[u][2].y, a[u][2].z, a[u][2].w);
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)key, off), hi = __shfl_xor((unsigned)(key >> 32), off);
      const unsigned long long o = ((unsigned long long)hi << 32) | lo;
      key = o < key ? o : key;
    }
    if (lane == 0) atomicMin(&s_bf[i], key);
  }
  __syncthreads();
  if ((unsigned)t < nleft) {
    const f4 w = s_queue[s_left[t]];
    const int dir = (__float_as_int(w.w) >> 30) & 1;
    const int wj = __float_as_int(w.w) & 0x3fffffff;
    const unsigned long long key = s_bf[t];
    // (an empty scan cannot happen: the other cloud has >= 8192 finite points)
    (dir ? dist2 : dist1)[(size_t)b * (dir ? M : N) + wj] = __uint_as_float((unsigned)(key >> 32));
    (dir ? idx2 : idx1)[(size_t)b * (dir ? M : N) + wj] = (int)(unsigned)key;
  }
  if (t == 0) *my_state = kServed;
}

}  // namespace ppslab

namespace pp {

bool chamfer_slab_applies(const float* xyz1, const float* xyz2, int B, int N, int M) {
  // the slabs' images hold 3584 points of a cloud (4 of 32 layers + 2 halo layers of an evenly sampled cloud: 3/16 of
  // it, + 17 % of room): clouds of config 2's size class; read as 16-byte pieces, four points to a lane
  return B >= 1 && N >= 8192 && M >= 8192 && N <= 17408 && M <= 17408 && N % 4 == 0 && M % 4 == 0 &&
         (reinterpret_cast<uintptr_t>(xyz1) & 15) == 0 && (reinterpret_cast<uintptr_t>(xyz2) & 15) == 0;
}

int chamfer_slab_launch(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2, int* idx2,
                        unsigned* state, int B, int N, int M, hipStream_t s) {
  static pp::DeviceFlags lds_ok;
  hipError_t e = pp::allow_big_lds(ppslab::chamfer_slab_kernel, (int)ppslab::kLdsBytes, lds_ok);
  if (e != hipSuccess) return (int)e;
  const int grid = 8 * ((B * ppslab::kSlabs + 7) / 8);
  ppslab::chamfer_slab_kernel<<<dim3(grid), dim3(ppslab::kThreads), ppslab::kLdsBytes, s>>>(xyz1, xyz2, dist1, idx1, dist2,
                                                                                          idx2, state, B, N, M);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

}  // namespace pp
