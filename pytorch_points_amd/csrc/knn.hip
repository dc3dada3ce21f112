// knn.hip -- K nearest neighbours of every point of p1 (B,N,3) among p2 (B,M,3): the operator the
// reference obtains from pytorch3d.ops.knn_points at eight call sites (network/model_loss.py:120,147,378,
// geo_operations.py:112,139, layers.py:52,99,115; pytorch3d itself is not vendored -- SURVEY.md §8f N4).
//
// Result per query: the K smallest pairs (d, index) in lexicographic order, d = squared distance in
// the sequential form pytorch3d's kernels use (dx*dx, then fma(dy,dy,.), fma(dz,dz,.) = pp::chamfer_d3),
// ascending; ties go to the lower index.  Slots beyond the number of valid reference points hold
// (0, 0), as do the rows of queries beyond lengths1 (pytorch3d pads with zeros).
//   * knn_scan_kernel<KT>: one lane per query, reference point wave-uniform, sorted K-list in registers.
//   * knn_grid_kernel<KT>: the exact grid search of three_nn_grid.hip with a K-list: reference points
//     counting-sorted into the grid, queries Morton-sorted, a lane scans the cell box
//     [cell(q - R), cell(q + R)] and stops when its K-th best is below 0.9999 reach^2 (reach = what the
//     rounded box bounds guarantee on every axis); otherwise R doubles.  Same bits as the scan.
#include "grid_common.h"

namespace {

using pp::GridSet;
using pp::cell_coord;
using pp::kGridCells;
using pp::kBuildThreads;

// ascending K-list in registers; (d, k) enters if it is lexicographically smaller than an entry
template <int KT>
struct KList {
  float d[KT];
  int i[KT];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      d[j] = __builtin_inff();
      i[j] = 0x7fffffff;
    }
  }
  __device__ __forceinline__ bool beats_last(float nd, int ni) const {
    // (bitwise operators here and in insert(): hipcc turns the short-circuit forms into exec-mask branches)
    return (nd < d[KT - 1]) | ((nd == d[KT - 1]) & (ni < i[KT - 1]));
  }
  __device__ __forceinline__ void insert(float nd, int ni) {
    bool lt[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) lt[j] = (nd < d[j]) | ((nd == d[j]) & (ni < i[j]));
#pragma unroll
    for (int j = KT - 1; j >= 1; --j) {
      d[j] = lt[j - 1] ? d[j - 1] : (lt[j] ? nd : d[j]);
      i[j] = lt[j - 1] ? i[j - 1] : (lt[j] ? ni : i[j]);
    }
    d[0] = lt[0] ? nd : d[0];
    i[0] = lt[0] ? ni : i[0];
  }
};

// The same list with (distance bits << 32 | index) keys: for distances that are >= +0 and not NaN -- sums of squares --
// the unsigned order of the keys IS the (distance, index) order, so an entry is one 64-bit compare and two 64-bit
// selects instead of three compares, two logic operations and four selects (the grid kernel's K-list: a NaN distance
// gets the all-ones key and never enters, as in the two-array form).
template <int KT>
struct KList64 {
  unsigned long long k[KT];
  static __device__ __forceinline__ unsigned long long key(float nd, int ni) {
    return nd == nd ? (((unsigned long long)__float_as_uint(nd) << 32) | (unsigned)ni) : ~0ull;
  }
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < KT; ++j) k[j] = ((unsigned long long)0x7f800000u << 32) | 0x7fffffffu;  // (inf, no index)
  }
  __device__ __forceinline__ bool beats_last(unsigned long long nk) const { return nk < k[KT - 1]; }
  __device__ __forceinline__ void insert(unsigned long long nk) {
    bool lt[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) lt[j] = nk < k[j];
#pragma unroll
    for (int j = KT - 1; j >= 1; --j) k[j] = lt[j - 1] ? k[j - 1] : (lt[j] ? nk : k[j]);
    k[0] = lt[0] ? nk : k[0];
  }
  __device__ __forceinline__ float dist(int j) const { return __uint_as_float((unsigned)(k[j] >> 32)); }
  __device__ __forceinline__ int index(int j) const { return (int)(unsigned)k[j]; }
};

template <int KT>
__device__ __forceinline__ void knn_store(const KList<KT>& L, float* __restrict__ od, int* __restrict__ oi, int K,
                                          int nvalid) {
  // slots beyond the valid reference points: (0, 0)
#pragma unroll
  for (int j = 0; j < KT; ++j)
    if (j < K) {
      const bool real = j < nvalid;
      od[j] = real ? L.d[j] : 0.0f;
      oi[j] = (real && L.i[j] != 0x7fffffff) ? L.i[j] : 0;  // a slot no point entered (NaN distances): (inf, 0)
    }
}

template <int KT>
__global__ __launch_bounds__(256) void knn_scan_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                       const int* __restrict__ len1, const int* __restrict__ len2,
                                                       float* __restrict__ dist, int* __restrict__ idx, int N, int M,
                                                       int K, int tiles_per_b, const GridSet* __restrict__ skip) {
  const int b = blockIdx.x / tiles_per_b;
  if (skip && skip[b].pad[0]) return;  // this batch element was handled by the grid search
  const int tile = blockIdx.x - b * tiles_per_b;
  const int n = tile * 256 + threadIdx.x;
  const int n1 = len1 ? min(max(len1[b], 0), N) : N;
  const int m2 = len2 ? min(max(len2[b], 0), M) : M;
  if (n >= N) return;
  float* od = dist + ((size_t)b * N + n) * K;
  int* oi = idx + ((size_t)b * N + n) * K;
  if (n >= n1) {  // a padded query row
    for (int j = 0; j < K; ++j) {
      od[j] = 0.0f;
      oi[j] = 0;
    }
    return;
  }
  const float* __restrict__ q = p1 + ((size_t)b * N + n) * 3;
  const float* __restrict__ r = p2 + (size_t)b * M * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  KList<KT> L;
  L.clear();
  for (int k = 0; k < m2; ++k) {  // r[...] is wave-uniform: scalar loads
    const float d = pp::chamfer_d3(r[3 * (size_t)k], r[3 * (size_t)k + 1], r[3 * (size_t)k + 2], qx, qy, qz);
    if (__any(L.beats_last(d, k))) L.insert(d, k);
  }
  knn_store<KT>(L, od, oi, K, m2);
}

// Any point dimension D and any K up to 128 (the reference's DenseEdgeConv searches in FEATURE space:
// network/layers.py:52,99, D = channel count, K = k + 1).  One wave per 64 queries, a lane per query.  The
// queries' coordinates sit transposed in LDS ([D/4][64][4]: a lane reads four of its own dimensions with one
// conflict-free ds_read_b128); reference points are wave-uniform (scalar loads), eight at a time, each with its
// own accumulator, so the distance is the SEQUENTIAL fma chain over the dimensions -- d = fma(t_c, t_c, d),
// c = 0 .. D-1, the order pytorch3d's kernels and oracle.knn use -- and the neighbour order is bit-exact.
// Padding dimensions (D up to the next multiple of 8) are zeros on both sides: fma(0, 0, d) == d.
template <int KT>
__global__ __launch_bounds__(256) void knn_nd_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                     const int* __restrict__ len1, const int* __restrict__ len2,
                                                     float* __restrict__ dist, int* __restrict__ idx, int N, int M,
                                                     int D, int K, int tiles_per_b) {
  // [Dp / 4][64][4] query coordinates, then one wave's K-list in transit: [KT][64] distances, [KT][64] indices
  extern __shared__ __attribute__((aligned(16))) float s_q[];
  const int b = blockIdx.x / tiles_per_b;
  const int tile = blockIdx.x - b * tiles_per_b;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = pp::wave_id_uniform();
  const int n = tile * 64 + lane;
  const int n1 = len1 ? min(max(len1[b], 0), N) : N;
  const int m2 = len2 ? min(max(len2[b], 0), M) : M;
  const int Dp = (D + 7) & ~7;
  float* s_ld = s_q + (size_t)Dp * 64;
  int* s_li = reinterpret_cast<int*>(s_ld + KT * 64);
  // stage the tile's queries (coalesced over the tile's 64 * D contiguous floats)
  for (int e = t; e < 64 * Dp; e += 256) s_q[e] = 0.0f;
  __syncthreads();
  const int rows = min(64, N - tile * 64);
  const float* __restrict__ qbase = p1 + ((size_t)b * N + (size_t)tile * 64) * D;
  for (int e = t; e < rows * D; e += 256) {
    const int q = e / D, d = e - q * D;
    s_q[(d >> 2) * 256 + q * 4 + (d & 3)] = qbase[e];
  }
  __syncthreads();
  const float* __restrict__ r = p2 + (size_t)b * M * D;
  const pp::f4* __restrict__ sq = reinterpret_cast<const pp::f4*>(s_q) + lane;  // + 64 per group of four dimensions
  KList<KT> L;
  L.clear();
  // the four waves share the tile's 64 queries and take a quarter of the reference cloud each (groups of eight)
  const int groups = (m2 + 7) / 8;
  const int g0 = (int)(((long long)groups * wave) / 4), g1 = (int)(((long long)groups * (wave + 1)) / 4);
  const int Dfull = D & ~7;
  for (int k0 = 8 * g0; k0 < 8 * g1; k0 += 8) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
    const float* rp[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rp[u] = r + (size_t)min(k0 + u, m2 - 1) * D;  // wave-uniform
    for (int db = 0; db < Dfull; db += 8) {
      const pp::f4 qa = sq[(db >> 2) * 64], qb = sq[((db >> 2) + 1) * 64];
      const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float tt = rp[u][db + c] - qv[c];
          acc[u] = __builtin_fmaf(tt, tt, acc[u]);
        }
      }
    }
    if (Dfull < D) {  // last, partial block of dimensions
      const pp::f4 qa = sq[(Dfull >> 2) * 64], qb = sq[((Dfull >> 2) + 1) * 64];
      const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (Dfull + c < D) {  // wave-uniform
            const float tt = rp[u][Dfull + c] - qv[c];
            acc[u] = __builtin_fmaf(tt, tt, acc[u]);
          }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < m2) {  // wave-uniform
        if (__any(L.beats_last(acc[u], k0 + u))) L.insert(acc[u], k0 + u);
      }
  }
  // merge: waves 1..3 hand their lists to wave 0 through LDS, one after the other (one list's worth of LDS:
  // occupancy); the (distance, index) order does not depend on who inserts first
  float* od = dist + ((size_t)b * N + min(n, N - 1)) * K;
  int* oi = idx + ((size_t)b * N + min(n, N - 1)) * K;
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        s_ld[j * 64 + lane] = L.d[j];
        s_li[j * 64 + lane] = L.i[j];
      }
    }
    __syncthreads();
    if (wave == 0)
      for (int j = 0; j < KT; ++j) {
        const float d = s_ld[j * 64 + lane];
        const int i = s_li[j * 64 + lane];
        if (!__any(L.beats_last(d, i))) break;  // the list is ascending: nothing further of it can enter either
        L.insert(d, i);
      }
    __syncthreads();
  }
  if (wave > 0 || n >= N) return;
  if (n >= n1) {  // a padded query row
    for (int j = 0; j < K; ++j) {
      od[j] = 0.0f;
      oi[j] = 0;
    }
    return;
  }
  knn_store<KT>(L, od, oi, K, m2);
}

struct KnLayout {
  size_t sets, cell_start, sorted, qsorted, total;
};
__host__ __device__ inline KnLayout kn_layout(int B, int N, int M) {
  KnLayout L;
  L.sets = 0;  // [2B]: sets of the reference clouds, then the (unused) sets of the query sort
  L.cell_start = ((size_t)64 * 2 * B + 255) / 256 * 256;
  L.sorted = L.cell_start + ((size_t)4 * (kGridCells + 1) * B + 255) / 256 * 256;
  L.qsorted = L.sorted + ((size_t)16 * B * M + 255) / 256 * 256;
  L.total = L.qsorted + (size_t)16 * B * N;
  return L;
}

template <bool VEC>
__global__ __launch_bounds__(kBuildThreads) void kn_build_kernel(const float* __restrict__ p2,
                                                                 const float* __restrict__ p1,
                                                                 unsigned char* __restrict__ ws, int B, int N,
                                                                 int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_cnt[];
  const KnLayout L = kn_layout(B, N, M);
  // both sets of a batch element are built on the XCD that will search it (the query kernel's batch ->
  // XCD mapping): virtual order (batch, cloud | queries, slab)
  const int V = pp::xcd_virtual_block(blockIdx.x, (2 * B * pp::kBuildSlabs + 7) / 8);
  if (V >= 2 * B * pp::kBuildSlabs) return;
  const int slab = V % pp::kBuildSlabs;
  const int set = ((V / pp::kBuildSlabs) & 1) * B + V / (2 * pp::kBuildSlabs);
  GridSet* gs = reinterpret_cast<GridSet*>(ws + L.sets) + set;
  if (set >= B) {
    const int b = set - B;
    pp::grid_build_set<true, VEC>(p1 + (size_t)b * N * 3, N, gs, nullptr,
                             reinterpret_cast<pp::f4*>(ws + L.qsorted) + (size_t)b * N, nullptr, s_cnt, nullptr,
                             nullptr, slab, pp::kBuildSlabs);
    return;
  }
  const int b = set;
  pp::grid_build_set_plain<VEC>(p2 + (size_t)b * M * 3, M, gs,
                                reinterpret_cast<unsigned*>(ws + L.cell_start) + (size_t)b * (kGridCells + 1),
                                reinterpret_cast<pp::f4*>(ws + L.sorted) + (size_t)b * M, s_cnt, slab, pp::kBuildSlabs);
}

template <int KT>
__global__ __launch_bounds__(256) void knn_grid_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                       float* __restrict__ dist, int* __restrict__ idx,
                                                       unsigned char* __restrict__ ws, int B, int N, int M, int K,
                                                       int tiles_per_b, int per_xcd) {
  const int vb = pp::xcd_virtual_block(blockIdx.x, per_xcd);  // a batch element stays on one XCD's L2
  if (vb >= B * tiles_per_b) return;
  const int b = vb / tiles_per_b;
  const int tile = vb - b * tiles_per_b;
  const KnLayout L = kn_layout(B, N, M);
  const GridSet g = reinterpret_cast<const GridSet*>(ws + L.sets)[b];
  const bool usable = !pp::grid_useless(g);
  const int n = tile * 256 + threadIdx.x;
  if (!usable) {
    // A batch element without a usable grid (every point identical, a non-finite coordinate): every pair, a lane per
    // query of the ORIGINAL order, here -- round 4; until then a second launch of the scan kernel followed every call to
    // pick these up, 4-5 us to find, nearly always, nothing to do.
    const int nc = min(n, N - 1);
    const float* __restrict__ qp = p1 + ((size_t)b * N + nc) * 3;
    const float* __restrict__ r = p2 + (size_t)b * M * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    KList<KT> Ls;
    Ls.clear();
    for (int k = 0; k < M; ++k) {  // r[...] is wave-uniform: scalar loads
      const float d = pp::chamfer_d3(r[3 * (size_t)k], r[3 * (size_t)k + 1], r[3 * (size_t)k + 2], qx, qy, qz);
      if (__any(Ls.beats_last(d, k))) Ls.insert(d, k);
    }
    if (n < N) knn_store<KT>(Ls, dist + ((size_t)b * N + n) * K, idx + ((size_t)b * N + n) * K, K, M);
    return;
  }
  if (n >= N) return;
  const unsigned* __restrict__ cell_start =
      reinterpret_cast<const unsigned*>(ws + L.cell_start) + (size_t)b * (kGridCells + 1);
  const pp::f4* __restrict__ sorted = reinterpret_cast<const pp::f4*>(ws + L.sorted) + (size_t)b * M;
  const pp::f4 q = (reinterpret_cast<const pp::f4*>(ws + L.qsorted) + (size_t)b * N)[n];
  const int qorig = __float_as_int(q.w);
  const bool finite_q = __builtin_isfinite(q.x) && __builtin_isfinite(q.y) && __builtin_isfinite(q.z);
  // first box: about 2K points expected if the cloud filled its cells evenly (2 per cell), at least one cell
  float R = finite_q ? g.h * fmaxf(1.0f, 0.5f * cbrtf((float)K)) : 2.0e38f;
#ifndef PP_KNN_KEY64
#define PP_KNN_KEY64 1
#endif
#ifndef PP_KNN_BALL
#define PP_KNN_BALL 1  // rounds after the first walk the ball of the K-th best found, not its box (0: the box)
#endif
#ifndef PP_KNN_FLY
#define PP_KNN_FLY 2
#endif
#if PP_KNN_KEY64
  KList64<KT> Lk;
#else
  KList<KT> Lk;
#endif
  const int kth = min(K, M) - 1;  // the entry that decides when to stop
  // (round 5) from the second round on only the BALL of the K-th best of the round before can matter, not its box: rows
  // beyond it are passed over, the others cut along x (three_nn_grid.hip / chamfer_grid.hip's ball stages: same rule)
  const float px = (q.x - g.minx) * g.invh, py = (q.y - g.miny) * g.invh, pz = (q.z - g.minz) * g.invh;
  const int cy = cell_coord(q.y, g.miny, g.invh, g.gy), cz = cell_coord(q.z, g.minz, g.invh, g.gz);
  float bk2 = __builtin_inff();
  while (true) {
    Lk.clear();
    const float lx = q.x - R, hx = q.x + R, ly = q.y - R, hy = q.y + R, lz = q.z - R, hz = q.z + R;
    const bool everything = !(R < 1.0e38f);  // last round (also: non-finite q): the whole grid, no questions
    const int x0 = everything ? 0 : cell_coord(lx, g.minx, g.invh, g.gx);
    const int x1 = everything ? g.gx - 1 : cell_coord(hx, g.minx, g.invh, g.gx);
    const int y0 = everything ? 0 : cell_coord(ly, g.miny, g.invh, g.gy);
    const int y1 = everything ? g.gy - 1 : cell_coord(hy, g.miny, g.invh, g.gy);
    const int z0 = everything ? 0 : cell_coord(lz, g.minz, g.invh, g.gz);
    const int z1 = everything ? g.gz - 1 : cell_coord(hz, g.minz, g.invh, g.gz);
    // what the box really guarantees, from the rounded bounds themselves (|q| may dwarf R)
    const float reach = fminf(fminf(fminf(hx - q.x, q.x - lx), fminf(hy - q.y, q.y - ly)), fminf(hz - q.z, q.z - lz));
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const int c = (z * g.gy + y) * g.gx;
        int xa = x0, xb = x1;
#if PP_KNN_BALL
        {
          if (bk2 < 3.0e38f && !everything) {  // (a lane in its first round has no bound yet: nothing of this for it)
            const float dy = y < cy ? py - (float)(y + 1) : (y > cy ? (float)y - py : 0.0f);
            const float dz = z < cz ? pz - (float)(z + 1) : (z > cz ? (float)z - pz : 0.0f);
            const float w2 = bk2 - (dy * dy + dz * dz);
            if (w2 < 0.0f) continue;  // the row lies beyond the ball
            const float w = __builtin_amdgcn_sqrtf(w2) * 1.00001f;
            xa = max(xa, max(min((int)(px - w), g.gx - 1), 0));
            xb = min(xb, max(min((int)(px + w), g.gx - 1), 0));
            if (xa > xb) continue;
          }
        }
#endif
        // (one 16-byte load for both bounds of a row up to three cells wide: see three_nn_grid.hip)
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 v;
        __builtin_memcpy(&v, cell_start + c + xa, sizeof(v));
        const int wd = xb + 1 - xa;
        unsigned e = wd == 1 ? v.y : (wd == 2 ? v.z : v.w);
        if (wd > 3) e = cell_start[c + xb + 1];
        constexpr int kFly = PP_KNN_FLY;  // loads in flight per lane (the tail of a row repeats its last point)
        for (unsigned i = v.x; i < e; i += kFly) {
          pp::f4 pf[kFly];
#pragma unroll
          for (int u = 0; u < kFly; ++u) pf[u] = sorted[min(i + u, e - 1)];
#pragma unroll
          for (int u = 0; u < kFly; ++u) {
            const pp::f4 p = pf[u];
            const float d = pp::chamfer_d3(p.x, p.y, p.z, q.x, q.y, q.z);
            const int id = __float_as_int(p.w);
#if PP_KNN_KEY64
            const unsigned long long nk = (u == 0 || i + u < e) ? KList64<KT>::key(d, id) : ~0ull;  // (~0: never enters)
            if (Lk.beats_last(nk)) Lk.insert(nk);
#else
            if (u == 0 || i + u < e)
              if (Lk.beats_last(d, id)) Lk.insert(d, id);
#endif
          }
        }
      }
    const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.gx - 1 && y1 == g.gy - 1 && z1 == g.gz - 1;
#if PP_KNN_KEY64
    float dk = Lk.dist(0);
#pragma unroll
    for (int j = 1; j < KT; ++j) dk = j <= kth ? Lk.dist(j) : dk;
#else
    float dk = Lk.d[0];
#pragma unroll
    for (int j = 1; j < KT; ++j) dk = j <= kth ? Lk.d[j] : dk;
#endif
    if (whole || dk < 0.9999f * (reach * reach)) break;
    // the next box: the K-th best found so far bounds the answer, so a box of half-width sqrt(dk) (+ 0.05 %) ends the
    // search -- doubling blindly walked 125 cells where a few dozen do (three_nn_grid.hip: the same); at least 25 %
    // wider than the last one; fewer than K points found: double
    R = dk < 3.0e38f ? fmaxf(sqrtf(dk) * 1.0005f, 1.25f * R) : 2.0f * R;
    bk2 = dk < 3.0e38f ? dk * (g.invh * g.invh) * (1.0f / 0.999f) : __builtin_inff();
  }
#if PP_KNN_KEY64
  KList<KT> out;
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    out.d[j] = Lk.dist(j);
    out.i[j] = Lk.index(j);
  }
  knn_store<KT>(out, dist + ((size_t)b * N + qorig) * K, idx + ((size_t)b * N + qorig) * K, K, M);
#else
  knn_store<KT>(Lk, dist + ((size_t)b * N + qorig) * K, idx + ((size_t)b * N + qorig) * K, K, M);
#endif
}

template <int KT>
int knn_scan_launch(const float* p1, const float* p2, const int* len1, const int* len2, float* dist, int* idx,
                    int B, int N, int M, int K, const GridSet* skip, hipStream_t s) {
  const int tiles = (N + 255) / 256;
  const long long blocks = (long long)B * tiles;
  if (blocks > 0x7fffffffLL) return PP_EINVAL;
  knn_scan_kernel<KT><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p1, p2, len1, len2, dist, idx, N, M, K, tiles,
                                                                    skip);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

int knn_scan_dispatch(const float* p1, const float* p2, const int* len1, const int* len2, float* dist, int* idx,
                      int B, int N, int M, int K, const GridSet* skip, hipStream_t s) {
  if (K <= 1) return knn_scan_launch<1>(p1, p2, len1, len2, dist, idx, B, N, M, K, skip, s);
  if (K <= 4) return knn_scan_launch<4>(p1, p2, len1, len2, dist, idx, B, N, M, K, skip, s);
  if (K <= 8) return knn_scan_launch<8>(p1, p2, len1, len2, dist, idx, B, N, M, K, skip, s);
  if (K <= 16) return knn_scan_launch<16>(p1, p2, len1, len2, dist, idx, B, N, M, K, skip, s);
  return knn_scan_launch<32>(p1, p2, len1, len2, dist, idx, B, N, M, K, skip, s);
}

template <int KT>
int knn_grid_launch(const float* p1, const float* p2, float* dist, int* idx, unsigned char* ws, int B, int N, int M, int K,
                    hipStream_t s) {
  const int tiles = (N + 255) / 256;
  const long long per_xcd = ((long long)B * tiles + 7) / 8;
  if (per_xcd * 8 > 0x7fffffffLL) return PP_EINVAL;
  knn_grid_kernel<KT><<<dim3((unsigned)(per_xcd * 8)), dim3(256), 0, s>>>(p1, p2, dist, idx, ws, B, N, M, K, tiles,
                                                                          (int)per_xcd);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

}  // namespace

// 0 = automatic (grid when a workspace is given); 1 = scan kernel only (tests and tuning)
static pp::Knob g_knn_grid_mode;
extern "C" void pp_debug_set_knn_search(int v) { g_knn_grid_mode.set(v); }

static bool knn_args_ok(const float* p1, const float* p2, const float* dist, const int* idx, int B, int N, int M,
                        int K) {
  return B >= 0 && N >= 0 && M >= 0 && K >= 1 && K <= 32 && (B == 0 || N == 0 || (p1 && dist && idx && (M == 0 || p2)));
}

extern "C" int pp_knn_f32(const float* p1, const float* p2, const int* lengths1, const int* lengths2, float* dist2,
                          int* idx, int B, int N, int M, int K, void* stream) {
  if (!knn_args_ok(p1, p2, dist2, idx, B, N, M, K)) return PP_EINVAL;
  if (B == 0 || N == 0) return PP_OK;
  return knn_scan_dispatch(p1, p2, lengths1, lengths2, dist2, idx, B, N, M, K, nullptr, (hipStream_t)stream);
}

extern "C" size_t pp_knn_workspace_bytes(int B, int N, int M, int K) {
  if (B <= 0 || N < 1024 || M < 1024 || K < 1 || K > 32 || M < 4 * K) return 0;
  if ((long long)B * N >= (1LL << 31) || (long long)B * M >= (1LL << 31)) return 0;
  return kn_layout(B, N, M).total;
}

extern "C" int pp_knn_ws_f32(const float* p1, const float* p2, const int* lengths1, const int* lengths2,
                             float* dist2, int* idx, int B, int N, int M, int K, void* workspace,
                             size_t workspace_bytes, void* stream) {
  const size_t need = pp_knn_workspace_bytes(B, N, M, K);
  // ragged batches (lengths given) take the scan: the grid is built over all M points of a cloud
  if (g_knn_grid_mode == 1 || need == 0 || !workspace || workspace_bytes < need || lengths1 || lengths2)
    return pp_knn_f32(p1, p2, lengths1, lengths2, dist2, idx, B, N, M, K, stream);
  if (!knn_args_ok(p1, p2, dist2, idx, B, N, M, K)) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  static pp::DeviceFlags lds_ok;
  const size_t lds = pp::grid_build_lds_bytes(pp::kBuildSlabs) > pp::grid_build_fast_lds_bytes() ? pp::grid_build_lds_bytes(pp::kBuildSlabs)
                                                                                                 : pp::grid_build_fast_lds_bytes();
  static pp::DeviceFlags lds_ok_vec;
  const bool vec = pp::clouds_vec_aligned(p1, N, B) && pp::clouds_vec_aligned(p2, M, B);
  hipError_t e = vec ? pp::allow_big_lds(kn_build_kernel<true>, (int)lds, lds_ok_vec) : pp::allow_big_lds(kn_build_kernel<false>, (int)lds, lds_ok);
  if (e != hipSuccess) return (int)e;
  (vec ? kn_build_kernel<true> : kn_build_kernel<false>)<<<dim3(8 * ((2 * B * pp::kBuildSlabs + 7) / 8)), dim3(kBuildThreads), lds, s>>>(p2, p1, ws, B, N, M);
  PP_RETURN_IF_LAUNCH_FAILED();
  // (batch elements whose grid is of no use are served by the same kernel, every pair: no second launch)
  if (K <= 1) return knn_grid_launch<1>(p1, p2, dist2, idx, ws, B, N, M, K, s);
  if (K <= 4) return knn_grid_launch<4>(p1, p2, dist2, idx, ws, B, N, M, K, s);
  if (K <= 8) return knn_grid_launch<8>(p1, p2, dist2, idx, ws, B, N, M, K, s);
  if (K <= 16) return knn_grid_launch<16>(p1, p2, dist2, idx, ws, B, N, M, K, s);
  return knn_grid_launch<32>(p1, p2, dist2, idx, ws, B, N, M, K, s);
}

// pytorch3d.ops.knn_points for any point dimension D (1 <= D <= 512) and 1 <= K <= 128: brute force with the
// sequential distance chain (knn_nd_kernel).  D == 3 with K <= 32 is better served by pp_knn_ws_f32.
extern "C" int pp_knn_nd_f32(const float* p1, const float* p2, const int* lengths1, const int* lengths2, float* dist2,
                             int* idx, int B, int N, int M, int D, int K, void* stream) {
  if (B < 0 || N < 0 || M < 0 || D < 1 || D > 512 || K < 1 || K > 128) return PP_EINVAL;
  if (B == 0 || N == 0) return PP_OK;
  if (!p1 || !dist2 || !idx || (M > 0 && !p2)) return PP_EINVAL;
  const int tiles = (N + 63) / 64;
  const long long blocks = (long long)B * tiles;
  if (blocks > 0x7fffffffLL) return PP_EINVAL;
  const size_t lds_q = (size_t)((D + 7) & ~7) * 64 * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define PP_KNN_ND(KT)                                                                                          \
  do {                                                                                                         \
    static pp::DeviceFlags ok;                                                                                 \
    const size_t lds = lds_q + (size_t)KT * 64 * 8;                                                        \
    if (lds > 160 * 1024) return PP_EINVAL;                                                                    \
    const hipError_t e = pp::allow_big_lds(knn_nd_kernel<KT>, 160 * 1024, ok);                                 \
    if (e != hipSuccess) return (int)e;                                                                        \
    knn_nd_kernel<KT><<<dim3((unsigned)blocks), dim3(256), lds, s>>>(p1, p2, lengths1, lengths2, dist2, idx, N, M, \
                                                                     D, K, tiles);                             \
  } while (0)
  if (K <= 4) PP_KNN_ND(4);
  else if (K <= 8) PP_KNN_ND(8);
  else if (K <= 16) PP_KNN_ND(16);
  else if (K <= 32) PP_KNN_ND(32);
  else if (K <= 64) PP_KNN_ND(64);
  else PP_KNN_ND(128);
#undef PP_KNN_ND
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
