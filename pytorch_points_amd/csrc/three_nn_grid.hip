// three_nn_grid.hip -- three_nn through the uniform grid of grid_common.h, with the scan kernel of
// sampling.hip as the fallback.  Same output as the scan (ref _ext/interpolate_gpu.cu:9-52): per
// unknown point the three smallest dist3 over the known points and their indices, ascending,
// earlier index first among equal distances (the reference's strict `<` in index order) -- i.e. the
// three smallest pairs (dist3, index) in lexicographic order.
//
// The known points of a batch element are counting-sorted into the grid (cell side h); the unknown
// points are sorted along a Morton curve so that the lanes of a wave walk neighbouring cells.  A
// lane searches the cell box [cell(q - R), cell(q + R)], R = h first, keeping the three smallest
// (dist3, index) pairs of the box (same arithmetic as the scan: pp::dist3).  A known point outside
// that box lies beyond one of the (rounded) bounds q -+ R along some axis (the cell coordinate is
// monotone in the coordinate), so it differs from q by more than reach = min over the axes of
// fl(q + R) - q and q - fl(q - R), and its fp32 dist3 is >= reach^2 (1 - 1e-6): once the third
// best is < 0.9999 reach^2 no outside point can enter the result or tie with it, and the lane is
// done.  (reach, not R: coordinates may be large against the cell size.)  Otherwise R doubles
// and the lane searches again; a box that covers the whole grid ends the search unconditionally
// (that is the scan).  Batch elements whose grid is useless go to the scan kernel.
#include "grid_common.h"

#ifndef PP_TN_BALL
#define PP_TN_BALL 1  // rounds after the first walk the ball of the third best found, not its box (0: the box)
#endif

namespace {

using pp::GridSet;
using pp::cell_coord;
using pp::kGridCells;
using pp::kBuildThreads;

struct TnLayout {
  size_t sets, cell_start, sorted, qsorted, total;
};
__host__ __device__ inline TnLayout tn_layout(int B, int N, int M) {
  TnLayout L;
  L.sets = 0;  // [2B]: sets of the known clouds, then the (unused) sets of the query sort
  L.cell_start = ((size_t)64 * 2 * B + 255) / 256 * 256;
  L.sorted = L.cell_start + ((size_t)4 * (kGridCells + 1) * B + 255) / 256 * 256;
  L.qsorted = L.sorted + ((size_t)16 * B * M + 255) / 256 * 256;
  L.total = L.qsorted + (size_t)16 * B * N;
  return L;
}

// workgroups [0, S*B): slab s of the known points of batch element b into their grid; [S*B, 2*S*B):
// slab s of the unknown points of batch element b into Morton order (S = kBuildSlabs)
template <bool VEC>
__global__ __launch_bounds__(kBuildThreads) void tn_build_kernel(const float* __restrict__ known,
                                                                 const float* __restrict__ unknown,
                                                                 unsigned char* __restrict__ ws, int B, int N,
                                                                 int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_cnt[];
  const TnLayout L = tn_layout(B, N, M);
  // both sets of a batch element are built on the XCD that will search it (the query kernel's batch ->
  // XCD mapping): virtual order (batch, cloud | queries, slab)
  const int V = pp::xcd_virtual_block(blockIdx.x, (2 * B * pp::kBuildSlabs + 7) / 8);
  if (V >= 2 * B * pp::kBuildSlabs) return;
  const int slab = V % pp::kBuildSlabs;
  const int set = ((V / pp::kBuildSlabs) & 1) * B + V / (2 * pp::kBuildSlabs);
  GridSet* gs = reinterpret_cast<GridSet*>(ws + L.sets) + set;
  if (set >= B) {
    const int b = set - B;
    pp::grid_build_set<true, VEC>(unknown + (size_t)b * N * 3, N, gs, nullptr,
                             reinterpret_cast<pp::f4*>(ws + L.qsorted) + (size_t)b * N, nullptr, s_cnt, nullptr,
                             nullptr, slab, pp::kBuildSlabs);
    return;
  }
  const int b = set;
  pp::grid_build_set_plain<VEC>(known + (size_t)b * M * 3, M, gs,
                                reinterpret_cast<unsigned*>(ws + L.cell_start) + (size_t)b * (kGridCells + 1),
                                reinterpret_cast<pp::f4*>(ws + L.sorted) + (size_t)b * M, s_cnt, slab, pp::kBuildSlabs);
}

// (d, k) enters the ascending triple if it is lexicographically smaller than an entry -- the reference's strict `<` in
// index order.  With (distance bits << 32 | index) keys: for distances >= +0 that are not NaN the unsigned order of the
// keys is the (distance, index) order -- three 64-bit compares instead of nine compares and six logic operations
// (7 % of the query kernel's time at the interpolation shape).  A NaN distance gets the all-ones key and never enters.
__device__ __forceinline__ void insert3_key(float d, int k, unsigned long long& k1, unsigned long long& k2,
                                            unsigned long long& k3) {
  const unsigned long long nk = d == d ? (((unsigned long long)__float_as_uint(d) << 32) | (unsigned)k) : ~0ull;
  const bool l1 = nk < k1, l2 = nk < k2, l3 = nk < k3;
  k3 = l2 ? k2 : (l3 ? nk : k3);
  k2 = l1 ? k1 : (l2 ? nk : k2);
  k1 = l1 ? nk : k1;
}

__global__ __launch_bounds__(256) void tn_query_kernel(const float* __restrict__ unknown,
                                                       const float* __restrict__ known, float* __restrict__ dist2,
                                                       int* __restrict__ idx, unsigned char* __restrict__ ws, int B,
                                                       int N, int M, int tiles_per_b, int per_xcd) {
  const int vb = pp::xcd_virtual_block(blockIdx.x, per_xcd);  // a batch element stays on one XCD's L2
  if (vb >= B * tiles_per_b) return;
  const int b = vb / tiles_per_b;
  const int tile = vb - b * tiles_per_b;
  const TnLayout L = tn_layout(B, N, M);
  const GridSet g = reinterpret_cast<const GridSet*>(ws + L.sets)[b];
  const bool usable = !pp::grid_useless(g);
  const int n = tile * 256 + threadIdx.x;
  if (!usable) {
    // A batch element without a usable grid (every known point identical, a non-finite coordinate): every pair, a lane
    // per unknown point of the ORIGINAL order, the scan kernel's loop (sampling.hip: three_nn_kernel) -- here, since
    // round 4; until then a second launch followed every call to pick these up (4-5 us to find nothing to do).
    const int nc = min(n, N - 1);
    const float* __restrict__ u = unknown + ((size_t)b * N + nc) * 3;
    const float* __restrict__ kn = known + (size_t)b * M * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < M; ++k) {  // kn[...] is wave-uniform: scalar loads
      const float d = pp::dist3(ux, uy, uz, kn[3 * (size_t)k], kn[3 * (size_t)k + 1], kn[3 * (size_t)k + 2]);
      if (__any(d < b3)) {  // (ascending k: strict < is the (distance, index) order)
        const bool l1 = d < b1, l2 = d < b2, l3 = d < b3;
        b3 = l2 ? b2 : (l3 ? d : b3);
        i3 = l2 ? i2 : (l3 ? k : i3);
        b2 = l1 ? b1 : (l2 ? d : b2);
        i2 = l1 ? i1 : (l2 ? k : i2);
        b1 = l1 ? d : b1;
        i1 = l1 ? k : i1;
      }
    }
    if (n < N) {
      float* od = dist2 + ((size_t)b * N + n) * 3;
      int* oi = idx + ((size_t)b * N + n) * 3;
      od[0] = b1; od[1] = b2; od[2] = b3;
      oi[0] = i1; oi[1] = i2; oi[2] = i3;
    }
    return;
  }
  if (n >= N) return;
  const unsigned* __restrict__ cell_start =
      reinterpret_cast<const unsigned*>(ws + L.cell_start) + (size_t)b * (kGridCells + 1);
  const pp::f4* __restrict__ sorted = reinterpret_cast<const pp::f4*>(ws + L.sorted) + (size_t)b * M;
  const pp::f4 q = (reinterpret_cast<const pp::f4*>(ws + L.qsorted) + (size_t)b * N)[n];
  const int qorig = __float_as_int(q.w);

  unsigned long long k1, k2, k3;  // the three best (distance, index) pairs so far, as keys (insert3_key)
  const bool finite_q = __builtin_isfinite(q.x) && __builtin_isfinite(q.y) && __builtin_isfinite(q.z);
  float R = finite_q ? g.h : 2.0e38f;
  // (round 5) from the second round on the third best of the round before bounds the answer: of the box only the BALL of
  // that radius can matter -- a cell row (y, z) whose slab lies beyond it is passed over, the others are cut to the cells
  // within what is left of the radius along x (the rule and the arithmetic of chamfer_grid.hip's ball stages: distances
  // in cells from the query's own cell coordinates, a factor 0.999 of slack on the squares, rim cells open outwards).
  // bk2 = that bound in cells^2 (+inf in the first round: every row, whole width).
  const float px = (q.x - g.minx) * g.invh, py = (q.y - g.miny) * g.invh, pz = (q.z - g.minz) * g.invh;
  const int cy = cell_coord(q.y, g.miny, g.invh, g.gy), cz = cell_coord(q.z, g.minz, g.invh, g.gz);
  float bk2 = __builtin_inff();
  while (true) {
    k1 = k2 = k3 = (unsigned long long)0x7f800000u << 32;  // (inf, 0)
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
#if PP_TN_BALL
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
        // a row's two bounds are a few table entries apart: up to three cells wide (the usual first round)
        // ONE 16-byte load fetches both (the entries after a set's table are the next set's or the sorted
        // cloud: valid memory); this kernel is bound by the L1's handling of scattered loads
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 v;
        __builtin_memcpy(&v, cell_start + c + xa, sizeof(v));
        const int wd = xb + 1 - xa;
        unsigned e = wd == 1 ? v.y : (wd == 2 ? v.z : v.w);
        if (wd > 3) e = cell_start[c + xb + 1];
        for (unsigned i = v.x; i < e; i += 2) {
          const pp::f4 p0 = sorted[i];
          const pp::f4 p1 = sorted[min(i + 1, e - 1)];
          insert3_key(pp::dist3(q.x, q.y, q.z, p0.x, p0.y, p0.z), __float_as_int(p0.w), k1, k2, k3);
          if (i + 1 < e) insert3_key(pp::dist3(q.x, q.y, q.z, p1.x, p1.y, p1.z), __float_as_int(p1.w), k1, k2, k3);
        }
      }
    const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.gx - 1 && y1 == g.gy - 1 && z1 == g.gz - 1;
    const float d3 = __uint_as_float((unsigned)(k3 >> 32));
    if (whole || d3 < 0.9999f * (reach * reach)) break;
    // The next box: the third best found so far bounds the answer -- nothing farther than sqrt(d3) can enter -- so a box
    // of that half-width (+ 0.05 %: its reach must exceed sqrt(d3 / 0.9999)) ends the search; doubling blindly walked
    // 125 cells where ~40 do (the rounds after the first were 35 % of this kernel's time: 24 -> 1x us).  At least 25 %
    // wider than the last one (coordinates far from the origin round the box's reach down); fewer than three points
    // found: double.
    R = d3 < 3.0e38f ? fmaxf(sqrtf(d3) * 1.0005f, 1.25f * R) : 2.0f * R;
    bk2 = d3 < 3.0e38f ? d3 * (g.invh * g.invh) * (1.0f / 0.999f) : __builtin_inff();
  }
  float* od = dist2 + ((size_t)b * N + qorig) * 3;
  int* oi = idx + ((size_t)b * N + qorig) * 3;
  od[0] = __uint_as_float((unsigned)(k1 >> 32)); od[1] = __uint_as_float((unsigned)(k2 >> 32));
  od[2] = __uint_as_float((unsigned)(k3 >> 32));
  oi[0] = (int)(unsigned)k1; oi[1] = (int)(unsigned)k2; oi[2] = (int)(unsigned)k3;
}

}  // namespace

// 0 = automatic (grid when a workspace is given); 1 = scan kernel only (tests and tuning)
static pp::Knob g_tn_grid_mode;
extern "C" void pp_debug_set_three_nn_search(int v) { g_tn_grid_mode.set(v); }

extern "C" size_t pp_three_nn_workspace_bytes(int B, int N, int M) {
  if (B <= 0 || N < 1024 || M < 1024) return 0;
  if ((long long)B * N >= (1LL << 31) || (long long)B * M >= (1LL << 31)) return 0;
  return tn_layout(B, N, M).total;
}

extern "C" int pp_three_nn_ws_f32(const float* unknown, const float* known, float* dist2, int* idx, int B, int N,
                                  int M, void* workspace, size_t workspace_bytes, void* stream) {
  const size_t need = pp_three_nn_workspace_bytes(B, N, M);
  if (g_tn_grid_mode == 1 || need == 0 || !workspace || workspace_bytes < need)
    return pp_three_nn_f32(unknown, known, dist2, idx, B, N, M, stream);
  if (!unknown || !known || !dist2 || !idx) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  static pp::DeviceFlags lds_ok;
  const size_t lds = pp::grid_build_lds_bytes(pp::kBuildSlabs) > pp::grid_build_fast_lds_bytes() ? pp::grid_build_lds_bytes(pp::kBuildSlabs)
                                                                                                 : pp::grid_build_fast_lds_bytes();
  static pp::DeviceFlags lds_ok_vec;
  const bool vec = pp::clouds_vec_aligned(unknown, N, B) && pp::clouds_vec_aligned(known, M, B);
  hipError_t e = vec ? pp::allow_big_lds(tn_build_kernel<true>, (int)lds, lds_ok_vec) : pp::allow_big_lds(tn_build_kernel<false>, (int)lds, lds_ok);
  if (e != hipSuccess) return (int)e;
  (vec ? tn_build_kernel<true> : tn_build_kernel<false>)<<<dim3(8 * ((2 * B * pp::kBuildSlabs + 7) / 8)), dim3(kBuildThreads), lds, s>>>(known, unknown, ws, B, N, M);
  PP_RETURN_IF_LAUNCH_FAILED();
  const int tiles = (N + 255) / 256;
  const long long per_xcd = ((long long)B * tiles + 7) / 8;
  if (per_xcd * 8 > 0x7fffffffLL) return PP_EINVAL;
  // (batch elements whose grid is of no use are served by the same kernel, every pair: no second launch)
  tn_query_kernel<<<dim3((unsigned)(per_xcd * 8)), dim3(256), 0, s>>>(unknown, known, dist2, idx, ws, B, N, M, tiles,
                                                                      (int)per_xcd);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
