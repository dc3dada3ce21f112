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
//   * a query that cannot stop at rho = 2 (far from the cloud, or degenerate data) is appended to
//     a list and resolved by the brute-force kernel (LIST mode), as is every query of a set whose
//     grid is useless (non-finite coordinates, almost all points in one cell).
#include "grid_common.h"

namespace pp {
// implemented in chamfer.hip: brute force over the queries listed in `qlist` (per set: count in
// qcount[set], indices in qlist[set_offset ...])
int nmdist_fwd_c3_list(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2,
                       int* idx2, int B, int N, int M, const int* qlist, const int* qcount,
                       hipStream_t s, const float* label1, const float* label2, unsigned long long* lkey,
                       int* ldone);
}  // namespace pp

namespace {

using pp::GridSet;
using pp::cell_coord;
using pp::kGridCells;
using pp::kGridMax;
using pp::kBuildThreads;

constexpr float kBoundSlack = 0.999f;

// Workspace layout (bytes), S = 2*B sets, T = B*(N+M) points:
//   [0, 64*S)                      GridSet[S]
//   [.., +4*S)  (padded to 256)    int qcount[S]
//   [.., +4*(kGridCells+1)*S)      unsigned cell_start[S][kGridCells+1]
//   [.., +16*T)                    float4 sorted[T]   (x, y, z, original index bits)
//   [.., +4*T)                     int qlist[T]           queries left to the brute force, per set
//   [.., +8*T)                     u64 lkey[T]            brute-force list: (distance, index) keys merged across slices
//   [.., +4*S*tiles_l)             int ldone[S][tiles_l]  brute-force list: slices finished per tile of 128
//   [.., +4*T)                     float slab[T]          labels in sorted order (labeled Chamfer only)
struct Layout {
  size_t sets, qcount, cell_start, sorted, qlist, lkey, ldone, slab, total;
};
__host__ __device__ inline Layout make_layout(int B, int N, int M, bool labeled = false) {
  Layout L;
  const size_t S = (size_t)2 * B, T = (size_t)B * ((size_t)N + M);
  L.sets = 0;
  L.qcount = L.sets + 64 * S;
  L.cell_start = L.qcount + ((4 * S + 255) / 256) * 256;
  L.sorted = L.cell_start + ((4 * (size_t)(kGridCells + 1) * S + 255) / 256) * 256;
  L.qlist = L.sorted + 16 * T;
  L.lkey = L.qlist + 4 * T;
  L.lkey = (L.lkey + 7) / 8 * 8;
  L.ldone = L.lkey + 8 * T;
  L.slab = L.ldone + ((4 * S * (size_t)(((N > M ? N : M) + 127) / 128) + 255) / 256) * 256;
  L.total = L.slab + (labeled ? 4 * T : 0);
  return L;
}
// set s = 2*b + dir; dir 0: queries = cloud 1 (N), references = cloud 2 (M)
__host__ __device__ inline size_t set_point_offset(int b, int dir, int N, int M) {
  return (size_t)b * ((size_t)N + M) + (dir ? (size_t)M : 0);  // references of (b,0) first (M), then (b,1) (N)
}
__host__ __device__ inline size_t set_query_offset(int b, int dir, int N, int M) {
  return (size_t)b * ((size_t)N + M) + (dir ? (size_t)N : 0);  // queries of (b,0) first (N), then (b,1) (M)
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
  // a set is built on the XCD that will search it (grid_query_kernel's set -> XCD mapping): its sorted
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
  if (threadIdx.x == 0 && slab == 0) {  // the lists this set's queries may be appended to start empty
    int* counts = reinterpret_cast<int*>(ws + L.qcount);
    counts[set] = 0;  // brute-force list of this set
  }
  if (slab == 0) {  // "slices finished" counters of this set's brute-force tiles
    const int tiles_l = ((N > M ? N : M) + 127) / 128;
    int* done = reinterpret_cast<int*>(ws + L.ldone) + (size_t)set * tiles_l;
    for (int i = threadIdx.x; i < tiles_l; i += kBuildThreads) done[i] = 0;
  }
  pp::grid_build_set<false, VEC>(ref, nr, reinterpret_cast<GridSet*>(ws + L.sets) + set,
                     reinterpret_cast<unsigned*>(ws + L.cell_start) + (size_t)set * (kGridCells + 1),
                     reinterpret_cast<pp::f4*>(ws + L.sorted) + set_point_offset(b, dir, N, M),
                     nullptr, s_cnt, lab,
                     labeled ? reinterpret_cast<float*>(ws + L.slab) + set_point_offset(b, dir, N, M) : nullptr,
                     slab, pp::kBuildSlabs);
}

// Append `value` to list[counter++] for the lanes with `want`: one atomic per wave.  `keys`, if given,
// is the brute-force list's key array: the new entry's key starts at "nothing found".
__device__ __forceinline__ void wave_append(bool want, int* counter, int* list, int value,
                                            unsigned long long* keys = nullptr) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0) return;
  const int lane = threadIdx.x & 63;
  const int leader = (int)__builtin_ctzll(mask);
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, (int)__builtin_popcountll(mask));
  base = __shfl(base, leader);
  if (want) {
    const int pos = base + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
    list[pos] = value;
    if (keys) keys[pos] = ~0ull;
  }
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
template <bool LAB>
__device__ __forceinline__ unsigned long long wave_scan_rows(int nrows, unsigned rs, unsigned re,
                                                             const pp::f4* __restrict__ sorted,
                                                             const float* __restrict__ slab, float qx, float qy,
                                                             float qz, float ql, unsigned long long key) {
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
template <bool LAB>
__device__ __forceinline__ bool wide_stages_wave(float qx, float qy, float qz, float ql, const GridSet& g,
                                                 const unsigned* __restrict__ cell_start,
                                                 const pp::f4* __restrict__ sorted, const float* __restrict__ slab,
                                                 float& best, int& bidx) {
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
#pragma unroll
  for (int rho = 1; rho <= 2; ++rho) {  // cube of radius 1, then 2 (re-examining cells is harmless)
    if (resolved) break;
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
    key = wave_scan_rows<LAB>(side * side, rs, re, sorted, slab, qx, qy, qz, ql, key);
    const float kbest = __uint_as_float((unsigned)(key >> 32));
    const int kidx = (int)(unsigned)key;
    const bool all = cz - rho <= 0 && cz + rho >= g.gz - 1 && cy - rho <= 0 && cy + rho >= g.gy - 1 &&
                     cx - rho <= 0 && cx + rho >= g.gx - 1;
    const float reach = g.h * reach_cube(rho);
    resolved = all ? (LAB || kidx != 0x7fffffff) : (kbest < reach * reach * kBoundSlack);
  }
  best = __uint_as_float((unsigned)(key >> 32));
  bidx = (int)(unsigned)key;
  if (LAB && resolved && bidx == 0x7fffffff) {  // whole grid examined, nobody carries this label
    best = 0.0f;                                  // (ref nmdistance_cuda.cu:110-113)
    bidx = -1;
  }
  return resolved;
}

// Group k of a lane's stage-A sequence (see grid_query_kernel): four points of the row it falls in.
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

// Stage A for every query, one lane per query (dense launch); the few queries it cannot settle are then
// taken through the wide stages by their wave, one at a time (see the end of the kernel).
// LAB (labeled Chamfer): a reference point is a candidate only if its label equals the query's; the
// stopping rule is unchanged (it bounds EVERY unexamined point, whatever its label).
template <bool LAB>
__global__ __launch_bounds__(256) void grid_query_kernel(const float* __restrict__ xyz1,
                                                         const float* __restrict__ xyz2,
                                                         float* __restrict__ dist1, int* __restrict__ idx1,
                                                         float* __restrict__ dist2, int* __restrict__ idx2,
                                                         unsigned char* __restrict__ ws, int B, int N, int M,
                                                         int tiles1, int tiles2, int total, int per_xcd,
                                                         const float* __restrict__ label1,
                                                         const float* __restrict__ label2) {
  // workgroups of one set on one XCD: its cells and points (384 KiB) stay in that L2
  const int V = pp::xcd_virtual_block(blockIdx.x, per_xcd);
  if (V >= total) return;
  const int per_b = tiles1 + tiles2;
  const int b = V / per_b;
  const int r = V - b * per_b;
  const int dir = r >= tiles1 ? 1 : 0;
  const int tile = dir ? r - tiles1 : r;
  const int nq = dir ? M : N;
  // every lane stays alive (the waves cooperate on their leftover queries below): the lanes beyond a
  // ragged last tile repeat the cloud's last query and are kept from storing or listing anything
  const bool valid = tile * 256 + (int)threadIdx.x < nq;
  const int jj = valid ? tile * 256 + (int)threadIdx.x : nq - 1;
  const int set = 2 * b + dir;
  const Layout L = make_layout(B, N, M, LAB);
  const GridSet g = reinterpret_cast<const GridSet*>(ws + L.sets)[set];
  // The query cloud is the reference cloud of the partner set (b, 1-dir), already sorted by cell
  // there: walking the queries in that order makes the lanes of a wave spatial neighbours, so
  // they read the same cells (coalesced, L1-resident) and run similar trip counts.
  const GridSet gp = reinterpret_cast<const GridSet*>(ws + L.sets)[set ^ 1];
  const pp::f4* __restrict__ qsorted =
      reinterpret_cast<const pp::f4*>(ws + L.sorted) + set_point_offset(b, dir ^ 1, N, M);
  int* counts = reinterpret_cast<int*>(ws + L.qcount);
  int* qlist = reinterpret_cast<int*>(ws + L.qlist) + set_query_offset(b, dir, N, M);
  const bool g_useless = pp::grid_useless(g), gp_useless = pp::grid_useless(gp);
  if (g_useless) {  // uniform over the workgroup (one set per workgroup)
    wave_append(valid, counts + set, qlist, jj,  // every query exactly once, any order
                reinterpret_cast<unsigned long long*>(ws + L.lkey) + set_query_offset(b, dir, N, M));
    return;
  }
  const unsigned* __restrict__ cell_start =
      reinterpret_cast<const unsigned*>(ws + L.cell_start) + (size_t)set * (kGridCells + 1);
  const pp::f4* __restrict__ sorted =
      reinterpret_cast<const pp::f4*>(ws + L.sorted) + set_point_offset(b, dir, N, M);
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
  const float* __restrict__ slab =
      LAB ? reinterpret_cast<const float*>(ws + L.slab) + set_point_offset(b, dir, N, M) : nullptr;
  const int cx = cell_coord(qx, g.minx, g.invh, g.gx);
  const int cy = cell_coord(qy, g.miny, g.invh, g.gy);
  const int cz = cell_coord(qz, g.minz, g.invh, g.gz);
  float best = __builtin_inff();
  int bidx = 0x7fffffff;
  bool resolved = false;
  bool live = valid;  // after the re-deal below: this lane holds a real query
  // Stage A: the 2x2x2 block of cells nearest to q' (own cell + the neighbour on the side of the
  // cell q' lies in, per axis).  A point outside that block is beyond the far face of q''s cell
  // along some axis (>= h/2 away) or beyond the neighbour (>= h away): true distance >= h/2.
  {
    const float fx = (qx - g.minx) * g.invh - (float)cx, fy = (qy - g.miny) * g.invh - (float)cy,
                fz = (qz - g.minz) * g.invh - (float)cz;  // position inside the cell, in cells
    const int sx = fx < 0.5f ? -1 : 1, sy = fy < 0.5f ? -1 : 1, sz = fz < 0.5f ? -1 : 1;
    const int x0 = max(min(cx, cx + sx), 0), x1 = min(max(cx, cx + sx), g.gx - 1);
    // (named scalars, not arrays: hipcc turns a select between array elements into an indexed load
    // from scratch memory)
    auto row_range = [&](int a, int bq, unsigned& s_out, unsigned& e_out) {
      const int z = cz + a * sz, y = cy + bq * sy;
      const bool ok = z >= 0 && z < g.gz && y >= 0 && y < g.gy;
      const int c = pp::cell_linear(0, min(max(y, 0), g.gy - 1), min(max(z, 0), g.gz - 1), g.gx, g.gy);
      // the row is one or two cells wide: its two bounds are at most two entries apart, so ONE 12-byte load
      // fetches both (the entry after a set's table is the next set's or the sorted cloud: valid memory) --
      // four scattered load instructions per query less
      typedef unsigned u3 __attribute__((ext_vector_type(3)));
      u3 v;
      __builtin_memcpy(&v, cell_start + c + x0, sizeof(v));
      s_out = ok ? v.x : 0u;
      e_out = ok ? (x1 > x0 ? v.z : v.y) : 0u;
    };
    unsigned rs0, rs1, rs2, rs3, re0, re1, re2, re3;
    row_range(0, 0, rs0, re0);
    row_range(0, 1, rs1, re1);
    row_range(1, 0, rs2, re2);
    row_range(1, 1, rs3, re3);
    // What the block guarantees for THIS query: along each axis the nearer face of the block that has
    // grid beyond it (beyond the grid there are no points).  Lower face: 1 + f cells away when the
    // block includes cell c-1, f when it starts at c; upper face: 1 - f or 2 - f.  Never below h/2.
    auto reach1 = [](float f, int s, int c, int gdim) {
      const float lo = s < 0 ? (c >= 1 ? f + 1.0f : __builtin_inff()) : (c >= 1 ? f : __builtin_inff());
      const float hi = s < 0 ? (c + 1 <= gdim - 1 ? 1.0f - f : __builtin_inff())
                             : (c + 1 <= gdim - 1 ? 2.0f - f : __builtin_inff());
      return fminf(lo, hi);
    };
    const float reach = g.h * fminf(reach1(fx, sx, cx, g.gx), fminf(reach1(fy, sy, cy, g.gy), reach1(fz, sz, cz, g.gz)));
    float thr = reach * reach * kBoundSlack;  // the query is settled if its best distance is below this
    // The walk below runs, per wave, as long as the wave's LONGEST candidate list, and the lists differ a
    // lot between neighbours (rocprofv3 counters at config 2: 13.5 groups walked per wave for a mean of
    // ~5 per query; lanes that have finished keep re-examining point 0).  The lists' lengths are known
    // here, before the first point is touched, so the workgroup's 256 queries are first re-dealt to the
    // lanes in order of length (a counting sort through LDS: 32 buckets by number of groups, 13 words of
    // state per query): each wave then walks queries of about the same length.  The queries of a tile are
    // close in space whatever the order, so the candidates stay cache-resident.
    {
      constexpr int kWords = LAB ? 14 : 13;
      __shared__ unsigned s_hist[32], s_base[32];
      __shared__ unsigned s_st[kWords][256];
      const int t = threadIdx.x;
      auto groups_of = [](unsigned s_, unsigned e_) { return (e_ - s_ + 3) >> 2; };
      const unsigned ng = valid ? groups_of(rs0, re0) + groups_of(rs1, re1) + groups_of(rs2, re2) + groups_of(rs3, re3) : 0u;
      const unsigned key = min(ng, 31u);
      if (t < 32) s_hist[t] = 0u;
      __syncthreads();
      const unsigned pos = atomicAdd(&s_hist[key], 1u);
      __syncthreads();
      if (t < 64) {  // exclusive prefix of the 32 bucket counts
        const unsigned h = t < 32 ? s_hist[t] : 0u;
        unsigned incl = h;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
          const unsigned o = __shfl_up(incl, off);
          if (t >= off) incl += o;
        }
        if (t < 32) s_base[t] = incl - h;
      }
      __syncthreads();
      const unsigned dest = s_base[key] + pos;
      s_st[0][dest] = __float_as_uint(qx); s_st[1][dest] = __float_as_uint(qy); s_st[2][dest] = __float_as_uint(qz);
      s_st[3][dest] = (unsigned)(valid ? j : -1);
      s_st[4][dest] = __float_as_uint(thr);
      s_st[5][dest] = rs0; s_st[6][dest] = rs1; s_st[7][dest] = rs2; s_st[8][dest] = rs3;
      s_st[9][dest] = re0; s_st[10][dest] = re1; s_st[11][dest] = re2; s_st[12][dest] = re3;
      if (LAB) s_st[kWords - 1][dest] = __float_as_uint(ql);
      __syncthreads();
      qx = __uint_as_float(s_st[0][t]); qy = __uint_as_float(s_st[1][t]); qz = __uint_as_float(s_st[2][t]);
      j = (int)s_st[3][t];
      thr = __uint_as_float(s_st[4][t]);
      rs0 = s_st[5][t]; rs1 = s_st[6][t]; rs2 = s_st[7][t]; rs3 = s_st[8][t];
      re0 = s_st[9][t]; re1 = s_st[10][t]; re2 = s_st[11][t]; re3 = s_st[12][t];
      if (LAB) ql = __uint_as_float(s_st[kWords - 1][t]);
      live = j >= 0;
    }
    // The four rows are walked as ONE sequence of groups of four points: a lane's rows hold t_r =
    // ceil(len_r / 4) groups each, group k of the sequence belongs to the row r with T_r <= k < T_{r+1}
    // (T = running sums) and starts at rs_r + 4 (k - T_r).  k is wave-uniform, so the wave runs
    // max over lanes of (t_0 + .. + t_3) steps instead of the sum over rows of the per-row maxima, and
    // the loads of group k + 1 are in flight while group k is evaluated (a row-by-row walk waits for
    // every group's loads before it can issue the next: 50 -> 3x us at config 2).  The clamped duplicates
    // of a ragged tail are the same candidate again: harmless.
    const unsigned t0 = (re0 - rs0 + 3) >> 2, t1 = (re1 - rs1 + 3) >> 2, t2 = (re2 - rs2 + 3) >> 2,
                   t3 = (re3 - rs3 + 3) >> 2;
    const unsigned T1 = t0, T2 = T1 + t1, T3 = T2 + t2, T4 = T3 + t3;
    const unsigned adj0 = rs0, adj1 = rs1 - 4 * T1, adj2 = rs2 - 4 * T2, adj3 = rs3 - 4 * T3;
    const unsigned last0 = re0 - 1, last1 = re1 - 1, last2 = re2 - 1, last3 = re3 - 1;
    auto fetch = [&](unsigned k, pp::f4 (&p)[4], float (&pl)[4]) {
      stage_a_fetch<LAB>(k, T1, T2, T3, T4, adj0, adj1, adj2, adj3, last0, last1, last2, last3, sorted, slab, p, pl);
    };
    auto examine = [&](const pp::f4 (&p)[4], const float (&pl)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = pp::chamfer_d3(p[u].x, p[u].y, p[u].z, qx, qy, qz);
        const int id = __float_as_int(p[u].w);
        // (bitwise, not short-circuit: the compiler turns `||` / `&&` into exec-mask branches, five scalar
        // instructions and two branches per candidate for a tie that almost never happens)
        const bool take = (!LAB || pl[u] == ql) & ((d < best) | ((d == best) & (id < bidx)));
        best = take ? d : best;
        bidx = take ? id : bidx;
      }
    };
    pp::f4 pa[4], pb[4];
    float la[4] = {0.0f, 0.0f, 0.0f, 0.0f}, lb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    fetch(0, pa, la);
    for (unsigned k = 0; __any(k < T4); k += 2) {  // two steps per trip: the buffers swap roles without copies
      fetch(k + 1, pb, lb);
      examine(pa, la);
      fetch(k + 2, pa, la);
      examine(pb, lb);
    }
    resolved = best < thr;
  }
  // Results go straight to the query's original position: two scattered 4-byte stores per query.
  // (Measured against leaving them in walked order, coalesced, plus an inverse permutation written by
  // the build and an unsort pass: the direct form is 4 us faster per forward at config 2 and needs
  // 12 bytes less workspace per point.)
  if (resolved && live) {
    (dir ? dist2 : dist1)[(size_t)b * nq + j] = best;
    (dir ? idx2 : idx1)[(size_t)b * nq + j] = bidx;
  }
  // The queries stage A could not settle (0.3 % at config 2: one in five waves has one) are served on the
  // spot, one after the other, by the whole wave (wide_stages_wave): no second list, no second launch
  // (a separate kernel over the compacted leftovers cost 7 us of launch and tail per forward).  What the
  // cube of radius 2 cannot settle either goes to the brute-force list.
  unsigned long long pending = __ballot(!resolved && live);
  while (pending) {  // wave-uniform
    const int l = (int)__builtin_ctzll(pending);
    pending &= pending - 1;
    const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), l));
    const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), l));
    const float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), l));
    const float wl = LAB ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ql), l)) : 0.0f;
    const int wj = __builtin_amdgcn_readlane(j, l);
    float wbest;
    int widx;
    const bool settled = wide_stages_wave<LAB>(wx, wy, wz, wl, g, cell_start, sorted, slab, wbest, widx);
    if ((threadIdx.x & 63) == 0) {
      if (settled) {
        (dir ? dist2 : dist1)[(size_t)b * nq + wj] = wbest;
        (dir ? idx2 : idx1)[(size_t)b * nq + wj] = widx;
      } else {
        const int at = atomicAdd(counts + set, 1);
        qlist[at] = wj;
        (reinterpret_cast<unsigned long long*>(ws + L.lkey) + set_query_offset(b, dir, N, M))[at] = ~0ull;
      }
    }
  }
}

}  // namespace

// 0 = automatic (grid when a workspace is given and the problem is large enough to pay for its four
// launches); 1 = brute force; 2 = grid wherever it is structurally possible (tests)
static int g_grid_mode = 0;
extern "C" void pp_debug_set_nmdistance_search(int v) { g_grid_mode = v; }

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

// build -> stage A for every query, wide stages for what it leaves -> brute force over what is left
template <bool LAB>
static int grid_forward(const float* xyz1, const float* xyz2, const float* label1, const float* label2,
                        float* dist1, int* idx1, float* dist2, int* idx2, int B, int N, int M,
                        unsigned char* ws, hipStream_t s) {
  const Layout L = make_layout(B, N, M, LAB);
  hipError_t e;
  static bool lds_ok[64] = {}, lds_ok_vec[64] = {};
  const size_t lds = pp::grid_build_lds_bytes(pp::kBuildSlabs);
  const bool vec = pp::clouds_vec_aligned(xyz1, N, B) && pp::clouds_vec_aligned(xyz2, M, B);
  e = vec ? pp::allow_big_lds(grid_build_kernel<true>, (int)lds, lds_ok_vec)
          : pp::allow_big_lds(grid_build_kernel<false>, (int)lds, lds_ok);
  if (e != hipSuccess) return (int)e;
  (vec ? grid_build_kernel<true> : grid_build_kernel<false>)<<<dim3(8 * ((2 * B * pp::kBuildSlabs + 7) / 8)), dim3(kBuildThreads), lds, s>>>(
      xyz1, xyz2, ws, B, N, M, LAB ? label1 : nullptr, LAB ? label2 : nullptr);
  PP_RETURN_IF_LAUNCH_FAILED();
  const int tiles1 = (N + 255) / 256, tiles2 = (M + 255) / 256;
  const long long blocks = (long long)B * (tiles1 + tiles2);
  if (blocks > 0x7fffffffLL) return PP_EINVAL;
  const int per_xcd = (int)((blocks + 7) / 8);
  grid_query_kernel<LAB><<<dim3((unsigned)(per_xcd * 8)), dim3(256), 0, s>>>(
      xyz1, xyz2, dist1, idx1, dist2, idx2, ws, B, N, M, tiles1, tiles2, (int)blocks, per_xcd, label1, label2);
  PP_RETURN_IF_LAUNCH_FAILED();
  return pp::nmdist_fwd_c3_list(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M,
                                reinterpret_cast<const int*>(ws + L.qlist),
                                reinterpret_cast<const int*>(ws + L.qcount), s, LAB ? label1 : nullptr,
                                LAB ? label2 : nullptr, reinterpret_cast<unsigned long long*>(ws + L.lkey),
                                reinterpret_cast<int*>(ws + L.ldone));
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
