// ball_grid.hip -- ball_query through the uniform grid of grid_common.h, with the scan kernels of
// sampling.hip as the fallback.  Same output as the scan (ref _ext/sampling_cuda.cu:340-376): per
// centre the first `nsample` in-radius indices in ASCENDING INDEX order, padded with the first one,
// zeros when the ball is empty.
//
// A hit needs dist3 < r^2 in fp32, hence |p_a - q_a| <= r(1 + 1e-6) on every axis; the cell
// coordinate is a monotone function of the coordinate, so every hit of a centre lies in its cell
// box [cell(lo), cell(hi)] with lo <= q - r' and hi >= q + r' in exact arithmetic, r' = r(1 + 1e-5)
// (the fp32 sums q -+ r' are pushed outwards by more than their rounding error: coordinates may be
// large against r).
//
// The reference's order (ascending index, cut at nsample) is what makes this more than a range
// query.  A wave takes 64 centres that are close in space (the build kernel also sorts the centres,
// along a Morton curve), and
//   1. marks the cells of the 64 boxes in an LDS bitmap (one bit per cell: the union),
//   2. marks the points of those cells in an LDS bitmap indexed by ORIGINAL point index,
//   3. walks that bitmap in order -- the candidates come out sorted by index for free -- and stages
//      them (coordinates from the cloud itself) in LDS, kBqCap at a time,
//   4. lets every lane scan the staged candidates in order with the scan kernel's test
//      (pp::dist3 < r^2) and append its hits to its row, exactly like the brute-force scan but over
//      a few hundred candidates instead of the whole cloud,
//   5. writes the 64 rows, padded, to the centres' original positions.
// Sets whose box would exceed 7x7x7 cells (large radius), or whose grid is useless, are left to the
// scan kernel, launched afterwards for those sets only.
#include "grid_common.h"

namespace pp {
// sampling.hip: the scan kernel over the batch elements whose grid set says "not usable"
int ball_query_scan_unusable(const float* new_xyz, const float* xyz, int* idx, int B, int N, int M,
                             float radius, int nsample, const GridSet* sets, hipStream_t s);
}  // namespace pp

#ifdef PP_BQ_PROBE
// diagnostic build only (tools/bq_phases.py): the 100 MHz clock at the step boundaries of every wave of the query kernel
__device__ unsigned long long g_bqphase[16384][8];
extern "C" int pp_debug_read_bq_phases(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bqphase), sizeof(g_bqphase)); }
#define PP_BQ_MARK(n)                                                                                   \
  do {                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_bqphase[blockIdx.x][n] = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  } while (0)
#else
#define PP_BQ_MARK(n)
#endif

namespace {

using pp::GridSet;
using pp::cell_coord;
using pp::kGridCells;
using pp::kBuildThreads;

constexpr int kBqMaxCells = 3;   // cell-box half-extent the grid path accepts
#ifndef PP_BQ_CAP
#define PP_BQ_CAP 448  // (round 6: 512 -> 448, 11.2 KB of LDS a wave, 14 waves a CU: 0.0960 -> 0.0944 ms at config 4; 384 / 416 / 480: 0.0950 / 0.0955 / 0.0972)
#endif
constexpr int kBqCap = PP_BQ_CAP;  // candidates staged per pass (16 bytes each)
constexpr int kBqMaxN = 524288;  // point bitmap <= 64 KiB

struct BqLayout {
  size_t sets, cell_start, sorted, csorted, total;
};
__host__ __device__ inline BqLayout bq_layout(int B, int N, int M) {
  BqLayout L;
  L.sets = 0;  // [2B]: cloud sets, then the (unused) sets of the centre sort
  L.cell_start = ((size_t)64 * 2 * B + 255) / 256 * 256;
  L.sorted = L.cell_start + ((size_t)4 * (kGridCells + 1) * B + 255) / 256 * 256;
  L.csorted = L.sorted + ((size_t)16 * B * N + 255) / 256 * 256;
  L.total = L.csorted + (size_t)16 * B * M;
  return L;
}

// workgroups [0, S*B): slab s of the cloud of batch element b into its grid; [S*B, 2*S*B): slab s of
// the centres of batch element b into Morton order (S = kBuildSlabs)
template <bool VEC>
__global__ __launch_bounds__(kBuildThreads) void bq_build_kernel(const float* __restrict__ xyz,
                                                                 const float* __restrict__ new_xyz,
                                                                 unsigned char* __restrict__ ws, int B, int N,
                                                                 int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_cnt[];
  const BqLayout L = bq_layout(B, N, M);
  // both sets of a batch element are built on the XCD that will search it (the query kernel's batch ->
  // XCD mapping): virtual order (batch, cloud | queries, slab)
  const int V = pp::xcd_virtual_block(blockIdx.x, (2 * B * pp::kBuildSlabs + 7) / 8);
  if (V >= 2 * B * pp::kBuildSlabs) return;
  const int slab = V % pp::kBuildSlabs;
  const int set = ((V / pp::kBuildSlabs) & 1) * B + V / (2 * pp::kBuildSlabs);
  GridSet* gs = reinterpret_cast<GridSet*>(ws + L.sets) + set;
  if (set >= B) {
    const int b = set - B;
    pp::grid_build_set<true, VEC>(new_xyz + (size_t)b * M * 3, M, gs, nullptr,
                             reinterpret_cast<pp::f4*>(ws + L.csorted) + (size_t)b * M, nullptr, s_cnt, nullptr,
                             nullptr, slab, pp::kBuildSlabs);
    return;
  }
  const int b = set;
  pp::grid_build_set_plain<VEC>(xyz + (size_t)b * N * 3, N, gs,
                                reinterpret_cast<unsigned*>(ws + L.cell_start) + (size_t)b * (kGridCells + 1),
                                reinterpret_cast<pp::f4*>(ws + L.sorted) + (size_t)b * N, s_cnt, slab, pp::kBuildSlabs);
}

// the grid path serves this batch element (otherwise the scan kernel does): the grid exists and the
// cell box of a centre stays within 7 cells per axis
__device__ __forceinline__ bool bq_usable(const GridSet& g, float rpad) {
  return !pp::grid_useless(g) && rpad * g.invh <= (float)kBqMaxCells;
}

// One wave per workgroup; see the file header for the five steps.  LPC lanes share a centre: the
// wave serves G = 64 / LPC centres (a smaller union, more waves), and in step 4 the LPC lanes of a
// centre scan consecutive segments of the staged candidates, so their hits concatenate in order.
template <typename IT, int LPC>
__global__ __launch_bounds__(64) void bq_query_kernel(const float* __restrict__ xyz, int* __restrict__ idx,
                                                      unsigned char* __restrict__ ws, int B, int N, int M,
                                                      float radius2, float rpad, int nsample, int tiles_per_b,
                                                      int per_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int vb = pp::xcd_virtual_block(blockIdx.x, per_xcd);  // a batch element stays on one XCD's L2
  if (vb >= B * tiles_per_b) return;
  const int b = vb / tiles_per_b;
  const int tile = vb - b * tiles_per_b;
  const BqLayout L = bq_layout(B, N, M);
  const GridSet g = reinterpret_cast<const GridSet*>(ws + L.sets)[b];
  const bool usable = bq_usable(g, rpad);
  if (tile == 0 && threadIdx.x == 0)  // the scan kernel, launched next, skips the sets served here
    reinterpret_cast<GridSet*>(ws + L.sets)[b].pad[0] = usable ? 1 : 0;
  if (!usable) return;
  const int ncell = g.gx * g.gy * g.gz;
  const int ncw = (ncell + 31) >> 5;  // words of the cell bitmap
  const int npw = (N + 31) >> 5;      // words of the point bitmap
  // (the cell bitmap and the list of its non-empty words live in steps 1-2, the staged candidates from step 3 on: they
  //  share one region -- 14 KB per wave instead of 18, eleven waves per CU instead of eight)
  static_assert(kGridCells / 8 + kGridCells / 32 * 2 <= kBqCap * 16, "the bitmap and its word list fit the staging area");
  unsigned* s_cell = reinterpret_cast<unsigned*>(s_raw);                 // [kGridCells / 32]
  unsigned short* s_words = reinterpret_cast<unsigned short*>(s_cell + kGridCells / 32);  // step 2: non-empty bitmap words
  float* s_cx = reinterpret_cast<float*>(s_raw);                         // [kBqCap] each, 16-byte aligned
  float* s_cy = s_cx + kBqCap;
  float* s_cz = s_cy + kBqCap;
  int* s_cid = reinterpret_cast<int*>(s_cz + kBqCap);
  unsigned* s_pt = reinterpret_cast<unsigned*>(s_cid + kBqCap);          // [npw]
  IT* s_rows = reinterpret_cast<IT*>(s_pt + ((npw + 3) & ~3));           // [G][nsample + 1]
  __shared__ int s_rng[2];
  const int stride = nsample + 1;

  const unsigned* __restrict__ cell_start =
      reinterpret_cast<const unsigned*>(ws + L.cell_start) + (size_t)b * (kGridCells + 1);
  const pp::f4* __restrict__ sorted = reinterpret_cast<const pp::f4*>(ws + L.sorted) + (size_t)b * N;
  const pp::f4* __restrict__ csorted = reinterpret_cast<const pp::f4*>(ws + L.csorted) + (size_t)b * M;
  const float* __restrict__ cloud = xyz + (size_t)b * N * 3;
  constexpr int G = 64 / LPC;
  const int lane = threadIdx.x;
  const int ci = lane & (G - 1), sub = lane / G;  // centre of the tile; which of its LPC lanes
  const int m0 = tile * G;
  const bool valid = m0 + ci < M;
  PP_BQ_MARK(0);
  const pp::f4 q = csorted[valid ? m0 + ci : M - 1];
  const int qorig = __float_as_int(q.w);  // the centre's row in the output

  // (16-byte stores: both bitmaps start 16-byte aligned and are padded to whole quads of words)
  for (int w = 4 * lane; w < ((ncw + 3) & ~3); w += 256) *reinterpret_cast<uint4*>(&s_cell[w]) = make_uint4(0u, 0u, 0u, 0u);
  for (int w = 4 * lane; w < ((npw + 3) & ~3); w += 256) *reinterpret_cast<uint4*>(&s_pt[w]) = make_uint4(0u, 0u, 0u, 0u);
  if (lane == 0) {
    s_rng[0] = ncw;
    s_rng[1] = -1;
  }
  __syncthreads();
  // 1. cells of the 64 boxes: a row of cells along x is a run of consecutive bits
  if (valid) {
    // bounds rounded OUTWARDS by more than an ulp: q -+ r' is rounded to fp32, and |q| may dwarf r
    auto above = [](float v) { return v + (fabsf(v) * 1.2e-7f + 1e-37f); };
    auto below = [](float v) { return v - (fabsf(v) * 1.2e-7f + 1e-37f); };
    const int x0 = cell_coord(below(q.x - rpad), g.minx, g.invh, g.gx), x1 = cell_coord(above(q.x + rpad), g.minx, g.invh, g.gx);
    const int y0 = cell_coord(below(q.y - rpad), g.miny, g.invh, g.gy), y1 = cell_coord(above(q.y + rpad), g.miny, g.invh, g.gy);
    const int z0 = cell_coord(below(q.z - rpad), g.minz, g.invh, g.gz), z1 = cell_coord(above(q.z + rpad), g.minz, g.invh, g.gz);
    const unsigned long long run = (2ull << (x1 - x0)) - 1ull;  // x1 - x0 + 1 <= 8 ones
    const int ny = y1 - y0 + 1, nzy = (z1 - z0 + 1) * ny;
    const unsigned inv_ny = 65536u / (unsigned)ny + 1u;  // t / ny for t < 64, ny <= 7 by a multiplication (exact there)
    for (int t = sub; t < nzy; t += LPC) {  // the centre's rows, dealt to its LPC lanes
      const int zz = (int)(((unsigned)t * inv_ny) >> 16);
      const int c = ((z0 + zz) * g.gy + y0 + (t - zz * ny)) * g.gx + x0;
      const unsigned long long bits = run << (c & 31);
      atomicOr(&s_cell[c >> 5], (unsigned)bits);
      if (bits >> 32) atomicOr(&s_cell[(c >> 5) + 1], (unsigned)(bits >> 32));
    }
    atomicMin(&s_rng[0], ((z0 * g.gy + y0) * g.gx + x0) >> 5);
    atomicMax(&s_rng[1], ((z1 * g.gy + y1) * g.gx + x1) >> 5);
  }
  __syncthreads();
  PP_BQ_MARK(1);
  // 2. points of the marked cells.  The non-empty bitmap words are first compacted into a list so
  // that the lanes share them evenly; a run of marked cells is contiguous in `sorted`.
  int nwords = 0;
  for (int wb = s_rng[0]; wb <= s_rng[1]; wb += 64) {
    const int w = wb + lane;
    const bool nz = w <= s_rng[1] && s_cell[w] != 0;
    const unsigned long long bal = __ballot(nz);
    if (nz) s_words[nwords + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0))] =
        (unsigned short)w;
    nwords += __builtin_popcountll(bal);
  }
  __syncthreads();
  for (int k = lane; k < nwords; k += 64) {
    const int w = s_words[k];
    unsigned bits = s_cell[w];
    while (bits) {
      const int lo = __builtin_ctz(bits);
      const unsigned inv = ~(bits >> lo);  // bits >> lo starts with a one; its high `lo` bits are zero
      const int len = inv ? __builtin_ctz(inv) : 32;
      const int c = w * 32 + lo;
      const unsigned e = cell_start[c + len];
      for (unsigned i = cell_start[c]; i < e; i += 4) {
        int id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = __float_as_int(sorted[min(i + u, e - 1)].w);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i + u < e) atomicOr(&s_pt[id[u] >> 5], 1u << (id[u] & 31));
      }
      bits = (lo + len >= 32) ? 0u : (bits >> (lo + len)) << (lo + len);
    }
  }
  __syncthreads();
  PP_BQ_MARK(2);
  // 3. ranks: a lane owns a contiguous chunk of bitmap words
  const int per = (npw + 63) >> 6;
  const int w0 = min(npw, lane * per), w1 = min(npw, w0 + per);
  // (round 5: a lane's words are read ONCE, all loads in flight -- up to eight of them, N <= 16384 -- and kept in
  //  registers for the passes below; the prefix sum over the lanes through DPP instead of six ds_bpermute round trips)
  constexpr int kPw = 8;
  unsigned pw[kPw];
#pragma unroll
  for (int i = 0; i < kPw; ++i) pw[i] = (i < per && w0 + i < w1) ? s_pt[w0 + i] : 0u;
  int mine = 0;
  if (per <= kPw) {  // (uniform)
#pragma unroll
    for (int i = 0; i < kPw; ++i) mine += __builtin_popcount(pw[i]);
  } else {
    for (int w = w0; w < w1; ++w) mine += __builtin_popcount(s_pt[w]);
  }
  const int incl = (int)pp::wave_scan_u32_dpp((unsigned)mine);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int mybase = incl - mine;

  IT* myrow = s_rows + (size_t)ci * stride;
  int cnt = valid ? 0 : nsample;  // hits of the centre so far (same in its LPC lanes); idle lanes count as full
  const float r2v = radius2;
  static_assert(kBqCap % 32 == 0, "a pass is whole 32-candidate blocks");
  constexpr int NB = (kBqCap / 32 + LPC - 1) / LPC;  // 32-candidate blocks a lane scans per pass, at most
  for (int base = 0; base < total; base += kBqCap) {
    const int ncand = min(kBqCap, total - base);
    // 3a. indices of the candidates of this pass, ascending
    int r = mybase;
    auto emit = [&](unsigned bits, int w) {
      const int pc = __builtin_popcount(bits);
      if (r + pc <= base) {
        r += pc;
        return;
      }
      while (bits) {
        const int bit = __builtin_ctz(bits);
        bits &= bits - 1;
        if (r >= base && r < base + kBqCap) s_cid[r - base] = w * 32 + bit;
        ++r;
      }
    };
    if (per <= kPw) {  // (uniform) from the registers
#pragma unroll
      for (int i = 0; i < kPw; ++i)
        if (pw[i] != 0u && r < base + kBqCap) emit(pw[i], w0 + i);
    } else {
      for (int w = w0; w < w1 && r < base + kBqCap; ++w) emit(s_pt[w], w);
    }
    __syncthreads();
    if (base == 0) PP_BQ_MARK(3);
    // 3b. coordinates; the tail up to a multiple of 32 gets +inf (never in radius)
    const int padded = (ncand + 31) & ~31;
    for (int c = lane; c < padded; c += 256) {
      int id[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) id[u] = s_cid[min(c + 64 * u, ncand - 1)];
      float v[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // (one 12-byte load per candidate, not three scattered 4-byte ones)
        typedef float f3 __attribute__((ext_vector_type(3)));
        f3 p;
        __builtin_memcpy(&p, cloud + 3 * (size_t)id[u], sizeof(p));
        v[u][0] = p.x; v[u][1] = p.y; v[u][2] = p.z;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = c + 64 * u;
        if (cc < padded) {
          const bool real = cc < ncand;
          s_cx[cc] = real ? v[u][0] : __builtin_inff();
          s_cy[cc] = real ? v[u][1] : __builtin_inff();
          s_cz[cc] = real ? v[u][2] : __builtin_inff();
        }
      }
    }
    __syncthreads();
    if (base == 0) PP_BQ_MARK(4);
    // 4. ordered scan.  Lane `sub` of a centre takes the sub-th segment of the staged candidates, 32
    // at a time: the in-radius flags are shifted into a mask (first candidate of the block ends up
    // in bit 31).  The hit counts of the LPC lanes give each its offset in the row; then the set
    // bits are appended in order.
    const int seg = ((padded / 32 + LPC - 1) / LPC) * 32;
    const int cbeg = sub * seg;
    unsigned hits[NB];
    int mine_hits = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      hits[j] = 0;
      const int c0 = cbeg + 32 * j;
      if (32 * j < seg && c0 < padded) {  // wave-uniform for LPC = 1; per-sub otherwise
#pragma unroll
        for (int u = 0; u < 32; u += 4) {
          const pp::f4 X = *reinterpret_cast<const pp::f4*>(s_cx + c0 + u);
          const pp::f4 Y = *reinterpret_cast<const pp::f4*>(s_cy + c0 + u);
          const pp::f4 Z = *reinterpret_cast<const pp::f4*>(s_cz + c0 + u);
          // (two candidates per instruction: the packed forms give dist3's bits)
          const pp::f2 dA = pp::dist3_pk(q.x, q.y, q.z, X.xy, Y.xy, Z.xy), dB = pp::dist3_pk(q.x, q.y, q.z, X.zw, Y.zw, Z.zw);
          const float d0 = dA.x, d1 = dA.y, d2 = dB.x, d3 = dB.y;
          // hits = 2 * hits + (d < r^2), four times (a pure function of its operands: not volatile, so that the
          // compiler may issue the next groups' LDS reads ahead of it)
          asm(
              "v_cmp_lt_f32 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_lt_f32 vcc, %2, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_lt_f32 vcc, %3, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_lt_f32 vcc, %4, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
              : "+v"(hits[j])
              : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(r2v)
              : "vcc");
        }
        mine_hits += __builtin_popcount(hits[j]);
      }
    }
    int off = cnt, all_hits = 0;
#pragma unroll
    for (int s2 = 0; s2 < LPC; ++s2) {
      const int v = __shfl(mine_hits, ci + G * s2);
      if (s2 < sub) off += v;
      all_hits += v;
    }
    if (off < nsample) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        unsigned h = hits[j];
        while (h && off < nsample) {
          const int p = __builtin_clz(h);
          h &= ~(0x80000000u >> p);
          myrow[off++] = (IT)s_cid[cbeg + 32 * j + p];
        }
      }
    }
    cnt = min(nsample, cnt + all_hits);
    if (__all(cnt >= nsample)) break;
    __syncthreads();
  }
  __syncthreads();
  PP_BQ_MARK(5);
  // 5. rows to the centres' original positions; slot >= count: the pad (first hit, 0 if none)
  if (!valid) cnt = 0;
  const int first = cnt > 0 ? (int)myrow[0] : 0;
  const int nrows = min(G, M - m0);
  const int total_out = nrows * nsample;
  int* __restrict__ gout = idx + (size_t)b * M * nsample;
  // (round 5: a row's count, pad and destination are left in LDS by the row's first lane and read from there -- three
  //  ds_bpermute round trips per output word before, sixteen output words per lane)
  int* s_rowinfo = reinterpret_cast<int*>(s_raw);  // [G][3] at the start of the staging area: the candidates have been scanned
  if (sub == 0) {
    s_rowinfo[3 * ci] = cnt;
    s_rowinfo[3 * ci + 1] = first;
    s_rowinfo[3 * ci + 2] = qorig;
  }
  __syncthreads();
  const bool pow2 = (nsample & (nsample - 1)) == 0;  // (uniform) a shift instead of a division per output word
  const int lg = __builtin_ctz((unsigned)nsample);
  // (round 6) four consecutive words of a row per lane and store where the rows allow it (nsample a multiple of four, the
  // output 16-byte aligned): 4 stores of 1 KB a wave for 16 rows of 64 instead of 16 of 256 bytes -- the step was 11 of the
  // kernel's 80 us
  if ((nsample & 3) == 0 && (reinterpret_cast<uintptr_t>(gout) & 15) == 0) {  // (uniform)
    for (int f0 = 0; f0 < total_out; f0 += 256) {
      const int f = f0 + 4 * lane;
      if (f < total_out) {
        const int row = pow2 ? (f >> lg) : f / nsample;
        const int slot = f - row * nsample;
        const int rc = s_rowinfo[3 * row], rf = s_rowinfo[3 * row + 1], ro = s_rowinfo[3 * row + 2];
        const IT* __restrict__ src = s_rows + (size_t)row * stride + slot;
        int4 v;
        v.x = slot < rc ? (int)src[0] : rf;
        v.y = slot + 1 < rc ? (int)src[1] : rf;
        v.z = slot + 2 < rc ? (int)src[2] : rf;
        v.w = slot + 3 < rc ? (int)src[3] : rf;
        {  // (non-temporal: nothing of the 33 MB is read again by this launch -- 0.0981 -> 0.0960 ms)
          const pp::i4 nv = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(nv, reinterpret_cast<pp::i4*>(gout + (size_t)ro * nsample + slot));
        }
      }
    }
  } else
#pragma unroll 4
  for (int f0 = 0; f0 < total_out; f0 += 64) {
    const int f = min(f0 + lane, total_out - 1);
    const int row = pow2 ? (f >> lg) : f / nsample;
    const int slot = f - row * nsample;
    const int rc = s_rowinfo[3 * row], rf = s_rowinfo[3 * row + 1], ro = s_rowinfo[3 * row + 2];
    const int v = slot < rc ? (int)s_rows[(size_t)row * stride + slot] : rf;
    if (f0 + lane < total_out) gout[(size_t)ro * nsample + slot] = v;
  }
  PP_BQ_MARK(6);
}

}  // namespace

// 0 = automatic (grid when a workspace is given and N >= 4096: below, the scan's ~27 us are less than
// the grid's two launches); 1 = scan kernels only; 2 = grid from N = 2048 (tests)
static pp::Knob g_bq_grid_mode;
static pp::Knob g_bq_lpc;
extern "C" void pp_debug_set_ball_query_lpc(int v) { g_bq_lpc.set(v); }
extern "C" void pp_debug_set_ball_query_search(int v) { g_bq_grid_mode.set(v); }

static size_t bq_query_lds(int N, int nsample, int G) {
  const size_t npw = ((size_t)N + 31) / 32;
  return ((npw + 3) & ~(size_t)3) * 4 + (size_t)kBqCap * 16 + (size_t)G * (nsample + 1) * (N <= 65536 ? 2 : 4);
}

extern "C" size_t pp_ball_query_workspace_bytes(int B, int N, int M, int nsample) {
  if (B <= 0 || M <= 0 || N < (g_bq_grid_mode == 2 ? 2048 : 4096) || N > kBqMaxN || nsample < 1) return 0;
  if ((long long)B * N >= (1LL << 31) || (long long)B * M >= (1LL << 31)) return 0;
  if (bq_query_lds(N, nsample, 16) > 128 * 1024) return 0;  // it has to fit (with the static LDS)
  return bq_layout(B, N, M).total;
}

template <typename IT, int LPC>
static int bq_launch_query(const float* xyz, int* idx, unsigned char* ws, int B, int N, int M, float radius2,
                           float rpad, int nsample, hipStream_t s) {
  constexpr int G = 64 / LPC;
  const int tiles = (M + G - 1) / G;
  const long long per_xcd = ((long long)B * tiles + 7) / 8;
  if (per_xcd * 8 > 0x7fffffffLL) return PP_EINVAL;
  static pp::DeviceFlags ok;
  hipError_t e = pp::allow_big_lds(bq_query_kernel<IT, LPC>, 152 * 1024, ok);
  if (e != hipSuccess) return (int)e;
  bq_query_kernel<IT, LPC><<<dim3((unsigned)(per_xcd * 8)), dim3(64), bq_query_lds(N, nsample, G), s>>>(
      xyz, idx, ws, B, N, M, radius2, rpad, nsample, tiles, (int)per_xcd);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_ball_query_ws_f32(const float* new_xyz, const float* xyz, int* idx, int B, int N, int M,
                                    float radius, int nsample, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  const size_t need = pp_ball_query_workspace_bytes(B, N, M, nsample);
  if (g_bq_grid_mode == 1 || need == 0 || !workspace || workspace_bytes < need || !(radius > 0.0f))
    return pp_ball_query_f32(new_xyz, xyz, idx, B, N, M, radius, nsample, stream);
  if (!new_xyz || !xyz || !idx) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  const float radius2 = radius * radius;  // fp32, as the reference (sampling_cuda.cu:354)
  const float rpad = radius * 1.00001f + 1e-30f;
  static pp::DeviceFlags lds_ok;
  const size_t lds = pp::grid_build_lds_bytes(pp::kBuildSlabs) > pp::grid_build_fast_lds_bytes() ? pp::grid_build_lds_bytes(pp::kBuildSlabs)
                                                                                                 : pp::grid_build_fast_lds_bytes();
  static pp::DeviceFlags lds_ok_vec;
  const bool vec = pp::clouds_vec_aligned(xyz, N, B) && pp::clouds_vec_aligned(new_xyz, M, B);
  hipError_t e = vec ? pp::allow_big_lds(bq_build_kernel<true>, (int)lds, lds_ok_vec) : pp::allow_big_lds(bq_build_kernel<false>, (int)lds, lds_ok);
  if (e != hipSuccess) return (int)e;
  (vec ? bq_build_kernel<true> : bq_build_kernel<false>)<<<dim3(8 * ((2 * B * pp::kBuildSlabs + 7) / 8)), dim3(kBuildThreads), lds, s>>>(xyz, new_xyz, ws, B, N, M);
  PP_RETURN_IF_LAUNCH_FAILED();
  // lanes per centre: 4 unless forced (tuning knob; at config 4, with the 14 KB footprint of round 4: 1 -> 0.122 ms,
  // 2 -> 0.115, 4 -> 0.111, 8 -> see DESIGN 5.3b)
  int lpc = g_bq_lpc ? g_bq_lpc : 4;
  while (lpc < 8 && bq_query_lds(N, nsample, 64 / lpc) > 128 * 1024) lpc *= 2;  // fewer rows per wave if the LDS is short
  int rc;
  if (N <= 65536)
    rc = lpc == 1 ? bq_launch_query<unsigned short, 1>(xyz, idx, ws, B, N, M, radius2, rpad, nsample, s)
       : lpc == 2 ? bq_launch_query<unsigned short, 2>(xyz, idx, ws, B, N, M, radius2, rpad, nsample, s)
       : lpc == 4 ? bq_launch_query<unsigned short, 4>(xyz, idx, ws, B, N, M, radius2, rpad, nsample, s)
                  : bq_launch_query<unsigned short, 8>(xyz, idx, ws, B, N, M, radius2, rpad, nsample, s);
  else
    rc = lpc == 1 ? bq_launch_query<unsigned, 1>(xyz, idx, ws, B, N, M, radius2, rpad, nsample, s)
       : lpc == 2 ? bq_launch_query<unsigned, 2>(xyz, idx, ws, B, N, M, radius2, rpad, nsample, s)
       : lpc == 4 ? bq_launch_query<unsigned, 4>(xyz, idx, ws, B, N, M, radius2, rpad, nsample, s)
                  : bq_launch_query<unsigned, 8>(xyz, idx, ws, B, N, M, radius2, rpad, nsample, s);
  if (rc != PP_OK) return rc;
  const BqLayout L = bq_layout(B, N, M);
  return pp::ball_query_scan_unusable(new_xyz, xyz, idx, B, N, M, radius, nsample,
                                      reinterpret_cast<const GridSet*>(ws + L.sets), s);
}
