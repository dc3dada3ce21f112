// chamfer_slab.hip -- nndistance forward for evenly sampled clouds of BASELINE config 2's size class, as ONE kernel
// that sorts and searches inside the LDS (round 4; replaces the reference's NmDistanceKernel x 2,
// _ext/nmdistance_cuda.cu:7-49,118-140, with the same outputs bit for bit).  OPT-IN (pp_debug_set_nmdistance_tile(-2) /
// PP_NMDISTANCE_TILE=-2): measured at 62 us against the 61 us of the three launches it would replace, plus the two
// early-exit launches behind it -- DESIGN.md 5.1f has the phase times and what they say about the three-launch search.
//
// The three-launch search of chamfer_grid.hip (build -> stage A by tiles -> list kernel) writes both clouds to HBM in
// sorted order and reads them back.  Here a batch element is cut into kSlabs slabs of grid layers along z, one
// 1024-thread workgroup each, and nothing sorted ever leaves the chip.  A workgroup
//   1. reads a sample of both clouds (every 16th group of four points: the same sample in all eight workgroups) for a
//      box -> a 32^3 grid of cubic cells over it; ANY box gives a valid grid (pp::cell_coord clamps into the rim cells,
//      every bound treats the rim as open), the workgroups only have to agree on it;
//   2. reads ITS eighth of the two clouds, once, and hands every point to the slab that owns its layer (and to the
//      neighbour whose halo the layer is): 16-byte records (x, y, z, index | launch tag) in its own part of that slab's
//      hand-off area in the workspace, then the counts, tagged with the launch's nonce;
//   3. waits (bounded) for the eight counts addressed to it, loads its ~6000 records into registers, counts them per
//      cell (16-bit counters packed in pairs, LDS atomics), scans, and scatters them into the LDS in cell order: two
//      sorted images of ~3000 points, its own four layers and a halo layer either side;
//   4. answers the queries of its own layers (both directions) from those images with the walk of chamfer_grid.hip's
//      stage A: the 2x2x2 block of cells nearest to the query, settled if the best candidate lies strictly within what
//      the block guarantees (`reach`, the same expression and the same 0.999 slack); else the cube of Chebyshev radius 1
//      (inside the halo by construction), a wave per query; else -- a few queries in a million on a surface -- an
//      every-pair scan of the other cloud by the whole workgroup.  Candidates are compared in the exact (distance,
//      original index) order with pp::chamfer_d3, so the result is the brute force's, bit for bit.
// The hand-off is the only communication between workgroups, and correctness does not depend on how it goes: a count is
// taken only with this launch's nonce, a record only with this launch's tag (re-read a bounded number of times), and a
// slab that does not get what it waits for DECLINES, as does one whose images do not fit, whose clouds are degenerate
// or non-finite, or which meets too many unsettled queries (volumes, clusters, far clouds: not what this form is for).
// Every workgroup leaves a word saying "served" or "declined" (written unconditionally: no initialisation), and the
// launches that follow -- chamfer_grid.hip's build and whole-search kernels -- skip the batch elements all of whose
// slabs were served and redo the others in full.  Placement (the eight workgroups of an element on one XCD, started in
// order) is speed only.
#include <chrono>
#include <random>

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
constexpr int kCapSrc = 640;                   // records one workgroup may hand to one slab, per cloud (an eighth of kCap, + 43 %)
constexpr int kGroupBatch = 4;                 // groups in flight per lane in the every-pair scan
constexpr int kSpinLimit = 1 << 14;            // polls of a hand-off word before the slab gives up (and declines): ~30 ms
constexpr int kRereads = 64;                   // ... and of records whose tag is not this launch's yet
constexpr int kTagShift = 15;                  // a record's index (< 2^15: N, M <= 17408) shares its word with 17 bits of the launch's tag
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

// A group = four consecutive points of a cloud = three 16-byte loads (the clouds are 16-byte aligned and N, M multiples
// of four: the host checks).  A 12-byte load per lane costs the texture unit three passes of a 768-byte span each
// (measured: 8 us per pass over the two clouds of config 2, against 2.7 for this form).
struct Group {
  f4 a, b, c;
  __device__ __forceinline__ void point(int u, float& x, float& y, float& z) const {
    if (u == 0) { x = a.x; y = a.y; z = a.z; }
    else if (u == 1) { x = a.w; y = b.x; z = b.y; }
    else if (u == 2) { x = b.z; y = b.w; z = c.x; }
    else { x = c.y; y = c.z; z = c.w; }
  }
};
__device__ __forceinline__ Group load_group(const float* __restrict__ c1, const float* __restrict__ c2, int G1, int g) {
  const f4* __restrict__ src = g < G1 ? reinterpret_cast<const f4*>(c1) + 3 * (size_t)g
                                      : reinterpret_cast<const f4*>(c2) + 3 * (size_t)(g - G1);
  return Group{src[0], src[1], src[2]};
}

// The hand-off between workgroups goes through memory with per-access coherence (sc0 sc1: a store is written through,
// a load looks past every cache that could hold a stale copy) instead of fences: a device-scope release / acquire on
// gfx950 writes back / invalidates the WHOLE L2 (measured here: 110 us and 30 us per workgroup hand-off), and leaving
// the fences out would tie correctness to the eight workgroups sharing an XCD's L2, which is placement, not a promise.
// (asm: the compiler does not count these loads -- an explicit s_waitcnt vmcnt(0) precedes every use.)
#ifndef PP_SLAB_SCOPE
#define PP_SLAB_SCOPE "sc1"
#endif

__device__ __forceinline__ void store16_coherent(f4* p, f4 v) {
  // (s_nop: a store of more than eight bytes reads its data a cycle after it issues, and the hazard recogniser does not
  //  look inside an asm statement: without it the VALU instruction that follows may overwrite the record on its way out)
  asm volatile("global_store_dwordx4 %0, %1, off " PP_SLAB_SCOPE "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// (four loads and the wait for them in ONE statement: the compiler takes an asm's outputs to be valid when the statement
//  ends -- it may copy them at once --, so a load whose wait is a statement of its own is a load of garbage)
__device__ __forceinline__ void load16x4_coherent(f4& v0, f4& v1, f4& v2, f4& v3, const f4* p0, const f4* p1, const f4* p2,
                                                  const f4* p3) {
  asm volatile("global_load_dwordx4 %0, %4, off " PP_SLAB_SCOPE "\n\tglobal_load_dwordx4 %1, %5, off " PP_SLAB_SCOPE
               "\n\tglobal_load_dwordx4 %2, %6, off " PP_SLAB_SCOPE "\n\tglobal_load_dwordx4 %3, %7, off " PP_SLAB_SCOPE
               "\n\ts_waitcnt vmcnt(0)"
               : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
               : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
               : "memory");
}
__device__ __forceinline__ void store8_coherent(unsigned long long* p, unsigned long long v) {
  asm volatile("global_store_dwordx2 %0, %1, off " PP_SLAB_SCOPE ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long load8_coherent(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off " PP_SLAB_SCOPE "\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The workspace of the kernel: verdicts, hand-off words, hand-off records (chamfer_slab_workspace_bytes)
__host__ __device__ inline size_t off_sync(int B) { return ((size_t)B * kSlabs * 4 + 255) / 256 * 256; }
__host__ __device__ inline size_t off_staging(int B) { return off_sync(B) + (size_t)B * kSlabs * 2 * kSlabs * 8; }
__host__ __device__ inline size_t ws_bytes(int B) { return off_staging(B) + (size_t)B * kSlabs * 2 * kSlabs * kCapSrc * sizeof(f4); }

__global__ __launch_bounds__(kThreads) void chamfer_slab_kernel(const float* __restrict__ xyz1,
                                                                const float* __restrict__ xyz2,
                                                                float* __restrict__ dist1, int* __restrict__ idx1,
                                                                float* __restrict__ dist2, int* __restrict__ idx2,
                                                                unsigned char* __restrict__ wsl, unsigned long long nonce,
                                                                int B, int N, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  f4* s_pts = reinterpret_cast<f4*>(s_raw);                      // [2][kCap + kPad]
  unsigned* s_tab = reinterpret_cast<unsigned*>(s_raw + kOffTab);  // [2][kTabWords]
  f4* s_queue = reinterpret_cast<f4*>(s_raw + kOffQueue);        // [kQueue]: x, y, z, original index | direction << 30
  __shared__ float s_box[kWaves][6];
  __shared__ unsigned s_cnt[2 * kSlabs];  // records handed to slab d of cloud c: [2 d + c]
  __shared__ unsigned s_src[2 * kSlabs];  // records received from workgroup s for cloud c: [8 c + s] (0xFFFF: none, give up)
  __shared__ unsigned s_bad;
  __shared__ unsigned s_wsum[kWaves];
  __shared__ unsigned s_qn, s_ln;
  __shared__ unsigned s_left[kMaxLeft];
  __shared__ unsigned long long s_bf[kMaxLeft];

  // the slabs of a batch element share an XCD (the records they hand to each other stay in that L2): speed only
  const int per_xcd = (B * kSlabs + 7) / 8;
  const int V = pp::xcd_virtual_block((int)blockIdx.x, per_xcd);
  if (V >= B * kSlabs) return;
  const int b = V / kSlabs, slab = V - b * kSlabs;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = pp::wave_id_uniform();
  const float* __restrict__ c1 = xyz1 + (size_t)b * N * 3;
  const float* __restrict__ c2 = xyz2 + (size_t)b * M * 3;
  unsigned* __restrict__ my_state = reinterpret_cast<unsigned*>(wsl) + (size_t)b * kSlabs + slab;
  unsigned long long* __restrict__ sync = reinterpret_cast<unsigned long long*>(wsl + off_sync(B));
  f4* __restrict__ staging = reinterpret_cast<f4*>(wsl + off_staging(B));
  // hand-off (d, c, s): what workgroup s of this batch element found for slab d of cloud c
  auto hand = [&](int d, int c, int sw) { return (((size_t)b * kSlabs + d) * 2 + c) * kSlabs + sw; };
  auto decline = [&](unsigned why) {  // (the low four bits say why: pp_debug_nmdistance_slab_state)
    if (t == 0) *my_state = kDeclined | why;
  };
  PP_SLAB_PHASE_END(9)
  const int G1 = N >> 2, G = G1 + (M >> 2);
  const unsigned tag = (unsigned)nonce & 0x1FFFFu;  // of this launch's records

  // ---------------------------------------------------------------- 1. a box for the grid, from a sample
  // Any box gives a valid grid (pp::cell_coord clamps: a point outside lies in a rim cell, and every bound treats the
  // rim as open); the eight workgroups must only agree on it, so each reads the same sample -- every 16th group of
  // four points -- instead of the clouds.  An evenly sampled cloud loses a thousandth of its extent to the rim.
  {
    float v[6] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};  // -lo, hi
    for (int g = 16 * t; g < G; g += 16 * kThreads) {
      const Group gr = load_group(c1, c2, G1, g);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float x, y, z;
        gr.point(u, x, y, z);
        v[0] = fmaxf(v[0], -x); v[1] = fmaxf(v[1], -y); v[2] = fmaxf(v[2], -z);
        v[3] = fmaxf(v[3], x);  v[4] = fmaxf(v[4], y);  v[5] = fmaxf(v[5], z);
      }
    }
    pp::wave_reduce6_dpp<false, 6>(v);
    if (lane == 63)
      for (int a = 0; a < 6; ++a) s_box[wave][a] = v[a];
  }
  if (t == 0) {
    s_qn = 0u;
    s_ln = 0u;
    s_bad = 0u;
  }
  if (t < 2 * kSlabs) s_cnt[t] = 0u;
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
  // (a NaN in the sample is dropped by the maxima, an infinity is not: either way the workgroups agree)
  const bool box_ok = ext > 0.0f && ext < INFINITY;
  PP_SLAB_PHASE_END(1)
  const float h = ext * (1.0f / (float)kG) * 1.0001f, invh = 1.0f / h;
  const int zbase = slab * kOwnLayers - 1;  // the layer that is local layer 0 (the lower halo; -1 for slab 0)

  // ---------------------------------------------------------------- 2. an eighth of the points, handed to the slabs
  // Workgroup s reads groups [s Gs, (s + 1) Gs) of the two clouds laid end to end and hands every point to the slab
  // that owns its layer -- and to a neighbour's halo when the layer is a slab's first or last: record (x, y, z, index)
  // at a slot of ITS part of that slab's hand-off area (slots from a counter in LDS; nothing is shared between the
  // writers).  Then the count, tagged with this launch's nonce, is published: the reader waits for the tag, so the
  // words need no initialisation, and a stale word (another launch, another process: the nonce is 48 random bits
  // plus a counter) is never taken for this launch's.
  {
    bool bad = !box_ok;
#pragma unroll 1
    for (int cl = 0; cl < 2; ++cl) {  // (a cloud at a time: every lane of a wave then hands to the same counters)
      const int Gc = cl ? G - G1 : G1;
      const int Gs = (Gc + kSlabs - 1) / kSlabs;
      const int gbeg = slab * Gs, gend = min(gbeg + Gs, Gc);
      for (int g0 = gbeg; g0 < gend; g0 += kThreads) {  // (uniform)
        const int g = g0 + t;
        const bool valid = g < gend;
        const Group gr = load_group(c1, c2, G1, (cl ? G1 : 0) + (valid ? g : gend - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float x, y, z;
          gr.point(u, x, y, z);
          bad |= valid && !((x - x) + (y - y) + (z - z) == 0.0f);  // a non-finite coordinate: not for this kernel
          const int cz = cell_coord(z, minz, invh, kG);
          const int d = cz / kOwnLayers, r = cz % kOwnLayers;
          const int dlo = (r == 0 && d > 0) ? d - 1 : d, dhi = (r == kOwnLayers - 1 && d < kSlabs - 1) ? d + 1 : d;
          const f4 rec = {x, y, z, __int_as_float((4 * g + u) | (int)(tag << kTagShift))};
          // slots: one add per slab and wave (lane `dst` adds the wave's count for slab dst), ranks from the ballots
          unsigned mine = 0;
#pragma unroll
          for (int dst = 0; dst < kSlabs; ++dst) {
            const unsigned long long m = __ballot(valid && dst >= dlo && dst <= dhi);
            mine = lane == dst ? (unsigned)__builtin_popcountll(m) : mine;
          }
          unsigned basev = 0;
          if (lane < kSlabs && mine != 0u) basev = atomicAdd(&s_cnt[2 * lane + cl], mine);
#pragma unroll
          for (int dst = 0; dst < kSlabs; ++dst) {
            const unsigned long long m = __ballot(valid && dst >= dlo && dst <= dhi);
            if (m == 0ull) continue;  // (uniform)
            const unsigned base = (unsigned)__builtin_amdgcn_readlane((int)basev, dst);
            const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (valid && dst >= dlo && dst <= dhi && slot < (unsigned)kCapSrc) store16_coherent(&staging[hand(dst, cl, slab) * kCapSrc + slot], rec);
          }
        }
      }
    }
    if (bad) s_bad = 1u;
    wait_vm();  // the records (written through), before the counts
    __syncthreads();
    if (t < 2 * kSlabs) {
      const unsigned cnt = s_cnt[t];
      const unsigned long long word = (nonce << 16) | ((s_bad != 0u || cnt > (unsigned)kCapSrc) ? 0xFFFFull : (unsigned long long)cnt);
      store8_coherent(&sync[hand(t >> 1, t & 1, slab)], word);
    }
  }
  PP_SLAB_PHASE_END(2)
  // ---------------------------------------------------------------- 3. this slab's records, from the eight workgroups
  // (a workgroup that never shows up -- it cannot happen while workgroups start in order and at least 64 of them fit
  //  the device, but nothing here depends on that -- is waited for kSpinLimit polls; then the slab declines)
  if (t < 2 * kSlabs) {
    const unsigned long long* w = &sync[hand(slab, t >> 3, t & 7)];
    unsigned got = 0xFFFFu;
    for (int spin = 0; spin < kSpinLimit; ++spin) {
      const unsigned long long v = load8_coherent(w);
      if ((v >> 16) == nonce) {
        got = (unsigned)(v & 0xFFFFull);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    s_src[t] = got;
  }
  __syncthreads();
  unsigned pre[2][kSlabs + 1];
  bool give_up = false;
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) {
    pre[cl][0] = 0u;
#pragma unroll
    for (int sw = 0; sw < kSlabs; ++sw) {
      const unsigned n = s_src[cl * kSlabs + sw];
      give_up |= n == 0xFFFFu;
      pre[cl][sw + 1] = pre[cl][sw] + (n == 0xFFFFu ? 0u : n);
    }
  }
  const unsigned ns0 = pre[0][kSlabs], ns1 = pre[1][kSlabs];
  if (give_up || ns0 > (unsigned)kCap || ns1 > (unsigned)kCap) {  // (uniform) the images do not fit, a cloud has a
    decline(give_up ? 1u : 2u);                                   // non-finite point, a hand-off overflowed or never came
    return;
  }
  // records -> registers (at most four per cloud and thread), cells counted (16-bit counters, two to a word)
  constexpr int kPer = (kCap + kThreads - 1) / kThreads;  // 4
  f4 rec[2][kPer];
  int cel[2][kPer];
  auto local_cell = [&](float x, float y, float z) -> int {
    return pp::cell_linear(cell_coord(x, minx, invh, kG), cell_coord(y, miny, invh, kG), cell_coord(z, minz, invh, kG) - zbase, kG, kG);
  };
  // (a record whose tag is not this launch's has not arrived yet -- the count may overtake it on its way through the
  //  memory system --: read again, a bounded number of times; then the slab declines.  Whatever the placement of the
  //  workgroups and the coherence of the caches between them, a record that is taken is this launch's.)
  for (int attempt = 0;; ++attempt) {
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const unsigned tot = cl ? ns1 : ns0;
      const f4* src[kPer];
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        const unsigned i = (unsigned)(j * kThreads + t);
        cel[cl][j] = -1;
        const unsigned ii = min(i, tot ? tot - 1u : 0u);  // (a lane without a record re-reads a valid slot)
        int sw = 0;
#pragma unroll
        for (int q = 1; q < kSlabs; ++q) sw += ii >= pre[cl][q] ? 1 : 0;
        unsigned first = 0;
#pragma unroll
        for (int q = 1; q < kSlabs; ++q) first = ii >= pre[cl][q] ? pre[cl][q] : first;
        src[j] = &staging[hand(slab, cl, sw) * kCapSrc + (ii - first)];
      }
      static_assert(kPer == 4, "load16x4_coherent");
      load16x4_coherent(rec[cl][0], rec[cl][1], rec[cl][2], rec[cl][3], src[0], src[1], src[2], src[3]);
    }
    bool late = false;
#pragma unroll
    for (int cl = 0; cl < 2; ++cl)
#pragma unroll
      for (int j = 0; j < kPer; ++j)
        late |= (unsigned)(j * kThreads + t) < (cl ? ns1 : ns0) && ((unsigned)__float_as_int(rec[cl][j].w) >> kTagShift) != tag;
    if (!__syncthreads_or(late ? 1 : 0)) break;
    if (attempt == kRereads) {
      decline(5u);
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
#pragma unroll
  for (int cl = 0; cl < 2; ++cl)
#pragma unroll
    for (int j = 0; j < kPer; ++j) rec[cl][j].w = __int_as_float(__float_as_int(rec[cl][j].w) & ((1 << kTagShift) - 1));
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) {
    const unsigned tot = cl ? ns1 : ns0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if ((unsigned)(j * kThreads + t) < tot) {
        const int c = local_cell(rec[cl][j].x, rec[cl][j].y, rec[cl][j].z);
        cel[cl][j] = c;
        atomicAdd(&s_tab[cl * kTabWords + (c >> 1)], (c & 1) ? 0x10000u : 1u);
      }
    }
  }
  __syncthreads();
  PP_SLAB_PHASE_END(3)
  // scan: entry c := END of cell c (thread t: twelve cells of cloud t / 512)
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
    if (tt == 511) tab[6] = run;  // the sentinel (entry kLocCells): the image's size
  }
  __syncthreads();
  if (t < 2 * kPad) {  // the padding: points whose distance to anything is NaN
    const float qn = __builtin_nanf("");
    const f4 nanp = {qn, qn, qn, __int_as_float(0x7fffffff)};
    s_pts[(t >> 2) * (kCap + kPad) + ((t >> 2) ? ns1 : ns0) + (t & 3)] = nanp;
  }
  // a record takes the slot below its cell's END and lowers it: afterwards entry c is the START of cell c
#pragma unroll
  for (int cl = 0; cl < 2; ++cl)
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int c = cel[cl][j];
      if (c >= 0) {
        const unsigned old = atomicSub(&s_tab[cl * kTabWords + (c >> 1)], (c & 1) ? 0x10000u : 1u);
        const unsigned pos = ((c & 1) ? (old >> 16) : (old & 0xFFFFu)) - 1u;
        s_pts[cl * (kCap + kPad) + pos] = rec[cl][j];
      }
    }
  __syncthreads();
  PP_SLAB_PHASE_END(4)

  // ---------------------------------------------------------------- 4. the 2x2x2 blocks, both directions
  // (chamfer_grid.hip's stage A: the same walk -- groups of four consecutive points of the block's four rows, the
  //  running minimum and the group that last lowered it, the exact order recovered afterwards -- over an image that
  //  was never in memory)
  const float inf = INFINITY;
  constexpr int g1 = kG - 1;
  for (int dir = 0; dir < 2; ++dir) {
    if (dir == 1) { PP_SLAB_PHASE_END(5) }
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
  PP_SLAB_PHASE_END(6)
  // ---------------------------------------------------------------- 5. what the blocks left: the cube of radius 1, a wave per query
  // (about one query in 150 on an evenly sampled surface; the cube lies inside the image by construction)
  const unsigned nqueue = s_qn;
  if (nqueue > (unsigned)kQueue) {  // not a cloud for this kernel (uniform): the batch element is redone by the launches that follow
    decline(3u);
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
    decline(4u);
    return;
  }
  for (unsigned i = 0; i < nleft; ++i) {  // (uniform)
    const f4 w = s_queue[s_left[i]];
    const int dir = (__float_as_int(w.w) >> 30) & 1;
    const f4* __restrict__ rv = reinterpret_cast<const f4*>(dir ? c1 : c2);
    const int Gr = (dir ? N : M) >> 2;
    unsigned long long key = ~0ull;
    auto cand = [&](int id, float x, float y, float z) {
      const float d = pp::chamfer_d3(x, y, z, w.x, w.y, w.z);
      const unsigned long long c = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id;
      key = (d == d && c < key) ? c : key;  // (never a NaN distance)
    };
    for (int g0 = 0; g0 < Gr; g0 += kGroupBatch * kThreads) {
      f4 a[kGroupBatch][3];
#pragma unroll
      for (int u = 0; u < kGroupBatch; ++u) {
        const f4* __restrict__ src = rv + 3 * (size_t)min(g0 + u * kThreads + t, Gr - 1);
        a[u][0] = src[0];
        a[u][1] = src[1];
        a[u][2] = src[2];
      }
#pragma unroll
      for (int u = 0; u < kGroupBatch; ++u) {
        const int g = g0 + u * kThreads + t;
        if (g < Gr) {
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
  return chamfer_slab_shape_ok(B, N, M) && (reinterpret_cast<uintptr_t>(xyz1) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(xyz2) & 15) == 0;
}

static_assert(ppslab::kCapSrc == pp::kSlabKernelCapSrc, "grid_common.h: chamfer_slab_workspace_bytes");
static_assert(pp::kSlabKernelMaxPoints <= (1 << ppslab::kTagShift), "a record's index shares its word with the launch's tag");
static_assert(pp::kSlabKernelMaxPoints < 65536, "16-bit cell tables, a hand-off count in 16 bits");

int chamfer_slab_launch(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2, int* idx2,
                        unsigned char* ws_slab, int B, int N, int M, hipStream_t s) {
  static pp::DeviceFlags lds_ok;
  hipError_t e = pp::allow_big_lds(ppslab::chamfer_slab_kernel, (int)ppslab::kLdsBytes, lds_ok);
  if (e != hipSuccess) return (int)e;
  // this launch's tag of the hand-off words: 48 bits, a random start (per process) plus a counter
  static std::atomic<unsigned long long> tag{[] {
    std::random_device rd;
    return ((unsigned long long)rd() << 32) ^ (unsigned long long)rd() ^
           (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
  }()};
  const unsigned long long nonce = tag.fetch_add(1, std::memory_order_relaxed) & ((1ull << 48) - 1);
  const int grid = 8 * ((B * ppslab::kSlabs + 7) / 8);
  ppslab::chamfer_slab_kernel<<<dim3(grid), dim3(ppslab::kThreads), ppslab::kLdsBytes, s>>>(xyz1, xyz2, dist1, idx1, dist2,
                                                                                          idx2, ws_slab, nonce, B, N, M);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

}  // namespace pp
