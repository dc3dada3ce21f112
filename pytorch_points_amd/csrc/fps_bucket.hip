// fps_bucket.hip -- exact furthest point sampling over a spatially bucketed cloud (gfx950).
// Third decomposition of the reference's furthest_point_sampling_forward_kernel
// (_ext/sampling_cuda.cu:162-233); same picks, same temp, bit for bit (semantics: fps.hip's header,
// SURVEY.md Appendix A.3).
//
// The reference's step visits every point: d2 = min(dist3(x_k, x_old), temp[k]).  A point's temp only
// changes if the new pick is closer than every earlier one, i.e. for the points of the pick's own
// neighbourhood -- N / j of them on average at step j.  So the cloud is sorted ONCE into buckets of 64
// (64 m) points that are neighbours in space, and a step only visits the buckets it can change:
//
//   * per bucket, in the registers of ONE thread of the workgroup: its bounding box, the 64-bit key
//     {float_bits(max temp) : 32 | ~tie_rank : 32} of its best point, and that point's coordinates;
//   * a pick p SKIPS a bucket when box_d2(p) >= max temp, where box_d2 is dist3's own instruction
//     sequence applied to the per-axis gaps max(lo - p, p - hi, 0).  Every fp32 operation of that
//     sequence is monotone, so box_d2 <= dist3(x_k, p) for every point of the box EXACTLY (no slack):
//     then min(d, temp[k]) = temp[k] for all of them -- the step would not have changed a bit there;
//   * a touched bucket is re-evaluated by one wave (a lane per point: the reference's arithmetic,
//     temp written back only where it changed, as the reference does) and its key refreshed;
//   * the next pick is the maximum of the bucket keys: one DPP reduction per wave, one barrier, one
//     16-value reduction.  The winner's coordinates travel with its key through LDS, so the chain of a
//     step holds no dependent global load except the touched buckets' points (L2 hits).
//
// One 1024-thread workgroup per batch element does all of it -- sort, bucket summaries, the serial
// chain -- so nothing waits for another workgroup: no cluster, no polling, no co-residency condition,
// no timeout.  Late steps touch 4-10 buckets (a few hundred points instead of N).
//
// The sort is a counting sort by a 15-bit cell key (LDS histogram, 32768 bins).  The 15 bits are dealt
// to the axes greedily (always halve the axis whose cells are longest, at most 8 bits per axis), and
// interleaved in that order, so surfaces, slabs and volumes all get roughly cubic cells in a
// Z-order-like sequence.  Any order is CORRECT (the boxes are computed from the points); a good one
// is fast.
#include "fps_common.h"

// phase clocks and touched-bucket counts of the chain (tools/fps_bucket_probe.hip defines these; nothing otherwise)
#ifndef PP_FPSB_DOUBLE
#define PP_FPSB_DOUBLE 0  // probe builds: bit k set = one link of the chain is executed twice (same results)
#endif
#ifndef PP_FPSB_STOP
#define PP_FPSB_STOP 99  // probe builds: leave after phase n of the set-up (what each phase costs)
#endif
#ifndef PP_FPSB_PROBE_DECL
#define PP_FPSB_PROBE_DECL
#define PP_FPSB_MARK(n)
#define PP_FPSB_TOUCHED(mask)
#define PP_FPSB_PICKS(k)
#define PP_FPSB_ELIG(e, k)
#define PP_FPSB_AT(j)
#define PP_FPSB_END()
#endif

namespace {

using pp::dist3;
using pp::f4;
using namespace ppfps;

constexpr int kBkThreads = 1024;
constexpr int kBkWaves = kBkThreads / 64;
constexpr int kBkBits = 15;
constexpr int kBkBins = 1 << kBkBits;
constexpr int kBkAxisBits = 10;  // at most this many key bits per axis
constexpr int kBkFine = 1024;    // fine bins per axis: the per-axis histograms the cell boundaries are taken from
constexpr int kBkLdsBytes = kBkBins * 4;

struct BucketGeom {
  int m;     // a bucket = 64 m consecutive points of the sorted order
  int nb;    // buckets per batch element (<= 1024)
  int npad;  // nb * 64 * m
  int naux;  // entries of `aux` per batch element: npad, or 65536 when m == 1 (a register per bucket and lane)
};

__device__ __forceinline__ int wave_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

struct __attribute__((packed, aligned(4))) P3 {
  float x, y, z;
};

__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Wave-wide maximum of 64-bit keys {hi, lo} and the lane that holds it.  One chain of six `v_max_u32_dpp` over the
// high words; the low words only matter when several lanes hold the maximal high word (exactly equal distances:
// duplicates, lattices) -- then, and only then, a second chain over those lanes' low words.  Keys are distinct
// (the tie ranks are), except all-zero keys of padding lanes, which never win against a real point.
__device__ __forceinline__ u64 wave_argmax_key(unsigned hi, unsigned lo, int& src) {
  const unsigned mh = wave_max_u32<6>(hi);
  u64 tied = __ballot(hi == mh);
  unsigned ml;
  if (__builtin_popcountll(tied) > 1) {
    ml = wave_max_u32<6>(hi == mh ? lo : 0u);
    tied = __ballot(hi == mh && lo == ml);
    src = __builtin_ctzll(tied);
  } else {
    src = __builtin_ctzll(tied);
    ml = (unsigned)__builtin_amdgcn_readlane((int)lo, src);
  }
  return ((u64)mh << 32) | ml;
}

// Wave-wide largest and second largest (as a multiset: equal when two lanes hold the largest) of 32-bit unsigned
// values.  One chain of six steps over the pair (x, y) = (largest, second largest so far): t = min(x, x'), x = max(x, x'),
// y = max(y, y', t).  bound_ctrl:0 makes a lane without a source read 0 -- the identity of both max and of the
// min's contribution; the order of the four operations leaves two instructions between every VALU write and the DPP
// read of the same register (the two wait states DPP needs): no s_nop inside the chain.  In the row_bcast steps the
// rows outside the row mask keep a stale t, which their y has already absorbed (t <= y after every step): harmless,
// and only lane 63 is read.
__device__ __forceinline__ void wave_max2_u32(unsigned x, unsigned& m1, unsigned& m2) {
  unsigned y = 0u, t = 0u;
#define PP_MAX2_STEP(CTRL, RM)                                                                \
  "v_min_u32_dpp %2, %0, %0 " CTRL " row_mask:" RM " bank_mask:0xf bound_ctrl:0\n\t"          \
  "v_max_u32_dpp %0, %0, %0 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"                       \
  "v_max_u32_dpp %1, %1, %1 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"                       \
  "v_max_u32 %1, %1, %2\n\t"
  asm volatile("s_nop 1\n\t"
               PP_MAX2_STEP("row_shr:1", "0xf")
               PP_MAX2_STEP("row_shr:2", "0xf")
               PP_MAX2_STEP("row_shr:4", "0xf")
               PP_MAX2_STEP("row_shr:8", "0xf")
               PP_MAX2_STEP("row_bcast:15", "0xa")
               PP_MAX2_STEP("row_bcast:31", "0xc")
               "s_nop 1"
               : "+v"(x), "+v"(y), "+v"(t));
#undef PP_MAX2_STEP
  m1 = (unsigned)__builtin_amdgcn_readlane((int)x, 63);
  m2 = (unsigned)__builtin_amdgcn_readlane((int)y, 63);
}

// wave_argmax_key, and the second largest high word beside it (what every other lane's key is bounded by)
__device__ __forceinline__ u64 wave_argmax_key2(unsigned hi, unsigned lo, int& src, unsigned& hi2) {
  unsigned mh;
  wave_max2_u32(hi, mh, hi2);
  u64 tied = __ballot(hi == mh);
  unsigned ml;
  if (__builtin_popcountll(tied) > 1) {
    ml = wave_max_u32<6>(hi == mh ? lo : 0u);
    tied = __ballot(hi == mh && lo == ml);
    src = __builtin_ctzll(tied);
  } else {
    src = __builtin_ctzll(tied);
    ml = (unsigned)__builtin_amdgcn_readlane((int)lo, src);
  }
  return ((u64)mh << 32) | ml;
}

// The same over each DPP row of sixteen lanes, the results in EVERY lane of the row: four rotations (row_ror:1, 2, 4,
// 8) -- lane i gathers the cyclic window of 2, 4, 8, 16 lanes that ends at i, and the window it merges in is always
// the disjoint one before its own, so the pair stays an exact multiset pair.
__device__ __forceinline__ void row_max2_u32(unsigned x, unsigned& m1, unsigned& m2) {
  unsigned y = 0u, t = 0u;
#define PP_RMAX2_STEP(CTRL)                                                  \
  "v_min_u32_dpp %2, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"          \
  "v_max_u32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"          \
  "v_max_u32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"          \
  "v_max_u32 %1, %1, %2\n\t"
  asm volatile("s_nop 1\n\t"
               PP_RMAX2_STEP("row_ror:1")
               PP_RMAX2_STEP("row_ror:2")
               PP_RMAX2_STEP("row_ror:4")
               PP_RMAX2_STEP("row_ror:8")
               "s_nop 1"
               : "+v"(x), "+v"(y), "+v"(t));
#undef PP_RMAX2_STEP
  m1 = x;
  m2 = y;
}
__device__ __forceinline__ unsigned row_max_u32(unsigned x) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(x));
  return x;
}

// ---- the sort's key plan, shared by the one-workgroup set-up of fps_bucket_kernel and the pre-sort kernels ----
struct KeyPlan {
  float blo[3], bsc[3];  // box origin; fine bins per unit length
  int nbits[3];          // key bits per axis
  unsigned plan;         // the axis of every key bit, two bits per step, first step in the low bits
  int cubic;             // five bits per axis: the cells are walked along a Hilbert curve instead of the Z-order
};
// 15 bits dealt to the axes greedily (always halve the axis whose cells are longest); bv = (-lo, hi) of the cloud
__device__ __forceinline__ KeyPlan make_key_plan(const float (&bv)[6]) {
  KeyPlan kp;
  float e0 = bv[3] + bv[0], e1 = bv[4] + bv[1], e2 = bv[5] + bv[2];  // extents (hi - lo)
  const float ext0 = e0, ext1 = e1, ext2 = e2;
  int n0 = 0, n1 = 0, n2 = 0;
  unsigned plan = 0;
  for (int s = 0; s < kBkBits; ++s) {
    const float c0 = n0 < kBkAxisBits ? e0 : -INFINITY, c1 = n1 < kBkAxisBits ? e1 : -INFINITY,
                c2 = n2 < kBkAxisBits ? e2 : -INFINITY;
    int a = 0;
    float best = c0;
    if (c1 > best) { a = 1; best = c1; }
    if (c2 > best) { a = 2; best = c2; }
    // (nothing compares greater when no axis has a finite extent: take the first axis with room)
    if (a == 0 && n0 >= kBkAxisBits) a = n1 < kBkAxisBits ? 1 : 2;
    plan |= (unsigned)a << (2 * s);
    if (a == 0) { ++n0; e0 *= 0.5f; } else if (a == 1) { ++n1; e1 *= 0.5f; } else { ++n2; e2 *= 0.5f; }
  }
  kp.blo[0] = -bv[0]; kp.blo[1] = -bv[1]; kp.blo[2] = -bv[2];
  kp.nbits[0] = n0; kp.nbits[1] = n1; kp.nbits[2] = n2;
  kp.plan = plan;
  kp.cubic = n0 == 5 && n1 == 5 && n2 == 5;
  // fine bins per unit length; an empty or unbounded extent puts everything into bin 0 of that axis
  kp.bsc[0] = (ext0 > 0.0f && ext0 < INFINITY) ? (float)kBkFine / ext0 : 0.0f;
  kp.bsc[1] = (ext1 > 0.0f && ext1 < INFINITY) ? (float)kBkFine / ext1 : 0.0f;
  kp.bsc[2] = (ext2 > 0.0f && ext2 < INFINITY) ? (float)kBkFine / ext2 : 0.0f;
  return kp;
}
__device__ __forceinline__ int fine_bin(const KeyPlan& kp, float v, int a) {
  return min(max((int)((v - kp.blo[a]) * kp.bsc[a]), 0), kBkFine - 1);
}
// Cells of equal COUNT along an axis: a wave turns the axis' 1024-bin histogram `cnt` into the table `lut` (cell =
// floor(2^n * (points below the bin's middle) / N), its bits at their places in the key; a cubic plan keeps the plain
// coordinate).  `cnt` and `lut` may be the same array.  Every lane of the wave.
__device__ __forceinline__ void axis_lut(const KeyPlan& kp, int a, int N, const unsigned* cnt, unsigned* lut, int lane) {
  const int nbits = kp.nbits[a];
  int carry = 0;
  for (int r = 0; r < kBkFine / 64; ++r) {
    const int i = 64 * r + lane;
    const int c = (int)cnt[i];
    const int inc = wave_scan_incl(c);
    const unsigned below = (unsigned)(carry + inc - c) + (unsigned)c / 2u;
    carry += __builtin_amdgcn_readlane(inc, 63);
    // (below < N <= 2^22 and at most ten bits: 32-bit arithmetic up to 2^21 points -- a 64-bit division is a hundred
    //  instructions in each of the sixteen dependent rounds)
    unsigned q = N <= (1 << 21) ? (below << nbits) / (unsigned)(N > 0 ? N : 1)
                                : (unsigned)(((u64)below << nbits) / (u64)N);
    q = min(q, (1u << nbits) - 1u);
    int left = nbits;
    unsigned val = 0;
    for (int st = 0; st < kBkBits; ++st)
      if ((int)((kp.plan >> (2 * st)) & 3u) == a) {
        --left;
        val |= ((q >> left) & 1u) << (kBkBits - 1 - st);
      }
    lut[i] = kp.cubic ? q : val;
  }
}
// the 15-bit cell key of a point; lut = the three axis tables, [3][kBkFine]
__device__ __forceinline__ unsigned cell_key(const KeyPlan& kp, const unsigned (*lut)[kBkFine], float x, float y, float z) {
  const int qx = fine_bin(kp, x, 0), qy = fine_bin(kp, y, 1), qz = fine_bin(kp, z, 2);
  if (!kp.cubic) return lut[0][qx] | lut[1][qy] | lut[2][qz];
  // Hilbert index of the cell (Skilling's transform, 3 axes x 5 bits): consecutive cells of the curve are always
  // neighbours in space, where the Z-order jumps -- a bucket is a run of 64 points of this order, and a run that
  // spans a jump has a box that covers two distant patches (every pick near either visits it)
  unsigned X0 = lut[0][qx], X1 = lut[1][qy], X2 = lut[2][qz];
#pragma unroll
  for (unsigned Q = 16u; Q > 1u; Q >>= 1) {
    const unsigned P = Q - 1u;
    X0 ^= (X0 & Q) ? P : 0u;
    {
      const unsigned hit = (X1 & Q) ? 0xFFFFFFFFu : 0u;
      const unsigned tt = (X0 ^ X1) & P & ~hit;
      X0 ^= (P & hit) ^ tt;
      X1 ^= tt;
    }
    {
      const unsigned hit = (X2 & Q) ? 0xFFFFFFFFu : 0u;
      const unsigned tt = (X0 ^ X2) & P & ~hit;
      X0 ^= (P & hit) ^ tt;
      X2 ^= tt;
    }
  }
  X1 ^= X0;
  X2 ^= X1;
  unsigned tg = 0u;
#pragma unroll
  for (unsigned Q = 16u; Q > 1u; Q >>= 1) tg ^= (X2 & Q) ? Q - 1u : 0u;
  X0 ^= tg; X1 ^= tg; X2 ^= tg;
  auto spread = [](unsigned v) {  // bit i -> bit 3 i
    v = (v | (v << 8)) & 0x100Fu;
    v = (v | (v << 4)) & 0x10C3u;
    v = (v | (v << 2)) & 0x1249u;
    return v;
  };
  return (spread(X0) << 2) | (spread(X1) << 1) | spread(X2);
}
__device__ __forceinline__ int ordered_int(float v) {  // order-preserving integer image of a float (an involution on the bits)
  const int b = (int)__float_as_uint(v);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float ordered_float(int k) { return __uint_as_float((unsigned)(k ^ ((k >> 31) & 0x7fffffff))); }

constexpr unsigned kRcMax = 0x0FFFFFFFu;  // ~tie rank in 28 bits (rank < N + 512 <= 2^22 + 2^9); four bits below it
                                          // carry the wave number in the workgroup-wide maximum

// REG (buckets of exactly 64 points, N <= 65536): the running minima live in REGISTERS -- lane i of wave w keeps
// temp of point i of each of the wave's 64 buckets (td[l], indexed by the wave-uniform bucket number) -- so a visit is
// one 16-byte load per lane (x, y, z, ~tie rank) and no store at all: nothing in the chain waits for a write to be
// acknowledged.  Otherwise (larger clouds) temp travels in the record's fourth word and ~tie rank in an array of its
// own (`aux`); with REG `aux` holds the incoming temp in sorted order, read once.
template <bool REG, bool BATCH>
__global__ __launch_bounds__(kBkThreads) void fps_bucket_kernel(
    const float* __restrict__ xyz, float* __restrict__ temp, int* __restrict__ idx, int N, int npoint,
    int seed, TieOrder order, BucketGeom geo, f4* __restrict__ sorted_all, unsigned* __restrict__ aux_all,
    float* __restrict__ sampled, int cf, int presorted) {
  extern __shared__ unsigned s_hist[];  // kBkBins counters, then cursors (the sort only)
  __shared__ float s_box[kBkWaves][6];
  __shared__ int s_wsum[kBkWaves];
  __shared__ unsigned s_lut[3][kBkFine];
  __shared__ u64 s_g[3];
  __shared__ float s_c[3][kBkWaves][4];
  // BATCH: the candidate table of a round -- one per DPP row of every wave (slot 4 w + row): key, then (x, y, z, the bound
  // of the other temps of its bucket) -- the largest second-best temp of any row (BOUND) and the largest failing
  // candidate key (FAIL); two deep (two barriers per round)
  __shared__ u64 s_key[2][4 * kBkWaves];
  __shared__ f4 s_rec[2][4 * kBkWaves];
  __shared__ unsigned s_bound[2];
  __shared__ u64 s_fail[2];
  // BATCH: "super-box" l = the sixteen buckets 16 l .. 16 l + 15 (1024 consecutive points of the curve: a compact
  // patch), i.e. lane l of EVERY wave.  Their union boxes (ordered-integer images of -lo, hi: LDS integer maxima), the
  // largest temp inside each (raised by every wave before the round's first barrier) and, per super-box, the candidate
  // slots whose pick can reach it (set by each candidate's OWN wave) -- the last two two deep, like the tables above.
  __shared__ int s_sbox[64][6];
  __shared__ unsigned s_sbmax[2][64];
  __shared__ u64 s_cover[2][64];

  const int b = blockIdx.x;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  float* __restrict__ tmp = temp ? temp + (size_t)b * N : nullptr;  // nullptr: every point starts at 1e10, nothing is written back
  int* __restrict__ out = idx + (size_t)b * npoint;
  f4* __restrict__ sorted = sorted_all + (size_t)b * geo.npad;
  unsigned* __restrict__ rc = aux_all + (size_t)b * geo.naux;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = pp::wave_id_uniform();

  if (t < 3) s_g[t] = 0ull;  // (the one-pick chain's ring: read after the barriers below)
  if (!presorted) {  // (else the pre-sort kernels below have filled `sorted` / `aux`: the chip sorts, not one CU)
  // ---------------------------------------------------------------- A. bounding box of the cloud
  {
    float v[6] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};  // -lo, hi
    for (int k = t; k < N; k += kBkThreads) {
      const float x = p[3 * (size_t)k], y = p[3 * (size_t)k + 1], z = p[3 * (size_t)k + 2];
      v[0] = fmaxf(v[0], -x); v[1] = fmaxf(v[1], -y); v[2] = fmaxf(v[2], -z);
      v[3] = fmaxf(v[3], x);  v[4] = fmaxf(v[4], y);  v[5] = fmaxf(v[5], z);
    }
    pp::wave_reduce6_dpp<false, 6>(v);
    if (lane == 63)
      for (int a = 0; a < 6; ++a) s_box[wave][a] = v[a];
  }
  __syncthreads();
  KeyPlan kp;
  {
    float bv[6];
    for (int a = 0; a < 6; ++a) {
      float m = s_box[0][a];
      for (int w = 1; w < kBkWaves; ++w) m = fmaxf(m, s_box[w][a]);
      bv[a] = m;
    }
    // ---------------------------------------------------------------- B. key plan: 15 bits dealt to the axes
    kp = make_key_plan(bv);
    for (int i = t; i < 3 * kBkFine; i += kBkThreads) (&s_lut[0][0])[i] = 0u;
    for (int i = t; i < kBkBins; i += kBkThreads) s_hist[i] = 0u;
    __syncthreads();
    if (PP_FPSB_STOP <= 1) return;
    // ---------------------------------------------------------------- B2. cells of equal COUNT along every axis
    // A uniform grid gives a cluster a handful of cells, and its buckets -- random subsets of it -- boxes as large
    // as the cluster: every pick there visits all of them.  So the 2^n cells of an axis are cut where its points
    // are: a 1024-bin histogram per axis, its running sum, cell = floor(2^n * (points below the bin's middle) / N).
    // A separable approximation of a k-d split, built in one extra pass; an evenly sampled cloud gets the uniform
    // grid back.
    for (int k = t; k < N; k += kBkThreads) {
      atomicAdd(&s_lut[0][fine_bin(kp, p[3 * (size_t)k], 0)], 1u);
      atomicAdd(&s_lut[1][fine_bin(kp, p[3 * (size_t)k + 1], 1)], 1u);
      atomicAdd(&s_lut[2][fine_bin(kp, p[3 * (size_t)k + 2], 2)], 1u);
    }
    __syncthreads();
    if (wave < 3) axis_lut(kp, wave, N, s_lut[wave], s_lut[wave], lane);
  }
  __syncthreads();
  auto key_of = [&](float x, float y, float z) -> unsigned { return cell_key(kp, s_lut, x, y, z); };
  if (PP_FPSB_STOP <= 2) return;
  // ---------------------------------------------------------------- C. count
  // (a point's key is computed once: it waits for the scatter in the first words of the record buffer, which nothing
  //  else uses before the gather pass -- the Hilbert transform is a hundred instructions a point)
  unsigned* __restrict__ keys = reinterpret_cast<unsigned*>(sorted);
  for (int k = t; k < N; k += kBkThreads) {
    const unsigned key = key_of(p[3 * (size_t)k], p[3 * (size_t)k + 1], p[3 * (size_t)k + 2]);
    keys[k] = key;
    atomicAdd(&s_hist[key], 1u);
  }
  __syncthreads();
  if (PP_FPSB_STOP <= 3) return;
  // ---------------------------------------------------------------- D. exclusive scan (a wave per 2048 bins)
  {
    constexpr int kRows = kBkBins / kBkWaves / 64;
    int carry = 0;
    for (int r = 0; r < kRows; ++r) {
      const int i = wave * (kBkBins / kBkWaves) + 64 * r + lane;
      const int c = (int)s_hist[i];
      const int inc = wave_scan_incl(c);
      s_hist[i] = (unsigned)(carry + inc - c);
      carry += __builtin_amdgcn_readlane(inc, 63);
    }
    if (lane == 0) s_wsum[wave] = carry;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    for (int r = 0; r < kRows; ++r) s_hist[wave * (kBkBins / kBkWaves) + 64 * r + lane] += (unsigned)base;
  }
  __syncthreads();
  if (PP_FPSB_STOP <= 4) return;
  // ---------------------------------------------------------------- E. scatter, as a permutation + a gather
  // (Scattering the 16-byte records themselves -- every lane of a store in a line of its own -- took 183 us of a
  //  470 us set-up; 4-byte scattered stores of the source index, then coalesced record stores, takes a third.)
#ifndef PP_FPSB_E_UNROLL
#define PP_FPSB_E_UNROLL 4
#endif
#pragma unroll PP_FPSB_E_UNROLL
  for (int k = t; k < N; k += kBkThreads) {
    const unsigned pos = atomicAdd(&s_hist[keys[k]], 1u);
    rc[pos] = (unsigned)k;
  }
  __syncthreads();
  if (PP_FPSB_STOP == 45) return;
#pragma unroll PP_FPSB_E_UNROLL
  for (int pos = t; pos < N; pos += kBkThreads) {
    const int k = (int)rc[pos];
    const P3 v = *(const P3*)(p + 3 * (size_t)k);  // (one 12-byte load: a gathered access costs per instruction)
    f4 rec;
    rec.x = v.x; rec.y = v.y; rec.z = v.z;
    const float t0 = tmp ? tmp[k] : 1e10f;  // (ref network/geo_operations.py:33: the caller's fill)
    if (REG) {
      rec.w = __uint_as_float(kRcMax - order.rank(k));
      rc[pos] = __float_as_uint(t0);
    } else {
      rec.w = t0;
      rc[pos] = kRcMax - order.rank(k);
    }
    sorted[pos] = rec;
  }
  // padding behind the last point: temp 0 and the lowest key -- never picked, never rewritten
  for (int pos = N + t; pos < geo.npad; pos += kBkThreads) {
    f4 rec;
    rec.x = 0.0f; rec.y = 0.0f; rec.z = 0.0f; rec.w = 0.0f;
    sorted[pos] = rec;
  }
  for (int pos = N + t; pos < geo.naux; pos += kBkThreads) rc[pos] = 0u;
  __syncthreads();  // (the stores are drained before the barrier; one CU, one L1: visible to every wave)
  }  // !presorted

  if (PP_FPSB_STOP <= 5) return;
  // ---------------------------------------------------------------- F. bucket summaries + the seed's step
  // thread (wave w, lane l) owns bucket l * 16 + w: neighbouring buckets live in different waves, so the
  // handful of buckets a pick touches are re-evaluated side by side.
  auto slot_bucket = [&](int l) -> int { return l * kBkWaves + wave; };
  const int m = geo.m;
  const int bsize = 64 * m;
  float lox = INFINITY, loy = INFINITY, loz = INFINITY, hix = -INFINITY, hiy = -INFINITY, hiz = -INFINITY;
  u64 bkey = 0ull;
  // (BATCH) an upper bound of every temp of the bucket but its best point's: the second-largest temp when the key was
  // last reduced (the others only fall)
  unsigned bsec = 0u;
  float ax = 0.0f, ay = 0.0f, az = 0.0f;
  float ox = p[3 * (size_t)seed], oy = p[3 * (size_t)seed + 1], oz = p[3 * (size_t)seed + 2];
  ox = rl(ox, 0); oy = rl(oy, 0); oz = rl(oz, 0);
  float* __restrict__ smp = sampled ? sampled + (size_t)b * npoint * 3 : nullptr;
  auto put = [&](int j, float x, float y, float z) {
    if (cf) {
      smp[j] = x; smp[(size_t)npoint + j] = y; smp[2 * (size_t)npoint + j] = z;
    } else {
      smp[3 * (size_t)j] = x; smp[3 * (size_t)j + 1] = y; smp[3 * (size_t)j + 2] = z;
    }
  };
  if (t == 0) {
    out[0] = seed;
    if (smp) put(0, ox, oy, oz);
  }
  // evaluate one bucket against the pick (ox, oy, oz): the reference's step for its 64 m points; a lane's best
  // point as (distance bits, ~tie rank) and its coordinates
  // td: lane i's running minimum for point i of each of the wave's 64 buckets.  Buckets 0..31 of the wave: one
  // 32-element register vector, read and written with a WAVE-UNIFORM index (the compiler indexes the register file
  // for that: s_set_gpr_idx_on / v_mov_b32).  Buckets 32..63: the 128 KB of LDS the sort's histogram occupied
  // (a word per lane: conflict-free).  (A 64-entry array indexed by a variable went to scratch memory; a 64-way
  // switch over constant indices cost a thousand cycles a visit in register shuffling; two register vectors met in
  // 32-register copies or one of them in scratch.)
  typedef float f32x32 __attribute__((ext_vector_type(32)));
  f32x32 td_lo;
  float* const s_td = (float*)s_hist + (wave * 32 * 64 + lane);
  auto td_get = [&](int l) -> float { return l < 32 ? td_lo[l] : s_td[(l - 32) * 64]; };
  auto td_set = [&](int l, float v) {
    if (l < 32) td_lo[l] = v; else s_td[(l - 32) * 64] = v;
  };
  auto td_min = [&](int l, float d) -> float {
    float d2 = 0.0f;
    if (REG) {
      d2 = __builtin_fminf(d, td_get(l));
      td_set(l, d2);
    }
    return d2;
  };
  auto visit = [&](int l, int bk, unsigned& whi, unsigned& wlo, float& wx, float& wy, float& wz) {
    f4* __restrict__ sp = sorted + (unsigned)(bk * bsize + lane);
    if (REG) {
      const f4 q = sp[0];
      const float d = dist3(q.x, q.y, q.z, ox, oy, oz);
      const float d2 = td_min(l, d);
      whi = __float_as_uint(d2);
      wlo = __float_as_uint(q.w);
      wx = q.x; wy = q.y; wz = q.z;
      return;
    }
    const unsigned* __restrict__ rp = rc + (unsigned)(bk * bsize + lane);
    if (m == 1) {
      const f4 q = sp[0];
      wlo = rp[0];
      const float d = dist3(q.x, q.y, q.z, ox, oy, oz);
      const float d2 = __builtin_fminf(d, q.w);
      if (d2 != q.w) ((float*)sp)[3] = d2;  // (ref: written only when changed, :203-205)
      whi = __float_as_uint(d2);
      wx = q.x; wy = q.y; wz = q.z;
      return;
    }
    u64 wk = 0ull;
    wx = wy = wz = 0.0f;
    for (int r = 0; r < m; ++r) {
      const f4 q = sp[64 * r];
      const unsigned c = rp[64 * r];
      const float d = dist3(q.x, q.y, q.z, ox, oy, oz);
      const float d2 = __builtin_fminf(d, q.w);
      if (d2 != q.w) ((float*)(sp + 64 * r))[3] = d2;
      const u64 k = ((u64)__float_as_uint(d2) << 32) | c;
      if (k > wk) { wk = k; wx = q.x; wy = q.y; wz = q.z; }
    }
    whi = (unsigned)(wk >> 32);
    wlo = (unsigned)wk;
  };
  if (REG) {  // (`aux` has 65536 entries per batch element in this form; behind the last point: zeros)
#pragma unroll
    for (int l = 0; l < 64; ++l) td_set(l, __uint_as_float(rc[(unsigned)(slot_bucket(l) * 64 + lane)]));
  }
  if (REG && npoint > 1) {
    // (one record per lane and bucket: loaded once, the next bucket's while this one is reduced)
    f4 q = sorted[(unsigned)(slot_bucket(0) * 64 + lane)];  // (inside the padded buffer for every wave: nb >= 32)
    for (int l = 0; l < 64; ++l) {
      const int bk = slot_bucket(l);
      const int bkn = l < 63 ? slot_bucket(l + 1) : geo.nb;
      f4 qn = q;
      if (bkn < geo.nb) qn = sorted[(unsigned)(bkn * 64 + lane)];
      if (bk >= geo.nb) {  // (uniform; an empty slot keeps the empty box and the zero key)
        q = qn;
        continue;
      }
      const bool real = bk * 64 + lane < N;  // (the box is the box of the bucket's real points)
      float v[6] = {real ? -q.x : -INFINITY, real ? -q.y : -INFINITY, real ? -q.z : -INFINITY,
                    real ? q.x : -INFINITY,  real ? q.y : -INFINITY,  real ? q.z : -INFINITY};
      pp::wave_reduce6_dpp<false, 6>(v);
      const float b0 = rl(v[0], 63), b1 = rl(v[1], 63), b2 = rl(v[2], 63), b3 = rl(v[3], 63), b4 = rl(v[4], 63),
                  b5 = rl(v[5], 63);
      const float d2 = td_min(l, dist3(q.x, q.y, q.z, ox, oy, oz));
      int src;
      unsigned sec = 0u;
      const u64 M = BATCH ? wave_argmax_key2(__float_as_uint(d2), __float_as_uint(q.w), src, sec)
                          : wave_argmax_key(__float_as_uint(d2), __float_as_uint(q.w), src);
      const float cx = rl(q.x, src), cy = rl(q.y, src), cz = rl(q.z, src);
      if (lane == l) {
        lox = -b0; loy = -b1; loz = -b2; hix = b3; hiy = b4; hiz = b5;
        bkey = M; ax = cx; ay = cy; az = cz; bsec = sec;
      }
      q = qn;
    }
  } else if (npoint > 1) {
    for (int l = 0; l < 64; ++l) {
      const int bk = l * kBkWaves + wave;
      if (bk >= geo.nb) break;  // (uniform)
      // box of the bucket's real points
      float v[6] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};
      for (int r = 0; r < m; ++r) {
        const int pos = bk * bsize + 64 * r + lane;
        if (pos < N) {
          const f4 q = sorted[pos];
          v[0] = fmaxf(v[0], -q.x); v[1] = fmaxf(v[1], -q.y); v[2] = fmaxf(v[2], -q.z);
          v[3] = fmaxf(v[3], q.x);  v[4] = fmaxf(v[4], q.y);  v[5] = fmaxf(v[5], q.z);
        }
      }
      pp::wave_reduce6_dpp<false, 6>(v);
      const float b0 = rl(v[0], 63), b1 = rl(v[1], 63), b2 = rl(v[2], 63), b3 = rl(v[3], 63), b4 = rl(v[4], 63),
                  b5 = rl(v[5], 63);
      unsigned whi, wlo;
      float wx, wy, wz;
      visit(l, bk, whi, wlo, wx, wy, wz);
      int src;
      const u64 M = wave_argmax_key(whi, wlo, src);
      const float cx = rl(wx, src), cy = rl(wy, src), cz = rl(wz, src);
      if (lane == l) {
        lox = -b0; loy = -b1; loz = -b2; hix = b3; hiy = b4; hiz = b5;
        bkey = M; ax = cx; ay = cy; az = cz;
      }
    }
  }

  if (PP_FPSB_STOP <= 6) return;
  PP_FPSB_PROBE_DECL
  if constexpr (REG && BATCH) {
    // -------------------------------------------------------------- G'. the chain, several picks per round
    // Keys only ever fall, and a pick only lowers the keys of points it is closer to than every earlier pick.  So if
    // c1 > c2 > ... are the largest keys of the cloud, c2 is the pick after c1 provided (i) c1 does not lower c2's own
    // key: !(dist3(c2, c1) < temp[c2]), and (ii) nothing that c1 leaves behind can overtake c2.  Everything outside
    // c1's bucket keeps or lowers a key that was below c2's already; inside c1's bucket every other point's temp is
    // at most the bucket's SECOND-largest temp (`sec`, taken from the bucket's running minima), so temp[c2] > sec(c1)
    // settles (ii).  By induction a whole prefix c1 .. ck of the sorted candidates is the next k picks when every c_j
    // passes (i) and (ii) against every c_i before it -- the very picks, in the very order, of one-at-a-time sampling:
    // the reference's tie rule is the 64-bit key order, which the test never leaves.  min commutes exactly, so
    // applying the k picks to the buckets they touch in one visit gives the same temp.
    //
    // Candidates: the best bucket key of every DPP row (sixteen buckets) of every wave, 64 in all, posted by the lanes
    // that hold them.  What a row did not post has a temp of at most the row's second-largest, so with BOUND = the
    // largest of those over all rows, a candidate whose temp is above BOUND is above every unposted key: the sorted
    // candidates above BOUND are the top of the sorted list of ALL keys.  (The largest candidate, TOP, is the next pick
    // whatever BOUND says.)  Round:
    //   post -> barrier -> a wave tests each ELIGIBLE candidate of its own against the lower ones (a lane per other
    //   candidate: one pair per lane, no sort): a lower one that fails (i) or (ii) raises FAIL (LDS atomic max); it
    //   also marks the super-boxes its pick can reach -> barrier -> the picks are the eligible candidates above FAIL,
    //   a pick's position is the number of eligible keys above it; box tests, one visit per touched bucket.
    // Box tests: 1024 buckets x k picks is what a round would cost with a pass over the wave's buckets per pick (the
    // whole kernel is bound by instruction issue -- sixteen waves, one instruction per wave every four cycles).  The
    // super-box masks are the same for every wave (super-box l = lane l of each), so a lane only tests ITS bucket
    // against the picks that reach ITS super-box: each lane fetches its own pick (ds_bpermute), one pass serves
    // every pick of the round (a second only where two picks reach one super-box).
    // The last pick is taken in a round of its own (every candidate but the largest fails; the rounds before it stop one
    // short of it), so that it is never applied: temp ends as the minimum over every pick but the last (ref :203-205).
    bool win = false;    // this lane's bucket is its row's candidate
    unsigned rsec = 0u;  // the row's second-largest temp
    bool redo = true;
    int buf = 0;
    if (t < 2) { s_bound[t] = 0u; s_fail[t] = 0ull; }
    if (t < 64) {
      for (int a = 0; a < 6; ++a) s_sbox[t][a] = (int)0x80000000;
      s_sbmax[0][t] = 0u; s_sbmax[1][t] = 0u;
      s_cover[0][t] = 0ull; s_cover[1][t] = 0ull;
    }
    __syncthreads();
    {  // (an empty slot's box is (+inf, -inf): the identity)
      const float v[6] = {-lox, -loy, -loz, hix, hiy, hiz};
      for (int a = 0; a < 6; ++a) atomicMax(&s_sbox[lane][a], ordered_int(v[a]));
    }
    __syncthreads();
    float sb[6];  // -lo, hi of super-box `lane`
    for (int a = 0; a < 6; ++a) sb[a] = ordered_float(s_sbox[lane][a]);
    const float sblox = -sb[0], sbloy = -sb[1], sbloz = -sb[2], sbhix = sb[3], sbhiy = sb[4], sbhiz = sb[5];
    const int myslot = 4 * wave + (lane >> 4);
    int j = 1;
    while (j < npoint) {
      PP_FPSB_MARK(0);
      PP_FPSB_AT(j);
      const unsigned khi = (unsigned)(bkey >> 32), klo = (unsigned)bkey;
      if (redo) {  // (a key of this wave changed)
        unsigned top;
        row_max2_u32(khi, top, rsec);
        win = khi == top;
        if (__builtin_popcountll(__ballot(win)) != 4) {  // (uniform) a temp shared inside a row -- or a row of empty
          const unsigned ml = row_max_u32(win ? klo : 0u);  // slots: the tie rank decides (empty slots post the same zeros)
          win = win && klo == ml;
        }
      }
      if (win) {
        f4 r;
        // (what this pick leaves behind in its bucket is also at most its distance to the box's farthest corner: dist3 of
        //  the pick and any point of the box, in dist3's own monotone arithmetic, is no larger -- early in a call, while
        //  every temp is far above a bucket's size, that is the bound that lets a second candidate through)
        const float ex = fmaxf(ax - lox, hix - ax), ey = fmaxf(ay - loy, hiy - ay), ez = fmaxf(az - loz, hiz - az);
        const unsigned diag = __float_as_uint(__builtin_fmaf(ez, ez, __builtin_fmaf(ex, ex, ey * ey)));
        r.x = ax; r.y = ay; r.z = az; r.w = __uint_as_float(bsec < diag ? bsec : diag);
        s_key[buf][myslot] = bkey;
        s_rec[buf][myslot] = r;
      }
      {  // BOUND: one atomic per wave (the rows' values meet in lane 63)
        unsigned wsec = rsec;
        asm volatile("s_nop 1\n\t"
                     "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                     : "+v"(wsec));
        if (lane == 63) __hip_atomic_fetch_max(&s_bound[buf], wsec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      // the largest temp of every super-box (lane l of the sixteen waves)
      __hip_atomic_fetch_max(&s_sbmax[buf][lane], khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      PP_FPSB_MARK(1);
      // LDS only (see the one-pick chain below): no store's acknowledgement is waited for
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PP_FPSB_MARK(2);
      // The last pick is taken alone (a round whose every candidate but the largest fails) and never applied; a round
      // before it takes at most npoint - 1 - j picks: a candidate with that many eligible keys above it fails.
      const int room = npoint - 1 - j;
      const bool last = room == 0;
      const u64 okey = s_key[buf][lane];  // a lane per candidate slot
      const f4 orec = s_rec[buf][lane];
      const unsigned bound = s_bound[buf];
      if (wave == kBkWaves - 1) {  // (the other half of the tables: read for the last time before this round's first
        if (lane == 0) s_fail[buf ^ 1] = 0ull;  // barrier; by the wave that is dealt the fewest candidates to test)
        s_sbmax[buf ^ 1][lane] = 0u;
        s_cover[buf ^ 1][lane] = 0ull;
      }
      const unsigned ohi = (unsigned)(okey >> 32);
      bool o_el = okey != 0ull && ohi > bound;
      u64 elm = __ballot(o_el);
      if (elm == 0ull) {  // (the largest temp is shared: the largest KEY is the next pick, alone)
        const u64 topk = wave_max_u64(okey);  // (every lane active: not behind the && below)
        o_el = okey != 0ull && okey == topk;
        elm = __ballot(o_el);
      }
      // the eligible candidates are dealt to the waves in turn (any wave can test any of them: everything is in the
      // tables): a wave tests the lower candidates against its own, a lane per other candidate
      u64 mym = __ballot(o_el && ((int)__builtin_amdgcn_mbcnt_hi((unsigned)(elm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)elm, 0u)) & (kBkWaves - 1)) == wave);
      if (mym) {
        const float sbm = __uint_as_float(s_sbmax[buf][lane]);
        while (mym) {
          const int ms = __builtin_ctzll(mym);
          mym &= mym - 1ull;
          const u64 mkey = ((u64)(unsigned)__builtin_amdgcn_readlane((int)ohi, ms) << 32) |
                           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)okey, ms);
          const float mx = rl(orec.x, ms), my = rl(orec.y, ms), mz = rl(orec.z, ms);
          const unsigned msec = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(orec.w), ms);
          // a lower eligible candidate fails against this one if this pick lowers its key, or if something this
          // pick leaves behind in its own bucket could still be above it
          const bool lower = o_el && okey < mkey;
          const float dd = dist3(orec.x, orec.y, orec.z, mx, my, mz);
          if (lower && (last || dd < __uint_as_float(ohi) || !(ohi > msec)))
            __hip_atomic_fetch_max(&s_fail[buf], okey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (room < 4 * kBkWaves && !last) {  // (uniform; only the call's last rounds can run out of room)
            const int rank = __builtin_popcountll(__ballot(o_el && okey > mkey));
            if (rank >= room && lane == 0)
              __hip_atomic_fetch_max(&s_fail[buf], mkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          // the super-boxes the pick can reach: the box test with the super-box's box and its largest temp (>= every
          // member's: the test only ever says "visit" more often), a lane per super-box
          const float gx = fmaxf(fmaxf(sblox - mx, mx - sbhix), 0.0f);
          const float gy = fmaxf(fmaxf(sbloy - my, my - sbhiy), 0.0f);
          const float gz = fmaxf(fmaxf(sbloz - mz, mz - sbhiz), 0.0f);
          const float bd = __builtin_fmaf(gz, gz, __builtin_fmaf(gx, gx, gy * gy));
          if (!(bd >= sbm)) atomicOr(&s_cover[buf][lane], 1ull << ms);
        }
      }
      PP_FPSB_MARK(3);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PP_FPSB_MARK(4);
      const u64 fail = s_fail[buf];
      u64 cover = s_cover[buf][lane];
      if (t == 0) s_bound[buf] = 0u;  // (read for the last time before the barrier above)
      const u64 accm = __ballot(o_el && okey > fail);
      const int k = __builtin_popcountll(accm);
      PP_FPSB_PICKS(k);
      PP_FPSB_ELIG(__builtin_popcountll(elm), k);
      // this wave's own candidates among the picks: stored at their positions -- the number of eligible keys above --
      // as ~tie rank (turned into the index after the loop)
      for (unsigned own = (unsigned)(accm >> (4 * wave)) & 15u; own; own &= own - 1u) {
        const int ms = 4 * wave + __builtin_ctz(own);
        const u64 mkey = ((u64)(unsigned)__builtin_amdgcn_readlane((int)ohi, ms) << 32) |
                         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)okey, ms);
        const int rank = __builtin_popcountll(__ballot(o_el && okey > mkey));
        if (lane == 0) {
          out[j + rank] = (int)((unsigned)mkey & kRcMax);
          if (smp) put(j + rank, rl(orec.x, ms), rl(orec.y, ms), rl(orec.z, ms));
        }
      }
      j += k;
      buf ^= 1;
      if (j >= npoint || k == 0) break;  // (the last round: its pick is never applied)
      // Which of my buckets can these picks change?  (Against the bucket's largest temp BEFORE the round: the test only
      // ever says "visit" more often than one pick at a time would.)  ... and does one of them lower the bucket's
      // BEST point?  Only then does the bucket's key change (every other temp only falls): a visit that leaves the
      // best point alone updates the running minima and nothing else -- no reduction, no new key.
      const float bmax = __uint_as_float(khi);
      cover &= accm;    // the picks (by candidate slot) that reach this lane's super-box
      u64 tm = 0ull;    // ... and this lane's bucket
      bool chg = false;
      // (one pick of the lane's list per pass; two per pass on packed fp32 measured slower, round 6: 2.86 against 2.81 ms --
      //  most lanes hold one pick or none, and the wider pass costs every lane)
      while (__ballot(cover != 0ull)) {
        const bool on = cover != 0ull;
        const int sl = on ? __builtin_ctzll(cover) : 0;
        cover &= cover - 1ull;  // (0 stays 0)
        const float px = __int_as_float(__builtin_amdgcn_ds_bpermute(sl << 2, __float_as_int(orec.x)));
        const float py = __int_as_float(__builtin_amdgcn_ds_bpermute(sl << 2, __float_as_int(orec.y)));
        const float pz = __int_as_float(__builtin_amdgcn_ds_bpermute(sl << 2, __float_as_int(orec.z)));
        const float gx = fmaxf(fmaxf(lox - px, px - hix), 0.0f);
        const float gy = fmaxf(fmaxf(loy - py, py - hiy), 0.0f);
        const float gz = fmaxf(fmaxf(loz - pz, pz - hiz), 0.0f);
        const float bd = __builtin_fmaf(gz, gz, __builtin_fmaf(gx, gx, gy * gy));
        const bool hit = on && !(bd >= bmax);
        tm |= hit ? 1ull << sl : 0ull;
        chg |= hit && dist3(ax, ay, az, px, py, pz) < bmax;  // (the visit's own arithmetic)
      }
      u64 mask = __ballot(tm != 0ull);
      const u64 chgm = __ballot(chg);
      redo = chgm != 0ull;
      PP_FPSB_TOUCHED(mask);
      PP_FPSB_MARK(5);
      {
        int l = -1;
        f4 q;
        q.x = q.y = q.z = q.w = 0.0f;
        float told = 0.0f;
        if (mask) {
          l = __builtin_ctzll(mask);
          mask &= mask - 1;
          q = sorted[(unsigned)(slot_bucket(l) * 64 + lane)];
          told = td_get(l);
        }
        while (l >= 0) {
          int ln = -1;
          f4 qn = q;
          float tn = 0.0f;
          if (mask) {  // (the next bucket's record is on its way while this one is evaluated)
            ln = __builtin_ctzll(mask);
            mask &= mask - 1;
            qn = sorted[(unsigned)(slot_bucket(ln) * 64 + lane)];
            tn = td_get(ln);
          }
          u64 tml = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(tm >> 32), l) << 32) |
                    (unsigned)__builtin_amdgcn_readlane((int)(unsigned)tm, l);
          float d2 = told;
          while (tml) {
            const int sl = __builtin_ctzll(tml);
            tml &= tml - 1;
            const float px = rl(orec.x, sl), py = rl(orec.y, sl), pz = rl(orec.z, sl);
            d2 = __builtin_fminf(dist3(q.x, q.y, q.z, px, py, pz), d2);
          }
          td_set(l, d2);
          if ((chgm >> l) & 1ull) {  // (uniform)
            int src;
            unsigned sec;
            const u64 M = wave_argmax_key2(__float_as_uint(d2), __float_as_uint(q.w), src, sec);
            const float cx = rl(q.x, src), cy = rl(q.y, src), cz = rl(q.z, src);
            if (lane == l) { bkey = M; ax = cx; ay = cy; az = cz; bsec = sec; }
          }
          l = ln;
          q = qn;
          told = tn;
        }
      }
      PP_FPSB_MARK(6);
    }
  } else {
    // ---------------------------------------------------------------- G. the chain
    // Keys only ever fall (temp = min(...)), so a wave's best bucket stays its best until that very bucket is
    // re-evaluated: only then is the wave's maximum taken again.  The workgroup's maximum is ONE LDS atomic per wave
    // (ds_max_u64 on a word of a three-deep ring, the wave's number in the key's low bits), one barrier, one read.
    __syncthreads();  // (the ring's words were cleared at the top: no barrier in between when the chip sorted)
    u64 wkey = 0ull;
    int wl = 0;  // the lane whose bucket holds wkey
    float wcx = 0.0f, wcy = 0.0f, wcz = 0.0f;
    bool redo = true;
    int buf = 1;  // j % 3
    for (int j = 1; j < npoint; ++j) {
      PP_FPSB_MARK(0);
      if (redo) {  // this wave's best bucket
        wkey = wave_argmax_key((unsigned)(bkey >> 32), (unsigned)bkey, wl);
        wcx = rl(ax, wl); wcy = rl(ay, wl); wcz = rl(az, wl);
      }
      if (lane == 0) {
        s_c[buf][wave][0] = wcx; s_c[buf][wave][1] = wcy; s_c[buf][wave][2] = wcz;
        __hip_atomic_fetch_max(&s_g[buf], (wkey & 0xFFFFFFFF00000000ull) | ((wkey & (u64)kRcMax) << 4) | (u64)wave,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      PP_FPSB_MARK(1);
      // LDS only: what the loop writes to global memory (temp, the picks) is read back by the SAME wave (a bucket is
      // always visited by its owner's wave) or after the loop's closing __syncthreads -- no store's acknowledgement is
      // waited for here
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (PP_FPSB_DOUBLE & 8) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PP_FPSB_MARK(2);
      const u64 G = s_g[buf];
      const int wv = (int)(G & 15ull);
      ox = s_c[buf][wv][0]; oy = s_c[buf][wv][1]; oz = s_c[buf][wv][2];
      // the pick, as its ~tie rank (turned into the index after the loop: a division, off the chain); one wave per
      // step stores, in turn; and the ring word of the step after the next is cleared (nobody reads it any more)
      const int nbuf = buf == 2 ? 0 : buf + 1;
      if (t == ((j & (kBkWaves - 1)) << 6)) {
        out[j] = (int)((unsigned)(G >> 4) & kRcMax);
        if (smp) put(j, ox, oy, oz);
        s_g[nbuf == 2 ? 0 : nbuf + 1] = 0ull;
      }
      buf = nbuf;
      if (j == npoint - 1) break;  // (ref: temp ends as the minimum over every pick but the last)
      PP_FPSB_MARK(3);
      // which of my buckets can this pick change?
      const float gx = fmaxf(fmaxf(lox - ox, ox - hix), 0.0f);
      const float gy = fmaxf(fmaxf(loy - oy, oy - hiy), 0.0f);
      const float gz = fmaxf(fmaxf(loz - oz, oz - hiz), 0.0f);
      const float bd = __builtin_fmaf(gz, gz, __builtin_fmaf(gx, gx, gy * gy));
      const float bmax = __uint_as_float((unsigned)(bkey >> 32));
      u64 mask = __ballot(!(bd >= bmax));
      if (PP_FPSB_DOUBLE & 16) {  // the box test and its ballot once more, dependent on the first
        float gx2 = fmaxf(fmaxf(lox - ox, ox - hix), __uint_as_float((unsigned)mask & 0u));
        asm volatile("" : "+v"(gx2));
        const float bd2 = __builtin_fmaf(gz, gz, __builtin_fmaf(gx2, gx2, gy * gy));
        mask = __ballot(!(bd2 >= bmax));
      }
      if (PP_FPSB_DOUBLE & 32) mask = 0;  // no visits at all (WRONG picks: the cost of a step without its visits)
      redo = (mask >> wl) & 1ull;
      PP_FPSB_TOUCHED(mask);
      PP_FPSB_MARK(4);
      if (REG) {
        // A step lasts as long as its busiest wave (one step in nine has a wave with two or more buckets to visit):
        // the next bucket's record is loaded before this one is evaluated, so every visit after the first costs its
        // arithmetic only.
        const unsigned at = (unsigned)(wave * 64 + lane);
        int l = 0;
        f4 q;
        float told = 0.0f;
        if (mask) {
          l = __builtin_ctzll(mask);
          mask &= mask - 1;
          q = sorted[at + (unsigned)l * (kBkWaves * 64)];
          told = td_get(l);
        } else {
          l = -1;
        }
        while (l >= 0) {
          int ln = -1;
          f4 qn = q;
          float tn = 0.0f;
          if (mask) {
            ln = __builtin_ctzll(mask);
            mask &= mask - 1;
            qn = sorted[at + (unsigned)ln * (kBkWaves * 64)];
            tn = td_get(ln);
          }
          if (PP_FPSB_DOUBLE & 1) {  // a second, dependent load of the same record
            const unsigned zero = __float_as_uint(q.x) & 0u;
            asm volatile("" ::: "memory");
            q = sorted[at + (unsigned)l * (kBkWaves * 64) + zero];
          }
          if (PP_FPSB_DOUBLE & 2) td_set(l, __builtin_fminf(INFINITY, td_get(l)));
          const float d2 = __builtin_fminf(dist3(q.x, q.y, q.z, ox, oy, oz), told);
          td_set(l, d2);
          const unsigned whi = __float_as_uint(d2);
          int src;
          if (PP_FPSB_DOUBLE & 4) {
            u64 M0 = wave_argmax_key(whi, __float_as_uint(q.w), src);
            asm volatile("" : "+s"(src), "+s"(M0));
          }
          const u64 M = wave_argmax_key(whi, __float_as_uint(q.w), src);
          const float cx = rl(q.x, src), cy = rl(q.y, src), cz = rl(q.z, src);
          if (lane == l) { bkey = M; ax = cx; ay = cy; az = cz; }
          l = ln;
          q = qn;
          told = tn;
        }
      } else {
        while (mask) {
          const int l = __builtin_ctzll(mask);
          mask &= mask - 1;
          unsigned whi, wlo;
          float wx, wy, wz;
          visit(l, l * kBkWaves + wave, whi, wlo, wx, wy, wz);
          int src;
          const u64 M = wave_argmax_key(whi, wlo, src);
          const float cx = rl(wx, src), cy = rl(wy, src), cz = rl(wz, src);
          if (lane == l) { bkey = M; ax = cx; ay = cy; az = cz; }
        }
      }
      PP_FPSB_MARK(5);
    }
  }
  PP_FPSB_END();
  __syncthreads();
  // ---------------------------------------------------------------- H. picks as indices; temp back in place
  for (int j = 1 + t; j < npoint; j += kBkThreads) out[j] = order.unrank(kRcMax - (unsigned)out[j]);
  if (!tmp) return;  // (the caller keeps no running minima: furthest_point_sample's own temp, never read back)
  if (REG) {
    unsigned first = (unsigned)lane;
    asm volatile("" : "+v"(first));  // (computed afresh: sixty-four addresses kept alive across the chain spilled)
#pragma unroll
    for (int l = 0; l < 64; ++l) {
      const unsigned pos = first + (unsigned)(slot_bucket(l) * 64);
      if (pos < (unsigned)N) tmp[order.unrank(kRcMax - __float_as_uint(sorted[pos].w))] = td_get(l);
    }
  } else {
    for (int pos = t; pos < N; pos += kBkThreads) tmp[order.unrank(kRcMax - rc[pos])] = sorted[pos].w;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The sort by the whole chip (clouds of 16384 points and more): the one-workgroup set-up above spends 0.28 ms of a
// 2.9 ms call at config 3 sorting 65536 points on ONE CU (gathered loads, one line per lane) while 240 CUs idle.  The
// same counting sort -- same plan, same cell tables, same keys -- as six short launches of kPre workgroups per
// batch element, its tables in the workspace:
//   box (a part per workgroup) -> per-axis fine histograms -> cell tables -> cell counts -> scan -> scatter.
// No atomic leaves a CU: a workgroup counts ITS share of the points in an LDS histogram and writes the histogram out as
// a slab of its own; the scan turns the kPre slabs into kPre sets of cursors (cell-major, slab-minor); the scatter
// loads its slab's cursors back into LDS.  (The first form had all kPre workgroups of an element add into ONE table with
// device-scope atomics: 54 + 85 us for the two passes at config 3 -- those atomics execute at the memory side of the
// eight L2s.)  The order of the points INSIDE a cell is by slab, then whatever the LDS atomics make it (as in the
// one-workgroup form): any order is correct, the boxes are computed from the points.
constexpr int kPre = 16;
constexpr int kPreThreads = 1024;
constexpr int kPreGroup = kBkBins / kPre;  // cells a workgroup of the scan owns
struct PreTables {  // per batch element
  float part[kPre][8];              // (-lo, hi) of every workgroup's share of the points
  unsigned fine[3][kBkFine];        // per-axis histograms, then the cell tables
  unsigned gsum[kPre][kPre];        // [slab][group of kPreGroup cells]: points of the slab in the group
  unsigned slab[kPre][kBkBins];     // cell counts of every workgroup's share, then its cursors
};

__device__ __forceinline__ void pre_range(int N, int& k0, int& k1) {
  const int per = (N + kPre - 1) / kPre;
  k0 = min(N, (int)blockIdx.x * per);
  k1 = min(N, k0 + per);
}
__device__ __forceinline__ KeyPlan pre_plan(const PreTables* tb) {  // (every thread of the workgroup)
  __shared__ float s_bv[6];
  if (threadIdx.x < 6) {
    float m = tb->part[0][threadIdx.x];
    for (int c = 1; c < kPre; ++c) m = fmaxf(m, tb->part[c][threadIdx.x]);
    s_bv[threadIdx.x] = m;
  }
  __syncthreads();
  float bv[6];
  for (int a = 0; a < 6; ++a) bv[a] = s_bv[a];
  return make_key_plan(bv);
}

__global__ __launch_bounds__(kPreThreads) void fps_pre_box_kernel(const float* __restrict__ xyz, int N, PreTables* tabs) {
  __shared__ float s_part[kPreThreads / 64][6];
  PreTables* tb = tabs + blockIdx.y;
  const float* __restrict__ p = xyz + (size_t)blockIdx.y * N * 3;
  const int t = threadIdx.x, lane = t & 63;
  // (this element's tables of the later passes are cleared here: nothing reads them before the next launch)
  for (int i = blockIdx.x * kPreThreads + t; i < 3 * kBkFine; i += kPre * kPreThreads) (&tb->fine[0][0])[i] = 0u;
  int k0, k1;
  pre_range(N, k0, k1);
  float v[6] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};  // -lo, hi
  for (int k = k0 + t; k < k1; k += kPreThreads) {
    const float x = p[3 * (size_t)k], y = p[3 * (size_t)k + 1], z = p[3 * (size_t)k + 2];
    v[0] = fmaxf(v[0], -x); v[1] = fmaxf(v[1], -y); v[2] = fmaxf(v[2], -z);
    v[3] = fmaxf(v[3], x);  v[4] = fmaxf(v[4], y);  v[5] = fmaxf(v[5], z);
  }
  pp::wave_reduce6_dpp<false, 6>(v);
  if (lane == 63)
    for (int a = 0; a < 6; ++a) s_part[t >> 6][a] = v[a];
  __syncthreads();
  if (t < 6) {
    float m = s_part[0][t];
    for (int w = 1; w < kPreThreads / 64; ++w) m = fmaxf(m, s_part[w][t]);
    tb->part[blockIdx.x][t] = m;
  }
}

__global__ __launch_bounds__(kPreThreads) void fps_pre_fine_kernel(const float* __restrict__ xyz, int N, PreTables* tabs) {
  __shared__ unsigned s_fine[3][kBkFine];
  PreTables* tb = tabs + blockIdx.y;
  const float* __restrict__ p = xyz + (size_t)blockIdx.y * N * 3;
  const int t = threadIdx.x;
  const KeyPlan kp = pre_plan(tb);
  for (int i = t; i < 3 * kBkFine; i += kPreThreads) (&s_fine[0][0])[i] = 0u;
  __syncthreads();
  int k0, k1;
  pre_range(N, k0, k1);
  for (int k = k0 + t; k < k1; k += kPreThreads) {
    atomicAdd(&s_fine[0][fine_bin(kp, p[3 * (size_t)k], 0)], 1u);
    atomicAdd(&s_fine[1][fine_bin(kp, p[3 * (size_t)k + 1], 1)], 1u);
    atomicAdd(&s_fine[2][fine_bin(kp, p[3 * (size_t)k + 2], 2)], 1u);
  }
  __syncthreads();
  for (int i = t; i < 3 * kBkFine; i += kPreThreads) {
    const unsigned c = (&s_fine[0][0])[i];
    if (c) atomicAdd(&(&tb->fine[0][0])[i], c);
  }
}

// the per-axis histograms to cell tables, in place (a wave per axis: sixteen dependent scans, once per element)
__global__ __launch_bounds__(192) void fps_pre_lut_kernel(int N, PreTables* tabs) {
  __shared__ unsigned s_cnt[3][kBkFine];  // (the sixteen rounds read their counts from LDS: one trip to memory, not sixteen)
  PreTables* tb = tabs + blockIdx.x;
  const KeyPlan kp = pre_plan(tb);
  const int lane = threadIdx.x & 63, wave = pp::wave_id_uniform();
  for (int i = lane; i < kBkFine; i += 64) s_cnt[wave][i] = tb->fine[wave][i];
  axis_lut(kp, wave, N, s_cnt[wave], tb->fine[wave], lane);
}

// SCATTER = false: this workgroup's cell counts, into its slab; true: its records to their places (after the scan).
// Dynamic LDS: the kBkBins counters / cursors.
template <bool SCATTER, bool REG>
__global__ __launch_bounds__(kPreThreads) void fps_pre_cell_kernel(const float* __restrict__ xyz, const float* __restrict__ temp,
                                                                   int N, TieOrder order, BucketGeom geo, PreTables* tabs,
                                                                   f4* __restrict__ sorted_all, unsigned* __restrict__ aux_all) {
  extern __shared__ unsigned s_cell[];  // kBkBins
  __shared__ unsigned s_lut[3][kBkFine];
  PreTables* tb = tabs + blockIdx.y;
  unsigned* __restrict__ mine = tb->slab[blockIdx.x];
  const float* __restrict__ p = xyz + (size_t)blockIdx.y * N * 3;
  const int t = threadIdx.x;
  const KeyPlan kp = pre_plan(tb);
#pragma unroll
  for (int i = 0; i < 3 * kBkFine / kPreThreads; ++i) (&s_lut[0][0])[i * kPreThreads + t] = (&tb->fine[0][0])[i * kPreThreads + t];
#pragma unroll 8
  for (int i = 4 * t; i < kBkBins; i += 4 * kPreThreads)
    *reinterpret_cast<uint4*>(&s_cell[i]) = SCATTER ? *reinterpret_cast<const uint4*>(&mine[i]) : make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  int k0, k1;
  pre_range(N, k0, k1);
  if (!SCATTER) {
#pragma unroll 4
    for (int k = k0 + t; k < k1; k += kPreThreads)
      atomicAdd(&s_cell[cell_key(kp, s_lut, p[3 * (size_t)k], p[3 * (size_t)k + 1], p[3 * (size_t)k + 2])], 1u);
    __syncthreads();
    // the slab, and its points per group of cells (what the scan's workgroups start from)
    unsigned part = 0u;  // thread t: cells [32 t, 32 t + 32), all inside group t / (kPreGroup / 32)
    for (int i = 0; i < kBkBins / kPreThreads; i += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(&s_cell[(kBkBins / kPreThreads) * t + i]);
      *reinterpret_cast<uint4*>(&mine[(kBkBins / kPreThreads) * t + i]) = v;
      part += v.x + v.y + v.z + v.w;
    }
    part = (unsigned)wave_scan_incl((int)part);  // (a wave = 2048 cells = one group)
    if ((t & 63) == 63) tb->gsum[blockIdx.x][t >> 6] = part;
    return;
  }
  f4* __restrict__ sorted = sorted_all + (size_t)blockIdx.y * geo.npad;
  unsigned* __restrict__ rc = aux_all + (size_t)blockIdx.y * geo.naux;
  const float* __restrict__ tmp = temp ? temp + (size_t)blockIdx.y * N : nullptr;
#pragma unroll 4
  for (int k = k0 + t; k < k1; k += kPreThreads) {
    f4 rec;
    rec.x = p[3 * (size_t)k]; rec.y = p[3 * (size_t)k + 1]; rec.z = p[3 * (size_t)k + 2];
    const unsigned pos = atomicAdd(&s_cell[cell_key(kp, s_lut, rec.x, rec.y, rec.z)], 1u);
    const float t0 = tmp ? tmp[k] : 1e10f;  // (ref network/geo_operations.py:33: the caller's fill)
    if (REG) {
      rec.w = __uint_as_float(kRcMax - order.rank(k));
      rc[pos] = __float_as_uint(t0);
    } else {
      rec.w = t0;
      rc[pos] = kRcMax - order.rank(k);
    }
    sorted[pos] = rec;
  }
  if (blockIdx.x == 0) {  // padding behind the last point: temp 0 and the lowest key -- never picked, never rewritten
    for (int pos = N + t; pos < geo.npad; pos += kPreThreads) {
      f4 rec;
      rec.x = 0.0f; rec.y = 0.0f; rec.z = 0.0f; rec.w = 0.0f;
      sorted[pos] = rec;
    }
    for (int pos = N + t; pos < geo.naux; pos += kPreThreads) rc[pos] = 0u;
  }
}
static_assert(kBkBins / kPreThreads == 32 && kPreGroup == 32 * 64 && kPreThreads / 64 == kPre, "a wave of the count = a group of the scan");

// The slabs' counts into cursors: position of (cell, slab) = points of lower cells in every slab + points of this cell in
// lower slabs.  Workgroup g of an element owns the cells [g kPreGroup, (g + 1) kPreGroup): two cells a thread.
__global__ __launch_bounds__(kPreThreads) void fps_pre_scan_kernel(PreTables* tabs) {
  __shared__ unsigned s_w[kPreThreads / 64];
  PreTables* tb = tabs + blockIdx.y;
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned base = 0u;  // points of the groups below this one
  for (int i = lane; i < kPre * kPre; i += 64) base += ((i % kPre) < g) ? (&tb->gsum[0][0])[i] : 0u;
  base = (unsigned)__builtin_amdgcn_readlane(wave_scan_incl((int)base), 63);
  const int c0 = g * kPreGroup + 2 * t;  // this thread's two cells
  unsigned v[kPre][2];
  unsigned tot = 0u;
#pragma unroll
  for (int s = 0; s < kPre; ++s) {
    const uint2 q = *reinterpret_cast<const uint2*>(&tb->slab[s][c0]);
    v[s][0] = q.x; v[s][1] = q.y;
    tot += q.x + q.y;
  }
  const unsigned inc = (unsigned)wave_scan_incl((int)tot);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  unsigned run = base + inc - tot;
  for (int w = 0; w < wave; ++w) run += s_w[w];
  unsigned at0 = run, at1 = run;
#pragma unroll
  for (int s = 0; s < kPre; ++s) at1 += v[s][0];
#pragma unroll
  for (int s = 0; s < kPre; ++s) {
    *reinterpret_cast<uint2*>(&tb->slab[s][c0]) = make_uint2(at0, at1);
    at0 += v[s][0];
    at1 += v[s][1];
  }
}

pp::DeviceFlags g_bucket_lds[3];
pp::Knob g_bucket_chain;  // 1: one pick per round (the round-4 chain)
pp::Knob g_bucket_sort;   // 1: always the one-workgroup sort

}  // namespace

namespace ppfps {

// buckets of 64 m points, at most 1024 of them (one per thread)
static BucketGeom bucket_geom(int N) {
  BucketGeom g;
  g.m = (N + 65535) / 65536;
  if (g.m < 1) g.m = 1;
  g.nb = (N + 64 * g.m - 1) / (64 * g.m);
  g.npad = g.nb * 64 * g.m;
  g.naux = g.m == 1 ? 65536 : g.npad;
  return g;
}

bool bucket_applies(int B, int N, int npoint) {
  (void)B;
  // the sort pays for itself after a few dozen steps; tie ranks and positions are 32-bit
  return N >= 2048 && N <= (1 << 22) && npoint >= 32;
}

// the sort by the whole chip pays from 32768 points (the one-workgroup set-up of a smaller cloud is a few dozen us)
static bool presort_sizes(int N) { return N >= 32768; }
static bool presort_applies(int N) { return presort_sizes(N) && (int)g_bucket_sort != 1; }

size_t bucket_workspace_bytes(int B, int N) {
  const BucketGeom g = bucket_geom(N);
  // (the pre-sort's tables wherever it can run, whatever the knob says: the size must not depend on a debug switch)
  return (size_t)B * ((size_t)g.npad * sizeof(f4) + (size_t)g.naux * sizeof(unsigned) +
                      (presort_sizes(N) ? sizeof(PreTables) : 0));
}

int bucket_launch(const float* xyz, float* temp, int* idx, int B, int N, int npoint, int seed, TieOrder order,
                  void* ws, float* sampled, int cf, hipStream_t s) {
  const BucketGeom g = bucket_geom(N);
  f4* sorted = (f4*)ws;
  unsigned* aux = (unsigned*)((char*)ws + (size_t)B * g.npad * sizeof(f4));
  PreTables* tabs = (PreTables*)((char*)aux + (size_t)B * g.naux * sizeof(unsigned));
  const int pre = presort_applies(N) ? 1 : 0;
  if (pre) {
    const dim3 grid(kPre, B);  // (the first launch clears the tables of the later ones: no memset)
    fps_pre_box_kernel<<<grid, dim3(kPreThreads), 0, s>>>(xyz, N, tabs);
    fps_pre_fine_kernel<<<grid, dim3(kPreThreads), 0, s>>>(xyz, N, tabs);
    fps_pre_lut_kernel<<<dim3(B), dim3(192), 0, s>>>(N, tabs);
    static pp::DeviceFlags pre_lds[4];
    hipError_t e = pp::allow_big_lds(fps_pre_cell_kernel<false, true>, kBkLdsBytes, pre_lds[0]);
    if (e == hipSuccess) e = pp::allow_big_lds(fps_pre_cell_kernel<false, false>, kBkLdsBytes, pre_lds[1]);
    if (e == hipSuccess) e = pp::allow_big_lds(fps_pre_cell_kernel<true, true>, kBkLdsBytes, pre_lds[2]);
    if (e == hipSuccess) e = pp::allow_big_lds(fps_pre_cell_kernel<true, false>, kBkLdsBytes, pre_lds[3]);
    if (e != hipSuccess) return (int)e;
    if (g.m == 1) fps_pre_cell_kernel<false, true><<<grid, dim3(kPreThreads), kBkLdsBytes, s>>>(xyz, temp, N, order, g, tabs, sorted, aux);
    else fps_pre_cell_kernel<false, false><<<grid, dim3(kPreThreads), kBkLdsBytes, s>>>(xyz, temp, N, order, g, tabs, sorted, aux);
    fps_pre_scan_kernel<<<grid, dim3(kPreThreads), 0, s>>>(tabs);
    if (g.m == 1) fps_pre_cell_kernel<true, true><<<grid, dim3(kPreThreads), kBkLdsBytes, s>>>(xyz, temp, N, order, g, tabs, sorted, aux);
    else fps_pre_cell_kernel<true, false><<<grid, dim3(kPreThreads), kBkLdsBytes, s>>>(xyz, temp, N, order, g, tabs, sorted, aux);
  }
  // several picks per round from 32768 points or 1024 picks (measured round 6: 0.906 against 0.933 ms at 16 x 16384 ->
  // 1024, 0.743 against 0.817 at 64 x 4096 -> 1024, but 0.508 against 0.490 at 32 x 8192 -> 512: the first fifty picks
  // of any call come one per round, at two barriers each)
  const int chain = (int)g_bucket_chain;
  if (g.m == 1 && (chain == 2 || (chain == 0 && (N >= 32768 || npoint >= 1024)))) {
    hipError_t e = pp::allow_big_lds(fps_bucket_kernel<true, true>, kBkLdsBytes, g_bucket_lds[2]);
    if (e != hipSuccess) return (int)e;
    fps_bucket_kernel<true, true><<<dim3(B), dim3(kBkThreads), kBkLdsBytes, s>>>(xyz, temp, idx, N, npoint, seed,
                                                                                order, g, sorted, aux, sampled, cf, pre);
  } else if (g.m == 1) {
    hipError_t e = pp::allow_big_lds(fps_bucket_kernel<true, false>, kBkLdsBytes, g_bucket_lds[0]);
    if (e != hipSuccess) return (int)e;
    fps_bucket_kernel<true, false><<<dim3(B), dim3(kBkThreads), kBkLdsBytes, s>>>(xyz, temp, idx, N, npoint, seed,
                                                                                 order, g, sorted, aux, sampled, cf, pre);
  } else {
    hipError_t e = pp::allow_big_lds(fps_bucket_kernel<false, false>, kBkLdsBytes, g_bucket_lds[1]);
    if (e != hipSuccess) return (int)e;
    fps_bucket_kernel<false, false><<<dim3(B), dim3(kBkThreads), kBkLdsBytes, s>>>(xyz, temp, idx, N, npoint, seed,
                                                                                  order, g, sorted, aux, sampled, cf, pre);
  }
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

}  // namespace ppfps

extern "C" void pp_debug_set_fps_bucket_chain(int form) { g_bucket_chain.set(form); }
extern "C" void pp_debug_set_fps_bucket_sort(int mode) { g_bucket_sort.set(mode); }
