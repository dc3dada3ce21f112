// chamfer.hip -- nearest-neighbour (Chamfer) distance forward / labeled forward / backward for
// gfx950.  Replaces the reference's _ext/nmdistance_cuda.cu (NmDistanceKernel :7-49,
// LabeledNmDistanceKernel :55-115, NmDistanceGradKernel :168-185 and their launchers).
//
// Forward, C == 3 (the hot kernel)
//   * one lane owns Q queries (coordinates in VGPRs); a workgroup = 4 waves = 64*Q queries; the
//     four waves scan the four quarters of the reference cloud and merge through LDS, so a
//     (batch, direction) pair yields 4x the waves of a one-wave-per-tile split;
//   * the reference point is wave-uniform: it is fetched with scalar loads (s_load) into SGPRs
//     and used directly as the SGPR operand of v_sub_f32 -- no LDS staging, no per-pair vector
//     memory traffic;
//   * exact "min, lowest index on ties" without a compare+2 selects per pair: reference points
//     are scanned in groups of G; inside a group only the running minimum is kept (v_min3_f32,
//     one op per two pairs) and after the group one compare/select records the id of the FIRST
//     group that lowered the minimum.  After the scan each query recomputes the G distances of
//     its recorded group (bit-identical arithmetic) and takes the first one equal to the minimum.
//     VALU cost per pair: 3 sub + 1 mul + 2 fma + 1/2 min3 + ~13/(Q*G) bookkeeping instead of 9
//     (and v_cmp/v_cndmask/v_min3 issue at half the rate of add/mul/fma on gfx950, so the saving
//     in issue cycles is larger than in instruction count: 14 cycles per pair per SIMD vs 24).
//   * both directions (xyz1->xyz2, xyz2->xyz1) are one launch; virtual block ids are dealt to the
//     8 XCDs in contiguous ranges so that workgroups sharing a reference cloud share an L2.
#include "pp_common.h"

// phase marks (tools/bwd_probe.hip defines PP_PHASE to record a clock; nothing otherwise)
#ifndef PP_PHASE
#define PP_PHASE(n)
#endif

namespace {

using pp::chamfer_d3;

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;

// PK: the distance arithmetic of two queries is issued as packed fp32 (v_pk_add/mul/fma_f32 on
// register pairs, the wave-uniform reference coordinate broadcast from an SGPR).  Same IEEE
// operations per element, so the bits do not change; measured on MI355X a v_sub_f32 with an SGPR
// operand issues at half the rate of the VGPR-only form while v_pk_add_f32 with an SGPR source
// does two subtractions in the same slot (tools/valu_microbench2.hip, DESIGN.md "VALU roof").
// LAB: labeled Chamfer (ref LabeledNmDistanceKernel, nmdistance_cuda.cu:55-115): a reference point
// is a candidate only if its label equals the query's (compared as floats, :89) -- its distance
// is replaced by +inf otherwise (one v_cmp_eq + one v_cndmask per pair); a query whose minimum is
// still +inf has no candidate: idx -1, dist 0 (:110-113).
template <int Q, int G, bool PK, bool PF, bool LAB>
__device__ __forceinline__ void nmdist_tile(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, float* __restrict__ dist1,
    int* __restrict__ idx1, float* __restrict__ dist2, int* __restrict__ idx2, int N, int M, const int b,
    const bool second, const int tile, const float* __restrict__ label1, const float* __restrict__ label2) {
  static_assert(!(LAB && PF), "labels are not combined with the prefetch form");
  static_assert(G % 2 == 0, "groups are consumed two reference points per v_min3");
  constexpr int TQ = 64 * Q;  // queries per workgroup
  __shared__ float s_best[kWavesPerBlock][TQ];
  __shared__ int s_idx[kWavesPerBlock][TQ];

  const int nq = second ? M : N;
  const int nr = second ? N : M;
  const float* __restrict__ qry = (second ? xyz2 : xyz1) + (size_t)b * nq * 3;
  const float* __restrict__ ref = (second ? xyz1 : xyz2) + (size_t)b * nr * 3;

  const int wave = pp::wave_id_uniform();
  const int lane = threadIdx.x & 63;
  const int count = nq;

  const float* __restrict__ qlab = LAB ? (second ? label2 : label1) + (size_t)b * nq : nullptr;
  const float* __restrict__ rlab = LAB ? (second ? label1 : label2) + (size_t)b * nr : nullptr;
  float qx[Q], qy[Q], qz[Q], best[Q], lq[Q];
  int gid[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    int j = tile * TQ + i * 64 + lane;
    j = j < count ? j : count - 1;  // clamp: out-of-range lanes compute a valid query, never stored
    lq[i] = LAB ? qlab[j] : 0.0f;
    qx[i] = qry[3 * (size_t)j + 0];
    qy[i] = qry[3 * (size_t)j + 1];
    qz[i] = qry[3 * (size_t)j + 2];
    best[i] = __builtin_inff();
    gid[i] = -1;
  }

  // ---- grouped scan of this wave's quarter of the reference cloud -----------------------------
  const int ngroups = nr / G;
  const int g0 = (int)(((long long)ngroups * wave) / kWavesPerBlock);  // this wave's quarter of the groups
  const int g1 = (int)(((long long)ngroups * (wave + 1)) / kWavesPerBlock);
  auto load_group = [&](float (&rr)[G * 3], int g) {
    const float* __restrict__ rp = ref + (size_t)g * (G * 3);  // wave-uniform -> s_load
#pragma unroll
    for (int e = 0; e < G * 3; ++e) rr[e] = rp[e];
  };
  auto scan_group = [&](const float (&rr)[G * 3], int g) {
    float nb[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) nb[i] = best[i];
    float rl[G];
    if constexpr (LAB) {
      const float* __restrict__ lp = rlab + (size_t)g * G;  // wave-uniform -> s_load
#pragma unroll
      for (int e = 0; e < G; ++e) rl[e] = lp[e];
    }
    if constexpr (PK) {
      static_assert(!PK || Q % 2 == 0, "packed form pairs the queries of a lane");
#pragma unroll
      for (int p = 0; p < G; p += 2) {
#pragma unroll
        for (int i = 0; i < Q; i += 2) {
          const pp::f2 x2 = {qx[i], qx[i + 1]}, y2 = {qy[i], qy[i + 1]}, z2 = {qz[i], qz[i + 1]};
          pp::f2 da = pp::chamfer_d3_pk(rr[3 * p + 0], rr[3 * p + 1], rr[3 * p + 2], x2, y2, z2);
          pp::f2 db = pp::chamfer_d3_pk(rr[3 * p + 3], rr[3 * p + 4], rr[3 * p + 5], x2, y2, z2);
          if constexpr (LAB) {
            const float inf = __builtin_inff();
            da.x = lq[i] == rl[p] ? da.x : inf;
            da.y = lq[i + 1] == rl[p] ? da.y : inf;
            db.x = lq[i] == rl[p + 1] ? db.x : inf;
            db.y = lq[i + 1] == rl[p + 1] ? db.y : inf;
          }
          nb[i] = pp::min3(da.x, db.x, nb[i]);
          nb[i + 1] = pp::min3(da.y, db.y, nb[i + 1]);
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < G; p += 2) {
#pragma unroll
        for (int i = 0; i < Q; ++i) {
          float da = chamfer_d3(rr[3 * p + 0], rr[3 * p + 1], rr[3 * p + 2], qx[i], qy[i], qz[i]);
          float db = chamfer_d3(rr[3 * p + 3], rr[3 * p + 4], rr[3 * p + 5], qx[i], qy[i], qz[i]);
          if constexpr (LAB) {
            da = lq[i] == rl[p] ? da : __builtin_inff();
            db = lq[i] == rl[p + 1] ? db : __builtin_inff();
          }
          nb[i] = pp::min3(da, db, nb[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      gid[i] = nb[i] < best[i] ? g : gid[i];  // first group that attains the running minimum
      best[i] = nb[i];
    }
  };
  if constexpr (PF) {
    // ping-pong SGPR sets: the scalar loads of group g+1 are in flight while group g is scanned.
    // Scalar loads return out of order (only lgkmcnt(0) exists), so each set is "used" before the
    // next loads are issued -- otherwise waiting for it would also wait for them.
    if (g0 < g1) {
      float ra[G * 3], rb[G * 3];
      load_group(ra, g0);
      int g = g0;
      for (; g + 1 < g1; g += 2) {
        load_group(rb, g + 1);
        scan_group(ra, g);
        asm volatile("" ::"s"(rb[0]), "s"(rb[G * 3 - 1]));
        load_group(ra, g + 2 < g1 ? g + 2 : g1 - 1);
        scan_group(rb, g + 1);
        asm volatile("" ::"s"(ra[0]), "s"(ra[G * 3 - 1]));
      }
      if (g < g1) scan_group(ra, g);
    }
  } else {
    for (int g = g0; g < g1; ++g) {
      float rr[G * 3];
      load_group(rr, g);
      scan_group(rr, g);
    }
  }

  // ---- recover the index inside the recorded group ------------------------------------------
  int bidx[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    int k = 0;
    if (gid[i] >= 0) {
      const int k0 = gid[i] * G;
      const float* rp = ref + 3 * (size_t)k0;
#pragma unroll
      for (int p = G - 1; p >= 0; --p) {  // descending: the last hit kept is the lowest index
        float d = chamfer_d3(rp[3 * p + 0], rp[3 * p + 1], rp[3 * p + 2], qx[i], qy[i], qz[i]);
        if constexpr (LAB) d = lq[i] == rlab[k0 + p] ? d : __builtin_inff();
        k = d == best[i] ? k0 + p : k;
      }
    }
    bidx[i] = k;
  }

  // ---- tail (nr % G points, highest indices): exact compare/select, done by the last wave ------
  if (wave == kWavesPerBlock - 1) {
    for (int k = ngroups * G; k < nr; ++k) {
      const float rx = ref[3 * (size_t)k + 0], ry = ref[3 * (size_t)k + 1], rz = ref[3 * (size_t)k + 2];
      const float rlk = LAB ? rlab[k] : 0.0f;
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        float d = chamfer_d3(rx, ry, rz, qx[i], qy[i], qz[i]);
        if constexpr (LAB) d = lq[i] == rlk ? d : __builtin_inff();
        const bool lt = d < best[i];
        best[i] = lt ? d : best[i];
        bidx[i] = lt ? k : bidx[i];
      }
    }
  }

  // ---- merge the four quarters (ascending index ranges => strict < keeps the lowest index) ----
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    s_best[wave][i * 64 + lane] = best[i];
    s_idx[wave][i * 64 + lane] = bidx[i];
  }
  __syncthreads();
  float* __restrict__ od = (second ? dist2 : dist1) + (size_t)b * nq;
  int* __restrict__ oi = (second ? idx2 : idx1) + (size_t)b * nq;
  for (int e = threadIdx.x; e < TQ; e += kBlock) {
    float bb = s_best[0][e];
    int bi = s_idx[0][e];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) {
      const float c = s_best[w][e];
      const int ci = s_idx[w][e];
      const bool lt = c < bb;
      bb = lt ? c : bb;
      bi = lt ? ci : bi;
    }
    int j = tile * TQ + e;
    if (j < count) {
      if constexpr (LAB) {
        const bool none = !(bb < __builtin_inff());  // no reference point with this query's label
        bb = none ? 0.0f : bb;
        bi = none ? -1 : bi;
      }
      od[j] = bb;
      oi[j] = bi;
    }
  }
}

template <int Q, int G, bool PK, bool PF, bool LAB = false>
__global__ __launch_bounds__(kBlock) void nmdist_fwd_c3_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, float* __restrict__ dist1,
    int* __restrict__ idx1, float* __restrict__ dist2, int* __restrict__ idx2, int N, int M,
    int tiles1, int tiles2, int total, int per_xcd, const float* __restrict__ label1 = nullptr,
    const float* __restrict__ label2 = nullptr) {
  const int V = pp::xcd_virtual_block(blockIdx.x, per_xcd);
  if (V >= total) return;  // uniform per workgroup
  const int per_b = tiles1 + tiles2;
  const int b = V / per_b;
  const int r = V - b * per_b;
  const bool second = r >= tiles1;
  nmdist_tile<Q, G, PK, PF, LAB>(xyz1, xyz2, dist1, idx1, dist2, idx2, N, M, b, second,
                                 second ? r - tiles1 : r, label1, label2);
}

// The same tile for the directions the grid search has routed here (chamfer_grid.hip, round 6: a direction whose
// queries see every reference point at nearly one distance cannot be pruned -- shell against a cluster at its centre,
// identical points, a degenerate reference set -- and the search's in-kernel scans take twice this kernel's time).
// routed[(2 b + direction) * stride] != 0: this kernel serves the direction; workgroups of the others leave at once.
template <int Q, int G, bool PK>
__global__ __launch_bounds__(kBlock) void nmdist_fwd_c3_routed_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, float* __restrict__ dist1,
    int* __restrict__ idx1, float* __restrict__ dist2, int* __restrict__ idx2, int N, int M,
    int tiles1, int tiles2, int total, int per_xcd, const unsigned* __restrict__ routed, int stride) {
  const int V = pp::xcd_virtual_block(blockIdx.x, per_xcd);
  if (V >= total) return;  // uniform per workgroup
  const int per_b = tiles1 + tiles2;
  const int b = V / per_b;
  const int r = V - b * per_b;
  const bool second = r >= tiles1;
  if (routed[(size_t)(2 * b + (second ? 1 : 0)) * stride] == 0u) return;
  nmdist_tile<Q, G, PK, false, false>(xyz1, xyz2, dist1, idx1, dist2, idx2, N, M, b, second,
                                      second ? r - tiles1 : r, nullptr, nullptr);
}

// Generic point dimension (C != 3): one lane per query, reference point wave-uniform, plain
// compare/select.  CT > 0: compile-time C, query in registers; CT == 0: run-time C, query re-read
// from memory (L1) for every reference point.  Correctness path, not a tuned one.
template <int CT, bool PAD>
__global__ __launch_bounds__(256) void nmdist_fwd_generic_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, float* __restrict__ dist1,
    int* __restrict__ idx1, float* __restrict__ dist2, int* __restrict__ idx2, int N, int M, int C,
    int tiles1, int tiles2) {
  // CT > 0: the query sits in CT registers.  PAD (CT = 16 serves 9 <= C <= 16): coordinates beyond C are
  // zeros on both sides, and fma(0, 0, d) == d leaves the distance untouched bit for bit
  const int c = (CT > 0 && !PAD) ? CT : C;
  const int per_b = tiles1 + tiles2;
  const int b = blockIdx.x / per_b;
  const int r = blockIdx.x - b * per_b;
  const bool second = r >= tiles1;
  const int tile = second ? r - tiles1 : r;
  const int nq = second ? M : N;
  const int nr = second ? N : M;
  const float* __restrict__ qry = (second ? xyz2 : xyz1) + (size_t)b * nq * c;
  const float* __restrict__ ref = (second ? xyz1 : xyz2) + (size_t)b * nr * c;
  const int j = tile * 256 + threadIdx.x;
  if (j >= nq) return;
  const float* qp = qry + (size_t)j * c;
  float q[CT > 0 ? CT : 1];
  if (CT > 0) {
#pragma unroll
    for (int e = 0; e < CT; ++e) q[e] = (!PAD || e < c) ? qp[e] : 0.0f;
  }
  float best = __builtin_inff();
  int bi = 0;
  for (int k = 0; k < nr; ++k) {
    const float* rp = ref + (size_t)k * c;
    float d = 0.0f;
    if (CT > 0) {
#pragma unroll
      for (int e = 0; e < CT; ++e) {
        const float t = ((!PAD || e < c) ? rp[e] : 0.0f) - q[e];  // rp is wave-uniform: scalar loads
        d = __builtin_fmaf(t, t, d);
      }
    } else {
      for (int e = 0; e < c; ++e) {
        const float t = rp[e] - qp[e];
        d = __builtin_fmaf(t, t, d);
      }
    }
    const bool lt = d < best;
    best = lt ? d : best;
    bi = lt ? k : bi;
  }
  ((second ? dist2 : dist1) + (size_t)b * nq)[j] = best;
  ((second ? idx2 : idx1) + (size_t)b * nq)[j] = bi;
}

// Labeled forward (ref _ext/nmdistance_cuda.cu:55-115): candidates restricted to equal labels
// (compared as floats, :89); no candidate -> idx -1, dist 0 (:110-113).
__global__ __launch_bounds__(256) void labeled_nmdist_fwd_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, const float* __restrict__ label1,
    const float* __restrict__ label2, float* __restrict__ dist1, int* __restrict__ idx1,
    float* __restrict__ dist2, int* __restrict__ idx2, int N, int M, int C, int tiles1, int tiles2) {
  const int per_b = tiles1 + tiles2;
  const int b = blockIdx.x / per_b;
  const int r = blockIdx.x - b * per_b;
  const bool second = r >= tiles1;
  const int tile = second ? r - tiles1 : r;
  const int nq = second ? M : N;
  const int nr = second ? N : M;
  const float* __restrict__ qry = (second ? xyz2 : xyz1) + (size_t)b * nq * C;
  const float* __restrict__ ref = (second ? xyz1 : xyz2) + (size_t)b * nr * C;
  const float* __restrict__ ql = (second ? label2 : label1) + (size_t)b * nq;
  const float* __restrict__ rl = (second ? label1 : label2) + (size_t)b * nr;
  const int j = tile * 256 + threadIdx.x;
  if (j >= nq) return;
  const float* qp = qry + (size_t)j * C;
  const float l1 = ql[j];
  float best = __builtin_inff();
  int bi = -1;
  if (C == 3) {
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    for (int k = 0; k < nr; ++k) {
      const float d = chamfer_d3(ref[3 * (size_t)k], ref[3 * (size_t)k + 1], ref[3 * (size_t)k + 2], qx, qy, qz);
      const bool take = (l1 == rl[k]) & ((bi < 0) | (d < best));  // (bitwise: no exec-mask branches)
      best = take ? d : best;
      bi = take ? k : bi;
    }
  } else {
    for (int k = 0; k < nr; ++k) {
      const float* rp = ref + (size_t)k * C;
      float d = 0.0f;
      for (int e = 0; e < C; ++e) {
        const float t = rp[e] - qp[e];
        d = __builtin_fmaf(t, t, d);
      }
      const bool take = (l1 == rl[k]) & ((bi < 0) | (d < best));  // (bitwise: no exec-mask branches)
      best = take ? d : best;
      bi = take ? k : bi;
    }
  }
  ((second ? dist2 : dist1) + (size_t)b * nq)[j] = bi < 0 ? 0.0f : best;
  ((second ? idx2 : idx1) + (size_t)b * nq)[j] = bi;
}

// Backward (ref _ext/nmdistance_cuda.cu:168-185,195-221).  Two passes on one stream:
//   own:     gradxyzA[j] = 2*gdA[j] * (xA[j] - xB[idxA[j]])           plain stores, one writer per row
//            (this overwrites, so the reference's zero_() of the outputs is folded in)
//   scatter: gradxyzB[idxA[j]] -= the same value                      fp32 atomics
// for A,B = (1,2) and (2,1).  g = gd*2 then g*(xa-xb): two roundings in the reference's order (:176,:179).
template <bool SCATTER>
__global__ __launch_bounds__(256) void nmdist_bwd_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, const float* __restrict__ gd1,
    const float* __restrict__ gd2, const int* __restrict__ idx1, const int* __restrict__ idx2,
    float* __restrict__ gx1, float* __restrict__ gx2, int N, int M, int C, long long total1,
    long long total2) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total1 + total2) return;
  const bool second = t >= total1;
  const long long p = second ? t - total1 : t;  // flat (b, j)
  const int na = second ? M : N, nb = second ? N : M;
  const long long b = p / na;
  const float* __restrict__ xa = (second ? xyz2 : xyz1) + p * C;
  const int j2 = (second ? idx2 : idx1)[p];
  float* __restrict__ ga = (second ? gx2 : gx1) + p * C;
  if (j2 < 0) {  // labeled variant: no neighbour (:175)
    if (!SCATTER)
      for (int e = 0; e < C; ++e) ga[e] = 0.0f;
    return;
  }
  const float g = (second ? gd2 : gd1)[p] * 2;
  const float* __restrict__ xb = (second ? xyz1 : xyz2) + (b * nb + j2) * C;
  float* __restrict__ gb = (second ? gx1 : gx2) + (b * nb + j2) * C;
  for (int e = 0; e < C; ++e) {
    const float v = g * (xa[e] - xb[e]);
    if (SCATTER)
      atomicAdd(gb + e, -v);
    else
      ga[e] = v;
  }
}

// Backward without global atomics: one workgroup per (batch, target cloud T, coordinate c) owns
// the whole column gradT[b, :, c] in LDS (4 bytes x n_T).  It seeds the column with the own-point
// terms of T, folds the scattered terms of the other cloud O in with LDS float atomics (on-chip,
// conflict cost only), and writes the column once.  Same terms and roundings as the reference; the
// summation order is unspecified there (global fp32 atomics) and here (LDS fp32 atomics).
// Replaces 3.1 M scattered global atomics (~160 us at config 2) by ~10 us of LDS traffic.
__global__ __launch_bounds__(1024) void nmdist_bwd_lds_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, const float* __restrict__ gd1,
    const float* __restrict__ gd2, const int* __restrict__ idx1, const int* __restrict__ idx2,
    float* __restrict__ gx1, float* __restrict__ gx2, int N, int M, int C) {
  extern __shared__ __attribute__((aligned(16))) float s_acc[];
  const int per_b = 2 * C;
  const int b = blockIdx.x / per_b;
  const int r = blockIdx.x - b * per_b;
  const bool second = r >= C;  // target cloud: false -> cloud 1, true -> cloud 2
  const int c = second ? r - C : r;
  const int nt = second ? M : N, no = second ? N : M;
  const float* __restrict__ xt = (second ? xyz2 : xyz1) + (size_t)b * nt * C;
  const float* __restrict__ xo = (second ? xyz1 : xyz2) + (size_t)b * no * C;
  const float* __restrict__ gt = (second ? gd2 : gd1) + (size_t)b * nt;
  const float* __restrict__ go = (second ? gd1 : gd2) + (size_t)b * no;
  const int* __restrict__ it = (second ? idx2 : idx1) + (size_t)b * nt;
  const int* __restrict__ io = (second ? idx1 : idx2) + (size_t)b * no;
  float* __restrict__ out = (second ? gx2 : gx1) + (size_t)b * nt * C;
  // own terms: +g*(x_T[k] - x_O[idx_T[k]])                   (ref nmdistance_cuda.cu:176-180)
  for (int k = threadIdx.x; k < nt; k += 1024) {
    const int j2 = it[k];
    float v = 0.0f;
    if (j2 >= 0) {
      const float g = gt[k] * 2;
      v = g * (xt[(size_t)k * C + c] - xo[(size_t)j2 * C + c]);
    }
    s_acc[k] = v;
  }
  __syncthreads();
  // scattered terms of the other direction: -g*(x_O[j] - x_T[idx_O[j]]) onto row idx_O[j]   (:181)
  for (int j = threadIdx.x; j < no; j += 1024) {
    const int k = io[j];
    if (k >= 0) {
      const float g = go[j] * 2;
      const float v = g * (xo[(size_t)j * C + c] - xt[(size_t)k * C + c]);
      atomicAdd(&s_acc[k], -v);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nt; k += 1024) out[(size_t)k * C + c] = s_acc[k];
}

// Backward with no floating-point atomics at all (C == 3).  ds_add_f32 runs at 0.36 lanes/clk/CU
// on gfx950 against 10 for ds_add_u32 (tools/lds_atomic_probe.hip), so the scattered terms are
// not added atomically but *listed*: a workgroup owns a slice [k0,k1) of one target cloud, counts
// with integer LDS atomics how many points of the other cloud chose each k as nearest neighbour,
// scans the counts, fills the lists (integer atomics again), and then one lane per k sums its own
// term and its list in registers and writes the 12-byte gradient row once.
constexpr int kCsrSlices = 4;  // workgroups per (batch, target cloud)
// ORDERED (deterministic mode, pp_nmdistance_backward_ordered_f32): every list is sorted by source index and
// the terms are added in the order of a sequential loop over the reference's two launches -- first launch
// (cloud 1 -> cloud 2): own terms into gradxyz1, scattered terms into gradxyz2 in ascending source order;
// second launch mirrored -- i.e. gradxyz1[j] = (0 + own) - s(i1) - s(i2) ... and gradxyz2[k] = ((0 - s(j1)) -
// s(j2) ...) + own.  No floating-point atomics anywhere, so the result is reproducible bit for bit, and it
// equals the CPU oracle's (the restatement under oracle/: oracle_chamfer_backward) bit for bit.
template <bool ORDERED>
__global__ __launch_bounds__(1024) void nmdist_bwd_csr_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, const float* __restrict__ gd1,
    const float* __restrict__ gd2, const int* __restrict__ idx1, const int* __restrict__ idx2,
    float* __restrict__ gx1, float* __restrict__ gx2, int N, int M, int slice_len) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_u[];  // start[slice_len+1] | cursor[slice_len] | list[no]
  __shared__ unsigned s_wave[16];
  unsigned* s_start = s_u;
  unsigned* s_cur = s_u + slice_len + 1;
  unsigned* s_list = s_cur + slice_len;
  const int per_b = 2 * kCsrSlices;
  const int b = blockIdx.x / per_b;
  const int r = blockIdx.x - b * per_b;
  const bool second = r >= kCsrSlices;  // target cloud: false -> cloud 1, true -> cloud 2
  const int slice = second ? r - kCsrSlices : r;
  const int nt = second ? M : N, no = second ? N : M;
  const int k0 = slice * slice_len;
  const int len = max(0, min(slice_len, nt - k0));
  if (len == 0) return;
  const float* __restrict__ xt = (second ? xyz2 : xyz1) + (size_t)b * nt * 3;
  const float* __restrict__ xo = (second ? xyz1 : xyz2) + (size_t)b * no * 3;
  const float* __restrict__ gt = (second ? gd2 : gd1) + (size_t)b * nt;
  const float* __restrict__ go = (second ? gd1 : gd2) + (size_t)b * no;
  const int* __restrict__ it = (second ? idx2 : idx1) + (size_t)b * nt;
  const int* __restrict__ io = (second ? idx1 : idx2) + (size_t)b * no;
  float* __restrict__ out = (second ? gx2 : gx1) + (size_t)b * nt * 3;
  const int t = threadIdx.x;
  for (int k = t; k <= len; k += 1024) s_start[k] = 0;
  // the other cloud's neighbour indices are needed twice (count, fill): keep up to KJ per thread in
  // registers (16384 points per pass), loads unconditional so that they are all in flight together
  constexpr int KJ = 16;
  const bool cached = no <= 1024 * KJ;
  int kj[KJ];
#pragma unroll
  for (int u = 0; u < KJ; ++u) {
    const int j = t + 1024 * u;
    kj[u] = (cached && j < no) ? io[j] - k0 : -1;
  }
  __syncthreads();
  if (cached) {
#pragma unroll
    for (int u = 0; u < KJ; ++u)
      if (kj[u] >= 0 && kj[u] < len) atomicAdd(&s_start[kj[u]], 1u);
  } else {
    for (int j = t; j < no; j += 1024) {
      const int k = io[j] - k0;
      if (k >= 0 && k < len) atomicAdd(&s_start[k], 1u);
    }
  }
  __syncthreads();
  // exclusive scan of s_start[0..len): thread t owns a contiguous run
  const int per = (len + 1023) / 1024;
  const int c0 = min(t * per, len), c1 = min(c0 + per, len);
  unsigned sum = 0;
  for (int c = c0; c < c1; ++c) sum += s_start[c];
  unsigned incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off);
    if ((t & 63) >= off) incl += o;
  }
  if ((t & 63) == 63) s_wave[t >> 6] = incl;
  __syncthreads();
  unsigned run = incl - sum;
  for (int w = 0; w < (t >> 6); ++w) run += s_wave[w];
  for (int c = c0; c < c1; ++c) {
    const unsigned v = s_start[c];
    s_start[c] = run;
    s_cur[c] = run;
    run += v;
  }
  if (t == 1023) s_start[len] = run;  // total (the last thread's run ends at the grand total)
  __syncthreads();
  if (cached) {
#pragma unroll
    for (int u = 0; u < KJ; ++u)
      if (kj[u] >= 0 && kj[u] < len) s_list[atomicAdd(&s_cur[kj[u]], 1u)] = (unsigned)(t + 1024 * u);
  } else {
    for (int j = t; j < no; j += 1024) {
      const int k = io[j] - k0;
      if (k >= 0 && k < len) s_list[atomicAdd(&s_cur[k], 1u)] = (unsigned)j;
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int kk = t; kk < len; kk += 1024) {
    const int k = k0 + kk;
    const float tx = xt[3 * (size_t)k], ty = xt[3 * (size_t)k + 1], tz = xt[3 * (size_t)k + 2];
    float ox_ = 0.0f, oy_ = 0.0f, oz_ = 0.0f;
    const int j2 = it[k];
    if (j2 >= 0) {  // own term: +g*(x_T[k] - x_O[idx_T[k]])          (ref nmdistance_cuda.cu:176-180)
      const float g = gt[k] * 2;
      ox_ = g * (tx - xo[3 * (size_t)j2]);
      oy_ = g * (ty - xo[3 * (size_t)j2 + 1]);
      oz_ = g * (tz - xo[3 * (size_t)j2 + 2]);
    }
    const unsigned e0 = s_start[kk], e1 = s_start[kk + 1];
    if (ORDERED) {  // ascending source index (the cursors filled the list in arrival order)
      unsigned* lst = s_list + e0;
      pp::lane_sort(
          e1 - e0, [&](unsigned i) { return lst[i]; },
          [&](unsigned i, unsigned j) {
            const unsigned v = lst[i];
            lst[i] = lst[j];
            lst[j] = v;
          });
    }
    // cloud 1's rows receive their own term first (first launch), cloud 2's last (second launch); a row
    // without an own term (labeled Chamfer, idx < 0) receives none at all, as in the reference (:175)
    const bool own_first = !ORDERED || !second;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    if (own_first && j2 >= 0) {
      ax += ox_; ay += oy_; az += oz_;
    }
    for (unsigned e = e0; e < e1; ++e) {  // scattered terms (:181)
      const unsigned j = s_list[e];
      const float g = go[j] * 2;
      ax += -(g * (xo[3 * (size_t)j] - tx));
      ay += -(g * (xo[3 * (size_t)j + 1] - ty));
      az += -(g * (xo[3 * (size_t)j + 2] - tz));
    }
    if (!own_first && j2 >= 0) {
      ax += ox_; ay += oy_; az += oz_;
    }
    out[3 * (size_t)k] = ax;
    out[3 * (size_t)k + 1] = ay;
    out[3 * (size_t)k + 2] = az;
  }
}

// Backward with DOUBLE accumulators in LDS (C == 3).  ds_add_f64 runs at 3.3 lanes/clk/CU on gfx950
// (ds_add_f32: 0.36, tools/lds_atomic_probe.hip), which makes the direct form -- seed the slice
// with the own-point terms, add the scattered terms atomically, round once at the end -- cheaper
// than listing them (the CSR kernel above needs a count pass, a scan and a fill pass before it can
// sum).  A workgroup owns a slice [k0,k1) of one target cloud: 3 doubles per point.  Every term is
// computed in fp32 exactly as the reference does; only the running sum is wider.
// VEC (clouds, gradients and indices 16-byte aligned, N, M and the slice length multiples of 4): a thread
// takes four CONSECUTIVE points and reads them with 16-byte loads -- 20 fully coalesced loads per thread
// in the scattered pass instead of 80 four-byte loads at a 12-byte lane stride, which is what bounded
// the kernel (the address path handles a wave's 4-byte loads lane group by lane group).
template <bool VEC>
__global__ __launch_bounds__(1024) void nmdist_bwd_lds64_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, const float* __restrict__ gd1,
    const float* __restrict__ gd2, const int* __restrict__ idx1, const int* __restrict__ idx2,
    float* __restrict__ gx1, float* __restrict__ gx2, int N, int M, int slice_len, int slices, int total,
    int per_xcd) {
  extern __shared__ __attribute__((aligned(16))) double s_acc64[];  // [slice_len][3] doubles, then [slice_len][3] floats
  // the workgroups of a batch element share an XCD -- the one whose L2 holds its clouds and the
  // indices the forward pass has just written (same batch -> XCD mapping as chamfer_grid.hip)
  const int V = pp::xcd_virtual_block(blockIdx.x, per_xcd);
  const int per_b = 2 * slices;
  if (V >= total) return;
  const int b = V / per_b;
  const int r = V - b * per_b;
  const bool second = r >= slices;  // target cloud: false -> cloud 1, true -> cloud 2
  const int slice = second ? r - slices : r;
  const int nt = second ? M : N, no = second ? N : M;
  const int k0 = slice * slice_len;
  const int len = max(0, min(slice_len, nt - k0));
  if (len == 0) return;
  const float* __restrict__ xt = (second ? xyz2 : xyz1) + (size_t)b * nt * 3;
  const float* __restrict__ xo = (second ? xyz1 : xyz2) + (size_t)b * no * 3;
  const float* __restrict__ gt = (second ? gd2 : gd1) + (size_t)b * nt;
  const float* __restrict__ go = (second ? gd1 : gd2) + (size_t)b * no;
  const int* __restrict__ it = (second ? idx2 : idx1) + (size_t)b * nt;
  const int* __restrict__ io = (second ? idx1 : idx2) + (size_t)b * no;
  float* __restrict__ out = (second ? gx2 : gx1) + (size_t)b * nt * 3;
  const int t = threadIdx.x;
  float* s_xt = reinterpret_cast<float*>(s_acc64 + 3 * (size_t)slice_len);  // the slice's coordinates [len][3]
  // Scattered pass, in batches of KJ elements per thread: everything a batch needs from memory (index,
  // gradient, coordinates of the other cloud's points) is loaded unconditionally up front (indices
  // clamped); the target coordinates it also needs come from the LDS copy the own-term pass leaves behind.
  // So a batch costs ONE memory round trip, and it is issued a batch ahead: the first before the own-term
  // pass starts, the next before the current one is added.  (Before: two dependent round trips per batch,
  // none overlapped: 17.8 -> 1x us at config 2.)
  constexpr int KJ = 8;
  struct Batch {
    int kj[KJ];
    float g[KJ], ox[KJ], oy[KJ], oz[KJ];
  };
  // batch starting at point jb of the other cloud: element u of thread t is point jb + 1024 u + t, or, in
  // the vector form, point jb + 4096 (u / 4) + 4 t + (u % 4)
  auto load_batch = [&](Batch& bt, int jb) {
    if constexpr (VEC) {
#pragma unroll
      for (int gq = 0; gq < KJ / 4; ++gq) {
        const int p0 = jb + 4096 * gq + 4 * t;
        const int pc = min(p0, no - 4);  // no is a multiple of 4 (>= 4): an aligned, in-range quad
        const pp::i4 k4 = *reinterpret_cast<const pp::i4*>(io + pc);
        const pp::f4 g4 = *reinterpret_cast<const pp::f4*>(go + pc);
        const pp::f4* __restrict__ src = reinterpret_cast<const pp::f4*>(xo + 3 * (size_t)pc);
        const pp::f4 a = src[0], b4 = src[1], c = src[2];
        const int kr[4] = {k4.x - k0, k4.y - k0, k4.z - k0, k4.w - k0};
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
        const float xs[4] = {a.x, a.w, b4.z, c.y}, ys[4] = {a.y, b4.x, b4.w, c.z}, zs[4] = {a.z, b4.y, c.x, c.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bt.kj[4 * gq + i] = (p0 < no && kr[i] >= 0 && kr[i] < len) ? kr[i] : -1;
          bt.g[4 * gq + i] = gg[i];
          bt.ox[4 * gq + i] = xs[i]; bt.oy[4 * gq + i] = ys[i]; bt.oz[4 * gq + i] = zs[i];
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < KJ; ++u) {
        const int j = jb + t + 1024 * u;
        const int jc = min(j, no - 1);
        const int kr = io[jc] - k0;
        bt.kj[u] = (j < no && kr >= 0 && kr < len) ? kr : -1;
        bt.g[u] = go[jc];
        bt.ox[u] = xo[3 * (size_t)jc]; bt.oy[u] = xo[3 * (size_t)jc + 1]; bt.oz[u] = xo[3 * (size_t)jc + 2];
      }
    }
  };
  auto add_batch = [&](const Batch& bt) {  // -g*(x_O[j] - x_T[idx_O[j]]) onto row idx_O[j]              (:181)
#pragma unroll
    for (int u = 0; u < KJ; ++u)
      if (bt.kj[u] >= 0) {
        const int kr = bt.kj[u];
        const float gg = bt.g[u] * 2;
        atomicAdd(&s_acc64[3 * kr], (double)(-(gg * (bt.ox[u] - s_xt[3 * kr]))));
        atomicAdd(&s_acc64[3 * kr + 1], (double)(-(gg * (bt.oy[u] - s_xt[3 * kr + 1]))));
        atomicAdd(&s_acc64[3 * kr + 2], (double)(-(gg * (bt.oz[u] - s_xt[3 * kr + 2]))));
      }
  };
  Batch ba, bb;
  PP_PHASE(0);
  load_batch(ba, 0);
  // own terms: +g*(x_T[k] - x_O[idx_T[k]])                              (ref nmdistance_cuda.cu:176-180)
  // (seeding the slice with plain stores and a barrier measured faster than zero-filling and adding
  // the own terms atomically in the same pass as the scattered ones)
  // The loads of several elements are issued before any is used (indices clamped, so no load is
  // conditional): the pass costs two memory round trips per batch instead of two per element.
  constexpr int KO = 4;
  for (int kb = 0; kb < len; kb += 1024 * KO) {
    int j2[KO], kkv[KO];
    float g[KO], tx[KO], ty[KO], tz[KO], ox[KO], oy[KO], oz[KO];
    if constexpr (VEC) {  // slice start and length are multiples of 4: quads never straddle the end
      const int p0 = kb + 4 * t;
      const int pc = k0 + min(p0, len - 4);
      const pp::i4 j4 = *reinterpret_cast<const pp::i4*>(it + pc);
      const pp::f4 g4 = *reinterpret_cast<const pp::f4*>(gt + pc);
      const pp::f4* __restrict__ src = reinterpret_cast<const pp::f4*>(xt + 3 * (size_t)pc);
      const pp::f4 a = src[0], b4 = src[1], c = src[2];
      j2[0] = j4.x; j2[1] = j4.y; j2[2] = j4.z; j2[3] = j4.w;
      g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
      tx[0] = a.x; ty[0] = a.y; tz[0] = a.z; tx[1] = a.w; ty[1] = b4.x; tz[1] = b4.y;
      tx[2] = b4.z; ty[2] = b4.w; tz[2] = c.x; tx[3] = c.y; ty[3] = c.z; tz[3] = c.w;
#pragma unroll
      for (int u = 0; u < KO; ++u) kkv[u] = p0 < len ? p0 + u : len;  // len: "nothing to store"
    } else {
#pragma unroll
      for (int u = 0; u < KO; ++u) {
        kkv[u] = min(kb + t + 1024 * u, len);
        const int k = k0 + min(kkv[u], len - 1);
        j2[u] = it[k];
        g[u] = gt[k];
        tx[u] = xt[3 * (size_t)k]; ty[u] = xt[3 * (size_t)k + 1]; tz[u] = xt[3 * (size_t)k + 2];
      }
    }
#pragma unroll
    for (int u = 0; u < KO; ++u) {
      const int jc = j2[u] >= 0 ? j2[u] : 0;
      // (one 12-byte load per neighbour instead of three scattered 4-byte ones: a third of the L1 requests)
      typedef float f3 __attribute__((ext_vector_type(3)));
      f3 v;
      __builtin_memcpy(&v, xo + 3 * (size_t)jc, sizeof(v));
      ox[u] = v.x; oy[u] = v.y; oz[u] = v.z;
    }
#pragma unroll
    for (int u = 0; u < KO; ++u) {
      const int kk = kkv[u];
      if (kk < len) {
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        if (j2[u] >= 0) {
          const float gg = g[u] * 2;
          ax = gg * (tx[u] - ox[u]);
          ay = gg * (ty[u] - oy[u]);
          az = gg * (tz[u] - oz[u]);
        }
        s_acc64[3 * kk] = (double)ax;
        s_acc64[3 * kk + 1] = (double)ay;
        s_acc64[3 * kk + 2] = (double)az;
        s_xt[3 * kk] = tx[u];
        s_xt[3 * kk + 1] = ty[u];
        s_xt[3 * kk + 2] = tz[u];
      }
    }
  }
  PP_PHASE(1);
  __syncthreads();
  PP_PHASE(2);
  for (int jb = 0; jb < no; jb += 2 * 1024 * KJ) {  // two batches per trip: the buffers swap roles without copies
    load_batch(bb, jb + 1024 * KJ);
    add_batch(ba);
    load_batch(ba, jb + 2 * 1024 * KJ);
    add_batch(bb);
  }
  PP_PHASE(3);
  __syncthreads();
  PP_PHASE(4);
  // (coalesced; non-temporal: nobody on this chip reads the gradients soon, and 12 MB of lines left dirty in the L2s are
  //  written back at the kernel's end, in front of the next launch -- PP_BWD_NT_STORES=0 at build time for comparison)
#ifndef PP_BWD_NT_STORES
#define PP_BWD_NT_STORES 1
#endif
  for (int e = t; e < 3 * len; e += 1024) {
    if (PP_BWD_NT_STORES)
      __builtin_nontemporal_store((float)s_acc64[e], &out[3 * (size_t)k0 + e]);
    else
      out[3 * (size_t)k0 + e] = (float)s_acc64[e];
  }
  PP_PHASE(5);
}

__global__ void fill_zero_kernel(float* __restrict__ a, int* __restrict__ b, long long n) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    a[t] = 0.0f;
    b[t] = 0;
  }
}

template <int Q, int G, bool PK = false, bool PF = false>
int launch_fwd_c3(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2,
                  int* idx2, int B, int N, int M, hipStream_t s) {
  constexpr int TQ = 64 * Q;
  const int tiles1 = (N + TQ - 1) / TQ, tiles2 = (M + TQ - 1) / TQ;
  const long long total = (long long)B * (tiles1 + tiles2);
  if (total > 0x7fffff00LL) return PP_EINVAL;
  const int per_xcd = (int)((total + 7) / 8);
  nmdist_fwd_c3_kernel<Q, G, PK, PF><<<dim3(per_xcd * 8), dim3(kBlock), 0, s>>>(
      xyz1, xyz2, dist1, idx1, dist2, idx2, N, M, tiles1, tiles2, (int)total, per_xcd);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

// Other point dimensions, tiled like the C == 3 kernel (the reference's kernel is generic in c,
// nmdistance_cuda.cu:31-35): a lane keeps Q queries of CT coordinates in registers, the four waves of a
// workgroup scan the four quarters of the reference cloud (reference point wave-uniform: scalar loads) in
// groups of G, tracking only the running minimum (v_min3: one op per two pairs) and the first group that
// lowered it; the index is recovered afterwards by re-evaluating that group with the same instruction
// sequence.  VALU per pair: CT sub + CT fma + 1/2 min3 + bookkeeping / (Q G), against CT sub + CT fma + cmp +
// 2 cndmask for the one-lane-per-query kernel above.  PAD (CT = 16 serves 9 <= C <= 16): coordinates beyond
// C are zeros on both sides and fma(0, 0, d) == d leaves every bit alone.
template <int CT, bool PAD, int Q, int G>
__global__ __launch_bounds__(kBlock) void nmdist_fwd_tiled_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, float* __restrict__ dist1,
    int* __restrict__ idx1, float* __restrict__ dist2, int* __restrict__ idx2, int N, int M, int C, int tiles1,
    int tiles2, int total, int per_xcd) {
  static_assert(G % 2 == 0, "groups are consumed two reference points per v_min3");
  constexpr int TQ = 64 * Q;
  __shared__ float s_best[kWavesPerBlock][TQ];
  __shared__ int s_idx[kWavesPerBlock][TQ];
  const int V = pp::xcd_virtual_block(blockIdx.x, per_xcd);
  if (V >= total) return;
  const int c = PAD ? C : CT;
  const int per_b = tiles1 + tiles2;
  const int b = V / per_b;
  const int r = V - b * per_b;
  const bool second = r >= tiles1;
  const int tile = second ? r - tiles1 : r;
  const int nq = second ? M : N, nr = second ? N : M;
  const float* __restrict__ qry = (second ? xyz2 : xyz1) + (size_t)b * nq * c;
  const float* __restrict__ ref = (second ? xyz1 : xyz2) + (size_t)b * nr * c;
  const int wave = pp::wave_id_uniform();
  const int lane = threadIdx.x & 63;
  float q[Q][CT], best[Q];
  int gid[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    int j = tile * TQ + i * 64 + lane;
    j = j < nq ? j : nq - 1;  // clamp: out-of-range lanes compute a valid query, never stored
#pragma unroll
    for (int e = 0; e < CT; ++e) q[i][e] = (!PAD || e < c) ? qry[(size_t)j * c + e] : 0.0f;
    best[i] = __builtin_inff();
    gid[i] = -1;
  }
  auto dist_to = [&](const float* __restrict__ rp, const float (&qq)[CT]) {  // rp wave-uniform: scalar loads
    float d = 0.0f;
#pragma unroll
    for (int e = 0; e < CT; ++e) {
      const float t = ((!PAD || e < c) ? rp[e] : 0.0f) - qq[e];
      d = __builtin_fmaf(t, t, d);
    }
    return d;
  };
  const int ngroups = nr / G;
  const int g0 = (int)(((long long)ngroups * wave) / kWavesPerBlock);
  const int g1 = (int)(((long long)ngroups * (wave + 1)) / kWavesPerBlock);
  for (int g = g0; g < g1; ++g) {
    const float* __restrict__ rp = ref + (size_t)g * G * c;
    float nb[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) nb[i] = best[i];
#pragma unroll
    for (int p = 0; p < G; p += 2) {
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const float da = dist_to(rp + (size_t)p * c, q[i]);
        const float db = dist_to(rp + (size_t)(p + 1) * c, q[i]);
        nb[i] = pp::min3(da, db, nb[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      gid[i] = nb[i] < best[i] ? g : gid[i];  // first group that attains the running minimum
      best[i] = nb[i];
    }
  }
  int bidx[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {  // recover the index inside the recorded group (descending: the lowest index is kept)
    int k = 0;
    if (gid[i] >= 0) {
      const int k0 = gid[i] * G;
#pragma unroll
      for (int p = G - 1; p >= 0; --p) {
        const float* rp = ref + (size_t)(k0 + p) * c;  // per-lane (gid differs between lanes): vector loads
        float d = 0.0f;
#pragma unroll
        for (int e = 0; e < CT; ++e) {
          const float t = ((!PAD || e < c) ? rp[e] : 0.0f) - q[i][e];
          d = __builtin_fmaf(t, t, d);
        }
        k = d == best[i] ? k0 + p : k;
      }
    }
    bidx[i] = k;
  }
  if (wave == kWavesPerBlock - 1) {  // tail (nr % G points, highest indices): exact compare/select
    for (int k = ngroups * G; k < nr; ++k) {
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const float d = dist_to(ref + (size_t)k * c, q[i]);
        const bool lt = d < best[i];
        best[i] = lt ? d : best[i];
        bidx[i] = lt ? k : bidx[i];
      }
    }
  }
  // merge the four quarters (ascending index ranges => strict < keeps the lowest index)
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    s_best[wave][i * 64 + lane] = best[i];
    s_idx[wave][i * 64 + lane] = bidx[i];
  }
  __syncthreads();
  float* __restrict__ od = (second ? dist2 : dist1) + (size_t)b * nq;
  int* __restrict__ oi = (second ? idx2 : idx1) + (size_t)b * nq;
  for (int e = threadIdx.x; e < TQ; e += kBlock) {
    float bb = s_best[0][e];
    int bi = s_idx[0][e];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) {
      const float cc = s_best[w][e];
      const int ci = s_idx[w][e];
      const bool lt = cc < bb;
      bb = lt ? cc : bb;
      bi = lt ? ci : bi;
    }
    const int j = tile * TQ + e;
    if (j < nq) {
      od[j] = bb;
      oi[j] = bi;
    }
  }
}

template <int CT, bool PAD, int Q, int G>
int launch_fwd_tiled(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2, int* idx2,
                     int B, int N, int M, int C, hipStream_t s) {
  constexpr int TQ = 64 * Q;
  const int tiles1 = (N + TQ - 1) / TQ, tiles2 = (M + TQ - 1) / TQ;
  const long long total = (long long)B * (tiles1 + tiles2);
  if (total > 0x7fffff00LL) return PP_EINVAL;
  const int per_xcd = (int)((total + 7) / 8);
  nmdist_fwd_tiled_kernel<CT, PAD, Q, G><<<dim3(per_xcd * 8), dim3(kBlock), 0, s>>>(
      xyz1, xyz2, dist1, idx1, dist2, idx2, N, M, C, tiles1, tiles2, (int)total, per_xcd);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

template <int CT, bool PAD = false>
int launch_fwd_generic(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2,
                       int* idx2, int B, int N, int M, int C, hipStream_t s) {
  const int tiles1 = (N + 255) / 256, tiles2 = (M + 255) / 256;
  const long long total = (long long)B * (tiles1 + tiles2);
  if (total > 0x7fffff00LL) return PP_EINVAL;
  nmdist_fwd_generic_kernel<CT, PAD><<<dim3((unsigned)total), dim3(256), 0, s>>>(
      xyz1, xyz2, dist1, idx1, dist2, idx2, N, M, C, tiles1, tiles2);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

int zero_outputs(float* dist1, int* idx1, float* dist2, int* idx2, int B, int N, int M,
                 hipStream_t s) {
  const long long n1 = (long long)B * N, n2 = (long long)B * M;
  if (n1 > 0) {
    fill_zero_kernel<<<dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s>>>(dist1, idx1, n1);
    PP_RETURN_IF_LAUNCH_FAILED();
  }
  if (n2 > 0) {
    fill_zero_kernel<<<dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s>>>(dist2, idx2, n2);
    PP_RETURN_IF_LAUNCH_FAILED();
  }
  return PP_OK;
}

}  // namespace

// Tuning override for benchmarking variants in one process (bench.py --variant); 0 = automatic.
static pp::Knob g_fwd_variant;
extern "C" void pp_debug_set_nmdistance_variant(int v) { g_fwd_variant.set(v); }

namespace pp {
// (chamfer_grid.hip) the every-pair kernel over the directions marked in `routed` (device memory, one word every
// `stride` words per (batch element, direction)); the other directions' workgroups leave at once
int nmdist_forward_routed(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2, int* idx2, int B,
                          int N, int M, const unsigned* routed, int stride, hipStream_t s) {
  constexpr int Q = 4, G = 16, TQ = 64 * Q;
  const int tiles1 = (N + TQ - 1) / TQ, tiles2 = (M + TQ - 1) / TQ;
  const long long total = (long long)B * (tiles1 + tiles2);
  if (total > 0x7fffff00LL) return PP_EINVAL;
  const int per_xcd = (int)((total + 7) / 8);
  nmdist_fwd_c3_routed_kernel<Q, G, true><<<dim3(per_xcd * 8), dim3(kBlock), 0, s>>>(
      xyz1, xyz2, dist1, idx1, dist2, idx2, N, M, tiles1, tiles2, (int)total, per_xcd, routed, stride);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
}  // namespace pp

extern "C" int pp_nmdistance_forward_f32(const float* xyz1, const float* xyz2, float* dist1,
                                         int* idx1, float* dist2, int* idx2, int B, int N, int M,
                                         int C, void* stream) {
  if (B < 0 || N < 0 || M < 0 || C < 1) return PP_EINVAL;
  if (B == 0 || (N == 0 && M == 0)) return PP_OK;
  if ((N > 0 && (!dist1 || !idx1)) || (M > 0 && (!dist2 || !idx2))) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0 || M == 0) return zero_outputs(dist1, idx1, dist2, idx2, B, N, M, s);
  if (!xyz1 || !xyz2) return PP_EINVAL;
  if (C == 3) {
    // queries per lane: enough workgroups to give every CU (256) several, else a smaller tile
    const long long q = (long long)B * ((long long)N + M);
    int variant = g_fwd_variant;
    if (variant == 0) variant = q >= 4LL * 256 * 1024 ? 1416 : (q >= 2LL * 256 * 512 ? 1002 : 1);
    switch (variant) {
      case 4: return launch_fwd_c3<4, 8>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 2: return launch_fwd_c3<2, 8>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 1: return launch_fwd_c3<1, 8>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 8: return launch_fwd_c3<8, 8>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 416: return launch_fwd_c3<4, 16>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 216: return launch_fwd_c3<2, 16>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 44: return launch_fwd_c3<4, 4>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 1002: return launch_fwd_c3<2, 8, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 1004: return launch_fwd_c3<4, 8, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 1008: return launch_fwd_c3<8, 8, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 1416: return launch_fwd_c3<4, 16, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 1816: return launch_fwd_c3<8, 16, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 2004: return launch_fwd_c3<4, 8, true, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 2008: return launch_fwd_c3<8, 8, true, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 2002: return launch_fwd_c3<2, 8, true, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 2416: return launch_fwd_c3<4, 16, true, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      case 3004: return launch_fwd_c3<4, 8, false, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, s);
      default: return PP_EINVAL;
    }
  }
  // the tiled kernel wherever a cloud gives its four waves a quarter each worth scanning (g_fwd_variant == 9:
  // the one-lane-per-query kernels, for tests)
  if (g_fwd_variant != 9 && N >= 256 && M >= 256) {
    switch (C) {
      case 1: return launch_fwd_tiled<1, false, 4, 8>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
      case 2: return launch_fwd_tiled<2, false, 4, 8>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
      case 4: return launch_fwd_tiled<4, false, 2, 4>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
      case 5: return launch_fwd_tiled<5, false, 2, 4>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
      case 6: return launch_fwd_tiled<6, false, 2, 4>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
      case 7: return launch_fwd_tiled<7, false, 2, 4>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
      case 8: return launch_fwd_tiled<8, false, 2, 4>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
#define PP_TILED_C(CC) \
      case CC: return launch_fwd_tiled<CC, false, 2, 4>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
      // (one instantiation per dimension: the zero-padded CT = 16 form spends its time on the run-time
      //  `e < C` selects -- 0.19 of the VALU roof at C = 9 against 0.77 for the exact instantiation)
      PP_TILED_C(9) PP_TILED_C(10) PP_TILED_C(11) PP_TILED_C(12) PP_TILED_C(13) PP_TILED_C(14) PP_TILED_C(15) PP_TILED_C(16)
#undef PP_TILED_C
      default: break;
    }
  }
  switch (C) {
    case 1: return launch_fwd_generic<1>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
    case 2: return launch_fwd_generic<2>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
    case 4: return launch_fwd_generic<4>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
    case 5: return launch_fwd_generic<5>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
    case 6: return launch_fwd_generic<6>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
    case 7: return launch_fwd_generic<7>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
    case 8: return launch_fwd_generic<8>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
    default:
      if (C <= 16) return launch_fwd_generic<16, true>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
      return launch_fwd_generic<0>(xyz1, xyz2, dist1, idx1, dist2, idx2, B, N, M, C, s);
  }
}

// 0 = automatic; 1 = force the one-lane-per-query kernel (tests and tuning)
static pp::Knob g_labeled_variant;
extern "C" void pp_debug_set_labeled_variant(int v) { g_labeled_variant.set(v); }

extern "C" int pp_labeled_nmdistance_forward_f32(const float* xyz1, const float* xyz2,
                                                 const float* label1, const float* label2,
                                                 float* dist1, int* idx1, float* dist2, int* idx2,
                                                 int B, int N, int M, int C, void* stream) {
  if (B < 0 || N < 0 || M < 0 || C < 1) return PP_EINVAL;
  if (B == 0 || (N == 0 && M == 0)) return PP_OK;
  if ((N > 0 && (!dist1 || !idx1)) || (M > 0 && (!dist2 || !idx2))) return PP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // M == 0: the reference's kernel leaves dist 0 / idx 0 from the wrapper's zeros, then its
  // post-pass (:110-113) sees idx 0 (not < 0) -- so zeros, like the unlabeled case.
  if (N == 0 || M == 0) return zero_outputs(dist1, idx1, dist2, idx2, B, N, M, s);
  if (!xyz1 || !xyz2 || !label1 || !label2) return PP_EINVAL;
  if (C == 3 && g_labeled_variant != 1) {  // the tiled scan with the label filter
    constexpr int Q = 4, G = 8, TQ = 64 * Q;
    const int t1 = (N + TQ - 1) / TQ, t2 = (M + TQ - 1) / TQ;
    const long long tot = (long long)B * (t1 + t2);
    if (tot > 0x7fffff00LL) return PP_EINVAL;
    const int per_xcd = (int)((tot + 7) / 8);
    nmdist_fwd_c3_kernel<Q, G, true, false, true><<<dim3(per_xcd * 8), dim3(kBlock), 0, s>>>(
        xyz1, xyz2, dist1, idx1, dist2, idx2, N, M, t1, t2, (int)tot, per_xcd, label1, label2);
    PP_RETURN_IF_LAUNCH_FAILED();
    return PP_OK;
  }
  const int tiles1 = (N + 255) / 256, tiles2 = (M + 255) / 256;
  const long long total = (long long)B * (tiles1 + tiles2);
  if (total > 0x7fffff00LL) return PP_EINVAL;
  labeled_nmdist_fwd_kernel<<<dim3((unsigned)total), dim3(256), 0, s>>>(
      xyz1, xyz2, label1, label2, dist1, idx1, dist2, idx2, N, M, C, tiles1, tiles2);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

// 0 = automatic (double LDS accumulators for C == 3); 1 = force the global-atomic form; 2 = force
// the fp32 LDS-column form; 3 = the CSR form; 4 = same as 0   (tests and tuning)
static pp::Knob g_bwd_variant;
extern "C" void pp_debug_set_nmdistance_backward_variant(int v) { g_bwd_variant.set(v); }

extern "C" int pp_nmdistance_backward_f32(const float* xyz1, const float* xyz2,
                                          const float* graddist1, const float* graddist2,
                                          const int* idx1, const int* idx2, float* gradxyz1,
                                          float* gradxyz2, int B, int N, int M, int C,
                                          void* stream) {
  if (B < 0 || N < 0 || M < 0 || C < 1) return PP_EINVAL;
  const long long t1 = (long long)B * N, t2 = (long long)B * M;
  if (t1 + t2 == 0) return PP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0 || M == 0) {
    // no pairs: gradients are zero (the reference zeroes and its loops do not run)
    float* g = N == 0 ? gradxyz2 : gradxyz1;
    const long long n = (N == 0 ? t2 : t1) * C;
    if (!g) return PP_EINVAL;
    hipError_t e = hipMemsetAsync(g, 0, (size_t)n * sizeof(float), s);
    return (int)e;
  }
  if (!xyz1 || !xyz2 || !graddist1 || !graddist2 || !idx1 || !idx2 || !gradxyz1 || !gradxyz2)
    return PP_EINVAL;
  // double LDS accumulators: C == 3; slices of at most 4096 points (96 KiB + 48 KiB of coordinates), at least 4 per cloud
  if ((g_bwd_variant == 0 || g_bwd_variant == 4) && C == 3 && N + M >= 4096) {
    const int big = N > M ? N : M;
    int slices = (big + 4095) / 4096;
    if (slices < 4) slices = 4;  // 4 measured best at config 2 (8: 38 us, 16: 40 us)
    const int slice_len = (big + slices - 1) / slices;
    if ((long long)B * 2 * slices <= 0x7fffffffLL) {
      static pp::DeviceFlags lds_ok;
      static pp::DeviceFlags lds_ok_vec;
      const uintptr_t align = reinterpret_cast<uintptr_t>(xyz1) | reinterpret_cast<uintptr_t>(xyz2) |
                              reinterpret_cast<uintptr_t>(graddist1) | reinterpret_cast<uintptr_t>(graddist2) |
                              reinterpret_cast<uintptr_t>(idx1) | reinterpret_cast<uintptr_t>(idx2);
      const bool vec = (align & 15) == 0 && N % 4 == 0 && M % 4 == 0 && slice_len % 4 == 0 && N >= 4 && M >= 4;
      const hipError_t e = vec ? pp::allow_big_lds(nmdist_bwd_lds64_kernel<true>, 152 * 1024, lds_ok_vec)
                               : pp::allow_big_lds(nmdist_bwd_lds64_kernel<false>, 152 * 1024, lds_ok);
      if (e != hipSuccess) return (int)e;
      const int total = B * 2 * slices, per_xcd = (total + 7) / 8;
      (vec ? nmdist_bwd_lds64_kernel<true> : nmdist_bwd_lds64_kernel<false>)<<<dim3((unsigned)(per_xcd * 8)), dim3(1024), (size_t)slice_len * 36, s>>>(
          xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, slice_len, slices, total, per_xcd);
      PP_RETURN_IF_LAUNCH_FAILED();
      return PP_OK;
    }
  }
  // CSR form (no fp atomics): C == 3, slice bookkeeping + the other cloud's list fit the LDS
  if (g_bwd_variant != 1 && g_bwd_variant != 2 && C == 3 && N + M >= 4096) {
    const int big = N > M ? N : M;
    const int slice_len = (big + kCsrSlices - 1) / kCsrSlices;
    const size_t lds = ((size_t)2 * slice_len + 1 + big) * sizeof(unsigned);
    if (lds <= 150 * 1024 && (long long)B * 2 * kCsrSlices <= 0x7fffffffLL) {
      static pp::DeviceFlags lds_ok;
      // (the kernel also has 64 bytes of static LDS: the dynamic limit must leave room for them)
      const hipError_t e = pp::allow_big_lds(nmdist_bwd_csr_kernel<false>, 152 * 1024, lds_ok);
      if (e != hipSuccess) return (int)e;
      nmdist_bwd_csr_kernel<false><<<dim3((unsigned)(B * 2 * kCsrSlices)), dim3(1024), lds, s>>>(
          xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, slice_len);
      PP_RETURN_IF_LAUNCH_FAILED();
      return PP_OK;
    }
  }
  // LDS-column form: the larger cloud's column must fit the LDS; enough (b, cloud, c) columns to
  // occupy the chip, and clouds large enough that scattered global atomics would hurt
  const size_t col_bytes = (size_t)(N > M ? N : M) * sizeof(float);
  if (g_bwd_variant != 1 && col_bytes <= 160 * 1024 && (long long)B * 2 * C <= 0x7fffffffLL &&
      (g_bwd_variant == 2 || ((long long)B * 2 * C >= 64 && N + M >= 4096))) {
    static pp::DeviceFlags lds_ok;
    const hipError_t e = pp::allow_big_lds(nmdist_bwd_lds_kernel, 160 * 1024, lds_ok);
    if (e != hipSuccess) return (int)e;
    nmdist_bwd_lds_kernel<<<dim3((unsigned)(B * 2 * C)), dim3(1024), col_bytes, s>>>(
        xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, C);
    PP_RETURN_IF_LAUNCH_FAILED();
    return PP_OK;
  }
  const long long blocks = (t1 + t2 + 255) / 256;
  if (blocks > 0x7fffff00LL) return PP_EINVAL;
  nmdist_bwd_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
      xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, C, t1, t2);
  PP_RETURN_IF_LAUNCH_FAILED();
  nmdist_bwd_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
      xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, C, t1, t2);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

// Deterministic mode: the CSR form with ordered lists (see nmdist_bwd_csr_kernel<true>).  C == 3 and clouds whose
// bookkeeping fits the LDS (max(N, M) up to ~19000 points); PP_ENOTSUP otherwise -- there is no deterministic
// substitute for other shapes.
extern "C" int pp_nmdistance_backward_ordered_f32(const float* xyz1, const float* xyz2, const float* graddist1,
                                                  const float* graddist2, const int* idx1, const int* idx2,
                                                  float* gradxyz1, float* gradxyz2, int B, int N, int M, int C,
                                                  void* stream) {
  if (B < 0 || N < 0 || M < 0 || C < 1) return PP_EINVAL;
  if ((long long)B * ((long long)N + M) == 0) return PP_OK;
  if (N == 0 || M == 0)  // no pairs: zeros, as in the unordered entry point
    return pp_nmdistance_backward_f32(xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, B, N, M, C, stream);
  if (!xyz1 || !xyz2 || !graddist1 || !graddist2 || !idx1 || !idx2 || !gradxyz1 || !gradxyz2) return PP_EINVAL;
  if (C != 3) return PP_ENOTSUP;
  const int big = N > M ? N : M;
  const int slice_len = (big + kCsrSlices - 1) / kCsrSlices;
  const size_t lds = ((size_t)2 * slice_len + 1 + big) * sizeof(unsigned);
  if (lds > 150 * 1024 || (long long)B * 2 * kCsrSlices > 0x7fffffffLL) return PP_ENOTSUP;
  static pp::DeviceFlags lds_ok;
  const hipError_t e = pp::allow_big_lds(nmdist_bwd_csr_kernel<true>, 152 * 1024, lds_ok);
  if (e != hipSuccess) return (int)e;
  nmdist_bwd_csr_kernel<true><<<dim3((unsigned)(B * 2 * kCsrSlices)), dim3(1024), lds, (hipStream_t)stream>>>(
      xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, N, M, slice_len);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
