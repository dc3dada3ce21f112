/*
 * pp_oracle.c -- CPU restatement of the pytorch_points `_ext` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pytorch_points_amd/ may import, link
 * or execute this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and there only as the checker / baseline.
 *
 * PARITY UNPINNED: the reference (yifita/pytorch_points) ships no tests, golden
 * vectors or fixtures for this path, has no CPU implementation of it, and its
 * extension cannot be built in this image (needs nvcc + the CUDA runtime, the
 * removed THC headers and cuSOLVER).  This file is therefore a line-by-line
 * restatement of the reference kernels' semantics, cross-checked against an
 * independent fp64 brute force (oracle/bruteforce.py), not against outputs of
 * the reference itself.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/pytorch_points/_ext/).
 *
 * Canonical fp32 arithmetic (compiled with -ffp-contract=off; every fused
 * operation is spelled fmaf):
 *   distc  (Chamfer, nmdistance_cuda.cu:31-35)  d = 0; d = fmaf(t_c, t_c, d) for c = 0..C-1,
 *                                               t_c = ref_c - query_c          (d += tmp*tmp contracted)
 *   dist3  (sampling_cuda.cu:202,364; interpolate_gpu.cu:36)
 *                                               fmaf(dz,dz, fmaf(dx,dx, dy*dy)) (a*a + b*b + c*c contracted
 *                                               the way LLVM/NVPTX contracts a left-associated sum)
 *   interp (interpolate_gpu.cu:96)              fmaf(w2,p2, fmaf(w0,p0, w1*p1))
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PP_CHUNK 512 /* nmdistance_cuda.cu:5  const int BATCH = 512 */

/* ---------------------------------------------------------------------------------------------
 * cuda_utils.h:11-16  opt_n_threads: 2^floor(log2(work)) clamped to [1, 512], computed in double
 * exactly as the reference does on the host.
 * ------------------------------------------------------------------------------------------- */
int oracle_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

/* cuda_utils.h:18-26 opt_block_config */
void oracle_opt_block_config(int x, int y, int* bx, int* by) {
  const int xt = oracle_opt_n_threads(x);
  int yt = oracle_opt_n_threads(y);
  if (yt > 512 / xt) yt = 512 / xt;
  if (yt < 1) yt = 1;
  *bx = xt;
  *by = yt;
}

static inline float distc(const float* ref, const float* q, int c) {
  float d = 0.0f;
  for (int k = 0; k < c; ++k) {
    const float t = ref[k] - q[k]; /* nmdistance_cuda.cu:33  buf - xyz */
    d = fmaf(t, t, d);             /* :34  d += tmp*tmp */
  }
  return d;
}

static inline float dist3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
}

/* ---------------------------------------------------------------------------------------------
 * K1  NmDistanceKernel, nmdistance_cuda.cu:7-49, one launch (one direction).
 * Literal restatement: reference set walked in 512-chunks (:20), per chunk the best is taken
 * unconditionally at k==0 and then on strict d<best (:36), chunks merged on strict
 * result>best (:41).  The CUDA thread/block decomposition only partitions the queries, so a
 * sequential loop over queries is equivalent.
 * ------------------------------------------------------------------------------------------- */
static void nmdistance_structural_1dir(int b, int n, int c, const float* xyz, int m,
                                       const float* xyz2, float* result, int* result_i) {
  for (int i = 0; i < b; ++i) {
    for (int k2 = 0; k2 < m; k2 += PP_CHUNK) {
      const int end_k = (m < k2 + PP_CHUNK ? m : k2 + PP_CHUNK) - k2;
      const float* buf = xyz2 + ((size_t)i * m + k2) * c;
      for (int j = 0; j < n; ++j) {
        const float* q = xyz + ((size_t)i * n + j) * c;
        int best_i = 0;
        float best = 0;
        for (int k = 0; k < end_k; ++k) {
          const float d = distc(buf + (size_t)k * c, q, c);
          if (k == 0 || d < best) {
            best = d;
            best_i = k + k2;
          }
        }
        if (k2 == 0 || result[(size_t)i * n + j] > best) {
          result[(size_t)i * n + j] = best;
          result_i[(size_t)i * n + j] = best_i;
        }
      }
    }
  }
}

/* chamfer_cuda_forward, nmdistance_cuda.cu:118-140: two launches with the roles swapped. */
void oracle_chamfer_forward_structural(const float* xyz1, const float* xyz2, float* dist1,
                                       int* idx1, float* dist2, int* idx2, int b, int n, int m,
                                       int c) {
  nmdistance_structural_1dir(b, n, c, xyz1, m, xyz2, dist1, idx1);
  nmdistance_structural_1dir(b, m, c, xyz2, n, xyz1, dist2, idx2);
}

/* ---------------------------------------------------------------------------------------------
 * Same semantics as above for NaN-free input (minimum distance, lowest index on exact ties;
 * Appendix A.1 of SURVEY.md), arranged for the CPU: queries on the SIMD axis, reference point
 * broadcast, OpenMP over (batch, query tile).  Used for large sizes and as the timed CPU
 * baseline.  tests/test_oracle.py checks it against the structural form.
 * ------------------------------------------------------------------------------------------- */
#define QT 64
static void nmdistance_fast_1dir(int b, int n, int c, const float* xyz, int m, const float* xyz2,
                                 float* result, int* result_i) {
  if (m <= 0 || n <= 0) return; /* reference loops do not run; outputs untouched */
  const int tiles = (n + QT - 1) / QT;
  if (c == 3) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < b; ++i) {
      for (int t = 0; t < tiles; ++t) {
        float qx[QT], qy[QT], qz[QT], best[QT];
        int bi[QT];
        const int j0 = t * QT;
        for (int u = 0; u < QT; ++u) {
          const int j = (j0 + u < n) ? j0 + u : n - 1;
          const float* q = xyz + ((size_t)i * n + j) * 3;
          qx[u] = q[0];
          qy[u] = q[1];
          qz[u] = q[2];
        }
        const float* r = xyz2 + (size_t)i * m * 3;
        {
          const float rx = r[0], ry = r[1], rz = r[2];
          for (int u = 0; u < QT; ++u) {
            const float t0 = rx - qx[u], t1 = ry - qy[u], t2 = rz - qz[u];
            best[u] = fmaf(t2, t2, fmaf(t1, t1, t0 * t0));
            bi[u] = 0;
          }
        }
        for (int k = 1; k < m; ++k) {
          const float rx = r[3 * (size_t)k], ry = r[3 * (size_t)k + 1], rz = r[3 * (size_t)k + 2];
#pragma omp simd
          for (int u = 0; u < QT; ++u) {
            const float t0 = rx - qx[u], t1 = ry - qy[u], t2 = rz - qz[u];
            const float d = fmaf(t2, t2, fmaf(t1, t1, t0 * t0));
            const int lt = d < best[u];
            best[u] = lt ? d : best[u];
            bi[u] = lt ? k : bi[u];
          }
        }
        for (int u = 0; u < QT && j0 + u < n; ++u) {
          result[(size_t)i * n + j0 + u] = best[u];
          result_i[(size_t)i * n + j0 + u] = bi[u];
        }
      }
    }
  } else {
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < b; ++i) {
      for (int j = 0; j < n; ++j) {
        const float* q = xyz + ((size_t)i * n + j) * c;
        const float* r = xyz2 + (size_t)i * m * c;
        float best = distc(r, q, c);
        int bi = 0;
        for (int k = 1; k < m; ++k) {
          const float d = distc(r + (size_t)k * c, q, c);
          if (d < best) {
            best = d;
            bi = k;
          }
        }
        result[(size_t)i * n + j] = best;
        result_i[(size_t)i * n + j] = bi;
      }
    }
  }
}

void oracle_chamfer_forward(const float* xyz1, const float* xyz2, float* dist1, int* idx1,
                            float* dist2, int* idx2, int b, int n, int m, int c) {
  nmdistance_fast_1dir(b, n, c, xyz1, m, xyz2, dist1, idx1);
  nmdistance_fast_1dir(b, m, c, xyz2, n, xyz1, dist2, idx2);
}

/* ---------------------------------------------------------------------------------------------
 * K2  LabeledNmDistanceKernel, nmdistance_cuda.cu:55-115 (one direction), literal.
 * Labels are compared as floats (:89); per chunk best=1e10, best_i=-1 (:82-84); inside the
 * label branch `k==0 || d<best` (:95); chunk merge `k2==0 || result>best` (:101); finally
 * queries with a negative index get distance 0 (:110-113).
 * ------------------------------------------------------------------------------------------- */
static void labeled_nmdistance_1dir(int b, int n, int c, const float* xyz, const float* label,
                                    int m, const float* xyz2, const float* label2, float* result,
                                    int* result_i) {
  for (int i = 0; i < b; ++i) {
    for (int k2 = 0; k2 < m; k2 += PP_CHUNK) {
      const int end_k = (m < k2 + PP_CHUNK ? m : k2 + PP_CHUNK) - k2;
      const float* buf = xyz2 + ((size_t)i * m + k2) * c;
      const float* lbuf = label2 + (size_t)i * m + k2;
      for (int j = 0; j < n; ++j) {
        const float* q = xyz + ((size_t)i * n + j) * c;
        const float l1 = label[(size_t)i * n + j];
        int best_i = -1;
        float best = 1e10f;
        for (int k = 0; k < end_k; ++k) {
          if (l1 == lbuf[k]) {
            const float d = distc(buf + (size_t)k * c, q, c);
            if (k == 0 || d < best) {
              best = d;
              best_i = k + k2;
            }
          }
        }
        if (k2 == 0 || result[(size_t)i * n + j] > best) {
          result[(size_t)i * n + j] = best;
          result_i[(size_t)i * n + j] = best_i;
        }
      }
    }
    for (int j = 0; j < n; ++j)
      if (result_i[(size_t)i * n + j] < 0) result[(size_t)i * n + j] = 0;
  }
}

/* labeled_chamfer_cuda_forward, nmdistance_cuda.cu:142-166 */
void oracle_labeled_chamfer_forward(const float* xyz1, const float* xyz2, const float* label1,
                                    const float* label2, float* dist1, int* idx1, float* dist2,
                                    int* idx2, int b, int n, int m, int c) {
  labeled_nmdistance_1dir(b, n, c, xyz1, label1, m, xyz2, label2, dist1, idx1);
  labeled_nmdistance_1dir(b, m, c, xyz2, label2, n, xyz1, label1, dist2, idx2);
}

/* ---------------------------------------------------------------------------------------------
 * K3  NmDistanceGradKernel + chamfer_cuda_backward, nmdistance_cuda.cu:168-185,195-221.
 * g = grad_dist*2 (:176), xyz_g = g*(x1 - x2) (:179) -- two roundings, in that order; +xyz_g to
 * the query row, -xyz_g to the matched row (:180-181); negative indices skipped (:175); outputs
 * zeroed first (:204-205).  The reference accumulates with fp32 atomics in arbitrary order; this
 * restatement adds in (launch, batch, query) order, so compare with a tolerance, not bitwise.
 * ------------------------------------------------------------------------------------------- */
static void nmdistance_grad_1dir(int b, int n, int c, const float* xyz1, int m, const float* xyz2,
                                 const float* grad_dist1, const int* idx1, float* grad_xyz1,
                                 float* grad_xyz2) {
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      const int j2 = idx1[(size_t)i * n + j];
      if (j2 < 0) continue;
      const float g = grad_dist1[(size_t)i * n + j] * 2;
      for (int k = 0; k < c; ++k) {
        const float xyz_g =
            g * (xyz1[((size_t)i * n + j) * c + k] - xyz2[((size_t)i * m + j2) * c + k]);
        grad_xyz1[((size_t)i * n + j) * c + k] += xyz_g;
        grad_xyz2[((size_t)i * m + j2) * c + k] += -xyz_g;
      }
    }
}

void oracle_chamfer_backward(const float* xyz1, const float* xyz2, const float* graddist1,
                             const float* graddist2, const int* idx1, const int* idx2,
                             float* gradxyz1, float* gradxyz2, int b, int n, int m, int c) {
  memset(gradxyz1, 0, sizeof(float) * (size_t)b * n * c);
  memset(gradxyz2, 0, sizeof(float) * (size_t)b * m * c);
  nmdistance_grad_1dir(b, n, c, xyz1, m, xyz2, graddist1, idx1, gradxyz1, gradxyz2);
  nmdistance_grad_1dir(b, m, c, xyz2, n, xyz1, graddist2, idx2, gradxyz2, gradxyz1);
}

/* ---------------------------------------------------------------------------------------------
 * K1 / K3 instantiated for double (AT_DISPATCH_FLOATING_TYPES_AND_HALF, nmdistance_cuda.cu:125,210:
 * scalar_t = double for coordinates, distances and gradients; indices stay int).  Literal, as the
 * structural fp32 form above: 512-chunks, `k==0 || d<best` (:36), `k2==0 || result>best` (:41);
 * d += tmp*tmp contracted to fma by nvcc (:34).  OpenMP over queries only (no data shared).
 * ------------------------------------------------------------------------------------------- */
static void nmdistance_f64_1dir(int b, int n, int c, const double* xyz, int m, const double* xyz2,
                                double* result, int* result_i) {
  for (int i = 0; i < b; ++i) {
    for (int k2 = 0; k2 < m; k2 += PP_CHUNK) {
      const int end_k = (m < k2 + PP_CHUNK ? m : k2 + PP_CHUNK) - k2;
      const double* buf = xyz2 + ((size_t)i * m + k2) * c;
#pragma omp parallel for schedule(static)
      for (int j = 0; j < n; ++j) {
        const double* q = xyz + ((size_t)i * n + j) * c;
        int best_i = 0;
        double best = 0;
        for (int k = 0; k < end_k; ++k) {
          double d = 0;
          for (int e = 0; e < c; ++e) {
            const double t = buf[(size_t)k * c + e] - q[e]; /* :33 */
            d = fma(t, t, d);                               /* :34 */
          }
          if (k == 0 || d < best) {
            best = d;
            best_i = k + k2;
          }
        }
        if (k2 == 0 || result[(size_t)i * n + j] > best) {
          result[(size_t)i * n + j] = best;
          result_i[(size_t)i * n + j] = best_i;
        }
      }
    }
  }
}

void oracle_chamfer_forward_f64(const double* xyz1, const double* xyz2, double* dist1, int* idx1,
                                double* dist2, int* idx2, int b, int n, int m, int c) {
  nmdistance_f64_1dir(b, n, c, xyz1, m, xyz2, dist1, idx1);
  nmdistance_f64_1dir(b, m, c, xyz2, n, xyz1, dist2, idx2);
}

static void nmdistance_grad_f64_1dir(int b, int n, int c, const double* xyz1, int m, const double* xyz2,
                                     const double* grad_dist1, const int* idx1, double* grad_xyz1,
                                     double* grad_xyz2) {
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      const int j2 = idx1[(size_t)i * n + j];
      if (j2 < 0) continue;
      const double g = grad_dist1[(size_t)i * n + j] * 2;
      for (int k = 0; k < c; ++k) {
        const double xyz_g =
            g * (xyz1[((size_t)i * n + j) * c + k] - xyz2[((size_t)i * m + j2) * c + k]);
        grad_xyz1[((size_t)i * n + j) * c + k] += xyz_g;
        grad_xyz2[((size_t)i * m + j2) * c + k] += -xyz_g;
      }
    }
}

void oracle_chamfer_backward_f64(const double* xyz1, const double* xyz2, const double* graddist1,
                                 const double* graddist2, const int* idx1, const int* idx2,
                                 double* gradxyz1, double* gradxyz2, int b, int n, int m, int c) {
  memset(gradxyz1, 0, sizeof(double) * (size_t)b * n * c);
  memset(gradxyz2, 0, sizeof(double) * (size_t)b * m * c);
  nmdistance_grad_f64_1dir(b, n, c, xyz1, m, xyz2, graddist1, idx1, gradxyz1, gradxyz2);
  nmdistance_grad_f64_1dir(b, m, c, xyz2, n, xyz1, graddist2, idx2, gradxyz2, gradxyz1);
}

/* ---------------------------------------------------------------------------------------------
 * K6  furthest_point_sampling_forward_kernel, sampling_cuda.cu:162-233, launched with
 * T = opt_n_threads(n) threads and one block per batch element (:239-243).
 * Literal restatement of the thread decomposition, because it fixes the tie-break:
 *   - thread t owns k = t, t+T, ... in ascending order, strict d2>best from best=-1,besti=0
 *     (:182-183,189,206-209);
 *   - binary tree over the T slots, slot i2 replaces i1 only on strict dists[i1]<dists[i2]
 *     (:214-226), so the lower slot wins exact ties.
 * temp is updated in place (:203-205).  The kernel's formal shared-memory race (read of
 * dists_i[0] at :228 vs the next iteration's write at :211-212) is not reproduced: the intended
 * barrier-separated semantics are.  The first min(T,n) points come from a shared-memory copy
 * (:176-178,193-196), the rest from global memory (:197-201): same values either way.
 * ------------------------------------------------------------------------------------------- */
void oracle_furthest_sampling(const float* input, float* temp, int* idx, int b, int n, int m,
                              int first_idx) {
  if (m <= 0) return;
  const int T = oracle_opt_n_threads(n);
  float* dists = (float*)malloc(sizeof(float) * T);
  int* dists_i = (int*)malloc(sizeof(int) * T);
  for (int i = 0; i < b; ++i) {
    const float* p = input + (size_t)i * n * 3;
    float* tmp = temp + (size_t)i * n;
    int old = first_idx;
    idx[(size_t)i * m] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
      for (int t = 0; t < T; ++t) {
        int besti = 0;
        float best = -1;
        for (int k = t; k < n; k += T) {
          const float td = tmp[k];
          const float d = dist3(p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2], x1, y1, z1);
          const float d2 = fminf(d, td);
          if (d2 != td) tmp[k] = d2;
          if (d2 > best) {
            best = d2;
            besti = k;
          }
        }
        dists[t] = best;
        dists_i[t] = besti;
      }
      for (int u = 0; (1 << u) < T; ++u) {
        for (int t = 0; t < (T >> (u + 1)); ++t) {
          const int i1 = (t * 2) << u, i2 = (t * 2 + 1) << u;
          if (dists[i1] < dists[i2]) {
            dists[i1] = dists[i2];
            dists_i[i1] = dists_i[i2];
          }
        }
      }
      old = dists_i[0];
      idx[(size_t)i * m + j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/* K4  gather_points_kernel_fast, sampling_cuda.cu:9-25: out[b,c,m] = points[b,c,idx[b,m]] */
void oracle_gather_forward(const float* points, const int* idx, float* out, int b, int c, int n,
                           int m) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + idx[(size_t)i * m + j]];
}

/* K5  gather_points_grad_kernel_fast, sampling_cuda.cu:47-64 (grad_points zeroed by the caller,
 * network/operations.py:76-77; here zeroed inside for convenience). */
void oracle_gather_backward(const float* grad_out, const int* idx, float* grad_points, int b,
                            int c, int n, int m) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        grad_points[((size_t)i * c + l) * n + idx[(size_t)i * m + j]] +=
            grad_out[((size_t)i * c + l) * m + j];
}

/* ---------------------------------------------------------------------------------------------
 * K7  ball_query_kernel_fast, sampling_cuda.cu:340-376.  radius2 = radius*radius in fp32 (:354);
 * strict d2<radius2 (:365); first hit fills all nsample slots (:366-369); stop at cnt>=nsample
 * (:373).  idx is zero-filled by the wrapper (sampling.cpp:93-94); done here.
 * Operand order of the difference is (new - x) (:364).
 * ------------------------------------------------------------------------------------------- */
void oracle_ball_query(const float* new_xyz, const float* xyz, int* idx, int b, int n, int m,
                       float radius, int nsample) {
  memset(idx, 0, sizeof(int) * (size_t)b * m * nsample);
  const float radius2 = radius * radius;
#pragma omp parallel for collapse(2) schedule(static)
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < m; ++j) {
      const float* q = new_xyz + ((size_t)i * m + j) * 3;
      const float* p = xyz + (size_t)i * n * 3;
      int* o = idx + ((size_t)i * m + j) * nsample;
      int cnt = 0;
      for (int k = 0; k < n; ++k) {
        const float d2 = dist3(q[0], q[1], q[2], p[k * 3], p[k * 3 + 1], p[k * 3 + 2]);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[l] = k;
          o[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
    }
}

/* K8  group_points_kernel, sampling_cuda.cu:447-467: out[b,l,j,k] = points[b,l,idx[b,j,k]] */
void oracle_group_points(const float* points, const int* idx, float* out, int b, int c, int n,
                         int npoints, int nsample) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          out[(((size_t)i * c + l) * npoints + j) * nsample + k] =
              points[((size_t)i * c + l) * n + idx[((size_t)i * npoints + j) * nsample + k]];
}

/* K9  group_points_grad_kernel, sampling_cuda.cu:482-503 (output zero-filled, sampling.cpp:148) */
void oracle_group_points_grad(const float* grad_out, const int* idx, float* grad_points, int b,
                              int c, int n, int npoints, int nsample) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          grad_points[((size_t)i * c + l) * n + idx[((size_t)i * npoints + j) * nsample + k]] +=
              grad_out[(((size_t)i * c + l) * npoints + j) * nsample + k];
}

/* ---------------------------------------------------------------------------------------------
 * K10  three_nn_kernel_fast, interpolate_gpu.cu:9-52.  bests are double, initialised 1e40
 * (:30), compared with the fp32 d (:37-48), strict <, cast back to float on store (:50).
 * Difference order is (u - x) (:36).
 * ------------------------------------------------------------------------------------------- */
void oracle_three_nn(const float* unknown, const float* known, float* dist2, int* idx, int b,
                     int n, int m) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      const float* u = unknown + ((size_t)i * n + j) * 3;
      const float* kn = known + (size_t)i * m * 3;
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float d = dist3(u[0], u[1], u[2], kn[k * 3], kn[k * 3 + 1], kn[k * 3 + 2]);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float* od = dist2 + ((size_t)i * n + j) * 3;
      int* oi = idx + ((size_t)i * n + j) * 3;
      od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
}

/* ---------------------------------------------------------------------------------------------
 * K nearest neighbours: what the reference gets from pytorch3d.ops.knn_points
 * (not vendored: no source to follow; version unpinned in environment.yml:11).  Restated from its
 * published contract: per point of p1 the K nearest points of p2, squared distances ascending,
 * computed sequentially over the coordinates (d = dx*dx; d += dy*dy; d += dz*dz with FMA
 * contraction = distc above); ties resolved to the lower index (pytorch3d leaves them unspecified);
 * slots beyond the valid points, and rows beyond lengths1, hold (0, 0).
 * ------------------------------------------------------------------------------------------- */
void oracle_knn_nd(const float* p1, const float* p2, const int* len1, const int* len2, float* dist, int* idx,
                   int b, int n, int m, int D, int K) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      float* od = dist + ((size_t)i * n + j) * K;
      int* oi = idx + ((size_t)i * n + j) * K;
      const int n1 = len1 ? (len1[i] < 0 ? 0 : (len1[i] > n ? n : len1[i])) : n;
      const int m2 = len2 ? (len2[i] < 0 ? 0 : (len2[i] > m ? m : len2[i])) : m;
      for (int k = 0; k < K; ++k) { od[k] = 0.0f; oi[k] = 0; }
      if (j >= n1) continue;
      const float* q = p1 + ((size_t)i * n + j) * D;
      int have = 0;
      for (int k = 0; k < m2; ++k) {
        const float d = distc(p2 + ((size_t)i * m + k) * D, q, D);
        if (d != d) continue; /* NaN never enters */
        /* insertion into the ascending list; strict <: the earlier index stays ahead among equals */
        int pos = have < K ? have : K;
        while (pos > 0 && d < od[pos - 1]) --pos;
        if (pos >= K) continue;
        const int last = have < K ? have : K - 1;
        for (int t = last; t > pos; --t) { od[t] = od[t - 1]; oi[t] = oi[t - 1]; }
        od[pos] = d; oi[pos] = k;
        if (have < K) ++have;
      }
      /* valid points that never entered (NaN distances): distance +inf, index 0 */
      const int valid = m2 < K ? m2 : K;
      for (int k = have; k < valid; ++k) { od[k] = INFINITY; oi[k] = 0; }
    }
}

void oracle_knn(const float* p1, const float* p2, const int* len1, const int* len2, float* dist, int* idx,
                int b, int n, int m, int K) {
  oracle_knn_nd(p1, p2, len1, len2, dist, idx, b, n, m, 3, K);
}

/* K11  three_interpolate_kernel_fast, interpolate_gpu.cu:77-97 */
void oracle_three_interpolate(const float* points, const int* idx, const float* weight, float* out,
                              int b, int c, int m, int n) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float* w = weight + ((size_t)i * n + j) * 3;
        const int* id = idx + ((size_t)i * n + j) * 3;
        const float* p = points + ((size_t)i * c + l) * m;
        out[((size_t)i * c + l) * n + j] = fmaf(w[2], p[id[2]], fmaf(w[0], p[id[0]], w[1] * p[id[1]]));
      }
}

/* K12  three_interpolate_grad_kernel_fast, interpolate_gpu.cu:120-142 (grad_points zeroed by the
 * caller, network/pointnet2_utils.py:82). */
void oracle_three_interpolate_grad(const float* grad_out, const int* idx, const float* weight,
                                   float* grad_points, int b, int c, int n, int m) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float* w = weight + ((size_t)i * n + j) * 3;
        const int* id = idx + ((size_t)i * n + j) * 3;
        const float g = grad_out[((size_t)i * c + l) * n + j];
        float* gp = grad_points + ((size_t)i * c + l) * m;
        gp[id[0]] += g * w[0];
        gp[id[1]] += g * w[1];
        gp[id[2]] += g * w[2];
      }
}

/* number of OpenMP threads the fast paths will use (reported as cpu_baseline.cores) */
#ifdef _OPENMP
#include <omp.h>
int oracle_num_threads(void) { return omp_get_max_threads(); }
#else
int oracle_num_threads(void) { return 1; }
#endif
