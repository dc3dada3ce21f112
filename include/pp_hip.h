/*
 * pp_hip.h -- C ABI of libpp_hip.so, the MI355X (gfx950) implementation of the pytorch_points
 * `_ext` hot path.
 *
 * Each entry point replaces one function of the reference's pybind modules
 * `pytorch_points._ext.losses` (_ext/nmdistance.cpp:30-34) and `pytorch_points._ext.sampling`
 * (_ext/sampling.cpp:205-216); the reference interface each one stands in for is cited beside it
 * (paths relative to /root/reference/pytorch_points/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous memory; fp32 data, int32 indices;
 *   - the caller allocates every output and every workspace; nothing is allocated, retained or
 *     synchronised inside the library.  Thread-safe: the only process-wide state is a set of atomic
 *     test / benchmark knobs, declared separately in pp_hip_debug.h and never set by product code,
 *     and idempotent per-device "LDS limit raised" flags;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); all work is
 *     enqueued on it, on the device that is current when the call is made;
 *   - return value: 0 on success, otherwise a hipError_t value (PP_EINVAL = hipErrorInvalidValue
 *     for bad sizes / null pointers).  Never exit()s, never prints.
 *   - NaN coordinates are outside the contract (the reference's behaviour for them is an
 *     artefact of its 512-point chunking).
 */
#ifndef PP_HIP_H
#define PP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_OK 0
#define PP_EINVAL 1    /* == hipErrorInvalidValue */
#define PP_ENOTSUP 801 /* == hipErrorNotSupported: an *_ordered_* entry point cannot serve this shape */

/* library / build identification: "pp_hip <version> gfx950" */
const char* pp_version(void);

/* ---- _ext.losses ------------------------------------------------------------------------- */

/* Replaces losses.nmdistance_forward(xyz1,xyz2,dist1,dist2,idx1,idx2)
 *   (_ext/nmdistance.cpp:13-15 -> chamfer_cuda_forward, _ext/nmdistance_cuda.cu:118-140).
 * xyz1 (B,N,C), xyz2 (B,M,C) -> dist1,idx1 (B,N): nearest xyz2 point of every xyz1 point
 * (squared distance, lowest index on exact ties); dist2,idx2 (B,M): the converse.
 * N == 0 or M == 0: outputs are zero-filled (what the reference's Python wrapper leaves). */
int pp_nmdistance_forward_f32(const float* xyz1, const float* xyz2, float* dist1, int* idx1,
                              float* dist2, int* idx2, int B, int N, int M, int C, void* stream);

/* The same operation with a caller-provided scratch buffer.  With a workspace of at least
 * pp_nmdistance_forward_workspace_bytes(...) bytes (0 = not applicable to these sizes) the search
 * is an exact uniform-grid search with the brute force as its fallback -- bit-identical outputs,
 * far fewer distance evaluations; with workspace == NULL it is pp_nmdistance_forward_f32. */
size_t pp_nmdistance_forward_workspace_bytes(int B, int N, int M, int C);
int pp_nmdistance_forward_ws_f32(const float* xyz1, const float* xyz2, float* dist1, int* idx1,
                                 float* dist2, int* idx2, int B, int N, int M, int C,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* Replaces losses.labeled_nmdistance_forward(xyz1,xyz2,label1,label2,dist1,dist2,idx1,idx2)
 *   (_ext/nmdistance.cpp:17-20 -> labeled_chamfer_cuda_forward, _ext/nmdistance_cuda.cu:142-166).
 * label1 (B,N), label2 (B,M) as fp32.  Unmatched query: idx -1, dist 0. */
int pp_labeled_nmdistance_forward_f32(const float* xyz1, const float* xyz2, const float* label1,
                                      const float* label2, float* dist1, int* idx1, float* dist2,
                                      int* idx2, int B, int N, int M, int C, void* stream);

/* The labeled forward through the exact uniform-grid search (same outputs; see
 * pp_nmdistance_forward_ws_f32).  The workspace is larger than the unlabeled one (the labels are
 * carried in sorted order); 0 bytes = not applicable, null / too small workspace = the call above. */
size_t pp_labeled_nmdistance_forward_workspace_bytes(int B, int N, int M, int C);
int pp_labeled_nmdistance_forward_ws_f32(const float* xyz1, const float* xyz2, const float* label1,
                                         const float* label2, float* dist1, int* idx1, float* dist2,
                                         int* idx2, int B, int N, int M, int C, void* workspace,
                                         size_t workspace_bytes, void* stream);

/* Replaces losses.nmdistance_backward(xyz1,xyz2,gradxyz1,gradxyz2,graddist1,graddist2,idx1,idx2)
 *   (_ext/nmdistance.cpp:23-27 -> chamfer_cuda_backward, _ext/nmdistance_cuda.cu:195-221).
 * gradxyz1 (B,N,C), gradxyz2 (B,M,C) are fully overwritten (the reference zeroes them first). */
int pp_nmdistance_backward_f32(const float* xyz1, const float* xyz2, const float* graddist1,
                               const float* graddist2, const int* idx1, const int* idx2,
                               float* gradxyz1, float* gradxyz2, int B, int N, int M, int C,
                               void* stream);

/* The same two operators for double clouds: the reference dispatches its kernels over the floating types
 * (AT_DISPATCH_FLOATING_TYPES_AND_HALF, _ext/nmdistance_cuda.cu:125,210), scalar_t = double for coordinates,
 * distances and gradients, int indices.  Every-pair scan (no workspace form); same tie rule and rounding order
 * as the reference's kernel instantiated for double. */
int pp_nmdistance_forward_f64(const double* xyz1, const double* xyz2, double* dist1, int* idx1,
                              double* dist2, int* idx2, int B, int N, int M, int C, void* stream);
int pp_nmdistance_backward_f64(const double* xyz1, const double* xyz2, const double* graddist1,
                               const double* graddist2, const int* idx1, const int* idx2,
                               double* gradxyz1, double* gradxyz2, int B, int N, int M, int C,
                               void* stream);
/* ... and for half clouds (scalar_t = at::Half, the third type of that dispatch): coordinates, distances and gradients
 * are IEEE binary16 words (torch.float16; opaque pointers here), every subtraction, product and sum rounded to half
 * separately as c10::Half's operators do (no fused operation), comparisons in half; the scattered gradient terms are
 * added with packed half atomics, each addition rounded, in arrival order (the reference: a CAS loop on at::Half).
 * bfloat16 is not part of the reference's dispatch and is not provided. */
int pp_nmdistance_forward_f16(const void* xyz1, const void* xyz2, void* dist1, int* idx1, void* dist2, int* idx2,
                              int B, int N, int M, int C, void* stream);
int pp_nmdistance_backward_f16(const void* xyz1, const void* xyz2, const void* graddist1, const void* graddist2,
                               const int* idx1, const int* idx2, void* gradxyz1, void* gradxyz2, int B, int N,
                               int M, int C, void* stream);

/* ---- _ext.sampling ----------------------------------------------------------------------- */

/* Replaces sampling.furthest_sampling(m, seedIdx, input, temp, idx)
 *   (_ext/sampling.cpp:68-82 -> furthest_sampling_cuda_forward, _ext/sampling_cuda.cu:235-325).
 * xyz (B,N,3); temp (B,N) in/out running min squared distance (caller pre-fills 1e10,
 * network/geo_operations.py:33); idx (B,npoint) out, idx[:,0] = seed_idx.  temp == NULL: every point starts at 1e10
 * and nothing is written back (the wrapper one level up allocates temp itself and never reads it) -- served where the
 * bucketed kernel runs, PP_ENOTSUP otherwise (the caller then passes a temp of its own).
 * workspace: pp_furthest_sampling_workspace_bytes(...) bytes of scratch (may be NULL if that is 0); a pure
 * host computation (an upper bound over devices).  Layout: a 256-byte status word | the ring of the cluster kernel |
 * the scratch of the bucketed kernel (the cloud re-ordered into spatial buckets: 16 B per point + one word per point,
 * 65536 words per batch element up to 65536 points).  With that scratch -- 2048 <= N <= 2^22 and 32 or more picks --
 * the sampling is ONE workgroup per batch element that visits, per pick, only the buckets the pick can change (an
 * exact test: same picks, same temp); nothing waits for another workgroup there.  Without it (or for N > 65536 when
 * the batch leaves room) the cluster kernel runs: the first 256 bytes hold its STICKY status word, which the caller
 * zeroes once after allocating the buffer.  That kernel shares a batch element between workgroups
 * that wait for each other; it is only launched with as many workgroups as the current device keeps resident
 * (its real CU count and the occupancy query), and every wait is bounded: if one times out (CUs held by a
 * kernel of another stream), the call leaves zeros from that step on in idx and sets the status word, which
 * stays set until the caller clears it. */
size_t pp_furthest_sampling_workspace_bytes(int B, int N, int npoint);
/* Reads the status word (synchronises `stream`): 0 = ok, 1 = a wait of some pp_furthest_sampling_f32 call
 * on this workspace has timed out since the word was last zeroed. */
int pp_furthest_sampling_status(const void* workspace, void* stream);
int pp_furthest_sampling_f32(const float* xyz, float* temp, int* idx, int B, int N, int npoint,
                             int seed_idx, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the picked points' coordinates written as well: the caller one level up,
 * network/geo_operations.py:59-63 (furthest_point_sample = the sampling + gather_points of the coordinates; SURVEY.md
 * §8f N3), in ONE launch.  sampled (B,npoint,3), or (B,3,npoint) with channels_first != 0; NULL = idx only. */
int pp_furthest_sampling_gather_f32(const float* xyz, float* temp, int* idx, float* sampled, int channels_first,
                                    int B, int N, int npoint, int seed_idx, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* Replaces sampling.gather_forward(b,c,n,npoints,points,idx,out)
 *   (_ext/sampling.cpp:19-28 -> _ext/sampling_cuda.cu:9-45).  points (B,C,N), idx (B,M) -> out (B,C,M) */
int pp_gather_forward_f32(const float* points, const int* idx, float* out, int B, int C, int N,
                          int M, void* stream);

/* Replaces sampling.gather_backward(b,c,n,npoints,grad_out,idx,grad_points)
 *   (_ext/sampling.cpp:31-41 -> _ext/sampling_cuda.cu:47-84).  ACCUMULATES into grad_points (B,C,N),
 * which the caller zero-fills (network/operations.py:76-77). */
int pp_gather_backward_f32(const float* grad_out, const int* idx, float* grad_points, int B, int C,
                           int N, int M, void* stream);

/* Replaces sampling.ball_query(new_xyz, xyz, radius, nsample)
 *   (_ext/sampling.cpp:85-104 -> _ext/sampling_cuda.cu:340-397).
 * new_xyz (B,M,3) centres, xyz (B,N,3) -> idx (B,M,nsample), fully written (rows with no hit are
 * written as zeros, which is what the reference's zero-filled allocation leaves). */
int pp_ball_query_f32(const float* new_xyz, const float* xyz, int* idx, int B, int N, int M,
                      float radius, int nsample, void* stream);

/* The same operation through a uniform grid over xyz (exact, same output), with the scan as the
 * fallback for batch elements whose radius spans too many cells.  workspace:
 * pp_ball_query_workspace_bytes(...) bytes (0 = not applicable); NULL = the scan. */
size_t pp_ball_query_workspace_bytes(int B, int N, int M, int nsample);
int pp_ball_query_ws_f32(const float* new_xyz, const float* xyz, int* idx, int B, int N, int M,
                         float radius, int nsample, void* workspace, size_t workspace_bytes,
                         void* stream);

/* Replaces sampling.group_points(points, idx)
 *   (_ext/sampling.cpp:113-138 -> _ext/sampling_cuda.cu:447-478).
 * points (B,C,N), idx (B,npoint,nsample) -> out (B,C,npoint,nsample), fully written. */
int pp_group_points_f32(const float* points, const int* idx, float* out, int B, int C, int N,
                        int npoint, int nsample, void* stream);

/* The same gather into a tensor whose batch stride (in elements) is larger than C*npoint*nsample,
 * e.g. the channel slice [3, 3+C) of QueryAndGroup's concatenated output
 * (network/operations.py:196-204): the fused caller writes its result once instead of
 * grouping and then torch.cat-ing 4 GiB. */
int pp_group_points_strided_f32(const float* points, const int* idx, float* out, int B, int C, int N,
                                int npoint, int nsample, long long out_batch_stride, void* stream);

/* Replaces sampling.group_points_grad(grad_out, idx, n)
 *   (_ext/sampling.cpp:140-161 -> _ext/sampling_cuda.cu:482-513).
 * ACCUMULATES into grad_points (B,C,N), which the caller zero-fills. */
int pp_group_points_grad_f32(const float* grad_out, const int* idx, float* grad_points, int B,
                             int C, int N, int npoint, int nsample, void* stream);

/* grad_out given as a channel slice of a wider tensor (batch stride in elements). */
int pp_group_points_grad_strided_f32(const float* grad_out, const int* idx, float* grad_points, int B,
                                     int C, int N, int npoint, int nsample,
                                     long long grad_out_batch_stride, void* stream);

/* Replaces sampling.three_nn_wrapper(b,n,m,unknown,known,dist2,idx)
 *   (_ext/sampling.cpp:163-172 -> _ext/interpolate_gpu.cu:9-74).
 * unknown (B,N,3), known (B,M,3) -> dist2 (B,N,3) squared distances ascending, idx (B,N,3).
 * M < 3: unused slots hold +inf / index 0. */
int pp_three_nn_f32(const float* unknown, const float* known, float* dist2, int* idx, int B, int N,
                    int M, void* stream);

/* The same three_nn through an exact uniform-grid search (three_nn_grid.hip): identical outputs,
 * a few dozen candidates per unknown point instead of M.  pp_three_nn_workspace_bytes returns 0
 * when the grid path does not apply (small N or M); with a null / too small workspace the call is
 * pp_three_nn_f32.  The workspace is scratch: no state is kept between calls. */
size_t pp_three_nn_workspace_bytes(int B, int N, int M);
int pp_three_nn_ws_f32(const float* unknown, const float* known, float* dist2, int* idx, int B, int N,
                       int M, void* workspace, size_t workspace_bytes, void* stream);

/* K nearest neighbours: replaces pytorch3d.ops.knn_points, which the reference calls at
 *   network/model_loss.py:120,147,378, geo_operations.py:112,139, layers.py:52,99,115
 *   (pytorch3d is an un-vendored dependency, environment.yml:11).
 * p1 (B,N,3), p2 (B,M,3) -> dist2 (B,N,K) squared distances ascending, idx (B,N,K); 1 <= K <= 32.
 * Ties go to the lower index.  lengths1 / lengths2 (B ints, may be null): valid points per cloud;
 * slots beyond the valid points of p2 and rows beyond lengths1 hold (0, 0).
 * pp_knn_ws_f32: the same through the exact uniform-grid search (identical outputs); ragged batches
 * and small clouds take the scan. */
int pp_knn_f32(const float* p1, const float* p2, const int* lengths1, const int* lengths2, float* dist2,
               int* idx, int B, int N, int M, int K, void* stream);
size_t pp_knn_workspace_bytes(int B, int N, int M, int K);
int pp_knn_ws_f32(const float* p1, const float* p2, const int* lengths1, const int* lengths2, float* dist2,
                  int* idx, int B, int N, int M, int K, void* workspace, size_t workspace_bytes,
                  void* stream);
/* The same operator for ANY point dimension D (1..512) and K up to 128 -- the reference's feature-space
 * searches, network/layers.py:52,99 (DenseEdgeConv: D = channel count, K = k + 1).  p1 (B,N,D), p2 (B,M,D).
 * Squared distance = the sequential chain d = fma(t_c, t_c, d), c = 0..D-1; ties to the lower index. */
int pp_knn_nd_f32(const float* p1, const float* p2, const int* lengths1, const int* lengths2, float* dist2,
                  int* idx, int B, int N, int M, int D, int K, void* stream);

/* Replaces sampling.three_interpolate_wrapper(b,c,m,n,points,idx,weight,out)
 *   (_ext/sampling.cpp:175-188 -> _ext/interpolate_gpu.cu:77-117).
 * points (B,C,M), idx (B,N,3), weight (B,N,3) -> out (B,C,N) */
int pp_three_interpolate_f32(const float* points, const int* idx, const float* weight, float* out,
                             int B, int C, int M, int N, void* stream);

/* Replaces sampling.three_interpolate_grad_wrapper(b,c,n,m,grad_out,idx,weight,grad_points)
 *   (_ext/sampling.cpp:190-203 -> _ext/interpolate_gpu.cu:120-160).
 * ACCUMULATES into grad_points (B,C,M), which the caller zero-fills
 * (network/pointnet2_utils.py:82). */
int pp_three_interpolate_grad_f32(const float* grad_out, const int* idx, const float* weight,
                                  float* grad_points, int B, int C, int N, int M, void* stream);

/* ---- the three scatter-add backwards with a caller-provided workspace -----------------------
 * Same contracts as pp_group_points_grad_strided_f32 / pp_gather_backward_f32 /
 * pp_three_interpolate_grad_f32 (they ACCUMULATE into caller-zeroed outputs).  With a workspace of
 * pp_scatter_workspace_bytes(B, triples per batch element, destinations per batch element,
 * triples per source element, weighted) bytes the (source, destination) pairs are sorted once per
 * call and no atomics are issued; without one (or when the size does not qualify: the function
 * returns 0 bytes) they are the atomic forms.
 *   group_points_grad:      triples = npoint*nsample, destinations = N, per_source = 1, weighted = 0
 *   gather_backward:        triples = M,              destinations = N, per_source = 1, weighted = 0
 *   three_interpolate_grad: triples = 3*N,            destinations = M, per_source = 3, weighted = 1 */
size_t pp_scatter_workspace_bytes(int B, long long triples_per_batch, int destinations, int per_source,
                                  int weighted);
int pp_group_points_grad_ws_f32(const float* grad_out, const int* idx, float* grad_points, int B, int C,
                                int N, int npoint, int nsample, long long grad_out_batch_stride,
                                void* workspace, size_t workspace_bytes, void* stream);
/* The same with grad_points (B,C,N) WRITTEN instead of accumulated into: it need not be initialised (the reference's
 * group_points_grad allocates a zero-filled tensor and adds into it, _ext/sampling.cpp:148-150 -- the fill and the read
 * of the output are 2 x 256 MB of traffic at B=32, C=128, N=16384 that this form does not move). */
int pp_group_points_grad_out_ws_f32(const float* grad_out, const int* idx, float* grad_points, int B, int C,
                                    int N, int npoint, int nsample, long long grad_out_batch_stride,
                                    void* workspace, size_t workspace_bytes, void* stream);
int pp_gather_backward_ws_f32(const float* grad_out, const int* idx, float* grad_points, int B, int C,
                              int N, int M, void* workspace, size_t workspace_bytes, void* stream);
int pp_three_interpolate_grad_ws_f32(const float* grad_out, const int* idx, const float* weight,
                                     float* grad_points, int B, int C, int N, int M, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* ---- deterministic ("ordered") backward passes --------------------------------------------------
 * The reference adds every gradient term with a global fp32 atomic, so its sums depend on the order
 * the hardware happens to serve (nmdistance_cuda.cu:180-181, sampling_cuda.cu:63,499-500,
 * interpolate_gpu.cu:139-141); the default entry points above keep the same freedom.  These four add
 * each destination's terms in ASCENDING SOURCE ORDER with no floating-point atomics: the result is
 * identical from run to run, and identical bit for bit to a sequential loop over the reference's
 * launches (the CPU oracle under oracle/).  The host side selects them when
 * torch.are_deterministic_algorithms_enabled().  They return PP_ENOTSUP for shapes the ordered forms
 * cannot serve (Chamfer: C != 3 or a cloud beyond ~19000 points; scatter ops: more than 20480
 * destinations per batch element, or a workspace smaller than pp_scatter_workspace_bytes): there is no
 * deterministic substitute.  Workspaces as for the *_ws_* entry points; grad_points must be zero-filled
 * by the caller for the three scatter ops (they accumulate), gradxyz is overwritten. */
int pp_nmdistance_backward_ordered_f32(const float* xyz1, const float* xyz2, const float* graddist1,
                                       const float* graddist2, const int* idx1, const int* idx2,
                                       float* gradxyz1, float* gradxyz2, int B, int N, int M, int C,
                                       void* stream);
int pp_group_points_grad_ordered_f32(const float* grad_out, const int* idx, float* grad_points, int B, int C,
                                     int N, int npoint, int nsample, long long grad_out_batch_stride,
                                     void* workspace, size_t workspace_bytes, void* stream);
int pp_gather_backward_ordered_f32(const float* grad_out, const int* idx, float* grad_points, int B, int C,
                                   int N, int M, void* workspace, size_t workspace_bytes, void* stream);
int pp_three_interpolate_grad_ordered_f32(const float* grad_out, const int* idx, const float* weight,
                                          float* grad_points, int B, int C, int N, int M, void* workspace,
                                          size_t workspace_bytes, void* stream);

/* The library also exports pp_debug_set_* switches that force one kernel variant or another; they
 * exist for the parity tests and for tuning and are deliberately not declared here. */

/* cuda_utils.h:11-16 opt_n_threads -- the FPS tie-break depends on it, so it is part of the ABI */
int pp_opt_n_threads(int work_size);

/* ---- batch-sharded execution (no counterpart in the reference, which has no multi-GPU code):
 * packing of a rank's Chamfer outputs for ONE collective per step, and unpacking of the gathered
 * buffer (pytorch_points_amd/sharded.py PackedShardGather).  n1 = B_local*N, n2 = B_local*M.
 * Packed layout: dist1 | dist2 | idx1 | idx2, indices as uint16 when compact != 0 (all < 65536).
 * pp_shard_packed_bytes: bytes of one rank's packed buffer (multiple of 16 = the row stride).
 * pp_shard_pack_f32 with dist1 == dist2 == NULL: the distances are IN PLACE already -- the caller gave the packed
 * buffer's first n1 + n2 floats to pp_nmdistance_forward*_f32 as its dist1 / dist2 outputs -- and only the indices are
 * narrowed in behind them.  pp_shard_unpack_f32 with dist1 == dist2 == NULL: indices only (the distances are read where
 * they were gathered, as strided views of the gathered buffer). */
size_t pp_shard_packed_bytes(long long n1, long long n2, int compact);
int pp_shard_pack_f32(const float* dist1, const float* dist2, const int* idx1, const int* idx2, void* packed,
                      long long n1, long long n2, int compact, void* stream);
int pp_shard_unpack_f32(const void* gathered, int world, long long stride_bytes, long long n1, long long n2,
                        int compact, float* dist1, float* dist2, int* idx1, int* idx2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PP_HIP_H */
