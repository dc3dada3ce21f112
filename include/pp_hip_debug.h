/*
 * pp_hip_debug.h -- test and benchmark knobs of libpp_hip.so.  NOT part of the drop-in C ABI (pp_hip.h).
 *
 * Every operator of pp_hip.h picks its kernel variant from the problem size.  The functions below override
 * that choice process-wide so that tests can compare the variants with each other bit for bit and benchmarks
 * can time them; 0 always restores the automatic choice.  They are the library's only global mutable state
 * besides idempotent per-device "LDS limit raised" flags: each is one relaxed std::atomic<int>, safe to set
 * from any thread, read by the next call of the operator.  Product code (pytorch_points_amd/network, _ext)
 * never calls them.
 */
#ifndef PP_HIP_DEBUG_H
#define PP_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Chamfer forward: 0 automatic, 1 brute force (every pair), 2 grid search wherever structurally possible */
void pp_debug_set_nmdistance_search(int mode);
/* brute-force kernel variant (Q queries per lane, G points per group, packed / prefetch forms; chamfer.hip) */
void pp_debug_set_nmdistance_variant(int variant);
/* grid search, wave-private form of the search kernel: staged points per wave (320, 384, 512; selecting one also
 * selects that form; 0 = the default, the tile form) */
void pp_debug_set_nmdistance_stage_cap(int points);
/* grid search, unlabeled: queries per workgroup of the stage-A kernel (0 = 512; 256, 1024); -1 = no stage-A kernel.
 * The same values are read once from the environment variable PP_NMDISTANCE_TILE when the knob is 0. */
void pp_debug_set_nmdistance_tile(int queries);
/* grid search, unlabeled: the build of sets of at most 16384 aligned points: 0 = sorted through the LDS, a slab owning
 * whole z-layers (default), 1 = the general build always (tests and A/B timing) */
void pp_debug_set_nmdistance_build(int general);
/* grid search, unlabeled: directions no search can prune (every reference point at nearly one distance from the queries)
 * are routed to the every-pair kernel, launched behind the search once an earlier call on the device has routed one:
 * 0 = that, 1 = never (the search serves every direction), 2 = the every-pair launch always follows */
void pp_debug_set_nmdistance_routing(int mode);
/* grid search, unlabeled: the far-field (group) search of the list kernel skips the cell rows that hold no points by a
 * per-set row bitmap written in the stage-A launch's tail: 0 = that (default), 1 = every row is looked up (tests, A/B) */
void pp_debug_set_nmdistance_row_bitmap(int off);
/* labeled Chamfer brute force: 1 = the one-lane-per-query kernel */
void pp_debug_set_labeled_variant(int variant);
/* Chamfer backward: 1 LDS doubles, 2 CSR lists, 3 LDS fp32 columns, 4 global atomics, 5 deterministic */
void pp_debug_set_nmdistance_backward_variant(int variant);
/* per-kernel HIP-event timing of the grid forward (build, search), read back after the call */
void pp_debug_set_nmdistance_kernel_timing(int on);
int pp_debug_nmdistance_kernel_ms(float* build_ms, float* search_ms);
/* the same with the search's two launches apart (stage A by tiles, then what it left); stage_a_ms = 0 without a stage-A kernel */
int pp_debug_nmdistance_kernel_ms3(float* build_ms, float* stage_a_ms, float* rest_ms);

/* ... and the build's and the stage-A kernel's OWN durations of the most recent forward timed with the knob at 2 (the
 * launches' begin / end stamps, hipExtLaunchKernelGGL: what rocprofv3 reports per dispatch) */
int pp_debug_nmdistance_kernel_own_ms(float* build_ms, float* stage_a_ms);

/* unlabeled grid forward: queries its stage-A kernel left to the list kernel, per direction (2 B values; synchronises) */
int pp_debug_nmdistance_pending(const void* workspace, int B, int N, int M, unsigned* totals);

void pp_debug_set_fps_v1(int form); /* 0 = the library's choice, 1 = one workgroup per batch element over all points,
                                      * 2 = the CU cluster over all points, 3 = the bucketed kernel */
/* the bucketed FPS kernel's serial chain (N <= 65536): 0 = the library's choice (several mutually independent picks
 * per barrier round from 32768 points or 1024 picks, one pick per round below), 1 = one pick per round, 2 = several per round */
void pp_debug_set_fps_bucket_chain(int form);
/* the bucketed FPS kernel's sort: 0 = by the whole chip from 32768 points (six short launches in front of the kernel),
 * below that inside the kernel by its one workgroup; 1 = always inside the kernel */
void pp_debug_set_fps_bucket_sort(int mode);
void pp_debug_set_gather_variant(int variant);
void pp_debug_set_ball_query_variant(int variant); /* scan kernels: 1 = one wave per 64 centres */
void pp_debug_set_ball_query_search(int mode);     /* 0 automatic, 1 scan, 2 grid wherever possible */
void pp_debug_set_ball_query_lpc(int lanes_per_centre);
void pp_debug_set_group_points_variant(int variant);
void pp_debug_set_group_points_grad_variant(int variant);
void pp_debug_set_three_nn_search(int mode);       /* 0 automatic, 1 scan */
void pp_debug_set_three_interpolate_variant(int variant);
void pp_debug_set_three_interpolate_grad_variant(int variant);
void pp_debug_set_scatter_mode(int mode);          /* 1 = never use the sorted scatter-add form */
void pp_debug_set_knn_search(int mode);            /* 0 automatic, 1 scan */
/* a pure streaming store of `bytes` (16-byte stores, non-temporal or plain, `workgroups` x 1024 threads): the store
 * ceiling bench.py reports beside group_points (roofline.peak_measured) */
int pp_debug_store_ceiling(void* buf, size_t bytes, int nontemporal, int workgroups, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PP_HIP_DEBUG_H */
