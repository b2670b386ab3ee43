/*
 * sfmhip.h -- C-ABI of libsfmhip.so: the MI355X (gfx950) implementation of the
 * matching -> triangulation -> bundle-adjustment hot path of CaptainEven/SFM_OpenCV.
 *
 * The reference has no FFI/plugin interface; its boundary is the set of free
 * functions in OpenCV_SFM/NViewReconstuct.cpp ("NView" below). Every entry point
 * here names the reference call it replaces.  Plain pointers and sizes only: no
 * OpenCV, torch or HIP types cross this boundary (streams travel as void*).
 *
 * Conventions
 *   - every function returns int: 0 = ok, negative = SFMHIP_E_*; nothing throws or exits
 *     (reference convention: int 0/-1 + "[Err]:" prints, NView:1122-1126, 301-306).
 *   - "host" entry points take host pointers, run H2D -> kernels -> D2H and return after a
 *     stream sync (reference calls are synchronous, NView:1369/1441/1491).
 *   - "_dev" entry points take device pointers (HBM-resident inputs/outputs) and enqueue on the
 *     context's stream without synchronising; the caller owns all buffers.
 *   - one context per process per GPU (one process per GPU; multi-GPU = N processes + an
 *     all-reduce hook, see sfmhip_ba_set_allreduce).
 */
#ifndef SFMHIP_H_
#define SFMHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFMHIP_OK          0
#define SFMHIP_E_ARG      (-1)  /* bad argument (mirrors the reference's -1) */
#define SFMHIP_E_HIP      (-2)  /* HIP runtime error (see sfmhip_last_error) */
#define SFMHIP_E_COMM     (-3)  /* all-reduce hook failed */
#define SFMHIP_E_NUMERIC  (-4)  /* non-finite value / factorisation failure */
#define SFMHIP_E_NODEVICE (-5)  /* no usable gfx950 device: the product path never falls back to CPU */

/* ---- POD mirrors of the OpenCV types that cross the reference's boundary ---- */
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } sfm_dmatch;      /* cv::DMatch, 16 B */
typedef struct { float x, y; } sfm_point2f;                                              /* cv::Point2f */
typedef struct { double x, y, z; } sfm_point3d;                                          /* cv::Point3d */
typedef struct { float x, y; float size, angle, response; int32_t octave, class_id; } sfm_keypoint; /* cv::KeyPoint, 28 B */
typedef struct { uint8_t b, g, r; } sfm_vec3b;                                           /* cv::Vec3b (BGR as sampled, NView:838) */

typedef struct sfmhip_ctx sfmhip_ctx;
typedef struct sfmhip_descset sfmhip_descset;   /* one image's descriptors, prepared and HBM-resident */
typedef struct sfmhip_ba sfmhip_ba;             /* one bundle-adjustment problem, HBM-resident */

/* ------------------------------------------------------------------------------------------ */
/* context                                                                                    */
/* ------------------------------------------------------------------------------------------ */
int  sfmhip_create(int device, sfmhip_ctx** out);
void sfmhip_destroy(sfmhip_ctx* ctx);
/* enqueue on an external stream (e.g. torch's current stream); NULL = the context's own stream */
int  sfmhip_set_stream(sfmhip_ctx* ctx, void* hip_stream);
int  sfmhip_synchronize(sfmhip_ctx* ctx);
/* The context keeps the device memory of destroyed bundle-adjustment problems (and the temporaries of their construction) for
 * the next one: handing gigabytes back to the driver stalls the following HIP calls for ~0.1 s.  sfmhip_trim releases what is
 * idle (sfmhip_destroy releases everything).  No reference counterpart. */
int  sfmhip_trim(sfmhip_ctx* ctx);
/* Measurement aid (no reference counterpart): with timing enabled every kNN launch sequence on this context is
 * bracketed by HIP events on its stream.  sfmhip_match_kernel_ms synchronises and returns, averaged over the calls
 * since the last query (at most 64): [0] the kNN kernel itself (knn2_i8 / exact f32 / hamming2), [1] merge + re-score,
 * [2] number of calls averaged, [3] reserved (0).  The same switch turns on the per-phase events of the bundle
 * adjustment loop (sfmhip_ba_phase_ms). */
int  sfmhip_set_kernel_timing(sfmhip_ctx* ctx, int enable);
int  sfmhip_match_kernel_ms(sfmhip_ctx* ctx, double out_ms[4]);
const char* sfmhip_last_error(sfmhip_ctx* ctx);
const char* sfmhip_version(void);

/* ------------------------------------------------------------------------------------------ */
/* matching: replaces cv::BFMatcher(norm).knnMatch(query, train, knn, 2) + the ratio tail of   */
/* match_features (NView:873-913; L2/SIFT twin TwoViewReconstruct.cpp:156-194)                */
/* ------------------------------------------------------------------------------------------ */

/* flags reported by sfmhip_descset_info */
#define SFMHIP_DESC_L2_F32      1   /* NORM_L2 on CV_32F rows (SIFT) */
#define SFMHIP_DESC_HAMMING2_U8 2   /* NORM_HAMMING2 on CV_8U rows (AKAZE MLDB, 61 B) */

/* Prepare one image's descriptor matrix (rows x dim, row stride ld elements).
 * L2: a device pass checks whether every value is an integer in [0,255] (OpenCV SIFT output is);
 *     if so the set also carries a biased int8 copy + squared norms and is matched on the int8
 *     MFMA path, which is exact; otherwise only the exact fp32 direct-difference path is used.
 * Hamming2 (nbytes <= 64): rows re-encoded for the popcount kernel (64 B per row) and, for nbytes <= 61 (AKAZE: 61), as 768 FP4
 *     values (384 B per row) for the matrix-core kernel: NORM_HAMMING2 out of a dot product, exactly (csrc/match.hip).
 * host variants copy the rows to HBM first. */
int sfmhip_descset_create_l2_host(sfmhip_ctx*, const float* desc, int rows, int dim, size_t ld, sfmhip_descset** out);
int sfmhip_descset_create_l2_dev (sfmhip_ctx*, const float* d_desc, int rows, int dim, size_t ld, sfmhip_descset** out);
int sfmhip_descset_create_hamming2_host(sfmhip_ctx*, const uint8_t* desc, int rows, int nbytes, size_t ld, sfmhip_descset** out);
int sfmhip_descset_create_hamming2_dev (sfmhip_ctx*, const uint8_t* d_desc, int rows, int nbytes, size_t ld, sfmhip_descset** out);
/* Many images in one call: what match_features_for_all (NView:850-871, called on descriptor_for_all at NView:1369) hands over.
 * desc[i]: rows[i] x dim on the host, row stride ld[i] elements (ld == NULL: dense).  One pass of the staging threads over all
 * rows, one transfer stream, one preparation launch; out[0..n) receives the sets (all NULL again on error).
 * L2: rows whose values are all integers in [0, 255] (cv::SIFT's) cross PCIe as BYTES -- the staging threads convert and verify on
 *     the way into the pinned ring, 128 B per SIFT row instead of 512, and the device re-creates the float rows -- for dim 32 / 64 /
 *     128; an image with any other value is uploaded as floats like sfmhip_descset_create_l2_host did in rounds 1-3.  Same sets,
 *     same matches either way (tests/test_match_gpu.py). */
int sfmhip_descsets_create_l2_host(sfmhip_ctx*, const float* const* desc, const int32_t* rows, int dim, const size_t* ld, int n, sfmhip_descset** out);
int sfmhip_descsets_create_hamming2_host(sfmhip_ctx*, const uint8_t* const* desc, const int32_t* rows, int nbytes, const size_t* ld, int n, sfmhip_descset** out);
void sfmhip_descset_destroy(sfmhip_descset*);
/* re-run the device preparation pass (int8 copy + norms) on the set's float rows; enqueues only */
int sfmhip_descset_refresh(sfmhip_descset*);
/* the same for n sets in ONE launch (what a new batch of frames costs before its pairs are matched) */
int sfmhip_descsets_refresh(sfmhip_ctx*, sfmhip_descset* const* sets, int n);
/* kind = SFMHIP_DESC_*; exact_u8 = 1 when the int8 MFMA path is usable (synchronises) */
int sfmhip_descset_info(sfmhip_descset*, int* kind, int* rows, int* dim, int* exact_u8);

/* kNN-2 of every query row against all train rows (cv::batchDistance semantics [3P]: ascending
 * distance, ties -> lower train index, missing neighbours idx=-1 / dist=FLT_MAX or INT_MAX).
 * idx2: rows_q x 2 int32.  dist2: rows_q x 2 float (L2: sqrtf of the squared distance;
 * Hamming2: the integer distance converted to float, as BFMatcher::knnMatchImpl does).
 * force_path: 0 = auto; L2 sets: 1 = exact fp32 direct-difference path, 2 = int8 MFMA path (E_ARG if unusable); Hamming2 sets: 3 = the
 * VALU popcount kernel, 4 = the FP4 matrix-core kernel (rows of <= 61 bytes; E_ARG otherwise).  Auto takes 4 whenever the rows fit. */
int sfmhip_knn2_dev(sfmhip_ctx*, const sfmhip_descset* query, const sfmhip_descset* train,
                    int32_t* d_idx2, float* d_dist2, int force_path);

/* host-buffer one-shots (what a reference-side binding of BFMatcher::knnMatch(k=2) calls) */
int sfmhip_knn2_l2_f32(sfmhip_ctx*, const float* q, int nq, const float* t, int nt, int dim,
                       size_t ldq, size_t ldt, int32_t* idx2, float* dist2);
int sfmhip_knn2_hamming2_u8(sfmhip_ctx*, const uint8_t* q, int nq, const uint8_t* t, int nt, int nbytes,
                            size_t ldq, size_t ldt, int32_t* idx2, float* dist2);

/* The ratio tail of match_features, NView:880-908, in the reference's arithmetic:
 * pass 1: min_dist = min d0 over rows with !(d0 > ratio*d1) (double compare, NView:884);
 * pass 2: keep row iff !(d0 > ratio*d1 || d0 > mult*max(min_dist, floor_)) (float gate, NView:900-901).
 * Pure host C (no device).  out must hold nq entries.  Rows with fewer than 2 neighbours are
 * dropped (the reference would read out of bounds). */
int sfmhip_ratio_filter(const int32_t* idx2, const float* dist2, int nq,
                        double ratio, float floor_, float mult, sfm_dmatch* out, int* n_out);

/* match_features (NView:873): kNN-2 + ratio tail, one pair, host buffers. */
int sfmhip_match_features_l2(sfmhip_ctx*, const float* q, int nq, const float* t, int nt, int dim,
                             size_t ldq, size_t ldt, sfm_dmatch* out, int* n_out);
int sfmhip_match_features_hamming2(sfmhip_ctx*, const uint8_t* q, int nq, const uint8_t* t, int nt, int nbytes,
                                   size_t ldq, size_t ldt, sfm_dmatch* out, int* n_out);

/* match_features_for_all (NView:850-871) generalised to a pair list: pairs[2*p] = query image,
 * pairs[2*p+1] = train image (reference: (i, i+1)).  All pairs are matched in batched launches;
 * the ratio tail and the ordered compaction run on the device (fp64 compare = the host's).
 * d_matches: n_pairs x max_per_pair sfm_dmatch (device), d_counts: n_pairs int32 (device).
 * max_per_pair must be >= the largest query row count. Enqueues only. */
int sfmhip_match_pairs_dev(sfmhip_ctx*, sfmhip_descset* const* sets, int n_sets,
                           const int32_t* pairs, int n_pairs,
                           double ratio, float floor_, float mult,
                           sfm_dmatch* d_matches, int max_per_pair, int32_t* d_counts);
/* host-output form: matches_out[n_pairs*max_per_pair], counts_out[n_pairs]; synchronises. */
int sfmhip_match_pairs(sfmhip_ctx*, sfmhip_descset* const* sets, int n_sets,
                       const int32_t* pairs, int n_pairs,
                       double ratio, float floor_, float mult,
                       sfm_dmatch* matches_out, int max_per_pair, int32_t* counts_out);

/* Materialised L2 distance matrix dist[i*ld + j] = sqrtf(sum_k (q[i,k]-t[j,k])^2), rows_q x rows_t
 * float32 in HBM (the "10k x 10k SIFT distance GEMM" roofline case; what cv::batchDistance(K=0)
 * would return [3P]). Enqueues only. */
int sfmhip_l2_distance_matrix_dev(sfmhip_ctx*, const sfmhip_descset* query, const sfmhip_descset* train,
                                  float* d_dist, size_t ld, int force_path);

/* Self-test of the distance epilogue: the materialised matrix takes sqrtf of exact integers < 2^24 with a short
 * correctly-rounded sequence (v_sqrt_f32 + neighbour test); this runs it over every such integer on the device and
 * returns in *mismatches how many results differ from sqrtf (must be 0).  Synchronises. */
int sfmhip_selftest_exact_sqrt(sfmhip_ctx*, int* mismatches);

/* ------------------------------------------------------------------------------------------ */
/* triangulation: replaces cv::triangulatePoints + the float32 de-homogenisation loop of       */
/* reconstruct (NView:1147-1156).  P1,P2: row-major 3x4 float32 (= float(K)*[float(R)|float(T)],*/
/* NView:1129-1143, built by the caller/wrapper).  xy1,xy2: n x 2 float32.                      */
/* xyzw: 4 x n float32 exactly like pts4d (row-major, may be NULL); xyz: n x 3 double holding   */
/* float32-exact values (Point3f -> Point3d, NView:1155; may be NULL).                          */
/* ------------------------------------------------------------------------------------------ */
int sfmhip_triangulate2_f32(sfmhip_ctx*, const float P1[12], const float P2[12],
                            const float* xy1, const float* xy2, int n, float* xyzw, double* xyz);
int sfmhip_triangulate2_f32_dev(sfmhip_ctx*, const float P1[12], const float P2[12],
                                const float* d_xy1, const float* d_xy2, int n, float* d_xyzw, double* d_xyz);
/* fused get_matched_points (NView:989-1003) + triangulation: points are gathered on the device
 * from keypoint arrays through the match list. kp1/kp2: sfm_keypoint arrays (device). */
int sfmhip_triangulate2_matches_dev(sfmhip_ctx*, const float P1[12], const float P2[12],
                                    const sfm_keypoint* d_kp1, const sfm_keypoint* d_kp2,
                                    const sfm_dmatch* d_matches, int n, float* d_xyzw, double* d_xyz);

/* N-view extension (SURVEY 8f rank 4 -- NOT reference behaviour: the reference triangulates a point once, from the pair
 * that created it, NView:1428-1453).  Multi-view DLT of every track from ALL its observations, on normalised image
 * coordinates ((u-cx)/fx, (v-cy)/fy) with [R|t] from the angle-axis extrinsics of the BA parameterisation (NView:151-183,
 * 1464-1487); points with fewer than two observations come back as NaN.  n_views_out (may be NULL) = observations used.
 * sfmhip_reprojection_errors: pixel error |K (R X + t)/z - uv| of every observation, e.g. to filter tracks after BA. */
int  sfmhip_triangulate_tracks(sfmhip_ctx* ctx, const double K4[4], const double* ext6, int n_cam,
                               const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs, int n_pt,
                               double* pts_out, int32_t* n_views_out);
int  sfmhip_reprojection_errors(sfmhip_ctx* ctx, const double K4[4], const double* ext6, int n_cam, const double* pts, int n_pt,
                                const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs, double* err_out);

/* ------------------------------------------------------------------------------------------ */
/* bundle adjustment: replaces bundle_adjustment (NView:1162-1244) = ceres::Solve on            */
/* AutoDiffCostFunction<ReprojectCost,2,4,6,3> (NView:142-184) + HuberLoss(4) + SPARSE_SCHUR.   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int    max_num_iterations;          /* 50   (Ceres default [3P]) */
    double initial_trust_region_radius; /* 1e4  */
    double max_trust_region_radius;     /* 1e16 */
    double min_trust_region_radius;     /* 1e-32 */
    double min_relative_decrease;       /* 1e-3 */
    double min_lm_diagonal;             /* 1e-6 */
    double max_lm_diagonal;             /* 1e32 */
    double function_tolerance;          /* 1e-6 */
    double gradient_tolerance;          /* 1e-10 */
    double parameter_tolerance;         /* 1e-8 */
    double huber_delta;                 /* 4.0  (NView:1184); <= 0 disables the loss */
    int    jacobi_scaling;              /* 1 */
    int    fix_first_camera;            /* 1    (NView:1178) */
    int    fix_intrinsics;              /* 0    (NView:1181: free, shared) */
    int    verbose;                     /* 0    (NView:1216-1217) */
    int    linearizer;                  /* 0    how the reduced system is built (same result up to rounding): 0 / 1 = per-observation
                                         *      kernels; 2 = run tiles (points sharing a camera list linearised once, reduced on the
                                         *      matrix pipe) when every point has 1..7 observations, else the per-observation kernels */
    int    solver;                      /* 0    reduced camera solve (same result up to rounding): 0 = the chain solver (csrc/ba_chain.hpp: fronts
                                         *      in LDS, a camera at a time, two launches) where the cameras form a chain of band width <= 3
                                         *      cameras and at most 640 of them are free, else 1; 1 = nested dissection, one launch per tree
                                         *      level (csrc/ba_solver.hpp; the only solver of rounds 1-3), dense blocked fallback */
} sfm_ba_options;

#define SFMHIP_BA_CONVERGENCE     0
#define SFMHIP_BA_NO_CONVERGENCE  1
#define SFMHIP_BA_FAILURE         2

typedef struct {
    int    termination;      /* SFMHIP_BA_* */
    int    iterations;       /* LM iterations taken (excluding iteration 0), successful + unsuccessful */
    int    successful_steps;
    int    num_residuals;    /* 2 * n_obs (NView:1236) */
    double initial_cost;     /* 1/2 sum rho(|r|^2) (NView:1237) */
    double final_cost;
    double final_radius;
    double final_gradient_max_norm;
    double total_time_s;          /* the whole call (ceres::Solver::Summary::total_time_in_seconds, the "Time (s)" of NView:1239) */
    double preprocessor_time_s;   /* sfmhip_ba_solve: problem construction (uploads, orderings, pair lists) + solver plan + scaling */
    double minimizer_time_s;      /* the LM loop */
    double postprocessor_time_s;  /* sfmhip_ba_solve: parameters back into the caller's arrays + teardown */
} sfm_ba_summary;

void sfmhip_ba_default_options(sfm_ba_options* o);

/* One-shot, in place on caller memory like the reference (NView:1174,1181,1209):
 * intrinsic4 = (fx,fy,cx,cy); ext6 = n_cam x 6 (angle-axis, t); pts = n_pt x 3;
 * observation k: camera obs_cam[k] sees point obs_pt[k] at obs_uv[2k..2k+1] (double, NView:1199). */
int sfmhip_ba_solve(sfmhip_ctx*, double* intrinsic4, double* ext6, int n_cam, double* pts, int n_pt,
                    const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                    const sfm_ba_options* opts, sfm_ba_summary* summary);

/* Resident form (bench, multi-GPU): create uploads + builds the per-point / per-camera orderings. */
int  sfmhip_ba_create(sfmhip_ctx*, const double* intrinsic4, const double* ext6, int n_cam,
                      const double* pts, int n_pt,
                      const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                      const sfm_ba_options* opts, sfmhip_ba** out);
void sfmhip_ba_destroy(sfmhip_ba*);
/* Multi-GPU: this rank holds a shard of the points (all their observations) and a replica of the
 * cameras/intrinsics.  The hook must sum `count` doubles at device pointer `d_buf` in place over all
 * ranks, ordered on `hip_stream`; return 0 on success.  (RCCL: ncclAllReduce(d_buf,d_buf,count,
 * ncclDouble,ncclSum,comm,stream); torch.distributed: dist.all_reduce on a tensor view.) */
typedef int (*sfmhip_allreduce_fn)(void* user, void* d_buf, size_t count, void* hip_stream);
/* rank / world: this process's index and the number of ranks (<= 64); resets the LM state.
 * On several ranks an LM iteration costs ONE call of the hook: the step's scalars (model cost change, candidate cost, step and
 * parameter norms, error flag) ride in the message of the next linearisation, which is built speculatively at the candidate
 * point; a rejected step (or a trust-region radius other than the guessed one) discards it and costs one more call. */
int  sfmhip_ba_set_allreduce(sfmhip_ba*, sfmhip_allreduce_fn fn, void* user, int rank, int world);
/* The production hook, inside the library: RCCL over xGMI, librccl.so looked up at run time (SFMHIP_E_COMM without it).
 * Rank 0 obtains the 128-byte unique id, the launcher hands it to every rank (torch.distributed / MPI / a file), every rank creates
 * its communicator on the context's device and installs it: the LM loop then calls ncclAllReduce(double, sum, in place) on the
 * context's stream itself.  No reference counterpart (the reference is one CPU process, NView:1334-1524). */
int  sfmhip_rccl_available(void);
int  sfmhip_rccl_get_unique_id(void* id128);
int  sfmhip_rccl_comm_create(sfmhip_ctx*, const void* id128, int rank, int world, void** comm);
int  sfmhip_rccl_comm_destroy(void* comm);
int  sfmhip_ba_set_rccl(sfmhip_ba*, void* comm, int rank, int world);
int  sfmhip_rccl_allreduce_f64(sfmhip_ctx*, void* comm, void* d_buf, size_t count);
/* bundle_adjustment (NView:1162-1244) on several GPUs of ONE process -- what a C++ caller like the reference's main() uses: one context per
 * device, same arguments and in-place semantics as sfmhip_ba_solve.  The points are sharded by the first camera that sees them (the cameras
 * cut into n_ctx consecutive ranges of equal observation count; a few parallel passes on the host), one host thread per context builds and
 * runs its shard, the reduced-system message is summed by RCCL; where two contexts share a device (one-card rehearsal) or librccl is missing,
 * by a host-staged exchange inside the process.  The communicators of a set of contexts are created by the first call (ncclCommInitAll) and
 * kept until one of the contexts is destroyed.  Every rank's shard is built before any rank enters a collective: a shard that fails
 * (e.g. out of memory on one device) returns its error on all ranks; a collective that does not complete within 60 s ends the LM loop with
 * SFMHIP_E_COMM and the communicators are dropped.  summary->preprocessor_time_s = sharding + the slowest shard's construction + plan.
 * n_ctx = 1 is sfmhip_ba_solve. */
int  sfmhip_ba_solve_multi(sfmhip_ctx* const* ctxs, int n_ctx, double* intrinsic4, double* ext6, int n_cam, double* pts, int n_pt,
                           const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                           const sfm_ba_options* opts, sfm_ba_summary* summary);
/* match_features_for_all (NView:850-871) from HOST matrices on several GPUs of one process: the pairs in n_ctx contiguous blocks, a block's
 * images uploaded to its context only over that device's own PCIe link (a chain: a block of images + one halo image), one host thread per
 * context, no exchange; matches[p * max_per_pair ...] / counts[p] in pair order, exactly what sfmhip_descsets_create_*_host +
 * sfmhip_match_pairs give on one context (n_ctx = 1 is that sequence in one call).  kind: SFMHIP_DESC_L2_F32 (desc[i]: float rows,
 * dim columns) or SFMHIP_DESC_HAMMING2_U8 (uint8 rows, dim = bytes per row); ld: row strides in elements or NULL (dense). */
int  sfmhip_match_pairs_multi(sfmhip_ctx* const* ctxs, int n_ctx, int kind, const void* const* desc, const int32_t* rows, int dim,
                              const size_t* ld, int n_images, const int32_t* pairs, int n_pairs,
                              double ratio, float floor_, float mult, sfm_dmatch* matches, int max_per_pair, int32_t* counts);
/* test hook: the next n device allocations of the context fail (SFMHIP_E_HIP) -- what an out-of-memory device looks like to its callers */
int  sfmhip_debug_fail_allocations(sfmhip_ctx*, int n);
/* run the LM loop to termination */
int  sfmhip_ba_run(sfmhip_ba*, sfm_ba_summary* summary);
/* run exactly n_iter LM iterations (tolerance checks disabled); state carries over between calls */
int  sfmhip_ba_iterate(sfmhip_ba*, int n_iter, sfm_ba_summary* summary);
/* restore the parameters given at create and reset the LM state */
int  sfmhip_ba_reset(sfmhip_ba*);
int  sfmhip_ba_get_params(sfmhip_ba*, double* intrinsic4, double* ext6, double* pts);
/* Test/diagnostic: linearise at the current parameters with trust-region radius `radius` and copy
 * out the reduced camera system (order n = 6*(n_cam - fixed) + 4*(!fix_intrinsics); S row-major n x n,
 * both triangles filled; rhs n) after the all-reduce.  Either pointer may be NULL.  radius < 0: the points are
 * damped with |radius| but the camera-side damping is skipped, so the result is additive over point shards. */
int  sfmhip_ba_reduced_system(sfmhip_ba*, double radius, double* S, double* rhs, int* n, double* cost);
/* Test/diagnostic: copy out one of the tables sfmhip_ba_create builds on the device (NView:1187-1210 is where the reference
 * hands the same observation list to ceres::Problem): "pt_slot" (caller's point -> storage slot), "pt_start", "ocam", "opt",
 * "ouv" (observations by slot, then camera), "cam_start", "cam_pt", "cam_uv" (camera-ordered copy), "blk_crange", "blk_cam",
 * "blk_chunk", "chunk_desc", "items" (camera-pair lists; int32 except the double pixel arrays), "setup_ms" (4 doubles: host
 * clock of sfmhip_ba_create, cumulative: inputs + observation sort, + orderings, + pair lists, whole call).
 * out == NULL: only *n_bytes is set.  Synchronises. */
int  sfmhip_ba_debug_table(sfmhip_ba*, const char* name, void* out, size_t cap_bytes, size_t* n_bytes);
/* average device time (ms) per LM iteration of the last sfmhip_ba_iterate call, measured with HIP events on the
 * context's stream WHILE sfmhip_set_kernel_timing(ctx, 1) is in effect (all zero otherwise: the thirteen event records
 * cost ~27 us per iteration, so they are off by default): [0]=linearise+Schur build, [1]=reduced solve, [2]=back-substitution+cost, [3]=total,
 * single kernels: [4]=ba_camera_kernel, [5]=ba_schur_kernel -- or, where the two share one launch, [4]=ba_camschur_kernel and [5]=0
 * ([4]=ba_tile_kernel with linearizer = 2) --, [6]=chol_node_forward_kernel of the leaf level (0 if unused);
 * [7] = number of non-zero 32x32 blocks of the Cholesky factor (not a time) */
int  sfmhip_ba_phase_ms(sfmhip_ba*, double out_ms[8]);

/* ------------------------------------------------------------------------------------------ */
/* "next" row 8f-1: normals for the .ply writer (estimate_normals NView:551-599 + PCAFitPlane   */
/* 601-690): brute-force kNN-K (self excluded) + 3x3 PCA, smallest-eigenvalue vector, flipped   */
/* so that n.mean <= 0, normalised.  pts: n x 3 double; normals: n x 3 double.                  */
/* ------------------------------------------------------------------------------------------ */
int sfmhip_estimate_normals(sfmhip_ctx*, const double* pts, int n, int K, double* normals);

#ifdef __cplusplus
}
#endif
#endif /* SFMHIP_H_ */
