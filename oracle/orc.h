/*
 * oracle/orc.h -- CPU restatement of the reference's matching -> triangulation -> BA path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sfm_opencv_amd/ links, imports or calls this; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may (as the checker / the
 * reported CPU baseline, never as the product).
 *
 * PARITY STATUS: the arithmetic of this path lives in third-party code that is not in
 * /root/reference (OpenCV 4.4.0 opencv_world440, Ceres [unpinned version], Eigen 3.3.7 --
 * OpenCV_SFM/OpenCV_SFM.vcxproj:94-95,121,130), the reference cannot be compiled here (needs
 * <io.h>, OpenCV, Ceres, glog, Eigen; none installed) and it ships no tests or golden vectors
 * for matching, triangulation or BA.  => "parity unpinned" for orc_knn2_*, orc_ratio_filter,
 * orc_triangulate2 and orc_ba_*: they restate the published algorithms ([3P] notes below) and
 * are cross-checked against independent numpy/scipy brute force in tests/.  Pinned by the
 * reference's own output files: orc_estimate_normals + the .ply/.yml writers
 * (Viewer/structure_ba.yml -> Viewer/structure_ba.ply), see tests/test_golden_outputs.py.
 *
 * "NView:L" = /root/reference/OpenCV_SFM/NViewReconstuct.cpp:L.
 */
#ifndef ORC_H_
#define ORC_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } orc_dmatch; /* cv::DMatch */

/* ---- matching (NView:873-913; TwoViewReconstruct.cpp:156-194 for NORM_L2) ---- */
/* cv::BFMatcher(NORM_L2).knnMatch(k=2) on CV_32F rows [3P batchDistance]. idx2/dist2: nq x 2. */
void orc_knn2_l2_f32(const float* q, int nq, const float* t, int nt, int dim,
                     size_t ldq, size_t ldt, int32_t* idx2, float* dist2);
/* cv::BFMatcher(NORM_HAMMING2).knnMatch(k=2) on CV_8U rows (NView:876-877). dist as float. */
void orc_knn2_hamming2_u8(const uint8_t* q, int nq, const uint8_t* t, int nt, int nbytes,
                          size_t ldq, size_t ldt, int32_t* idx2, float* dist2);
/* full distance matrix sqrtf(normL2Sqr) (batchDistance with K=0 [3P]); dist: nq x nt */
void orc_l2_distance_matrix_f32(const float* q, int nq, const float* t, int nt, int dim,
                                size_t ldq, size_t ldt, float* dist, size_t ldd);
/* ratio tail NView:880-908. returns number of matches written to out (<= nq). */
int  orc_ratio_filter(const int32_t* idx2, const float* dist2, int nq,
                      double ratio, float floor_, float mult, orc_dmatch* out);

/* ---- triangulation (NView:1117-1159 + cvTriangulatePoints [3P]) ---- */
void orc_triangulate2(const float P1[12], const float P2[12], const float* xy1, const float* xy2,
                      int n, float* xyzw /*4 x n or NULL*/, double* xyz /*n x 3 or NULL*/);
/* NView:1129-1143: P = float(K) * [float(R) | float(T)] in float32 (cv::Mat float gemm [3P]) */
void orc_projection_matrix(const double K[9], const double R[9], const double T[3], float P[12]);

/* N-view DLT on normalised coordinates + per-observation reprojection error (extension; see orc_triangulate.c) */
void orc_triangulate_tracks(const double K4[4], const double* ext6, int n_cam, const int32_t* obs_cam, const int32_t* obs_pt,
                            const double* obs_uv, int n_obs, int n_pt, double* pts, int32_t* n_views);
void orc_reprojection_errors(const double K4[4], const double* ext6, int n_cam, const double* pts, const int32_t* obs_cam,
                             const int32_t* obs_pt, const double* obs_uv, int n_obs, double* err);

/* ---- bundle adjustment (NView:142-184, 1162-1244 + Ceres defaults [3P]) ---- */
typedef struct {
    int    max_num_iterations;
    double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
    double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double huber_delta;
    int    jacobi_scaling, fix_first_camera, fix_intrinsics, verbose;
} orc_ba_options;   /* same layout as sfm_ba_options (include/sfmhip.h) */

typedef struct {
    int    termination, iterations, successful_steps, num_residuals;
    double initial_cost, final_cost, final_radius, final_gradient_max_norm, total_time_s;
} orc_ba_summary;   /* same layout as sfm_ba_summary */

void orc_ba_default_options(orc_ba_options* o);
/* ReprojectCost::operator() (NView:151-183) with 13-wide forward-mode duals like Ceres autodiff:
 * r[2], J[2][13] ordered [intrinsic 4 | extrinsic 6 | point 3]; no loss applied. */
void orc_reproject(const double K4[4], const double ext6[6], const double X[3], const double uv[2],
                   double r[2], double J[26]);
/* trace arrays (may be NULL): per LM iteration it>=0: cost after the iteration, radius, accepted flag.
 * force_iterations > 0: run exactly that many iterations with the tolerance checks disabled. */
int  orc_ba_solve(double* K4, double* ext6, int n_cam, double* pts, int n_pt,
                  const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                  const orc_ba_options* opts, orc_ba_summary* summary,
                  int force_iterations, double* trace_cost, double* trace_radius, int32_t* trace_ok,
                  int trace_cap);
/* One linearisation at the given parameters (jacobi scaling computed at these parameters):
 * reduced camera system S (n x n row-major, full symmetric), rhs (n), cost.
 * n = 6*(n_cam - fix_first) + 4*(!fix_intrinsics).  Returns n.  S/rhs may be NULL to query n.
 * radius < 0: points damped with |radius|, camera-side damping skipped (the form that adds up over point shards). */
int  orc_ba_reduced_system(const double* K4, const double* ext6, int n_cam, const double* pts, int n_pt,
                           const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                           const orc_ba_options* opts, double radius, double* S, double* rhs, double* cost);
void orc_set_num_threads(int n);
int  orc_get_max_threads(void);

/* ---- normals (estimate_normals NView:551-599, PCAFitPlane NView:601-690) ---- */
void orc_estimate_normals(const double* pts, int n, int K, double* normals);

#ifdef __cplusplus
}
#endif
#endif
