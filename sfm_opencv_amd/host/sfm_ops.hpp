// sfm_ops.hpp -- host-side mirror of the reference's hot-path interface (header-only C++17, no OpenCV).
//
// The reference (OpenCV_SFM/NViewReconstuct.cpp, "NView") has no plugin API: its boundary is the free functions
// declared at NView:35-138.  This header gives the same function names, argument order and error behaviour on POD
// mirrors of the OpenCV types, implemented over the C-ABI of libsfmhip.so (include/sfmhip.h).  A maintainer of the
// reference keeps his cv:: types and uses the shims shown in INTEGRATION.md; code without OpenCV uses this header.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/sfmhip.h"

namespace sfm {

// ---- POD mirrors (layouts identical to the cv:: types they replace) ------------------------------------------
struct Point2f { float x = 0, y = 0; };
struct Point2d { double x = 0, y = 0; };
struct Point3f { float x = 0, y = 0, z = 0; };
struct Point3d { double x = 0, y = 0, z = 0; Point3d() = default; Point3d(double a, double b, double c) : x(a), y(b), z(c) {}
                 Point3d(const Point3f& p) : x(p.x), y(p.y), z(p.z) {} };
struct Vec3b { uint8_t v[3] = { 0, 0, 0 }; uint8_t& operator[](int i) { return v[i]; } const uint8_t& operator[](int i) const { return v[i]; } };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };   // 28 bytes
struct DMatch { int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0; };                          // 16 bytes
static_assert(sizeof(KeyPoint) == sizeof(sfm_keypoint) && sizeof(DMatch) == sizeof(sfm_dmatch), "layout");

enum { CV_8U = 0, CV_32F = 5, CV_64F = 6 };
// minimal row-major matrix: descriptors (CV_8U / CV_32F), 3x3 R, 3x1 T, 6x1 extrinsic, 4x1 intrinsic, 3x3 K (CV_64F)
struct Mat {
    int rows = 0, cols = 0, type = CV_64F;
    std::vector<uint8_t> buf;
    Mat() = default;
    Mat(int r, int c, int t) : rows(r), cols(c), type(t), buf((size_t)r * c * elem(t), 0) {}
    static size_t elem(int t) { return t == CV_8U ? 1 : (t == CV_32F ? 4 : 8); }
    bool empty() const { return rows == 0 || cols == 0; }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(buf.data()) + (size_t)r * cols; }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(buf.data()) + (size_t)r * cols; }
    template <typename T> T& at(int r, int c = 0) { return ptr<T>(r)[c]; }
    template <typename T> const T& at(int r, int c = 0) const { return ptr<T>(r)[c]; }
    static Mat eye3() { Mat m(3, 3, CV_64F); m.at<double>(0, 0) = m.at<double>(1, 1) = m.at<double>(2, 2) = 1.0; return m; }
};

struct Pt3DPly { float x = 0, y = 0, z = 0, nx = 0, ny = 0, nz = 0; uint8_t r = 0, g = 0, b = 0; };   // NView:21-32

// ---- context -------------------------------------------------------------------------------------------------
inline sfmhip_ctx* context(int device = 0)
{
    static sfmhip_ctx* ctx = nullptr;
    if (!ctx) {
        const int rc = sfmhip_create(device, &ctx);
        if (rc != SFMHIP_OK) { printf("[Err]: sfmhip_create failed (%d): no usable MI355X device, and there is no CPU fallback.\n", rc); ctx = nullptr; }
    }
    return ctx;
}

// Devices match_features_for_all() spreads its image pairs over (sfmhip_match_pairs_multi: contiguous blocks of the chain, every block
// uploaded over its own device's PCIe link, no exchange) and bundle_adjustment() its points (one context each; the reduced-system
// sums go over RCCL inside the library, sfmhip_ba_solve_multi).  Default: the one context above.  Listing a device twice gives two
// contexts on it: the one-card rehearsal.
inline std::vector<sfmhip_ctx*>& ba_contexts()
{
    static std::vector<sfmhip_ctx*> v;
    return v;
}
inline bool set_ba_devices(const std::vector<int>& devices)
{
    std::vector<sfmhip_ctx*>& v = ba_contexts();
    for (size_t i = 1; i < v.size(); ++i) sfmhip_destroy(v[i]);
    v.clear();
    for (size_t i = 0; i < devices.size(); ++i) {
        sfmhip_ctx* c = nullptr;
        if (i == 0) c = context(devices[0]);
        else if (sfmhip_create(devices[i], &c) != SFMHIP_OK) c = nullptr;
        if (!c) { printf("[Err]: no context on device %d\n", devices[i]); for (size_t k = 1; k < v.size(); ++k) sfmhip_destroy(v[k]); v.clear(); return false; }
        v.push_back(c);
    }
    return true;
}

// ---- matching (NView:873-913, 850-871) ---------------------------------------------------------------------------
// CV_8U rows -> NORM_HAMMING2 (the live configuration, NView:876); CV_32F rows -> NORM_L2 (TwoViewReconstruct.cpp:159)
inline void match_features(const Mat& query, const Mat& train, std::vector<DMatch>& matches)
{
    matches.clear();
    sfmhip_ctx* ctx = context();
    if (!ctx || query.rows == 0) return;
    std::vector<DMatch> out((size_t)query.rows);
    int n = 0, rc;
    if (query.type == CV_8U)
        rc = sfmhip_match_features_hamming2(ctx, query.ptr<uint8_t>(), query.rows, train.ptr<uint8_t>(), train.rows, query.cols,
                                            (size_t)query.cols, (size_t)train.cols, reinterpret_cast<sfm_dmatch*>(out.data()), &n);
    else
        rc = sfmhip_match_features_l2(ctx, query.ptr<float>(), query.rows, train.ptr<float>(), train.rows, query.cols,
                                      (size_t)query.cols, (size_t)train.cols, reinterpret_cast<sfm_dmatch*>(out.data()), &n);
    if (rc != SFMHIP_OK) { printf("[Err]: match_features: %s\n", sfmhip_last_error(ctx)); return; }
    out.resize((size_t)n);
    matches.swap(out);
}

// one batched launch sequence for the whole chain instead of N-1 serial calls
inline void match_features_for_all(const std::vector<Mat>& descriptor_for_all, std::vector<std::vector<DMatch>>& matches_for_all)
{
    matches_for_all.clear();
    sfmhip_ctx* ctx = context();
    const int n = (int)descriptor_for_all.size();
    if (!ctx || n < 2) return;
    std::vector<sfmhip_descset*> sets((size_t)n, nullptr);
    int max_rows = 1, rc = SFMHIP_OK;
    // all images in one call (one pass of the staging threads, one transfer stream, one preparation launch) when they share a type
    // and a row length -- what extract_features produces; image by image otherwise
    bool uniform = true;
    for (int i = 0; i < n; ++i) {
        const Mat& d = descriptor_for_all[i];
        max_rows = d.rows > max_rows ? d.rows : max_rows;
        uniform = uniform && d.type == descriptor_for_all[0].type && d.cols == descriptor_for_all[0].cols && d.cols > 0;
    }
    if (uniform && ba_contexts().size() > 1) {
        // several devices (--gpus=): the chain in contiguous blocks over the contexts, host matrices straight into the one call
        std::vector<int32_t> rows((size_t)n), pairs, counts((size_t)n - 1, 0);
        std::vector<const void*> ptrs((size_t)n);
        for (int i = 0; i < n; ++i) { rows[i] = descriptor_for_all[i].rows; ptrs[i] = descriptor_for_all[i].buf.data(); }
        for (int i = 0; i + 1 < n; ++i) { printf("Matching images %d - %d\n", i, i + 1); pairs.push_back(i); pairs.push_back(i + 1); }
        std::vector<DMatch> out((size_t)(n - 1) * max_rows);
        const bool ham = descriptor_for_all[0].type == CV_8U;
        rc = sfmhip_match_pairs_multi(ba_contexts().data(), (int)ba_contexts().size(), ham ? SFMHIP_DESC_HAMMING2_U8 : SFMHIP_DESC_L2_F32, ptrs.data(), rows.data(),
                                      descriptor_for_all[0].cols, nullptr, n, pairs.data(), n - 1, 0.6, 10.0f, 5.0f,
                                      reinterpret_cast<sfm_dmatch*>(out.data()), max_rows, counts.data());
        if (rc != SFMHIP_OK) { printf("[Err]: match_features_for_all: %s\n", sfmhip_last_error(ctx)); return; }
        for (int i = 0; i + 1 < n; ++i) {
            matches_for_all.emplace_back(out.begin() + (size_t)i * max_rows, out.begin() + (size_t)i * max_rows + counts[i]);
            if (counts[i] == 0) printf("[Warning]: zero matches between %d and %d.\n", i, i + 1);
        }
        return;
    }
    if (uniform) {
        std::vector<int32_t> rows((size_t)n);
        for (int i = 0; i < n; ++i) rows[i] = descriptor_for_all[i].rows;
        if (descriptor_for_all[0].type == CV_8U) {
            std::vector<const uint8_t*> ptrs((size_t)n);
            for (int i = 0; i < n; ++i) ptrs[i] = descriptor_for_all[i].ptr<uint8_t>();
            rc = sfmhip_descsets_create_hamming2_host(ctx, ptrs.data(), rows.data(), descriptor_for_all[0].cols, nullptr, n, sets.data());
        } else {
            std::vector<const float*> ptrs((size_t)n);
            for (int i = 0; i < n; ++i) ptrs[i] = descriptor_for_all[i].ptr<float>();
            rc = sfmhip_descsets_create_l2_host(ctx, ptrs.data(), rows.data(), descriptor_for_all[0].cols, nullptr, n, sets.data());
        }
    } else {
        for (int i = 0; i < n && rc == SFMHIP_OK; ++i) {
            const Mat& d = descriptor_for_all[i];
            rc = d.type == CV_8U ? sfmhip_descset_create_hamming2_host(ctx, d.ptr<uint8_t>(), d.rows, d.cols, (size_t)d.cols, &sets[i])
                                 : sfmhip_descset_create_l2_host(ctx, d.ptr<float>(), d.rows, d.cols, (size_t)d.cols, &sets[i]);
        }
    }
    std::vector<int32_t> pairs, counts((size_t)n - 1, 0);
    for (int i = 0; i + 1 < n; ++i) { printf("Matching images %d - %d\n", i, i + 1); pairs.push_back(i); pairs.push_back(i + 1); }
    std::vector<DMatch> out((size_t)(n - 1) * max_rows);
    if (rc == SFMHIP_OK)
        rc = sfmhip_match_pairs(ctx, sets.data(), n, pairs.data(), n - 1, 0.6, 10.0f, 5.0f,
                                reinterpret_cast<sfm_dmatch*>(out.data()), max_rows, counts.data());
    for (auto* s : sets) sfmhip_descset_destroy(s);
    if (rc != SFMHIP_OK) { printf("[Err]: match_features_for_all: %s\n", sfmhip_last_error(ctx)); return; }
    for (int i = 0; i + 1 < n; ++i) {
        matches_for_all.emplace_back(out.begin() + (size_t)i * max_rows, out.begin() + (size_t)i * max_rows + counts[i]);
        if (counts[i] == 0) printf("[Warning]: zero matches between %d and %d.\n", i, i + 1);
    }
}

inline void get_matched_points(const std::vector<KeyPoint>& p1, const std::vector<KeyPoint>& p2, const std::vector<DMatch> matches,
                               std::vector<Point2f>& out_p1, std::vector<Point2f>& out_p2)
{
    out_p1.clear(); out_p2.clear();
    for (const auto& m : matches) { out_p1.push_back(p1[m.queryIdx].pt); out_p2.push_back(p2[m.trainIdx].pt); }
}
inline void get_matched_colors(const std::vector<Vec3b>& c1, const std::vector<Vec3b>& c2, const std::vector<DMatch> matches,
                               std::vector<Vec3b>& out_c1, std::vector<Vec3b>& out_c2)
{
    out_c1.clear(); out_c2.clear();
    for (const auto& m : matches) { out_c1.push_back(c1[m.queryIdx]); out_c2.push_back(c2[m.trainIdx]); }
}

// ---- triangulation (NView:1117-1159) ---------------------------------------------------------------------------
// proj = float(K) * [float(R) | float(T)] as a float32 product with double accumulation (cv::gemm on CV_32F [3P])
inline void projection_matrix(const Mat& K, const Mat& R, const Mat& T, float P[12])
{
    float fK[9], RT[12];
    for (int i = 0; i < 9; ++i) fK[i] = (float)K.ptr<double>()[i];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) RT[4 * r + c] = (float)R.at<double>(r, c); RT[4 * r + 3] = (float)T.at<double>(r, 0); }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (double)fK[3 * r + k] * (double)RT[4 * k + c];
            P[4 * r + c] = (float)s;
        }
}

inline int reconstruct(const Mat& K, Mat& R1, Mat& T1, Mat& R2, Mat& T2, std::vector<Point2f>& pts2d_1, std::vector<Point2f>& pts2d_2,
                       std::vector<Point3d>& structure)
{
    if (pts2d_1.size() == 0 || pts2d_2.size() == 0) { printf("[Err]: empty 2d points.\n"); return -1; }
    sfmhip_ctx* ctx = context();
    if (!ctx) return -1;
    float P1[12], P2[12];
    projection_matrix(K, R1, T1, P1); projection_matrix(K, R2, T2, P2);
    const int n = (int)pts2d_1.size();
    structure.clear(); structure.resize((size_t)n);
    const int rc = sfmhip_triangulate2_f32(ctx, P1, P2, &pts2d_1[0].x, &pts2d_2[0].x, n, nullptr, &structure[0].x);
    if (rc != SFMHIP_OK) { printf("[Err]: reconstruct: %s\n", sfmhip_last_error(ctx)); structure.clear(); return -1; }
    return 0;
}

// TwoViewReconstruct.cpp:231-250: first camera at the origin (proj1 = float(K) [I | 0]), second [R | T]; the result stays
// HOMOGENEOUS -- `structure` is the 4 x N CV_32F matrix cv::triangulatePoints fills (one column per point); the division
// by w happens when the file is written (TwoViewReconstruct.cpp:341-346).  void like the reference: errors only print.
inline void reconstruct(Mat& K, Mat& R, Mat& T, std::vector<Point2f>& p1, std::vector<Point2f>& p2, Mat& structure)
{
    const int n = (int)p1.size();
    structure = Mat(4, n, CV_32F);
    sfmhip_ctx* ctx = context();
    if (!ctx || n == 0 || p2.size() != p1.size()) { if (n == 0) structure = Mat(); return; }
    Mat R0 = Mat::eye3(), T0(3, 1, CV_64F);
    float P1[12], P2[12];
    projection_matrix(K, R0, T0, P1); projection_matrix(K, R, T, P2);
    const int rc = sfmhip_triangulate2_f32(ctx, P1, P2, &p1[0].x, &p2[0].x, n, structure.ptr<float>(), nullptr);
    if (rc != SFMHIP_OK) { printf("[Err]: reconstruct: %s\n", sfmhip_last_error(ctx)); structure = Mat(); }
}

// ---- track bookkeeping (host, integer; NView:959-983, 1246-1301) --------------------------------------------------
inline void init_correspondence(const std::vector<std::vector<KeyPoint>>& key_points_for_all, const std::vector<DMatch>& matches01,
                                const std::vector<uint8_t>& mask, std::vector<std::vector<int>>& correspond_struct_idx)
{
    correspond_struct_idx.clear();
    correspond_struct_idx.resize(key_points_for_all.size());
    for (size_t i = 0; i < key_points_for_all.size(); ++i) correspond_struct_idx[i].resize(key_points_for_all[i].size(), -1);
    int idx = 0;
    for (size_t i = 0; i < matches01.size(); ++i) {
        if (!mask.empty() && mask[i] == 0) continue;
        correspond_struct_idx[0][matches01[i].queryIdx] = idx;
        correspond_struct_idx[1][matches01[i].trainIdx] = idx;
        ++idx;
    }
    printf("Total %d 3D points from the first two frames' valid keypoint matches.\n", idx);
}
inline void get_obj_pts_and_img_pts(const std::vector<DMatch>& matches, const std::vector<int>& struct_indices,
                                    const std::vector<Point3d>& structure, const std::vector<KeyPoint>& key_points,
                                    std::vector<Point3f>& object_points, std::vector<Point2f>& image_points)
{
    object_points.clear(); image_points.clear();
    for (const auto& m : matches) {
        const int si = struct_indices[m.queryIdx];
        if (si < 0) continue;
        object_points.push_back(Point3f{ (float)structure[si].x, (float)structure[si].y, (float)structure[si].z });
        image_points.push_back(key_points[m.trainIdx].pt);
    }
}
inline void fuse_structure(const std::vector<DMatch>& matches, std::vector<int>& struct_indices, std::vector<int>& next_struct_indices,
                           std::vector<Point3d>& structure, std::vector<Point3d>& next_structure,
                           std::vector<Vec3b>& colors, std::vector<Vec3b>& next_colors)
{
    for (size_t i = 0; i < matches.size(); ++i) {
        const int q = matches[i].queryIdx, t = matches[i].trainIdx;
        const int si = struct_indices[q];
        if (si >= 0) { next_struct_indices[t] = si; continue; }
        structure.push_back(next_structure[i]);
        colors.push_back(next_colors[i]);
        struct_indices[q] = next_struct_indices[t] = (int)structure.size() - 1;
    }
}
inline void maskout_2d_pts_pair(const std::vector<uint8_t>& mask, std::vector<Point2f>& a, std::vector<Point2f>& b)
{
    std::vector<Point2f> ca = a, cb = b; a.clear(); b.clear();
    for (size_t i = 0; i < mask.size(); ++i) if (mask[i] > 0) { a.push_back(ca[i]); b.push_back(cb[i]); }
}
inline void maskout_colors(const std::vector<uint8_t>& mask, std::vector<Vec3b>& c)
{
    std::vector<Vec3b> cc = c; c.clear();
    for (size_t i = 0; i < mask.size(); ++i) if (mask[i] > 0) c.push_back(cc[i]);
}

// cv::Rodrigues 3x3 -> 3x1 (NView:1480) and back
inline void Rodrigues(const Mat& R, Mat& r)
{
    r = Mat(3, 1, CV_64F);
    const double* m = R.ptr<double>();
    double c = (m[0] + m[4] + m[8] - 1.0) * 0.5; c = c > 1 ? 1 : (c < -1 ? -1 : c);
    const double th = std::acos(c);
    double w[3] = { m[7] - m[5], m[2] - m[6], m[3] - m[1] };
    if (th < 1e-12) { for (int i = 0; i < 3; ++i) r.at<double>(i) = 0.5 * w[i]; return; }
    if (M_PI - th < 1e-6) {
        double A[3] = { (m[0] + 1) * 0.5, (m[4] + 1) * 0.5, (m[8] + 1) * 0.5 };
        int k = A[0] >= A[1] ? (A[0] >= A[2] ? 0 : 2) : (A[1] >= A[2] ? 1 : 2);
        double ax[3], d = std::sqrt(A[k] > 0 ? A[k] : 0);
        for (int i = 0; i < 3; ++i) ax[i] = ((m[3 * k + i] + (i == k ? 1.0 : 0.0)) * 0.5) / d;
        const double nn = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        for (int i = 0; i < 3; ++i) r.at<double>(i) = th * ax[i] / nn;
        return;
    }
    const double s = th / (2.0 * std::sin(th));
    for (int i = 0; i < 3; ++i) r.at<double>(i) = s * w[i];
}

// cv::Rodrigues 3x1 -> 3x3 (NView:1418): R = cos(th) I + (1 - cos(th)) r r' + sin(th) [r]x, r = v / |v| [3P]
inline void Rodrigues_vec(const Mat& rvec, Mat& R)
{
    R = Mat::eye3();
    const double x = rvec.at<double>(0), y = rvec.at<double>(1), z = rvec.at<double>(2);
    const double th = std::sqrt(x * x + y * y + z * z);
    if (th < 2.220446049250313e-16) return;
    const double c = std::cos(th), s = std::sin(th), c1 = 1.0 - c, rx = x / th, ry = y / th, rz = z / th;
    double* m = R.ptr<double>();
    m[0] = c + c1 * rx * rx;       m[1] = c1 * rx * ry - s * rz;  m[2] = c1 * rx * rz + s * ry;
    m[3] = c1 * rx * ry + s * rz;  m[4] = c + c1 * ry * ry;       m[5] = c1 * ry * rz - s * rx;
    m[6] = c1 * rx * rz - s * ry;  m[7] = c1 * ry * rz + s * rx;  m[8] = c + c1 * rz * rz;
}

// ---- bundle adjustment (NView:1162-1244): in place on intrinsic (4x1), extrinsics (6x1 each), structure -----------
inline void bundle_adjustment(Mat& intrinsic, std::vector<Mat>& extrinsics, std::vector<std::vector<int>>& inds_2d_to_3d,
                              std::vector<std::vector<KeyPoint>>& key_points_for_all, std::vector<Point3d>& pts3d)
{
    sfmhip_ctx* ctx = context();
    if (!ctx) { printf("Bundle Adjustment failed.\n"); return; }
    std::vector<int32_t> oc, op; std::vector<double> uv;
    for (size_t img = 0; img < inds_2d_to_3d.size(); ++img)
        for (size_t k = 0; k < inds_2d_to_3d[img].size(); ++k) {
            const int id = inds_2d_to_3d[img][k];
            if (id < 0) continue;
            oc.push_back((int32_t)img); op.push_back(id);
            uv.push_back((double)key_points_for_all[img][k].pt.x); uv.push_back((double)key_points_for_all[img][k].pt.y);   // Point2d observed (NView:1199)
        }
    const int nc = (int)extrinsics.size();
    std::vector<double> ext((size_t)6 * nc);
    for (int c = 0; c < nc; ++c) std::memcpy(&ext[6 * (size_t)c], extrinsics[c].ptr<double>(), 6 * sizeof(double));
    sfm_ba_options o; sfmhip_ba_default_options(&o);
    sfm_ba_summary s; std::memset(&s, 0, sizeof s);
    const std::vector<sfmhip_ctx*>& many = ba_contexts();
    const int rc = many.size() > 1
        ? sfmhip_ba_solve_multi(many.data(), (int)many.size(), intrinsic.ptr<double>(), ext.data(), nc, pts3d.empty() ? nullptr : &pts3d[0].x,
                                (int)pts3d.size(), oc.data(), op.data(), uv.data(), (int)oc.size(), &o, &s)
        : sfmhip_ba_solve(ctx, intrinsic.ptr<double>(), ext.data(), nc, pts3d.empty() ? nullptr : &pts3d[0].x, (int)pts3d.size(),
                          oc.data(), op.data(), uv.data(), (int)oc.size(), &o, &s);
    if (rc != SFMHIP_OK || s.termination == SFMHIP_BA_FAILURE) { printf("Bundle Adjustment failed.\n"); return; }
    for (int c = 0; c < nc; ++c) std::memcpy(extrinsics[c].ptr<double>(), &ext[6 * (size_t)c], 6 * sizeof(double));
    printf("\nBundle Adjustment statistics (approximated RMSE):\n #views: %d\n #residuals: %d\n Initial RMSE(pixel): %g\n"
           " Final   RMSE(pixel): %g\n Time (s): %g\n\n", nc, s.num_residuals,
           std::sqrt(s.initial_cost / (s.num_residuals > 0 ? s.num_residuals : 1)),
           std::sqrt(s.final_cost / (s.num_residuals > 0 ? s.num_residuals : 1)), s.total_time_s);
}

// ---- normals (NView:551-599) ------------------------------------------------------------------------------------
inline int estimate_normals(const std::vector<Point3d>& pts3d, const int K, std::vector<Point3d>& normals)
{
    sfmhip_ctx* ctx = context();
    normals.resize(pts3d.size());
    if (!ctx) return -1;
    if (pts3d.empty()) return 0;
    return sfmhip_estimate_normals(ctx, &pts3d[0].x, (int)pts3d.size(), K, &normals[0].x) == SFMHIP_OK ? 0 : -1;
}

// ---- outputs (NView:186-338) ------------------------------------------------------------------------------------
namespace detail {
// fs::doubleToString [3P] + the "%.16e" of the Windows CRT the reference's files were written with (exact decimal ties
// round half away from zero; glibc rounds them to even)
// fs::floatToString [3P]: "%.8e" of the float (integral values as "3."), same CRT rounding rule
inline std::string etoa(double v, int frac_digits);
inline std::string ftoa(float v)
{
    char buf[64];
    if (std::isnan(v)) return ".Nan";
    if (std::isinf(v)) return v < 0 ? "-.Inf" : ".Inf";
    if (std::fabs(v) < 2147483648.0f && (float)(long long)std::llround((double)v) == v) { snprintf(buf, sizeof buf, "%lld.", (long long)std::llround((double)v)); return buf; }
    return etoa((double)v, 8);
}
inline std::string dtoa(double v)
{
    char buf[512];
    if (std::isnan(v)) return ".Nan";
    if (std::isinf(v)) return v < 0 ? "-.Inf" : ".Inf";
    if (std::fabs(v) < 2147483648.0 && (double)(long long)std::llround(v) == v) { snprintf(buf, sizeof buf, "%lld.", (long long)std::llround(v)); return buf; }
    return etoa(v, 16);
}
// "%.<frac_digits>e" with exact decimal ties rounded half away from zero
inline std::string etoa(double v, int frac_digits)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%.60e", v);          // glibc prints the exact expansion
    std::string s(buf);
    const size_t epos = s.find('e');
    std::string mant = s.substr(0, epos), ex = s.substr(epos + 1);
    const bool neg = mant[0] == '-';
    if (neg) mant = mant.substr(1);
    std::string digits; digits += mant[0]; digits += mant.substr(2);          // d.ddddd -> "ddddd..."
    int e10 = std::atoi(ex.c_str());
    const int nd = frac_digits + 1;
    const bool up = digits[nd] >= '5';                                          // >= half -> away from zero (exact ties included)
    std::string dn = digits.substr(0, nd);
    if (up) {
        int i = nd - 1;
        while (i >= 0) { if (dn[i] == '9') { dn[i] = '0'; --i; } else { dn[i]++; break; } }
        if (i < 0) { dn = "1" + dn.substr(0, nd - 1); ++e10; }
    }
    snprintf(buf, sizeof buf, "%s%c.%se%c%02d", neg ? "-" : "", dn[0], dn.substr(1).c_str(), e10 < 0 ? '-' : '+', e10 < 0 ? -e10 : e10);
    return buf;
}
struct Emitter {
    std::string out, cur;
    void line(const std::string& s) { out += s; out += '\n'; }
    void flow(const std::string& prefix, const std::vector<std::string>& tok, size_t indent)
    {
        cur = prefix + "[";
        bool first = true;
        for (const auto& t : tok) {
            if (!first) cur += ",";
            if (cur.size() + t.size() > 71 && cur.size() > indent) { line(cur); cur.assign(indent, ' '); }
            else cur += " ";
            cur += t; first = false;
        }
        cur += " ]"; line(cur); cur.clear();
    }
};
}  // namespace detail

inline void save_structure(std::string file_name, std::vector<Mat>& rotations, std::vector<Mat>& motions,
                           std::vector<Point3d>& structure, std::vector<Vec3b>& colors)
{
    detail::Emitter e;
    e.line("%YAML:1.0"); e.line("---");
    e.line("Camera Count: " + std::to_string(rotations.size()));
    e.line("Point Count: " + std::to_string(structure.size()));
    auto mats = [&](const char* name, std::vector<Mat>& ms) {
        e.line(std::string(name) + ":");
        for (auto& m : ms) {
            e.line("   - !!opencv-matrix"); e.line("      rows: " + std::to_string(m.rows)); e.line("      cols: " + std::to_string(m.cols)); e.line("      dt: d");
            std::vector<std::string> tok;
            for (int i = 0; i < m.rows * m.cols; ++i) tok.push_back(detail::dtoa(m.ptr<double>()[i]));
            e.flow("      data: ", tok, 10);
        }
    };
    mats("Rotations", rotations); mats("Motions", motions);
    e.line("Points:");
    for (const auto& p : structure) e.flow("   - ", { detail::dtoa(p.x), detail::dtoa(p.y), detail::dtoa(p.z) }, 7);
    e.line("Colors:");
    for (const auto& c : colors) e.flow("   - ", { std::to_string(c[0]), std::to_string(c[1]), std::to_string(c[2]) }, 7);
    std::ofstream f(file_name, std::ios::out | std::ios::binary);
    f << e.out;
}

// TwoViewReconstruct.cpp:313-356: `structure` is the homogeneous 4 x N float matrix; each column is divided by its w in
// float32 (Mat_<float> c; c /= c(3)) and written as a Point3f, i.e. "%.8e" tokens (fs::floatToString [3P])
inline void save_structure(std::string file_name, std::vector<Mat>& rotations, std::vector<Mat>& motions, Mat& structure,
                           std::vector<Vec3b>& colors)
{
    detail::Emitter e;
    e.line("%YAML:1.0"); e.line("---");
    e.line("Camera Count: " + std::to_string(rotations.size()));
    e.line("Point Count: " + std::to_string(structure.cols));
    auto mats = [&](const char* name, std::vector<Mat>& ms) {
        e.line(std::string(name) + ":");
        for (auto& m : ms) {
            e.line("   - !!opencv-matrix"); e.line("      rows: " + std::to_string(m.rows)); e.line("      cols: " + std::to_string(m.cols)); e.line("      dt: d");
            std::vector<std::string> tok;
            for (int i = 0; i < m.rows * m.cols; ++i) tok.push_back(detail::dtoa(m.ptr<double>()[i]));
            e.flow("      data: ", tok, 10);
        }
    };
    mats("Rotations", rotations); mats("Motions", motions);
    e.line("Points:");
    for (int i = 0; i < structure.cols; ++i) {
        const float w = structure.at<float>(3, i);
        // cv::Mat /= scalar multiplies by the reciprocal taken in double and rounds to float [3P]
        const double rw = 1.0 / (double)w;
        const float x = (float)((double)structure.at<float>(0, i) * rw), y = (float)((double)structure.at<float>(1, i) * rw),
                    z = (float)((double)structure.at<float>(2, i) * rw);
        e.flow("   - ", { detail::ftoa(x), detail::ftoa(y), detail::ftoa(z) }, 7);
    }
    e.line("Colors:");
    for (const auto& c : colors) e.flow("   - ", { std::to_string(c[0]), std::to_string(c[1]), std::to_string(c[2]) }, 7);
    std::ofstream f(file_name, std::ios::out | std::ios::binary);
    f << e.out;
}

inline int get_ply_pts3d(const std::vector<Point3d>& pts3d, const std::vector<Point3d>& normals, const std::vector<Vec3b>& colors,
                         std::vector<Pt3DPly>& pts3d_ply)
{
    if (pts3d.size() != normals.size() || pts3d.size() != colors.size()) { printf("[Err]: items size not equal.\n"); return -1; }
    pts3d_ply.resize(pts3d.size());
    for (size_t i = 0; i < pts3d.size(); ++i) {
        Pt3DPly p;
        p.x = (float)pts3d[i].x; p.y = (float)pts3d[i].y; p.z = (float)pts3d[i].z;
        p.nx = (float)normals[i].x; p.ny = (float)normals[i].y; p.nz = (float)normals[i].z;
        p.b = colors[i][0]; p.g = colors[i][1]; p.r = colors[i][2];
        pts3d_ply[i] = p;
    }
    printf("Total %zd 3D points.\n", pts3d.size());
    return 0;
}

inline void write_ply_binary(const std::string& path, const std::vector<Pt3DPly>& points)
{
    auto bad = [](const Pt3DPly& p) { return std::isnan(p.x) || std::isnan(p.y) || std::isnan(p.z) || std::isnan(p.nx) || std::isnan(p.ny) || std::isnan(p.nz); };
    size_t valid = 0;
    for (const auto& p : points) if (!bad(p)) ++valid;
    std::ofstream f(path, std::ios::out | std::ios::binary);
    assert(f.is_open());
    f << "ply\nformat binary_little_endian 1.0\nelement vertex " << valid << "\nproperty float x\nproperty float y\nproperty float z\n"
         "property float nx\nproperty float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
    for (const auto& p : points) {
        if (bad(p)) continue;
        f.write((const char*)&p.x, 4); f.write((const char*)&p.y, 4); f.write((const char*)&p.z, 4);
        f.write((const char*)&p.nx, 4); f.write((const char*)&p.ny, 4); f.write((const char*)&p.nz, 4);
        f.write((const char*)&p.r, 1); f.write((const char*)&p.g, 1); f.write((const char*)&p.b, 1);
    }
}

}  // namespace sfm
