// sfm_pipeline.hpp -- what the reference's main() does around the hot path (NViewReconstuct.cpp:1334-1524), on the host
// mirror of sfm_ops.hpp / sfm_geometry.hpp.  Used by the two driver programs NViewReconstruct.cpp / TwoViewReconstruct.cpp.
//
// Input: the reference reads a directory of .jpg files and runs cv::AKAZE on them (NView:785-848); image decoding and
// feature extraction are out of this build's scope, so the drivers start one step later, from a FEATURES
// FILE holding exactly what extract_features() leaves behind -- key points, descriptor matrix and BGR colours per image --
// plus K (the reference hard-codes it, NView:1353-1356; its TODO at 1358 asks for it to become an input).
//
//   features file, little-endian:
//     char magic[8] = "SFMFEAT1";  int32 n_img;  double K[9];  int32 has_poses;
//     per image:  int32 n_kp, desc_type (0 = CV_8U, 5 = CV_32F, 100 = CV_32F rows stored as one byte per value: integer-valued
//                 SIFT descriptors), desc_cols;
//                 sfm_keypoint kp[n_kp];  descriptor rows;  uint8 bgr[n_kp][3];
//                 if has_poses: double R[9], T[3]      (world -> camera; used with --poses-from-file only)
#pragma once
#include <dirent.h>
#include <sys/stat.h>

#include <iostream>

#include "sfm_features.hpp"
#include "sfm_geometry.hpp"

namespace sfm {

struct Features {
    Mat K;
    std::vector<std::vector<KeyPoint>> key_points_for_all;
    std::vector<Mat> descriptor_for_all;
    std::vector<std::vector<Vec3b>> colors_for_all;
    std::vector<Mat> file_rotations, file_motions;      // optional
    bool has_poses = false;
};

inline bool read_features(const std::string& path, Features& f)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) { printf("[Err]: cannot open features file %s\n", path.c_str()); return false; }
    char magic[8];
    in.read(magic, 8);
    if (!in || std::memcmp(magic, "SFMFEAT1", 8) != 0) { printf("[Err]: %s is not a features file.\n", path.c_str()); return false; }
    int32_t n_img = 0, has_poses = 0;
    in.read((char*)&n_img, 4);
    f.K = Mat(3, 3, CV_64F);
    in.read((char*)f.K.ptr<double>(), 72);
    in.read((char*)&has_poses, 4);
    f.has_poses = has_poses != 0;
    for (int i = 0; i < n_img && in; ++i) {
        int32_t n_kp = 0, type = 0, cols = 0;
        in.read((char*)&n_kp, 4); in.read((char*)&type, 4); in.read((char*)&cols, 4);
        if (!in || n_kp < 0 || cols < 0 || (type != CV_8U && type != CV_32F && type != 100)) { printf("[Err]: corrupt features file.\n"); return false; }
        std::vector<KeyPoint> kp((size_t)n_kp);
        in.read((char*)kp.data(), (std::streamsize)sizeof(KeyPoint) * n_kp);
        Mat d(n_kp, cols, type == 100 ? (int)CV_32F : type);
        if (type == 100) {
            std::vector<uint8_t> packed((size_t)n_kp * cols);
            in.read((char*)packed.data(), (std::streamsize)packed.size());
            for (size_t q = 0; q < packed.size(); ++q) d.ptr<float>()[q] = (float)packed[q];
        } else in.read((char*)d.buf.data(), (std::streamsize)d.buf.size());
        std::vector<Vec3b> col((size_t)n_kp);
        in.read((char*)col.data(), (std::streamsize)3 * n_kp);
        Mat R(3, 3, CV_64F), T(3, 1, CV_64F);
        if (f.has_poses) { in.read((char*)R.ptr<double>(), 72); in.read((char*)T.ptr<double>(), 24); }
        if (!in) { printf("[Err]: truncated features file.\n"); return false; }
        printf("Extracting features for image %d...\n", i);
        if (kp.size() <= 10) continue;                       // extract_features drops such images (NView:820-823)
        printf("%zd 2D feature point detected.\n", kp.size());
        f.key_points_for_all.push_back(std::move(kp)); f.descriptor_for_all.push_back(std::move(d)); f.colors_for_all.push_back(std::move(col));
        f.file_rotations.push_back(R); f.file_motions.push_back(T);
    }
    return (bool)in;
}

// the same file from extracted features (CV_32F descriptors whose values are all integers in [0, 255] are stored packed)
inline bool write_features(const std::string& path, const Features& f)
{
    std::ofstream out(path, std::ios::binary);
    if (!out) return false;
    out.write("SFMFEAT1", 8);
    const int32_t n_img = (int32_t)f.key_points_for_all.size(), has_poses = 0;
    out.write((const char*)&n_img, 4); out.write((const char*)f.K.ptr<double>(), 72); out.write((const char*)&has_poses, 4);
    for (int i = 0; i < n_img; ++i) {
        const Mat& d = f.descriptor_for_all[i];
        bool packable = d.type == CV_32F;
        if (packable) for (size_t q = 0; q < (size_t)d.rows * d.cols && packable; ++q) { const float v = d.ptr<float>()[q]; packable = v >= 0 && v <= 255 && v == (float)(int)v; }
        const int32_t n_kp = d.rows, type = packable ? 100 : d.type, cols = d.cols;
        out.write((const char*)&n_kp, 4); out.write((const char*)&type, 4); out.write((const char*)&cols, 4);
        out.write((const char*)f.key_points_for_all[i].data(), (std::streamsize)sizeof(KeyPoint) * n_kp);
        if (packable) { std::vector<uint8_t> p((size_t)n_kp * cols); for (size_t q = 0; q < p.size(); ++q) p[q] = (uint8_t)d.ptr<float>()[q]; out.write((const char*)p.data(), (std::streamsize)p.size()); }
        else out.write((const char*)d.buf.data(), (std::streamsize)d.buf.size());
        out.write((const char*)f.colors_for_all[i].data(), (std::streamsize)3 * n_kp);
    }
    return (bool)out;
}

// get_files_format (NView:1303-1330) for a POSIX directory: regular files whose name ends in `format`, sorted by name
// (the reference relies on the directory order of _findfirst, alphabetical on NTFS; it does not recurse into this use)
inline int get_files_format(const std::string& path, const std::string& format, std::vector<std::string>& files)
{
    DIR* d = opendir(path.c_str());
    if (!d) return 0;
    std::vector<std::string> names;
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.size() <= format.size()) continue;
        std::string tail = n.substr(n.size() - format.size());
        for (auto& ch : tail) ch = (char)tolower((unsigned char)ch);       // the reference matches ".jpg" only (NView:1344); its own crazyhorse files are ".JPG"
        if (tail == format) names.push_back(n);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    for (const auto& n : names) files.push_back(path + "/" + n);
    return (int)files.size();
}

// K: the reference hard-codes it (NView:1353-1356) and asks for it to become an input (TODO at 1358): `<dir>/K.txt` with
// "fx fy cx cy" if present, else the reference's matrix
inline Mat load_K(const std::string& dir)
{
    Mat K = Mat::eye3();
    double v[4] = { 2826.561, 2826.519, 1835.259, 1370.103 };
    std::ifstream f(dir + "/K.txt");
    if (f) { double t[4]; if (f >> t[0] >> t[1] >> t[2] >> t[3]) for (int i = 0; i < 4; ++i) v[i] = t[i]; }
    K.at<double>(0, 0) = v[0]; K.at<double>(1, 1) = v[1]; K.at<double>(0, 2) = v[2]; K.at<double>(1, 2) = v[3];
    return K;
}

inline bool is_directory(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }

inline std::vector<uint8_t> mask_vector(const Mat& mask) { return std::vector<uint8_t>(mask.ptr<uint8_t>(), mask.ptr<uint8_t>() + (size_t)mask.rows * mask.cols); }

inline void print_mat(const char* name, const Mat& m)
{
    printf("%s:\n[", name);
    for (int r = 0; r < m.rows; ++r) { for (int c = 0; c < m.cols; ++c) printf("%s%.17g", c ? ", " : "", m.at<double>(r, c)); printf(r + 1 < m.rows ? ";\n " : "]\n"); }
}

// init_structure (NView:916-987).  given_R / given_T non-null: the pose of frame 1 comes from the caller (poses-from-file
// mode) and every match of the first pair is kept; otherwise find_transform + its inlier mask, as the reference.
inline int init_structure(const Mat& K, const std::vector<std::vector<KeyPoint>>& key_points_for_all,
                          const std::vector<std::vector<Vec3b>>& colors_for_all, const std::vector<std::vector<DMatch>>& matches_for_all,
                          std::vector<Point3d>& structure, std::vector<std::vector<int>>& correspond_struct_idx, std::vector<Vec3b>& colors,
                          std::vector<Mat>& rotations, std::vector<Mat>& motions, const Mat* given_R = nullptr, const Mat* given_T = nullptr)
{
    std::vector<Point2f> pts2d_1, pts2d_2;
    std::vector<Vec3b> c2;
    Mat R, T, mask;
    get_matched_points(key_points_for_all[0], key_points_for_all[1], matches_for_all[0], pts2d_1, pts2d_2);
    get_matched_colors(colors_for_all[0], colors_for_all[1], matches_for_all[0], colors, c2);
    std::vector<uint8_t> mv;
    if (given_R && given_T) { R = *given_R; T = *given_T; mv.assign(pts2d_1.size(), 1); }
    else {
        find_transform(K, pts2d_1, pts2d_2, R, T, mask);       // the reference ignores its verdict too (NView:935)
        if (R.empty() || T.empty()) { printf("[Err]: no transform between the first two frames.\n"); return -1; }
        mv = mask_vector(mask);
    }
    maskout_2d_pts_pair(mv, pts2d_1, pts2d_2);
    maskout_colors(mv, colors);
    Mat R0 = Mat::eye3(), T0(3, 1, CV_64F);
    const int ret = reconstruct(K, R0, T0, R, T, pts2d_1, pts2d_2, structure);
    if (ret < 0) return ret;
    rotations = { R0, R };
    motions = { T0, T };
    init_correspondence(key_points_for_all, matches_for_all[0], mv, correspond_struct_idx);
    return 0;
}

// Extension (NOT reference behaviour: the reference triangulates a point once, from the pair that created it,
// and never filters the matches of later pairs, NView:1428-1453): after a bundle adjustment, drop the observations whose
// reprojection error exceeds `max_px`, re-triangulate every track from ALL its remaining observations (N-view DLT on the
// GPU), drop tracks left with fewer than two observations or behind a camera, and adjust again.  Returns the number of
// observations removed, -1 on error.  correspond_struct_idx, structure and colors are compacted in place.
inline int refine_structure(Mat& intrinsic, std::vector<Mat>& extrinsics, std::vector<std::vector<int>>& correspond_struct_idx,
                            std::vector<std::vector<KeyPoint>>& key_points_for_all, std::vector<Point3d>& structure,
                            std::vector<Vec3b>& colors, double max_px = 4.0)
{
    sfmhip_ctx* ctx = context();
    if (!ctx || structure.empty()) return -1;
    const int nc = (int)extrinsics.size(), np = (int)structure.size();
    std::vector<double> ext((size_t)6 * nc);
    for (int c = 0; c < nc; ++c) std::memcpy(&ext[6 * (size_t)c], extrinsics[c].ptr<double>(), 6 * sizeof(double));
    std::vector<int32_t> oc, op; std::vector<double> uv; std::vector<std::pair<int, int>> where;
    auto gather = [&]() {
        oc.clear(); op.clear(); uv.clear(); where.clear();
        for (int img = 0; img < nc && img < (int)correspond_struct_idx.size(); ++img)
            for (size_t k = 0; k < correspond_struct_idx[img].size(); ++k) {
                const int id = correspond_struct_idx[img][k];
                if (id < 0) continue;
                oc.push_back(img); op.push_back(id); where.push_back({ img, (int)k });
                uv.push_back((double)key_points_for_all[img][k].pt.x); uv.push_back((double)key_points_for_all[img][k].pt.y);
            }
    };
    gather();
    std::vector<double> err(oc.size());
    if (sfmhip_reprojection_errors(ctx, intrinsic.ptr<double>(), ext.data(), nc, &structure[0].x, np, oc.data(), op.data(), uv.data(), (int)oc.size(), err.data()) != SFMHIP_OK)
        { printf("[Err]: refine_structure: %s\n", sfmhip_last_error(ctx)); return -1; }
    int removed = 0;
    for (size_t q = 0; q < err.size(); ++q)
        if (!(err[q] <= max_px)) { correspond_struct_idx[where[q].first][where[q].second] = -1; ++removed; }
    gather();
    std::vector<double> pts((size_t)3 * np); std::vector<int32_t> nviews((size_t)np);
    if (sfmhip_triangulate_tracks(ctx, intrinsic.ptr<double>(), ext.data(), nc, oc.data(), op.data(), uv.data(), (int)oc.size(), np, pts.data(), nviews.data()) != SFMHIP_OK)
        { printf("[Err]: refine_structure: %s\n", sfmhip_last_error(ctx)); return -1; }
    // keep tracks with >= 2 views and a finite position; compact
    std::vector<int> new_id((size_t)np, -1);
    std::vector<Point3d> kept; std::vector<Vec3b> kept_col;
    for (int p = 0; p < np; ++p) {
        const bool ok = nviews[p] >= 2 && std::isfinite(pts[3 * (size_t)p]) && std::isfinite(pts[3 * (size_t)p + 1]) && std::isfinite(pts[3 * (size_t)p + 2]);
        if (!ok) continue;
        new_id[p] = (int)kept.size();
        kept.emplace_back(pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2]);
        if ((size_t)p < colors.size()) kept_col.push_back(colors[p]);
    }
    for (auto& v : correspond_struct_idx) for (int& id : v) if (id >= 0) { if (new_id[id] < 0) ++removed; id = new_id[id]; }
    printf("refine_structure: %d observations dropped (> %.2f px or orphaned), %zu of %d points kept.\n", removed, max_px, kept.size(), np);
    structure.swap(kept); colors.swap(kept_col);
    bundle_adjustment(intrinsic, extrinsics, correspond_struct_idx, key_points_for_all, structure);
    return removed;
}

struct PipelineOptions {
    std::string out_dir = "../Viewer";     // the reference's relative output paths (NView:1458, 1505, 1511)
    bool poses_from_file = false;          // skip find_transform / solvePnPRansac, take R, T of every frame from the features file
    bool write_back_poses = false;         // structure_ba.yml with the optimised poses (the reference writes the pre-BA ones)
    bool print_offsets = true;             // the per-point "Point3d i offset" lines (NView:1494-1498)
    int max_features = 0;                  // directory input: keep the strongest N key points per image (0: all, like the reference)
    std::string save_features;             // directory input: also write the extracted features to this file
    bool features_only = false;            // ... and stop there
    bool akaze = false;                    // directory input: AKAZE + M-LDB rows (the reference's live extractor, NView:797) or SIFT (its commented twin,
                                           // TwoView:112).  driver_main sets the default per program: NViewReconstruct AKAZE, TwoViewReconstruct SIFT
    double refine_px = 0.0;                // > 0: after BA, refine_structure(max_px) + a second BA (extension, not reference behaviour)
};

// main() of NViewReconstuct.cpp from "match_features_for_all" on (NView:1369-1517)
inline int run_nview(Features& f, const PipelineOptions& opt)
{
    const Mat& K = f.K;
    auto& kpts_for_all = f.key_points_for_all; auto& colors_for_all = f.colors_for_all;
    std::vector<std::vector<DMatch>> matches_for_all;
    match_features_for_all(f.descriptor_for_all, matches_for_all);
    if (matches_for_all.empty()) { printf("[Err]: fewer than two usable images.\n"); return -1; }

    std::vector<Point3d> pts3d;
    std::vector<std::vector<int>> inds_2d_to_3d;
    std::vector<Vec3b> colors;
    std::vector<Mat> rotations, translations;

    printf("\nConstruct from the first two frames...\n");
    const int ret = init_structure(K, kpts_for_all, colors_for_all, matches_for_all, pts3d, inds_2d_to_3d, colors, rotations, translations,
                                   opt.poses_from_file ? &f.file_rotations[1] : nullptr, opt.poses_from_file ? &f.file_motions[1] : nullptr);
    if (ret < 0) return ret;
    if (opt.poses_from_file) { rotations[0] = f.file_rotations[0]; translations[0] = f.file_motions[0]; }

    printf("\nIncremental SFM...\n");
    for (int i = 1; i < (int)matches_for_all.size(); ++i) {
        std::vector<Point3f> obj_pts;
        std::vector<Point2f> img_pts;
        Mat r, R, T;
        get_obj_pts_and_img_pts(matches_for_all[i], inds_2d_to_3d[i], pts3d, kpts_for_all[i + 1], obj_pts, img_pts);
        if (opt.poses_from_file) { R = f.file_rotations[i + 1]; T = f.file_motions[i + 1]; }
        else {
            if (obj_pts.size() < 4 || img_pts.size() < 4) { printf("[Warning]: too few 3D-2D point pairs for frame %d.\n", i); continue; }
            if (!solvePnPRansac(obj_pts, img_pts, K, r, T)) { printf("[Warning]: no pose for frame %d.\n", i); continue; }
            Rodrigues_vec(r, R);
        }
        print_mat("R", R); print_mat("T", T);
        rotations.push_back(R);
        translations.push_back(T);
        if ((int)rotations.size() <= i + 1) {
            // after a skipped frame the reference's rotations[i] no longer refers to frame i and indexing
            // runs off the end; stop here instead of reading out of bounds
            printf("[Err]: frame %d has no pose (an earlier frame was skipped).\n", i); return -1;
        }
        std::vector<Point2f> pts2d_1, pts2d_2;
        std::vector<Vec3b> colors_1, colors_2;
        get_matched_points(kpts_for_all[i], kpts_for_all[i + 1], matches_for_all[i], pts2d_1, pts2d_2);
        get_matched_colors(colors_for_all[i], colors_for_all[i + 1], matches_for_all[i], colors_1, colors_2);
        std::vector<Point3d> next_structure;
        reconstruct(K, rotations[i], translations[i], R, T, pts2d_1, pts2d_2, next_structure);
        printf("Frame %d reconstructed.\n", i);
        fuse_structure(matches_for_all[i], inds_2d_to_3d[i], inds_2d_to_3d[i + 1], pts3d, next_structure, colors, colors_1);
        printf("Frame %d point cloud fused, total %d points now.\n", i, (int)pts3d.size());
    }

    save_structure(opt.out_dir + "/structure.yml", rotations, translations, pts3d, colors);

    printf("\nBundle adjustment fo SFM...\n");
    Mat intrinsic(4, 1, CV_64F);
    intrinsic.at<double>(0) = K.at<double>(0, 0); intrinsic.at<double>(1) = K.at<double>(1, 1);
    intrinsic.at<double>(2) = K.at<double>(0, 2); intrinsic.at<double>(3) = K.at<double>(1, 2);
    print_mat("intrinsic", intrinsic);
    std::vector<Mat> extrinsics;
    for (size_t i = 0; i < rotations.size(); ++i) {
        Mat extrinsic(6, 1, CV_64F), r;
        Rodrigues(rotations[i], r);
        for (int k = 0; k < 3; ++k) { extrinsic.at<double>(k) = r.at<double>(k); extrinsic.at<double>(3 + k) = translations[i].at<double>(k); }
        extrinsics.push_back(extrinsic);
    }
    // frames that never got a pose carry no camera: their index rows must not reach the solver
    std::vector<std::vector<int>> inds_ba(inds_2d_to_3d.begin(), inds_2d_to_3d.begin() + std::min(inds_2d_to_3d.size(), extrinsics.size()));
    std::vector<std::vector<KeyPoint>> kpts_ba(kpts_for_all.begin(), kpts_for_all.begin() + inds_ba.size());
    auto pts3d_old = pts3d;
    bundle_adjustment(intrinsic, extrinsics, inds_ba, kpts_ba, pts3d);
    if (opt.refine_px > 0.0) {
        pts3d_old.clear();
        refine_structure(intrinsic, extrinsics, inds_ba, kpts_ba, pts3d, colors, opt.refine_px);
        for (size_t i = 0; i < inds_ba.size(); ++i) inds_2d_to_3d[i] = inds_ba[i];
    }
    if (opt.print_offsets && pts3d_old.size() == pts3d.size())
        for (size_t i = 0; i < pts3d.size(); ++i)
            printf("Point3d %zu offset: [%.17g, %.17g, %.17g]\n", i, pts3d[i].x - pts3d_old[i].x, pts3d[i].y - pts3d_old[i].y, pts3d[i].z - pts3d_old[i].z);

    std::vector<Point3d> normals(pts3d.size());
    estimate_normals(pts3d, 10, normals);

    if (opt.write_back_poses)
        for (size_t i = 0; i < extrinsics.size(); ++i) {
            Mat r(3, 1, CV_64F);
            for (int k = 0; k < 3; ++k) { r.at<double>(k) = extrinsics[i].at<double>(k); translations[i].at<double>(k) = extrinsics[i].at<double>(3 + k); }
            Rodrigues_vec(r, rotations[i]);
        }
    save_structure(opt.out_dir + "/structure_ba.yml", rotations, translations, pts3d, colors);
    printf("structure_ba.yml saved.\n");
    printf("Saving structure to ply...\n");
    std::vector<Pt3DPly> pts3dply;
    get_ply_pts3d(pts3d, normals, colors, pts3dply);
    write_ply_binary(opt.out_dir + "/structure_ba.ply", pts3dply);
    printf("%s/structure_ba.ply saved.\n", opt.out_dir.c_str());
    std::cout << "Save structure done." << std::endl;
    return 0;
}

// main() of TwoViewReconstruct.cpp (lines 50-97): two images, L2 matching of SIFT rows, essential matrix, homogeneous
// triangulation, structure.yml with float points
inline int run_twoview(Features& f, const PipelineOptions& opt)
{
    if (f.descriptor_for_all.size() < 2) { printf("[Err]: two images needed.\n"); return -1; }
    std::vector<DMatch> matches;
    match_features(f.descriptor_for_all[0], f.descriptor_for_all[1], matches);
    std::vector<Point2f> p1, p2;
    std::vector<Vec3b> c1, c2;
    Mat R, T, mask;
    get_matched_points(f.key_points_for_all[0], f.key_points_for_all[1], matches, p1, p2);
    get_matched_colors(f.colors_for_all[0], f.colors_for_all[1], matches, c1, c2);
    std::vector<uint8_t> mv;
    if (opt.poses_from_file) { R = f.file_rotations[1]; T = f.file_motions[1]; mv.assign(p1.size(), 1); }
    else {
        find_transform(f.K, p1, p2, R, T, mask);
        if (R.empty() || T.empty()) { printf("[Err]: no transform between the two frames.\n"); return -1; }
        mv = mask_vector(mask);
    }
    Mat structure;                                           // 4 x N, homogeneous
    maskout_2d_pts_pair(mv, p1, p2);
    reconstruct(f.K, R, T, p1, p2, structure);
    std::vector<Mat> rotations = { Mat::eye3(), R };
    std::vector<Mat> motions = { Mat(3, 1, CV_64F), T };
    maskout_colors(mv, c1);
    save_structure(opt.out_dir + "/structure.yml", rotations, motions, structure, c1);
    std::cout << "successful!!!" << std::endl;
    return 0;
}

inline int driver_main(int argc, char** argv, bool nview)
{
    if (argc < 2 || std::string(argv[1]).empty()) {
        printf("[Warning]: empty dataset path.\nusage: %s <image directory (.jpg | .ppm | .pgm, K.txt beside them) | features file> [output dir = ../Viewer] [--poses-from-file] [--write-back-poses] [--quiet] [--akaze | --sift] [--gpus=DEV,DEV,...] [--max-features=N] [--save-features=FILE] [--features-only] [--refine[=PX]]\n", argv[0]);
        return 0;
    }
    PipelineOptions opt;
    // each program runs its reference's extractor + matcher unless told otherwise: NViewReconstuct.cpp is cv::AKAZE::create() +
    // BFMatcher(NORM_HAMMING2) (NView:797, 876); TwoViewReconstruct.cpp is cv::SIFT::create(0, 3, 0.04, 10) + NORM_L2 (TwoView:112, 159)
    opt.akaze = nview;
    int positional = 0;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--poses-from-file") opt.poses_from_file = true;
        else if (a == "--write-back-poses") opt.write_back_poses = true;
        else if (a == "--quiet") opt.print_offsets = false;
        else if (a == "--features-only") opt.features_only = true;
        else if (a.rfind("--gpus=", 0) == 0) {          // bundle adjustment over several devices of this process: --gpus=0,1,2,3
            std::vector<int> devs;
            for (size_t at = 7; at < a.size();) { size_t e = a.find(',', at); if (e == std::string::npos) e = a.size(); devs.push_back(std::atoi(a.substr(at, e - at).c_str())); at = e + 1; }
            if (devs.empty() || !set_ba_devices(devs)) return 1;         // (matching spreads its pairs over the same contexts)
        }
        else if (a == "--akaze") opt.akaze = true;
        else if (a == "--sift") opt.akaze = false;
        else if (a.rfind("--max-features=", 0) == 0) opt.max_features = std::atoi(a.c_str() + 15);
        else if (a == "--refine") opt.refine_px = 4.0;
        else if (a.rfind("--refine=", 0) == 0) opt.refine_px = std::atof(a.c_str() + 9);
        else if (a.rfind("--save-features=", 0) == 0) opt.save_features = a.substr(16);
        else if (positional++ == 0) opt.out_dir = a;
    }
    Features f;
    if (is_directory(argv[1])) {
        // the reference's own entry: a directory of images (.jpg / .jpeg through sfm_jpeg.hpp, else binary .ppm / .pgm)
        std::vector<std::string> img_names;
        int n_files = get_files_format(argv[1], ".jpg", img_names);
        if (n_files == 0) n_files = get_files_format(argv[1], ".jpeg", img_names);
        if (n_files == 0) n_files = get_files_format(argv[1], ".ppm", img_names);
        if (n_files == 0) n_files = get_files_format(argv[1], ".pgm", img_names);
        printf("Total %d image files.\n", n_files);
        f.K = load_K(argv[1]);
        extract_features(img_names, f.key_points_for_all, f.descriptor_for_all, f.colors_for_all, opt.max_features, opt.akaze ? EXTRACT_AKAZE : EXTRACT_SIFT);
        f.file_rotations.assign(f.key_points_for_all.size(), Mat()); f.file_motions.assign(f.key_points_for_all.size(), Mat());
        if (!opt.save_features.empty() && !write_features(opt.save_features, f)) printf("[Warning]: cannot write %s\n", opt.save_features.c_str());
    } else {
        if (!read_features(argv[1], f)) return 1;
        printf("Total %d image files.\n", (int)f.key_points_for_all.size());
    }
    if (opt.features_only) return 0;
    if (opt.poses_from_file && !f.has_poses) { printf("[Err]: the features file holds no poses.\n"); return 1; }
    if (f.key_points_for_all.size() < 2) { printf("[Err]: fewer than two usable images.\n"); return 1; }
    if (!context()) return 1;                               // no GPU: fail loudly, there is no CPU path
    const int rc = nview ? run_nview(f, opt) : run_twoview(f, opt);
    return rc == 0 ? 0 : 1;
}

}  // namespace sfm
