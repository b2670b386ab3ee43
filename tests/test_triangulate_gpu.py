"""GPU parity of two-view triangulation vs the oracle (float tolerance stated per test)."""
import numpy as np
import pytest

import oracle as orc
from sfm_opencv_amd import synth, api

pytestmark = pytest.mark.gpu


def test_triangulate_matches_oracle(ctx):
    s = synth.two_view_scene(5000)
    P1 = orc.projection_matrix(s["K"], s["R1"], s["T1"]); P2 = orc.projection_matrix(s["K"], s["R2"], s["T2"])
    assert np.array_equal(P1, api.projection_matrix(s["K"], s["R1"], s["T1"]))
    gw, gx = ctx.triangulate2(P1, P2, s["xy1"], s["xy2"])
    ow, ox = orc.triangulate2(P1, P2, s["xy1"], s["xy2"])
    # float32 outputs of an fp64 solve: tolerance 1e-5 relative; observed agreement is ~1 ulp of float32
    rel = np.linalg.norm(gx - ox, axis=1) / np.linalg.norm(ox, axis=1)
    assert rel.max() <= 1e-5
    assert np.median(rel) <= 2e-7
    # values are float32-exact (Point3f -> Point3d, NViewReconstuct.cpp:1155)
    assert np.array_equal(gx, gx.astype(np.float32).astype(np.float64))
    # homogeneous output agrees up to sign
    sgn = np.sign((gw * ow).sum(0))
    assert np.abs(gw * sgn - ow).max() <= 1e-6
    # and with the truth (noise 0.3 px)
    err = np.linalg.norm(gx - s["X"], axis=1) / np.linalg.norm(s["X"], axis=1)
    assert np.median(err) < 5e-3


def test_reconstruct_wrapper_error_and_values(ctx, capsys):
    s = synth.two_view_scene(64, noise_px=0.0)
    ret, xyz = api.reconstruct(s["K"], s["R1"], s["T1"], s["R2"], s["T2"], s["xy1"], s["xy2"], ctx=ctx)
    assert ret == 0
    assert np.abs(xyz - s["X"]).max() < 1e-3
    ret, xyz = api.reconstruct(s["K"], s["R1"], s["T1"], s["R2"], s["T2"], np.zeros((0, 2)), np.zeros((0, 2)), ctx=ctx)
    assert ret == -1 and "[Err]: empty 2d points." in capsys.readouterr().out


def test_fused_gather_triangulation(ctx):
    import torch
    s = synth.two_view_scene(300)
    rng = np.random.default_rng(0)
    kp1 = np.zeros(500, api.KEYPOINT); kp2 = np.zeros(600, api.KEYPOINT)
    qi = rng.permutation(500)[:300]; ti = rng.permutation(600)[:300]
    kp1["x"][qi] = s["xy1"][:, 0]; kp1["y"][qi] = s["xy1"][:, 1]
    kp2["x"][ti] = s["xy2"][:, 0]; kp2["y"][ti] = s["xy2"][:, 1]
    m = np.zeros(300, api.DMATCH); m["queryIdx"] = qi; m["trainIdx"] = ti
    a, b = api.get_matched_points(kp1, kp2, m)
    assert np.array_equal(a, s["xy1"]) and np.array_equal(b, s["xy2"])
    P1 = orc.projection_matrix(s["K"], s["R1"], s["T1"]); P2 = orc.projection_matrix(s["K"], s["R2"], s["T2"])
    d_kp1 = torch.from_numpy(kp1.view(np.uint8).reshape(-1)).cuda(); d_kp2 = torch.from_numpy(kp2.view(np.uint8).reshape(-1)).cuda()
    d_m = torch.from_numpy(m.view(np.uint8).reshape(-1)).cuda()
    xyz = torch.zeros((300, 3), dtype=torch.float64, device="cuda")
    ctx.triangulate2_matches_dev(P1, P2, d_kp1, d_kp2, d_m, 300, None, xyz)
    ctx.synchronize()
    _, ref = ctx.triangulate2(P1, P2, s["xy1"], s["xy2"])
    assert np.array_equal(xyz.cpu().numpy(), ref)


def test_nview_tracks_and_reprojection_errors_match_oracle(ctx):
    """N-view DLT extension (sfmhip_triangulate_tracks) and per-observation pixel errors vs the oracle; fp64, 1e-10."""
    sc = synth.ba_scene(40, 20000, perturb=False)
    rng = np.random.default_rng(5)
    perm = rng.permutation(sc["n_obs"])                          # the entry point must not depend on the observation order
    args = (sc["K_true"], sc["ext_true"], sc["obs_cam"][perm], sc["obs_pt"][perm], sc["obs_uv"][perm])
    pts, nv = ctx.triangulate_tracks(*args, sc["n_pt"])
    opts, onv = orc.triangulate_tracks(*args, sc["n_pt"])
    assert np.array_equal(nv, onv)
    scale = np.abs(opts).max()
    assert np.abs(pts - opts).max() <= 1e-10 * scale
    # with 0.5 px noise and 2 % outlier pixels most tracks land within a few 1e-3 of the truth
    assert np.median(np.linalg.norm(pts - sc["pts_true"], axis=1)) < 5e-3
    err = ctx.reprojection_errors(sc["K_true"], sc["ext_true"], pts, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    oerr = orc.reprojection_errors(sc["K_true"], sc["ext_true"], pts, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    assert np.abs(err - oerr).max() <= 1e-9 * max(oerr.max(), 1.0)
    # the gross outliers (2 % of the pixels were moved by up to 50 px) stand out
    assert 0.01 < (err > 5.0).mean() < 0.10      # an outlier pixel also drags the other observations of its track
    # ragged input: a point with a single observation and one with none
    p2, n2 = ctx.triangulate_tracks(sc["K_true"], sc["ext_true"], sc["obs_cam"][:1], np.zeros(1, np.int32), sc["obs_uv"][:1], 2)
    assert np.isnan(p2).all() and list(n2) == [1, 0]


def test_refine_structure_filters_outliers_and_retriangulates(ctx):
    """Extension: BA -> drop observations above 4 px -> N-view re-triangulation -> BA.  On a scene with 2 % gross
    outliers (+-50 px) the filter removes (nearly) exactly those observations and the second BA reaches the noise floor."""
    sc = synth.ba_scene(14, 2500, outlier_frac=0.02)
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    K, ext, pts, s = ctx.ba_solve(*args)
    clean_uv = synth.project(sc["K_true"], sc["ext_true"][sc["obs_cam"]], sc["pts_true"][sc["obs_pt"]])
    is_out = np.linalg.norm(sc["obs_uv"] - clean_uv, axis=1) > 6.0
    assert 0.01 < is_out.mean() < 0.03
    K2, ext2, pts2, ids, keep, s2 = api.refine_structure(K, ext, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], pts, max_px=4.0, ctx=ctx)
    assert (~keep[is_out]).mean() > 0.97                      # the gross outliers are gone ...
    assert keep[~is_out].mean() > 0.98                        # ... and hardly anything else
    rm1 = np.sqrt(s["final_cost"] / s["num_residuals"]); rm2 = np.sqrt(s2["final_cost"] / s2["num_residuals"])
    assert rm2 < rm1 and rm2 < 0.4                            # 0.5 px noise per axis: cost = sum r^2 / 2 -> rmse ~ 0.35
    assert len(ids) > 0.98 * sc["n_pt"] and pts2.shape == (len(ids), 3)
    e1 = np.linalg.norm(pts[ids] - sc["pts_true"][ids], axis=1); e2 = np.linalg.norm(pts2 - sc["pts_true"][ids], axis=1)
    assert np.median(e2) <= np.median(e1) * 1.05
