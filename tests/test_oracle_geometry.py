"""CPU: oracle triangulation vs numpy SVD; oracle BA vs finite differences, scipy and a hand-written IRLS step.
(parity unpinned by the reference: these pin the restatement to the published algorithms.)"""
import numpy as np
import pytest
import scipy.optimize

import oracle as orc
from sfm_opencv_amd import synth, api


def test_triangulate_vs_numpy_svd():
    s = synth.two_view_scene(400)
    P1 = orc.projection_matrix(s["K"], s["R1"], s["T1"]); P2 = orc.projection_matrix(s["K"], s["R2"], s["T2"])
    assert np.array_equal(P1, api.projection_matrix(s["K"], s["R1"], s["T1"]))
    xyzw, xyz = orc.triangulate2(P1, P2, s["xy1"], s["xy2"])
    for i in range(0, 400, 7):
        A = np.zeros((4, 4))
        for j, (P, pt) in enumerate(((P1, s["xy1"][i]), (P2, s["xy2"][i]))):
            A[2 * j] = float(pt[0]) * P[2].astype(np.float64) - P[0]; A[2 * j + 1] = float(pt[1]) * P[2].astype(np.float64) - P[1]
        v = np.linalg.svd(A)[2][3].astype(np.float32)
        ref = (v[:3] * np.float32(1.0 / np.float64(v[3]))).astype(np.float64)
        assert np.abs(xyz[i] - ref).max() <= 2e-6 * np.abs(ref).max()      # float32 ulp level
        assert np.abs(np.abs(xyzw[:, i]) - np.abs(v)).max() <= 1e-6
    assert np.array_equal(xyz, xyz.astype(np.float32).astype(np.float64))   # Point3f -> Point3d


@pytest.mark.parametrize("ext", [[0.1, -0.2, 0.3, 0.5, -0.4, 9.0], [1e-9, 0, 0, 0.1, 0.2, 8.0], [0, 0, 0, 0, 0, 5.0]])
def test_reproject_jacobian_vs_central_differences(ext):
    K4 = synth.K_REF; ext = np.array(ext, float); X = np.array([0.3, -0.5, 1.0]); uv = np.array([1800.0, 1300.0])
    r, J = orc.reproject(K4, ext, X, uv)
    assert np.allclose(r, synth.project(K4, ext[None], X[None])[0] - uv, rtol=0, atol=1e-9)
    x0 = np.concatenate([K4, ext, X])
    for k in range(13):
        h = 1e-6 * max(1.0, abs(x0[k])); xp = x0.copy(); xm = x0.copy(); xp[k] += h; xm[k] -= h
        rp, _ = orc.reproject(xp[:4], xp[4:10], xp[10:], uv); rm, _ = orc.reproject(xm[:4], xm[4:10], xm[10:], uv)
        assert np.abs((rp - rm) / (2 * h) - J[:, k]).max() <= 1e-6 * max(1.0, np.abs(J[:, k]).max())


def _residuals(x, sc, ncam):
    K4 = x[:4]; ext = np.vstack([sc["ext0"][:1], x[4:4 + 6 * (ncam - 1)].reshape(-1, 6)]); pts = x[4 + 6 * (ncam - 1):].reshape(-1, 3)
    return (synth.project(K4, ext[sc["obs_cam"]], pts[sc["obs_pt"]]) - sc["obs_uv"]).reshape(-1)


def test_ba_no_loss_reaches_scipy_least_squares_minimum():
    sc = synth.ba_scene(5, 60, outlier_frac=0.0)
    o = orc.ba_default_options(huber_delta=0.0, max_num_iterations=200, function_tolerance=1e-14, parameter_tolerance=1e-14)
    K, ext, pts, s, tr = orc.ba_solve(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], opts=o)
    x0 = np.concatenate([sc["K0"], sc["ext0"][1:].reshape(-1), sc["pts0"].reshape(-1)])
    ref = scipy.optimize.least_squares(_residuals, x0, args=(sc, 5), method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=300)
    assert abs(s["final_cost"] - ref.cost) <= 1e-6 * ref.cost       # both minimise 1/2 sum r^2 (gauge-free cost value)
    assert np.array_equal(ext[0], sc["ext0"][0])
    assert (np.diff(tr["cost"]) <= 1e-9).all()                       # monotone (rejected steps keep the cost)


def test_ba_huber_cost_definition_and_first_lm_step():
    sc = synth.ba_scene(5, 80)
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    S, rhs, cost = orc.ba_reduced_system(*args, 1e4)
    r = (synth.project(sc["K0"], sc["ext0"][sc["obs_cam"]], sc["pts0"][sc["obs_pt"]]) - sc["obs_uv"])
    s = (r ** 2).sum(1)
    rho = np.where(s > 16, 8 * np.sqrt(s) - 16, s)
    assert abs(cost - 0.5 * rho.sum()) <= 1e-12 * cost               # HuberLoss(4), cost = 1/2 sum rho
    assert np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max() and np.linalg.eigvalsh(S).min() > 0
    # one forced LM iteration decreases the cost and keeps camera 0 fixed
    K, ext, pts, su, tr = orc.ba_solve(*args, force_iterations=1)
    assert su["final_cost"] < su["initial_cost"] and abs(su["initial_cost"] - cost) <= 1e-12 * cost
    assert np.array_equal(ext[0], sc["ext0"][0])


def test_ba_reduced_system_is_schur_complement_of_dense_normal_equations():
    # build J'J + D^2 densely from the oracle's own per-observation Jacobians and eliminate the points with numpy
    sc = synth.ba_scene(4, 30, outlier_frac=0.0)
    o = orc.ba_default_options(jacobi_scaling=0, huber_delta=0.0)
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    radius = 50.0
    S, rhs, _ = orc.ba_reduced_system(*args, radius, opts=o)
    nc, npt = 4, 30
    n = 6 * (nc - 1) + 4; m = n + 3 * npt
    J = np.zeros((2 * sc["n_obs"], m)); rr = np.zeros(2 * sc["n_obs"])
    for k in range(sc["n_obs"]):
        c, p = sc["obs_cam"][k], sc["obs_pt"][k]
        r, j = orc.reproject(sc["K0"], sc["ext0"][c], sc["pts0"][p], sc["obs_uv"][k])
        rr[2 * k:2 * k + 2] = r
        J[2 * k:2 * k + 2, 6 * (nc - 1):n] = j[:, :4]
        if c > 0:
            J[2 * k:2 * k + 2, 6 * (c - 1):6 * c] = j[:, 4:10]
        J[2 * k:2 * k + 2, n + 3 * p:n + 3 * p + 3] = j[:, 10:]
    H = J.T @ J
    D2 = np.clip(np.diag(H), 1e-6, 1e32) / radius
    H = H + np.diag(D2); g = J.T @ rr
    A, B, C = H[:n, :n], H[:n, n:], H[n:, n:]
    Sref = A - B @ np.linalg.solve(C, B.T); rref = g[:n] - B @ np.linalg.solve(C, g[n:])
    assert np.abs(S - Sref).max() <= 1e-9 * np.abs(Sref).max()
    assert np.abs(rhs - rref).max() <= 1e-9 * np.abs(rref).max()


def test_ba_observation_order_invariance():
    sc = synth.ba_scene(6, 120)
    rng = np.random.default_rng(1); perm = rng.permutation(sc["n_obs"])
    a = orc.ba_solve(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], force_iterations=5)[3]
    b = orc.ba_solve(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"][perm], sc["obs_pt"][perm], sc["obs_uv"][perm], force_iterations=5)[3]
    assert abs(a["final_cost"] - b["final_cost"]) <= 1e-9 * a["final_cost"]


def test_nview_dlt_oracle_vs_numpy_svd_and_truth():
    """N-view extension: oracle vs numpy.linalg.svd of the stacked normalised system, and vs truth."""
    sc = synth.ba_scene(9, 400, noise_px=0.0, outlier_frac=0.0, perturb=False)
    pts, nv = orc.triangulate_tracks(sc["K_true"], sc["ext_true"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["n_pt"])
    assert np.array_equal(nv, np.bincount(sc["obs_pt"], minlength=sc["n_pt"]))
    assert np.abs(pts - sc["pts_true"]).max() < 1e-9            # noise-free: the exact point
    scn = synth.ba_scene(9, 400, outlier_frac=0.0, perturb=False)    # 0.5 px noise
    pts, _ = orc.triangulate_tracks(scn["K_true"], scn["ext_true"], scn["obs_cam"], scn["obs_pt"], scn["obs_uv"], scn["n_pt"])
    K = scn["K_true"]
    for p in range(0, 400, 37):
        sel = np.nonzero(scn["obs_pt"] == p)[0]
        rows = []
        for k in sel:
            R = synth.angle_axis_to_rotmat(scn["ext_true"][scn["obs_cam"][k], :3]); t = scn["ext_true"][scn["obs_cam"][k], 3:]
            Rt = np.hstack([R, t[:, None]])
            xn = (scn["obs_uv"][k, 0] - K[2]) / K[0]; yn = (scn["obs_uv"][k, 1] - K[3]) / K[1]
            rows += [xn * Rt[2] - Rt[0], yn * Rt[2] - Rt[1]]
        v = np.linalg.svd(np.array(rows))[2][-1]
        assert np.abs(pts[p] - v[:3] / v[3]).max() < 1e-9
    err = orc.reprojection_errors(K, scn["ext_true"], pts, scn["obs_cam"], scn["obs_pt"], scn["obs_uv"])
    uv = synth.project(K, scn["ext_true"][scn["obs_cam"]], pts[scn["obs_pt"]])
    assert np.abs(err - np.linalg.norm(uv - scn["obs_uv"], axis=1)).max() < 1e-9
    assert np.median(err) < 1.0
    # fewer than two observations -> NaN
    one = orc.triangulate_tracks(K, scn["ext_true"], scn["obs_cam"][:1], np.zeros(1, np.int32), scn["obs_uv"][:1], 2)
    assert np.isnan(one[0]).all() and list(one[1]) == [1, 0]
