"""Parity at BASELINE.json's full sizes (C3's 50-camera / 80k-point BA scene, C4 per-pair 5000 x 5000 x 128 and its
200-camera / 300k-point scene, C5's 10k x 10k pair and its 1000-camera / 2M-point / 8M-observation scene, normals of a
300k-point cloud) through size-independent properties plus oracle checks where the oracle finishes in about a minute.
Everything goes through the C-ABI."""
import time

import numpy as np
import pytest
import torch

import oracle as orc
from sfm_opencv_amd import synth, api
from sfm_opencv_amd import dist as sdist

pytestmark = pytest.mark.gpu


def test_c4_pair_self_match_permutation_and_oracle_slice(ctx):
    d = synth.sift_descriptor_chain(2, 5000, seed=4242)
    a, b = d[0], d[1]
    # (1) a set matched against itself: every row finds itself at distance 0 first (ties -> lower index: rows are distinct)
    gi, gd = ctx.knn2_l2(a, a)
    uniq = np.unique(a, axis=0).shape[0] == a.shape[0]
    if uniq:
        assert np.array_equal(gi[:, 0], np.arange(5000)) and not gd[:, 0].any()
    assert (gd[:, 1] >= gd[:, 0]).all()
    # (2) permuting the train rows permutes the answer (distances bit-identical) -- rows are distinct, so no tie order issue
    rng = np.random.default_rng(9)
    perm = rng.permutation(5000)
    gi1, gd1 = ctx.knn2_l2(a, b)
    gi2, gd2 = ctx.knn2_l2(a, b[perm])
    assert np.array_equal(gd1.view(np.uint32), gd2.view(np.uint32))
    clean = gd1[:, 0] != gd1[:, 1]                       # a tie between best and runner-up may swap under a permutation
    assert np.array_equal(perm[gi2[clean]], gi1[clean])
    # (3) oracle on a slice of the queries against all 5000 trains: bit-exact
    rows = rng.choice(5000, 300, replace=False)
    oi, od = orc.knn2_l2(a[rows], b)
    assert np.array_equal(gi1[rows], oi) and np.array_equal(gd1[rows].view(np.uint32), od.view(np.uint32))
    # (4) the ratio tail of the full pair equals the host tail applied to the device kNN
    m = api.match_features(a, b, ctx=ctx)
    ref = api.ratio_filter(gi1, gd1)
    assert np.array_equal(m["queryIdx"], ref["queryIdx"]) and np.array_equal(m["trainIdx"], ref["trainIdx"])


def test_10k_distance_matrix_properties(ctx):
    d = synth.sift_descriptor_chain(2, 10000, seed=777)
    q = torch.from_numpy(d[0]).cuda(); t = torch.from_numpy(d[1]).cuda()
    qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
    out = torch.empty((10000, 10000), dtype=torch.float32, device="cuda")
    out_t = torch.empty((10000, 10000), dtype=torch.float32, device="cuda")
    ctx.l2_distance_matrix_dev(qs, ts, out)
    ctx.l2_distance_matrix_dev(ts, qs, out_t)
    ctx.synchronize()
    # symmetry: d(q_i, t_j) computed from either side is the same float
    assert torch.equal(out, out_t.t())
    # row minima agree with the fused kNN kernel (same integers -> same floats)
    gi, gd = ctx.knn2_l2(d[0], d[1])
    mn = out.min(dim=1).values.cpu().numpy()
    assert np.array_equal(mn.view(np.uint32), gd[:, 0].view(np.uint32))
    # oracle spot check: 64 random rows, bit-exact
    rows = np.random.default_rng(1).choice(10000, 64, replace=False)
    ref = orc.l2_distance_matrix(d[0][rows], d[1])
    got = out[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_c4_triangulation_round_trip(ctx):
    s = synth.two_view_scene(200_000, noise_px=0.0)
    P1 = api.projection_matrix(s["K"], s["R1"], s["T1"]); P2 = api.projection_matrix(s["K"], s["R2"], s["T2"])
    _, X = ctx.triangulate2(P1, P2, s["xy1"], s["xy2"])
    # project -> triangulate returns the point: float32 pixels (1/4096 px quantisation at u ~ 2000) over a 1-unit baseline
    rel = np.linalg.norm(X - s["X"], axis=1) / np.linalg.norm(s["X"], axis=1)
    assert rel.max() < 2e-3 and np.median(rel) < 5e-5
    # scaling both pixel sets' homogeneous rows leaves the DLT null vector unchanged: a permutation of the matches permutes X
    perm = np.random.default_rng(2).permutation(200_000)
    _, Xp = ctx.triangulate2(P1, P2, s["xy1"][perm], s["xy2"][perm])
    assert np.array_equal(Xp, X[perm])


def test_c4_bundle_adjustment_full_size(ctx):
    cfg = synth.CONFIGS["C4"]
    sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    # (1) one linearisation against the oracle at full size (n = 1198): 1e-9 relative to the largest entry
    pb = ctx.ba_create(*args)
    S, rhs, cost = pb.reduced_system(1e4)
    orc.set_num_threads(16)
    So, rhso, costo = orc.ba_reduced_system(*args, 1e4)
    assert abs(cost - costo) <= 1e-12 * costo
    assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(rhs - rhso).max() <= 1e-9 * np.abs(rhso).max()
    # (2) LM trajectory: the cost never increases over accepted steps, and reaches the noise floor of the scene
    costs = [pb.iterate(1)["final_cost"] for _ in range(12)]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(costs, costs[1:]))
    rmse = np.sqrt(costs[-1] / (2 * sc["n_obs"]))
    assert rmse < 2.5
    K, ext, pts = pb.params()
    # (3) relabelling the points (a permutation of point ids and of the observation list) changes nothing beyond rounding
    rng = np.random.default_rng(3)
    pperm = rng.permutation(sc["n_pt"]); inv = np.empty_like(pperm); inv[pperm] = np.arange(sc["n_pt"])
    operm = rng.permutation(sc["n_obs"])
    pb2 = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"][pperm], sc["obs_cam"][operm], inv[sc["obs_pt"]][operm].astype(np.int32), sc["obs_uv"][operm])
    s2 = pb2.iterate(12)
    # (different fp64 summation orders through 12 LM iterations: 1e-7 on the cost, 1e-9 on the cameras; a point's update is
    #  V_p^-1 (...) with cond(V_p) up to ~1e10 for two-view points at a narrow baseline or with an outlier pixel, so the
    #  bulk of the points must agree to 1e-9 (median 1e-10) and the worst-conditioned few to 1e-3)
    assert abs(s2["final_cost"] - costs[-1]) <= 1e-7 * costs[-1]
    K2, ext2, pts2 = pb2.params()
    dp = np.abs(pts2 - pts[pperm]).max(axis=1)
    q = np.quantile(dp, [0.5, 0.99, 0.999, 1.0])
    assert np.abs(ext2 - ext).max() <= 1e-9 and np.abs(K2 - K).max() <= 1e-9 * np.abs(K).max()
    assert q[0] <= 1e-10 and q[1] <= 1e-9 and q[3] <= 1e-3, f"point differences, quantiles 0.5/0.99/0.999/1: {q}"
    # (4) partial systems over 4 point shards add up to the full one (the multi-GPU contract, radius < 0 form)
    #     (column scaling off: each shard would otherwise scale by its own column norms)
    o = ctx.ba_options(jacobi_scaling=0)
    full = ctx.ba_create(K, ext, pts, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], opts=o)
    Sf, rf, cf = full.reduced_system(-1e4)
    full.close()
    acc_S = np.zeros_like(Sf); acc_r = np.zeros_like(rf); acc_c = 0.0
    for r in range(4):
        pl, oc, op, uv, ids = sdist.shard_points(sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], pts, r, 4)
        sh = ctx.ba_create(K, ext, pl, oc, op, uv, opts=o)
        Ss, rs, cs = sh.reduced_system(-1e4)
        acc_S += Ss; acc_r += rs; acc_c += cs
        sh.close()
    assert abs(acc_c - cf) <= 1e-11 * cf
    assert np.abs(acc_S - Sf).max() <= 1e-10 * np.abs(Sf).max() and np.abs(acc_r - rf).max() <= 1e-10 * max(np.abs(rf).max(), 1e-300)
    pb.close(); pb2.close()


def test_c4_pair_hamming2_self_match_and_oracle_slice(ctx):
    """AKAZE-like 61-byte descriptors at the C4 per-pair size (the reference's live NORM_HAMMING2 configuration)."""
    d = synth.akaze_descriptor_chain(2, 5000, seed=99)
    a, b = d[0], d[1]
    gi, gd = ctx.knn2_hamming2(a, a)
    if np.unique(a, axis=0).shape[0] == a.shape[0]:
        assert np.array_equal(gi[:, 0], np.arange(5000)) and not gd[:, 0].any()
    gi1, gd1 = ctx.knn2_hamming2(a, b)
    rows = np.random.default_rng(4).choice(5000, 400, replace=False)
    oi, od = orc.knn2_hamming2(a[rows], b)
    assert np.array_equal(gi1[rows], oi) and np.array_equal(gd1[rows], od)
    # distances are counts of differing 2-bit cells: integers in [0, 244]
    assert (gd1 == np.round(gd1)).all() and gd1.min() >= 0 and gd1.max() <= 244


def test_kernel_timing_api_and_full_size_tracks(ctx):
    # per-kernel timing of the matching path: two timed calls -> averaged over 2, then the counter resets
    d = synth.sift_descriptor_chain(2, 2000, seed=5)
    ctx.set_kernel_timing(True)
    ctx.knn2_l2(d[0], d[1]); ctx.knn2_l2(d[0], d[1])
    t = ctx.match_kernel_ms()
    assert t[2] == 2 and t[0] > 0 and t[1] > 0
    assert ctx.match_kernel_ms()[2] == 0
    ctx.set_kernel_timing(False)
    # N-view DLT at the C4 track count: noise-free tracks come back exactly, whatever the observation order
    sc = synth.ba_scene(200, 300_000, noise_px=0.0, outlier_frac=0.0, perturb=False)
    perm = np.random.default_rng(8).permutation(sc["n_obs"])
    pts, nv = ctx.triangulate_tracks(sc["K_true"], sc["ext_true"], sc["obs_cam"][perm], sc["obs_pt"][perm], sc["obs_uv"][perm], sc["n_pt"])
    assert np.array_equal(nv, np.bincount(sc["obs_pt"], minlength=sc["n_pt"]))
    assert np.abs(pts - sc["pts_true"]).max() < 1e-7
    err = ctx.reprojection_errors(sc["K_true"], sc["ext_true"], pts, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    assert err.max() < 1e-6


def test_c3_bundle_adjustment_scene(ctx):
    """BASELINE.json configs[2]: 50 cameras / 80k points / ~320k observations (n = 298): the reduced system and six forced
    LM steps against the oracle, same bars as the small-scene tests."""
    cfg = synth.CONFIGS["C3"]
    sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    orc.set_num_threads(16)
    pb = ctx.ba_create(*args)
    S, rhs, cost = pb.reduced_system(1e4)
    So, rhso, costo = orc.ba_reduced_system(*args, 1e4)
    assert S.shape == (298, 298) and abs(cost - costo) <= 1e-12 * costo
    assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(rhs - rhso).max() <= 1e-9 * np.abs(rhso).max()
    s = pb.iterate(6)
    K, ext, pts = pb.params()
    Ko, exto, ptso, so, _ = orc.ba_solve(*args, force_iterations=6)
    assert s["iterations"] == so["iterations"] == 6 and s["successful_steps"] == so["successful_steps"]
    assert abs(s["initial_cost"] - so["initial_cost"]) <= 1e-12 * so["initial_cost"]
    assert abs(s["final_cost"] - so["final_cost"]) <= 1e-8 * so["final_cost"]
    assert np.abs(ext - exto).max() <= 1e-6 * 10.0 and np.abs(K - Ko).max() <= 1e-6 * 3000.0
    assert np.quantile(np.abs(pts - ptso).max(axis=1), 0.999) <= 1e-6 * 10.0
    pb.close()


def test_c5_pair_knn_oracle_slice(ctx):
    """BASELINE.json configs[4] per-pair size: 10k x 10k x 128 fused kNN-2 + ratio tail; oracle on a 200-row query slice."""
    d = synth.sift_descriptor_chain(2, 10000, seed=31337)
    a, b = d[0], d[1]
    gi, gd = ctx.knn2_l2(a, b)
    rows = np.random.default_rng(12).choice(10000, 200, replace=False)
    oi, od = orc.knn2_l2(a[rows], b)
    assert np.array_equal(gi[rows], oi) and np.array_equal(gd[rows].view(np.uint32), od.view(np.uint32))
    m = api.match_features(a, b, ctx=ctx)
    ref = api.ratio_filter(gi, gd)
    assert len(m) > 4000 and np.array_equal(m, ref)
    # a permutation of the train rows permutes the answer
    perm = np.random.default_rng(13).permutation(10000)
    gi2, gd2 = ctx.knn2_l2(a, b[perm])
    assert np.array_equal(gd.view(np.uint32), gd2.view(np.uint32))
    clean = gd[:, 0] != gd[:, 1]                          # (the runner-up may still tie with the third: its index is checked by distance)
    assert np.array_equal(perm[gi2[clean, 0]], gi[clean, 0])
    same = perm[gi2[:, 1]] == gi[:, 1]
    assert same.mean() > 0.99
    for i in np.nonzero(~same)[0]:
        assert np.sqrt(((a[i] - b[perm[gi2[i, 1]]]) ** 2).sum(dtype=np.float64)) == np.sqrt(((a[i] - b[gi[i, 1]]) ** 2).sum(dtype=np.float64))


def test_c5_pair_hamming2_two_windows_oracle_slice_and_permutation(ctx):
    """BASELINE.json configs[4] per-pair size on binary rows: 10k x 10k, 61 bytes -- the matrix-core kernel over two chunk windows
    (8192 + 2048 rows, 8-bit tile index): an oracle slice, the VALU kernel on the same data, and the size-independent properties
    (self match at distance 0; a permutation of the train rows permutes the answer up to ties)."""
    import torch
    d = synth.akaze_descriptor_chain(2, 10000, seed=4242)
    a, b = d[0], d[1]
    gi, gd = ctx.knn2_hamming2(a, b)
    rows = np.random.default_rng(7).choice(10000, 200, replace=False)
    oi, od = orc.knn2_hamming2(a[rows], b)
    assert np.array_equal(gi[rows], oi) and np.array_equal(gd[rows], od)
    qs = ctx.descset_hamming2(torch.from_numpy(a).cuda()); ts = ctx.descset_hamming2(torch.from_numpy(b).cuda())
    idx = torch.empty((10000, 2), dtype=torch.int32, device="cuda"); dist = torch.empty((10000, 2), dtype=torch.float32, device="cuda")
    ctx.knn2_dev(qs, ts, idx, dist, force_path=3)
    ctx.synchronize()
    assert np.array_equal(idx.cpu().numpy(), gi) and np.array_equal(dist.cpu().numpy(), gd)
    si, sd = ctx.knn2_hamming2(a, a)
    if np.unique(a, axis=0).shape[0] == a.shape[0]:
        assert np.array_equal(si[:, 0], np.arange(10000)) and not sd[:, 0].any()
    perm = np.random.default_rng(8).permutation(10000)
    gi2, gd2 = ctx.knn2_hamming2(a, b[perm])
    assert np.array_equal(gd, gd2)                        # distances are permutation-invariant
    clean = (gd[:, 0] != gd[:, 1])                        # integer distances tie often: compare the best index where it is unique
    third_free = clean & (gd2[:, 0] != gd2[:, 1])
    assert third_free.sum() > 5000 and np.array_equal(perm[gi2[third_free, 0]], gi[third_free, 0])
    m = api.match_features(a, b, ctx=ctx)
    assert len(m) > 4000 and np.array_equal(m, api.ratio_filter(gi, gd))


@pytest.fixture(scope="module")
def c5_scene():
    cfg = synth.CONFIGS["C5"]
    t0 = time.time()
    sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
    print(f"[c5] scene generated in {time.time() - t0:.1f} s: {sc['n_obs']} observations")
    return sc


def test_c5_bundle_adjustment_full_size(ctx, c5_scene):
    """BASELINE.json configs[4] on ONE GPU: 1000 cameras / 2M points / ~8M observations, n = 5998 (S dense = 288 MB).
    (1) one linearisation against the oracle on 16 threads; (2) monotone LM trajectory down to the noise floor;
    (3) bitwise-identical rerun."""
    sc = c5_scene
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    t0 = time.time()
    pb = ctx.ba_create(*args)
    t_create = time.time() - t0
    S, rhs, cost = pb.reduced_system(1e4)
    assert S.shape == (5998, 5998)
    orc.set_num_threads(16)
    t0 = time.time()
    So, rhso, costo = orc.ba_reduced_system(*args, 1e4)
    print(f"[c5] ba_create {t_create:.1f} s, oracle linearisation {time.time() - t0:.1f} s")
    assert abs(cost - costo) <= 1e-12 * costo
    assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(rhs - rhso).max() <= 1e-9 * np.abs(rhso).max()
    del So, rhso, S
    costs = [pb.iterate(1)["final_cost"] for _ in range(10)]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(costs, costs[1:]))
    assert np.sqrt(costs[-1] / (2 * sc["n_obs"])) < 2.5
    K, ext, pts = pb.params()
    pb.reset(); pb.iterate(10)
    K2, ext2, pts2 = pb.params()
    assert np.array_equal(K, K2) and np.array_equal(ext, ext2) and np.array_equal(pts, pts2)
    pb.close()


def test_c5_point_shards_add_up(ctx, c5_scene):
    """the multi-GPU contract at C5's size: the partial reduced systems of four point shards sum to the full one"""
    sc = c5_scene
    o = ctx.ba_options(jacobi_scaling=0)
    full = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], opts=o)
    Sf, rf, cf = full.reduced_system(-1e4)
    full.close()
    acc_r = np.zeros_like(rf); acc_c = 0.0
    acc_S = np.zeros_like(Sf)
    for r in range(4):
        pl, oc, op, uv, ids = sdist.shard_points(sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["pts0"], r, 4)
        sh = ctx.ba_create(sc["K0"], sc["ext0"], pl, oc, op, uv, opts=o)
        Ss, rs, cs = sh.reduced_system(-1e4)
        acc_S += Ss; acc_r += rs; acc_c += cs
        sh.close(); del Ss
    assert abs(acc_c - cf) <= 1e-11 * cf
    assert np.abs(acc_S - Sf).max() <= 1e-10 * np.abs(Sf).max() and np.abs(acc_r - rf).max() <= 1e-10 * max(np.abs(rf).max(), 1e-300)


def test_normals_at_300k_points(ctx):
    """estimate_normals (NView:551-599) at the C4 point count, where the reference's O(N^2 log N) host loop is unusable:
    (1) 400 random rows against a numpy restatement of the same rule (10 nearest other points by Euclidean distance,
    plane fit, flip towards the origin side); (2) on a noisy sphere the normals are radial and point inwards."""
    rng = np.random.default_rng(77)
    n = 300_000
    d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = d * (5.0 + 0.002 * rng.standard_normal((n, 1))) + np.array([0.3, -0.2, 0.1])
    t0 = time.time()
    nrm = ctx.estimate_normals(pts, 10)
    dt = time.time() - t0
    print(f"[normals] 300k points, K=10: {dt * 1e3:.0f} ms incl. H2D/D2H")
    assert np.isfinite(nrm).all() and np.abs(np.linalg.norm(nrm, axis=1) - 1.0).max() < 1e-12
    rows = rng.choice(n, 400, replace=False)
    for i in rows:
        dist = np.sqrt(((pts - pts[i]) ** 2).sum(1)); dist[i] = np.inf
        nb = np.argsort(dist, kind="stable")[:10]
        q = pts[nb]; mean = q.mean(0)
        C = (q - mean).T @ (q - mean) / 10.0
        w, V = np.linalg.eigh(C)
        v = V[:, 0]
        if v @ mean > 0:
            v = -v
        assert np.abs(nrm[i] - v).max() <= 1e-7, i           # eigenvector conditioning: gap / eps
    radial = -(pts - np.array([0.3, -0.2, 0.1])) / np.linalg.norm(pts - np.array([0.3, -0.2, 0.1]), axis=1, keepdims=True)
    cosang = (nrm * radial).sum(1)
    assert np.quantile(cosang, 0.01) > 0.95


def test_c3_on_the_bench_own_generator_follows_the_oracle(ctx):
    """bench.py times the std::mt19937_64 scenes of host/sfm_synth.cpp, the tests above the numpy PCG64 ones: the benchmarked inputs
    themselves under parity (round-3 review): C3 = 50 cameras / 80,000 points, six forced LM steps against the oracle, 1e-8 on the cost."""
    sc = synth.ba_scene_mt(50, 80000)
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    pb = ctx.ba_create(*args)
    sg = pb.iterate(6)
    pb.close()
    so = orc.ba_solve(*args, force_iterations=6)[3]
    assert sg["iterations"] == so["iterations"] == 6 and sg["successful_steps"] == so["successful_steps"]
    assert abs(sg["initial_cost"] - so["initial_cost"]) <= 1e-10 * so["initial_cost"]
    assert abs(sg["final_cost"] - so["final_cost"]) <= 1e-8 * so["final_cost"], (sg["final_cost"], so["final_cost"])
