"""GPU parity of bundle adjustment vs the oracle (fp64; tolerances stated per assertion)."""
import numpy as np
import pytest

import oracle as orc
from sfm_opencv_amd import synth, api

pytestmark = pytest.mark.gpu


def _args(sc):
    return sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"]


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("kw", [dict(), dict(huber_delta=0.0), dict(fix_intrinsics=1), dict(jacobi_scaling=0),
                                dict(fix_first_camera=0)])
def test_reduced_system_matches_oracle(ctx, kw):
    sc = synth.ba_scene(12, 700)
    for radius in (1e4, 3.0):
        pb = ctx.ba_create(*_args(sc), opts=ctx.ba_options(**kw))
        S, rhs, cost = pb.reduced_system(radius)
        So, rhso, costo = orc.ba_reduced_system(*_args(sc), radius, opts=orc.ba_default_options(**kw))
        assert S.shape == So.shape
        assert abs(cost - costo) <= 1e-12 * costo
        assert np.abs(S - S.T).max() <= 1e-12 * np.abs(S).max()
        # fp64 sums in different orders + analytic vs dual-number Jacobians: 1e-9 relative to the largest entry
        assert _relerr(S, So) <= 1e-9
        assert _relerr(rhs, rhso) <= 1e-9
        pb.close()


def test_reduced_system_duplicate_camera_and_single_obs_points(ctx):
    sc = synth.ba_scene(6, 80, outlier_frac=0.0)
    oc, op, uv = sc["obs_cam"].copy(), sc["obs_pt"].copy(), sc["obs_uv"].copy()
    # point 3 seen twice by its first camera (two keypoints mapped to one track), point 5 seen only once
    k = np.nonzero(op == 3)[0][0]
    oc = np.append(oc, oc[k]); op = np.append(op, 3); uv = np.vstack([uv, uv[k] + [0.7, -0.4]])
    keep = np.ones(len(oc), bool); keep[np.nonzero(op == 5)[0][1:]] = False
    oc, op, uv = oc[keep], op[keep], uv[keep]
    pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], oc, op, uv)
    S, rhs, cost = pb.reduced_system(100.0)
    So, rhso, costo = orc.ba_reduced_system(sc["K0"], sc["ext0"], sc["pts0"], oc, op, uv, 100.0)
    assert _relerr(S, So) <= 1e-9 and _relerr(rhs, rhso) <= 1e-9 and abs(cost - costo) <= 1e-12 * costo


def test_partial_systems_add_up_over_point_shards(ctx):
    # the multi-GPU contract: undamped camera-side systems of point shards sum to the full system
    sc = synth.ba_scene(10, 500)
    o = ctx.ba_options(jacobi_scaling=0)
    full = ctx.ba_create(*_args(sc), opts=o)
    S, rhs, cost = full.reduced_system(-50.0)
    acc_S = np.zeros_like(S); acc_r = np.zeros_like(rhs); acc_c = 0.0
    for r in range(3):
        sel = (sc["obs_pt"] % 3) == r
        pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"][sel], sc["obs_pt"][sel], sc["obs_uv"][sel], opts=o)
        s_, r_, c_ = pb.reduced_system(-50.0)
        acc_S += s_; acc_r += r_; acc_c += c_
    assert _relerr(acc_S, S) <= 1e-12 and _relerr(acc_r, rhs) <= 1e-12 and abs(acc_c - cost) <= 1e-12 * cost


def test_forced_iterations_follow_oracle_trajectory(ctx):
    sc = synth.ba_scene(16, 1500)
    pb = ctx.ba_create(*_args(sc))
    n_it = 6
    s = pb.iterate(n_it)
    K, ext, pts = pb.params()
    Ko, exto, ptso, so, tr = orc.ba_solve(*_args(sc), force_iterations=n_it)
    assert s["iterations"] == so["iterations"] == n_it
    assert s["successful_steps"] == so["successful_steps"]
    # per-iteration agreement degrades with the conditioning of each solve; after 6 LM steps: 1e-8 on the cost,
    # 1e-6 on the parameters relative to the scene scale
    assert abs(s["final_cost"] - so["final_cost"]) <= 1e-8 * so["final_cost"]
    assert abs(s["initial_cost"] - so["initial_cost"]) <= 1e-12 * so["initial_cost"]
    assert np.abs(pts - ptso).max() <= 1e-6 * 10.0
    assert np.abs(ext - exto).max() <= 1e-6 * 10.0
    assert np.abs(K - Ko).max() <= 1e-6 * 3000.0
    assert np.array_equal(ext[0], sc["ext0"][0])          # camera 0 never modified (NView:1178)
    # state carries over: two more iterations == oracle with 8
    s2 = pb.iterate(2)
    so8 = orc.ba_solve(*_args(sc), force_iterations=8)[3]
    assert s2["iterations"] == 8 and abs(s2["final_cost"] - so8["final_cost"]) <= 1e-7 * so8["final_cost"]
    # reset restores the start
    pb.reset()
    K2, ext2, pts2 = pb.params()
    assert np.array_equal(K2, sc["K0"]) and np.array_equal(ext2, sc["ext0"]) and np.array_equal(pts2, sc["pts0"])


def test_solve_to_convergence_matches_oracle(ctx):
    sc = synth.ba_scene(12, 600)
    K, ext, pts, s = ctx.ba_solve(*_args(sc))
    Ko, exto, ptso, so, _ = orc.ba_solve(*_args(sc))
    assert s["termination"] == so["termination"] == 0
    assert s["iterations"] == so["iterations"]
    assert abs(s["final_cost"] - so["final_cost"]) <= 1e-6 * so["final_cost"]          # 1e-6 at convergence
    assert np.abs(pts - ptso).max() <= 1e-5 * 10.0
    assert np.abs(ext - exto).max() <= 1e-5 * 10.0
    # and it actually reduced the reprojection error to the noise level (0.5 px + 2 % outliers under Huber)
    rmse = np.sqrt(s["final_cost"] / s["num_residuals"])
    assert rmse < 2.5


def test_observation_order_invariance(ctx):
    sc = synth.ba_scene(8, 300)
    rng = np.random.default_rng(0)
    perm = rng.permutation(sc["n_obs"])
    a = ctx.ba_create(*_args(sc)); b = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"][perm], sc["obs_pt"][perm], sc["obs_uv"][perm])
    sa = a.iterate(4); sb = b.iterate(4)
    assert abs(sa["final_cost"] - sb["final_cost"]) <= 1e-9 * sa["final_cost"]


@pytest.mark.parametrize("shape", [(10, 900), (120, 30000)])
def test_determinism_bitwise(ctx, shape):
    """Reruns are bit-identical: every reduction has a fixed order, and the solver's fp64 atomic adds touch each element
    from exactly one lane per panel with barriers between panels.  (120 cameras: four dissection segments + shared top.)"""
    sc = synth.ba_scene(*shape)
    outs = []
    for _ in range(2):
        pb = ctx.ba_create(*_args(sc))
        pb.iterate(4)
        outs.append(pb.params())
        pb.close()
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


def test_allreduce_hook_identity_and_failure(ctx):
    sc = synth.ba_scene(8, 300)
    ref = ctx.ba_create(*_args(sc)); sr = ref.iterate(3)
    calls = []
    pb = ctx.ba_create(*_args(sc))
    pb.set_allreduce(lambda ptr, count, stream: calls.append(count) or 0, 0, 1)
    s = pb.iterate(3)
    assert s["final_cost"] == sr["final_cost"] and len(calls) >= 6
    bad = ctx.ba_create(*_args(sc))
    bad.set_allreduce(lambda ptr, count, stream: 7, 0, 1)
    with pytest.raises(api.SfmHipError):
        bad.iterate(1)


def _run_sharded_on_one_gpu(sc, n_iter, world=2, opts_of_rank=None):
    """SURVEY 8e on a single card: `world` contexts, each with a shard of the points, driven by `world` threads whose
    all-reduce hook sums the libraries' packed messages -- the N>1 data path (packed non-zero blocks through the hook,
    replicated solve, sharded back-substitution).  Returns (summaries, params, point ids, message sizes) per rank."""
    import threading
    import torch
    from sfm_opencv_amd import dist as sdist
    ctxs = [api.Context(0, use_torch_stream=False) for _ in range(world)]
    probs, ids = [], []
    for r in range(world):
        pts_l, oc, op, uv, pid = sdist.shard_points(sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["pts0"], r, world)
        o = ctxs[r].ba_options(**(opts_of_rank(r) if opts_of_rank else {}))
        probs.append(ctxs[r].ba_create(sc["K0"], sc["ext0"], pts_l, oc, op, uv, opts=o)); ids.append(pid)
    bar = threading.Barrier(world)
    slots = [None] * world
    counts = [[] for _ in range(world)]

    def make_hook(r):
        def hook(ptr, count, stream):
            ctxs[r].synchronize()                                   # this rank's message is complete
            slots[r] = torch.as_tensor(sdist._CudaView(ptr, count), device="cuda")
            counts[r].append(count)
            bar.wait()
            if r == 0:
                total = slots[0].clone()
                for q in range(1, world):
                    total += slots[q]
                for q in range(world):
                    slots[q].copy_(total)
                torch.cuda.synchronize()
            bar.wait()
            return 0
        return hook

    out, errs = [None] * world, []

    def run(r):
        try:
            probs[r].set_allreduce(make_hook(r), r, world)
            out[r] = probs[r].iterate(n_iter)
        except Exception as e:                                      # pragma: no cover
            errs.append(e); bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(180)
    assert not errs and all(o is not None for o in out), errs
    params = [pb.params() for pb in probs]
    for pb in probs: pb.close()
    for c in ctxs: c.close()
    return out, params, ids, counts


@pytest.mark.parametrize("shape", [(24, 4000), (120, 12000)])
def test_two_point_shards_on_one_gpu_match_unsharded(ctx, shape):
    """Two point shards against the unsharded solve.  (120 cameras: the multi-segment solver path with the hook.)"""
    sc = synth.ba_scene(*shape)
    ref = ctx.ba_create(*_args(sc)); sr = ref.iterate(5); Kr, extr, ptsr = ref.params()
    out, params, ids, counts = _run_sharded_on_one_gpu(sc, 5)
    # start-up: camera adjacency, the scaling linearisation, |x|^2; then the first linearisation and ONE call per LM iteration
    # (the step scalars ride in the message of the speculative next linearisation; a miss would cost one more)
    assert counts[0] == counts[1] and 3 + 1 + 5 <= len(counts[0]) <= 3 + 1 + 5 + 1, counts[0]
    assert counts[0][2] == 1 and len(set(counts[0][3:])) == 1
    n_red = 6 * (sc["n_cam"] - 1) + 4
    assert max(counts[0]) < n_red * n_red                           # packed: fewer doubles than the dense square
    for r in range(2):
        assert out[r]["iterations"] == sr["iterations"] and out[r]["successful_steps"] == sr["successful_steps"]
        assert abs(out[r]["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
        K, ext, pts = params[r]
        assert np.abs(ext - extr).max() <= 1e-9 and np.abs(K - Kr).max() <= 1e-9 * np.abs(Kr).max()
        assert np.abs(pts - ptsr[ids[r]]).max() <= 1e-9


def test_sharded_error_flag_is_shared_by_all_ranks():
    """A rank whose shard trips the point kernel's error flag (non-SPD V of a LOCAL point) must not take a different
    accept / invalid branch than its peers (replicated cameras, radius and nu would diverge and the next all-reduce
    hang).  Injection: rank 1 damps its points with a negative max_lm_diagonal, so every V of its shard fails the
    Cholesky while the cost stays finite; rank 0's shard is healthy.  Every rank must count the same invalid steps."""
    sc = synth.ba_scene(24, 4000)
    out, params, ids, counts = _run_sharded_on_one_gpu(sc, 3, opts_of_rank=lambda r: dict(max_lm_diagonal=-1.0) if r == 1 else {})
    assert out[0]["iterations"] == out[1]["iterations"] == 3
    assert out[0]["successful_steps"] == out[1]["successful_steps"] == 0       # every step invalid on BOTH ranks
    assert out[0]["final_radius"] == out[1]["final_radius"]                    # both halved the radius three times
    for r in range(2):
        assert np.array_equal(params[r][1], sc["ext0"])                        # no rank moved its camera replicas


def test_bundle_adjustment_wrapper_in_place(ctx, capsys):
    sc = synth.ba_scene(6, 200, outlier_frac=0.0)
    kps, ids = [], []
    for c in range(sc["n_cam"]):
        sel = sc["obs_cam"] == c
        kp = np.zeros(int(sel.sum()) + 2, api.KEYPOINT)
        kp["x"][:-2] = sc["obs_uv"][sel, 0]; kp["y"][:-2] = sc["obs_uv"][sel, 1]
        kps.append(kp); ids.append(np.concatenate([sc["obs_pt"][sel], [-1, -1]]))
    K = sc["K0"].copy(); ext = sc["ext0"].copy(); pts = sc["pts0"].copy()
    s = api.bundle_adjustment(K, ext, ids, kps, pts, ctx=ctx)
    out = capsys.readouterr().out
    assert "Bundle Adjustment statistics (approximated RMSE):" in out and " #views: 6" in out
    assert s["final_cost"] < s["initial_cost"] and not np.array_equal(pts, sc["pts0"])
    assert np.array_equal(ext[0], sc["ext0"][0])


def test_non_chain_tracks_take_the_dense_solver_and_match_oracle(ctx):
    """Tracks through arbitrary (non-consecutive) cameras: the reduced system is not block-banded, the nested-dissection
    plan does not apply and the dense blocked Cholesky fallback runs.  Same parity bars as the chain case."""
    rng = np.random.default_rng(17)
    sc = synth.ba_scene(40, 3000)
    # re-draw each point's cameras at random (keeping the number of views), re-project with the true parameters + noise
    L = np.bincount(sc["obs_pt"], minlength=sc["n_pt"])
    obs_pt = np.repeat(np.arange(sc["n_pt"]), L)
    obs_cam = np.concatenate([rng.choice(sc["n_cam"], l, replace=False) for l in L])
    uv = synth.project(sc["K_true"], sc["ext_true"][obs_cam], sc["pts_true"][obs_pt]) + 0.5 * rng.standard_normal((obs_pt.size, 2))
    order = np.lexsort((obs_pt, obs_cam))
    args = (sc["K0"], sc["ext0"], sc["pts0"], obs_cam[order].astype(np.int32), obs_pt[order].astype(np.int32), np.ascontiguousarray(uv[order]))
    pb = ctx.ba_create(*args)
    S, rhs, cost = pb.reduced_system(1e4)
    So, rhso, costo = orc.ba_reduced_system(*args, 1e4)
    assert abs(cost - costo) <= 1e-12 * costo and _relerr(S, So) <= 1e-9 and _relerr(rhs, rhso) <= 1e-9
    assert (np.abs(So) > 0).mean() > 0.5                        # really dense
    s = pb.iterate(6)
    Ko, exto, ptso, so, tr = orc.ba_solve(*args, force_iterations=6)
    assert s["successful_steps"] == so["successful_steps"]
    assert abs(s["final_cost"] - so["final_cost"]) <= 1e-8 * so["final_cost"]
    K, ext, pts = pb.params()
    assert np.abs(ext - exto).max() <= 1e-7 and np.abs(pts - ptso).max() <= 1e-6


def test_phase_timing_is_opt_in(ctx):
    """sfmhip_ba_phase_ms reports only while sfmhip_set_kernel_timing is on (the events cost ~27 us per iteration)."""
    sc = synth.ba_scene(120, 30000)
    pb = ctx.ba_create(*_args(sc))
    pb.iterate(3)
    assert not any(pb.phase_ms()[:7])
    ctx.set_kernel_timing(True)
    try:
        pb.iterate(3)
        ph = pb.phase_ms()
    finally:
        ctx.set_kernel_timing(False)
    assert all(v > 0 for v in ph[:4]) and abs(ph[0] + ph[1] + ph[2] - ph[3]) <= 1e-6 * ph[3]    # float32 event times
    assert ph[6] > 0 and ph[7] > 0          # four dissection segments: the forward kernel is timed, the factor has blocks
    pb.iterate(2)
    assert not any(pb.phase_ms()[:7])       # "of the last iterate call"
    pb.close()


@pytest.mark.parametrize("seed", range(8))
def test_randomised_shapes_follow_oracle(ctx, seed):
    """Camera counts from one 32-block to four dissection segments, track lengths 2..9 (panels with 1..6 row blocks in the
    sparse solver), random option mixes: three forced LM steps against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    n_cam = int(rng.choice([5, 9, 23, 40, 64, 97, 130, 171]))
    n_pt = int(rng.integers(60, 120)) * n_cam
    max_len = int(rng.integers(3, 10))
    sc = synth.ba_scene(n_cam, n_pt, seed=int(rng.integers(1, 1 << 30)), max_len=max_len, outlier_frac=float(rng.choice([0.0, 0.02])))
    kw = dict(fix_intrinsics=int(rng.integers(0, 2)), fix_first_camera=int(rng.integers(0, 2)), jacobi_scaling=int(rng.integers(0, 2)),
              huber_delta=float(rng.choice([0.0, 1.0, 4.0])))
    pb = ctx.ba_create(*_args(sc), opts=ctx.ba_options(**kw))
    s = pb.iterate(3)
    K, ext, pts = pb.params()
    Ko, exto, ptso, so, _ = orc.ba_solve(*_args(sc), opts=orc.ba_default_options(**kw), force_iterations=3)
    pb.close()
    assert s["successful_steps"] == so["successful_steps"], (n_cam, n_pt, max_len, kw)
    assert abs(s["final_cost"] - so["final_cost"]) <= 1e-8 * so["final_cost"], (n_cam, n_pt, max_len, kw)
    # parameters: 1e-6 of the scene scale, as in test_forced_iterations_follow_oracle_trajectory (a free first camera leaves
    # the gauge to the damping: compare the cost only there)
    if kw["fix_first_camera"]:
        assert np.abs(ext - exto).max() <= 1e-6 * 10.0 and np.abs(K - Ko).max() <= 1e-6 * 3000.0, (n_cam, n_pt, max_len, kw)
        assert np.quantile(np.abs(pts - ptso).max(axis=1), 0.99) <= 1e-6 * 10.0, (n_cam, n_pt, max_len, kw)


@pytest.mark.parametrize("omega", [(0.0, 0.0, 0.0), (6e-9, -3e-9, 2e-9), (3e-8, 5e-8, -4e-8), (2e-5, -1e-5, 3e-5)])
def test_angle_axis_first_order_branch_and_tiny_angles(ctx, omega):
    """ceres::AngleAxisRotatePoint switches to p = X + w x X for theta^2 <= DBL_EPSILON; the kernels carry that branch as a
    per-camera flag (derivative e_m x X) and otherwise form d(RX)/dw_m = c_m x (R X): free cameras exactly at, just below,
    just above the switch and at a tiny angle must give the oracle's reduced system."""
    sc = synth.ba_scene(7, 400, outlier_frac=0.0)
    ext = sc["ext0"].copy()
    ext[2, :3] = omega; ext[5, :3] = np.array(omega) * -0.5
    args = (sc["K0"], ext, sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    for kw in (dict(), dict(fix_first_camera=0)):
        pb = ctx.ba_create(*args, opts=ctx.ba_options(**kw))
        S, rhs, cost = pb.reduced_system(1e4)
        So, rhso, costo = orc.ba_reduced_system(*args, 1e4, opts=orc.ba_default_options(**kw))
        pb.close()
        assert abs(cost - costo) <= 1e-12 * costo
        assert _relerr(S, So) <= 1e-9 and _relerr(rhs, rhso) <= 1e-9


# ---------------------------------------------------------------------------------------------------------------------
# linearizer = 2: the run-tile construction of the reduced system (csrc/ba_tiles.hpp) -- opt-in, same contract
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kw", [dict(), dict(huber_delta=0.0), dict(fix_intrinsics=1), dict(jacobi_scaling=0), dict(fix_first_camera=0)])
def test_run_tiles_reduced_system_matches_oracle(ctx, kw):
    # 12 cameras / 700 points: runs of ~12 points (partly filled 16-point batches); 40 / 9000: runs of ~50 (several batches per wave)
    for shape in ((12, 700), (40, 9000)):
        sc = synth.ba_scene(*shape)
        for radius in (1e4, 3.0):
            pb = ctx.ba_create(*_args(sc), opts=ctx.ba_options(linearizer=2, **kw))
            S, rhs, cost = pb.reduced_system(radius)
            So, rhso, costo = orc.ba_reduced_system(*_args(sc), radius, opts=orc.ba_default_options(**kw))
            assert abs(cost - costo) <= 1e-12 * costo
            assert np.abs(S - S.T).max() <= 1e-12 * np.abs(S).max()
            assert _relerr(S, So) <= 1e-9 and _relerr(rhs, rhso) <= 1e-9       # same tolerance as the per-observation kernels
            pb.close()


def test_run_tiles_duplicate_camera_single_obs_and_long_tracks(ctx):
    sc = synth.ba_scene(9, 300, outlier_frac=0.0, max_len=7)
    oc, op, uv = sc["obs_cam"].copy(), sc["obs_pt"].copy(), sc["obs_uv"].copy()
    cnt = np.bincount(op)
    assert cnt.max() == 7                               # 7 observations: three row tiles, two observations per lane
    pa, pb_ = int(np.nonzero(cnt <= 5)[0][0]), int(np.nonzero(cnt >= 3)[0][1])
    k = np.nonzero(op == pa)[0][0]                      # point pa seen twice by one camera, point pb_ seen once
    oc = np.append(oc, oc[k]); op = np.append(op, pa); uv = np.vstack([uv, uv[k] + [0.7, -0.4]])
    keep = np.ones(len(oc), bool); keep[np.nonzero(op == pb_)[0][1:]] = False
    oc, op, uv = oc[keep], op[keep], uv[keep]
    So, rhso, costo = orc.ba_reduced_system(sc["K0"], sc["ext0"], sc["pts0"], oc, op, uv, 100.0)
    pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], oc, op, uv, opts=ctx.ba_options(linearizer=2))
    S, rhs, cost = pb.reduced_system(100.0)
    assert _relerr(S, So) <= 1e-9 and _relerr(rhs, rhso) <= 1e-9 and abs(cost - costo) <= 1e-12 * costo
    pb.close()
    # a track longer than the tiles take (8 observations): the request falls back to the per-observation kernels
    sc8 = synth.ba_scene(10, 200, min_len=8, max_len=8)
    pb = ctx.ba_create(*_args(sc8), opts=ctx.ba_options(linearizer=2))
    S, rhs, cost = pb.reduced_system(100.0)
    So, rhso, costo = orc.ba_reduced_system(*_args(sc8), 100.0)
    assert _relerr(S, So) <= 1e-9 and _relerr(rhs, rhso) <= 1e-9
    pb.close()


@pytest.mark.parametrize("shape", [(24, 4000), (120, 30000)])
def test_run_tiles_iterations_follow_the_per_observation_path_and_repeat_bitwise(ctx, shape):
    sc = synth.ba_scene(*shape)
    ref = ctx.ba_create(*_args(sc)); sr = ref.iterate(6); Kr, extr, ptsr = ref.params(); ref.close()
    outs = []
    for _ in range(2):
        pb = ctx.ba_create(*_args(sc), opts=ctx.ba_options(linearizer=2))
        s = pb.iterate(6); outs.append(pb.params()); pb.close()
        assert s["successful_steps"] == sr["successful_steps"]
        assert abs(s["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
    K, ext, pts = outs[0]
    assert np.abs(ext - extr).max() <= 1e-8 and np.abs(pts - ptsr).max() <= 1e-8 and np.abs(K - Kr).max() <= 1e-8 * np.abs(Kr).max()
    for x, y in zip(*outs):
        assert np.array_equal(x, y)                     # fixed reduction orders: reruns are bit-identical


def test_run_tiles_two_point_shards_match_unsharded(ctx):
    sc = synth.ba_scene(24, 4000)
    ref = ctx.ba_create(*_args(sc)); sr = ref.iterate(5); Kr, extr, ptsr = ref.params(); ref.close()
    out, params, ids, counts = _run_sharded_on_one_gpu(sc, 5, opts_of_rank=lambda r: dict(linearizer=2))
    for r in range(2):
        assert out[r]["successful_steps"] == sr["successful_steps"]
        assert abs(out[r]["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
        K, ext, pts = params[r]
        assert np.abs(ext - extr).max() <= 1e-8 and np.abs(pts - ptsr[ids[r]]).max() <= 1e-8


def test_parameters_without_residuals_stay_put(ctx):
    """A camera no observation refers to (Ceres never sees such a block: only the blocks of added residuals enter the problem,
    NViewReconstuct.cpp:1187-1197) and points without observations: zero rows of the reduced system.  They must neither move nor make
    the damped system singular as the radius grows; everything else follows the oracle."""
    sc = synth.ba_scene(8, 300)
    keep = sc["obs_pt"] % 10 != 3                                   # every tenth point loses all its observations
    ext0 = np.vstack([sc["ext0"], sc["ext0"][5] + 0.01])            # a ninth camera that sees nothing
    a = (sc["K0"], ext0, sc["pts0"], sc["obs_cam"][keep], sc["obs_pt"][keep], sc["obs_uv"][keep])
    K, ext, pts, s = ctx.ba_solve(*a)
    Ko, exto, ptso, so, _ = orc.ba_solve(*a)
    assert s["termination"] == so["termination"] == 0 and s["iterations"] == so["iterations"]
    assert abs(s["final_cost"] - so["final_cost"]) <= 1e-9 * so["final_cost"]
    assert np.array_equal(ext[8], ext0[8]) and np.array_equal(pts[3::10], sc["pts0"][3::10])
    assert np.abs(ext - exto).max() <= 1e-7 and np.abs(pts - ptso).max() <= 1e-7
    # a factorisation that breaks down must not poison the following iterations: with single-observation points (camera 7's
    # observations dropped) S + D loses definiteness once the radius passes ~5e9; the step is invalid, the radius shrinks, LM goes on
    keep2 = keep & (sc["obs_cam"] != 7)
    b = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"][keep2], sc["obs_pt"][keep2], sc["obs_uv"][keep2])
    pb = ctx.ba_create(*b)
    s2 = pb.iterate(30)
    assert np.isfinite(s2["final_cost"]) and s2["final_cost"] < 2640.0 and s2["successful_steps"] >= 12
    Kb, extb, ptsb = pb.params(); pb.close()
    assert np.isfinite(extb).all() and np.isfinite(ptsb).all() and np.array_equal(extb[7], sc["ext0"][7])
    # no observations at all / no points: nothing to do, not an error
    for b in ((sc["K0"], sc["ext0"], sc["pts0"], np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2))),
              (sc["K0"], sc["ext0"], np.zeros((0, 3)), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2)))):
        _, e3, _, s3 = ctx.ba_solve(*b)
        assert s3["termination"] == 0 and s3["final_cost"] == 0.0 and np.array_equal(e3, sc["ext0"])


@pytest.mark.parametrize("linearizer", [0, 2])
def test_reused_message_tail_with_an_unobserved_camera(ctx, linearizer):
    """Round-2 advisor finding: after the first iteration only S is zero-filled for the next linearisation; the tail
    [rhs | diagU | graw | scalars] and the error flag are re-used and depend on every entry being stored again.  A free camera
    without observations is the case where a skipped store would leave the previous iteration's values behind: two separate
    iterate() calls must follow a fresh problem's two iterations bit for bit."""
    sc = synth.ba_scene(8, 500)
    keep = sc["obs_cam"] != 5
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"][keep], sc["obs_pt"][keep], sc["obs_uv"][keep])
    a = ctx.ba_create(*args, ctx.ba_options(linearizer=linearizer))
    a1 = a.iterate(1); a2 = a.iterate(1)
    b = ctx.ba_create(*args, ctx.ba_options(linearizer=linearizer))
    b2 = b.iterate(2)
    assert a2["final_cost"] == b2["final_cost"] and a2["final_gradient_max_norm"] == b2["final_gradient_max_norm"], (a1, a2, b2)
    assert a2["final_cost"] < a1["initial_cost"]
    for x, y in zip(a.params(), b.params()):
        assert np.array_equal(x, y)
    assert np.array_equal(a.params()[1][5], sc["ext0"][5])          # nothing pulls on the unobserved camera
    a.close(); b.close()


def test_solve_summary_times_cover_the_whole_call(ctx):
    """sfmhip_ba_solve is one call like bundle_adjustment() (NView:1162-1244); its total_time_s is the whole call the way
    Ceres' total_time_in_seconds is (NView:1239): construction + minimiser + write-back."""
    import time
    sc = synth.ba_scene(12, 4000)
    t0 = time.perf_counter()
    _, _, _, s = ctx.ba_solve(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    wall = time.perf_counter() - t0
    parts = s["preprocessor_time_s"] + s["minimizer_time_s"] + s["postprocessor_time_s"]
    assert s["preprocessor_time_s"] > 0 and s["minimizer_time_s"] > 0 and s["postprocessor_time_s"] > 0
    assert parts <= s["total_time_s"] * 1.001 + 1e-5 and s["total_time_s"] <= wall
    assert parts >= 0.9 * s["total_time_s"]


@pytest.mark.parametrize("shape,n_ctx", [((24, 4000), 2), ((120, 12000), 3)])
def test_single_process_multi_context_solve_matches_the_single_gpu_call(ctx, shape, n_ctx):
    """sfmhip_ba_solve_multi: what a C++ caller with several GPUs in ONE process uses (the reference's main() is one process,
    NView:1334-1524).  The box has one card, so the contexts share device 0 and the message goes through the library's host-staged
    in-process exchange instead of RCCL; shards by first camera, one thread per context, same result as sfmhip_ba_solve."""
    sc = synth.ba_scene(*shape)
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    Kr, extr, ptsr, sr = ctx.ba_solve(*args)
    ctxs = [api.Context(0, use_torch_stream=False) for _ in range(n_ctx)]
    K, ext, pts, s = api.ba_solve_multi(ctxs, *args)
    for c in ctxs:
        c.close()
    assert s["termination"] == sr["termination"] and s["iterations"] == sr["iterations"] and s["num_residuals"] == sr["num_residuals"]
    assert abs(s["final_cost"] - sr["final_cost"]) <= 1e-7 * sr["final_cost"]
    # run to convergence the two differ along the nearly flat directions of the cost (sums taken in another order): same cost, parameters to ~1e-4
    assert np.abs(ext - extr).max() <= 1e-3 and np.abs(K - Kr).max() <= 1e-5 * np.abs(Kr).max() and np.abs(pts - ptsr).max() <= 1e-3
    assert np.array_equal(ext[0], sc["ext0"][0])                   # camera 0 stays constant on every rank
    assert s["total_time_s"] > 0


@pytest.mark.parametrize("shape,n_ctx", [((24, 4000), 2), ((120, 12000), 3), ((60, 9000), 4)])
def test_multi_context_solve_follows_the_single_gpu_steps(ctx, shape, n_ctx):
    """The same comparison where it is sharp: five LM steps with every tolerance switched off (so both calls take exactly five), the
    sharded sums against the unsharded ones: 1e-9 on the cost and on every parameter (the converged comparison above cannot be
    that tight: sums taken in another order wander along the flat directions).  Also: the time split reported by the call."""
    sc = synth.ba_scene(*shape)
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    o = ctx.ba_options(max_num_iterations=5, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    Kr, extr, ptsr, sr = ctx.ba_solve(*args, opts=o)
    ctxs = [api.Context(0, use_torch_stream=False) for _ in range(n_ctx)]
    K, ext, pts, s = api.ba_solve_multi(ctxs, *args, opts=o)
    K2, ext2, pts2, s2 = api.ba_solve_multi(ctxs, *args, opts=o)          # again on the same contexts (communicators / blocks kept)
    for c in ctxs:
        c.close()
    assert sr["iterations"] == s["iterations"] == 5 and s["successful_steps"] == sr["successful_steps"]
    assert abs(s["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
    assert _relerr(ext, extr) <= 1e-9 and _relerr(K, Kr) <= 1e-9 and _relerr(pts, ptsr) <= 1e-9
    assert np.array_equal(ext, ext2) and np.array_equal(pts, pts2) and np.array_equal(K, K2)
    assert s["preprocessor_time_s"] > 0 and s["minimizer_time_s"] > 0 and s["preprocessor_time_s"] + s["minimizer_time_s"] <= s["total_time_s"] * 1.001


def test_multi_context_solve_reports_a_failed_shard_instead_of_hanging(ctx):
    """One rank cannot build its shard (its device allocations fail: sfmhip_debug_fail_allocations): every rank leaves with an error
    before anyone enters a collective -- the call returns SFMHIP_E_HIP naming the rank, it does not hang; the contexts stay usable."""
    sc = synth.ba_scene(24, 4000)
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    ctxs = [api.Context(0, use_torch_stream=False) for _ in range(3)]
    ctxs[0].trim(); ctxs[1].trim(); ctxs[2].trim()
    for bad in (1, 0):
        ctxs[bad].lib.sfmhip_debug_fail_allocations(ctxs[bad].h, 3)
        with pytest.raises(api.SfmHipError) as e:
            api.ba_solve_multi(ctxs, *args)
        assert "injected allocation failure" in str(e.value) and (bad == 0 or "rank 1" in str(e.value))
        ctxs[bad].lib.sfmhip_debug_fail_allocations(ctxs[bad].h, 0)
    K, ext, pts, s = api.ba_solve_multi(ctxs, *args)
    Kr, extr, ptsr, sr = ctx.ba_solve(*args)
    for c in ctxs:
        c.close()
    assert s["iterations"] == sr["iterations"] and abs(s["final_cost"] - sr["final_cost"]) <= 1e-7 * sr["final_cost"]


def test_problem_memory_is_reused_across_solves_and_trim_releases_it(ctx):
    """The context keeps the device blocks of destroyed problems (a hipFree of gigabytes stalls the following calls); reuse and
    sfmhip_trim must not change results: the same solve three times, trimming in between, bit for bit."""
    sc = synth.ba_scene(30, 20000)
    outs = []
    for rep in range(3):
        K, ext, pts, s = ctx.ba_solve(*_args(sc))
        outs.append((K, ext, pts, s["final_cost"], s["iterations"]))
        if rep == 0:
            ctx.trim()
        big = synth.ba_scene(12, 300 + 100 * rep)                   # another problem size in between: different blocks come and go
        ctx.ba_solve(*_args(big))
    for o in outs[1:]:
        assert o[3] == outs[0][3] and o[4] == outs[0][4]
        for a, b in zip(o[:3], outs[0][:3]):
            assert np.array_equal(a, b)


# ------------------------------------------------------------------------------------------------
# sfm_ba_options.solver = 0 on narrow bands (tracks of 2..4 frames: band 3 cameras): the chain solver of csrc/ba_chain.hpp
# (fronts in LDS, a camera at a time, two launches) instead of the level-per-launch nested dissection (solver = 1)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,kw", [((16, 1500), dict()), ((30, 4000), dict(fix_intrinsics=1)), ((61, 9000), dict(fix_first_camera=0)),
                                       ((120, 20000), dict(huber_delta=0.0)), ((200, 30000), dict())])
def test_chain_solver_on_narrow_bands(ctx, shape, kw):
    """Forced LM steps against the oracle (1e-8 on the cost, as for the other solver), against solver = 1 on the same problem
    (two factorisations of one system in different orders: 1e-10 on the parameters), bit-identical reruns.  16 cameras: one
    launch, one leaf per two waves; 200 cameras: 16 leaves, three tree levels in the second kernel."""
    sc = synth.ba_scene(*shape, max_len=4)
    runs = {}
    for solver in (0, 0, 1):
        pb = ctx.ba_create(*_args(sc), opts=ctx.ba_options(solver=solver, **kw))
        s = pb.iterate(5)
        runs.setdefault(solver, []).append((s, pb.params()))
        pb.close()
    so = orc.ba_solve(*_args(sc), force_iterations=5, opts=orc.ba_default_options(**kw))[3]
    (s0, p0), (s0b, p0b) = runs[0]
    (s1, p1), = runs[1]
    assert s0["iterations"] == so["iterations"] == 5 and s0["successful_steps"] == so["successful_steps"]
    assert abs(s0["final_cost"] - so["final_cost"]) <= 1e-8 * so["final_cost"]
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-11 * s1["final_cost"]
    for x, y in zip(p0, p1):
        assert _relerr(x, y) <= 1e-10
    for x, y in zip(p0, p0b):
        assert np.array_equal(x, y)


def test_chain_solver_survives_a_failed_factorisation(ctx):
    """A camera whose block of the damped system is not positive definite (a camera with two observations only, next to no damping):
    the chain solver raises the error flag like the other solver, LM halves the radius and goes on instead of returning NaN."""
    sc = synth.ba_scene(24, 2000, max_len=4)
    keep = np.ones(len(sc["obs_cam"]), bool)
    idx = np.nonzero(sc["obs_cam"] == 11)[0]
    keep[idx[2:]] = False
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"][keep], sc["obs_pt"][keep], sc["obs_uv"][keep])
    out = {}
    for solver in (0, 1):
        pb = ctx.ba_create(*args, opts=ctx.ba_options(solver=solver, initial_trust_region_radius=1e30, min_lm_diagonal=1e-300))
        out[solver] = pb.iterate(6)
        pb.close()
    assert np.isfinite(out[0]["final_cost"]) and out[0]["final_cost"] <= out[0]["initial_cost"]
    assert out[0]["successful_steps"] == out[1]["successful_steps"] < 6          # both solvers refuse the same steps
    assert abs(out[0]["final_cost"] - out[1]["final_cost"]) <= 1e-6 * out[1]["final_cost"]
