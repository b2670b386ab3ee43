"""sfmhip_ba_create builds its orderings on the device (csrc/ba_setup.hpp: radix sorts, scans, gathers).  The tables must
be, entry by entry, what the serial host code of rounds 1-2 produced; that construction is restated here in numpy
(lexsort = the stable sorts it used).  Reference: bundle_adjustment() hands the observation list to ceres::Problem in
NViewReconstuct.cpp:1187-1210; the orderings themselves are internal to this implementation."""
import numpy as np
import pytest

from sfm_opencv_amd import synth

pytestmark = pytest.mark.gpu


def host_tables(n_cam, n_pt, obs_cam, obs_pt, obs_uv, fix0=1, schur_chunk=512):
    obs_cam = np.asarray(obs_cam, np.int64); obs_pt = np.asarray(obs_pt, np.int64)
    n_obs = len(obs_cam)
    k = np.arange(n_obs)
    # observations by (point, camera, caller's index)
    by_pt = np.lexsort((k, obs_cam, obs_pt))
    cnt = np.bincount(obs_pt, minlength=n_pt)
    st = np.concatenate([[0], np.cumsum(cnt)])
    mmax = int(cnt.max()) if n_pt else 0
    # camera lists padded with -1 (a prefix sorts first), points sorted lexicographically, ties by index
    lists = np.full((n_pt, max(mmax, 1)), -1, np.int64)
    pos = np.arange(n_obs) - st[obs_pt[by_pt]]
    lists[obs_pt[by_pt], pos] = obs_cam[by_pt]
    order = np.lexsort((np.arange(n_pt),) + tuple(lists[:, j] for j in range(lists.shape[1] - 1, -1, -1)))
    slot = np.empty(n_pt, np.int64); slot[order] = np.arange(n_pt)
    pt_start = np.concatenate([[0], np.cumsum(cnt[order])])
    q_of = pt_start[slot[obs_pt[by_pt]]] + pos           # storage position of by_pt[i]
    ocam = np.empty(n_obs, np.int64); opt = np.empty(n_obs, np.int64); ouv = np.empty((n_obs, 2))
    ocam[q_of] = obs_cam[by_pt]; opt[q_of] = slot[obs_pt[by_pt]]; ouv[q_of] = np.asarray(obs_uv).reshape(-1, 2)[by_pt]
    by_cam = np.lexsort((np.arange(n_obs), ocam))
    cam_start = np.concatenate([[0], np.cumsum(np.bincount(ocam, minlength=n_cam))])
    cam_pt = opt[by_cam]; cam_uv = ouv[by_cam]
    nblk256 = max(1, -(-n_pt // 256))
    crange = np.empty((nblk256, 2), np.int64)
    for b in range(nblk256):
        seg = ocam[pt_start[min(n_pt, b * 256)]:pt_start[min(n_pt, (b + 1) * 256)]]
        crange[b] = (seg.min(), seg.max()) if len(seg) else (2**31 - 1, -1)
    # pair items in generation order (point, i, j > i), then stable by key
    keys, qi, qj = [], [], []
    for s in range(n_pt):
        lo, hi = pt_start[s], pt_start[s + 1]
        for i in range(lo, hi):
            for j in range(i + 1, hi):
                ci, cj, a, b = ocam[i], ocam[j], i, j
                if ci < cj:
                    ci, cj, a, b = cj, ci, b, a
                if fix0 and (ci == 0 or cj == 0):
                    continue
                keys.append(ci * n_cam + cj); qi.append(a); qj.append(b)
    keys = np.asarray(keys, np.int64); qi = np.asarray(qi, np.int64); qj = np.asarray(qj, np.int64)
    o = np.argsort(keys, kind="stable")
    keys, qi, qj = keys[o], qi[o], qj[o]
    items = np.stack([qi, qj, opt[qi] if len(qi) else qi, np.zeros_like(qi)], 1) if len(qi) else np.zeros((0, 4), np.int64)
    blk_cam, blk_chunk, chunk_desc = [], [], []
    t = 0
    while t < len(keys):
        u = t
        while u < len(keys) and keys[u] == keys[t]:
            u += 1
        ca, cb = int(keys[t] // n_cam), int(keys[t] % n_cam)
        blk_cam.append((ca, cb)); blk_chunk.append(len(chunk_desc))
        cntk = u - t; nch = -(-cntk // schur_chunk); per = -(-(-(-cntk // nch)) // 64) * 64
        for a in range(t, u, per):
            chunk_desc.append((ca, cb, a, min(u, a + per)))
        t = u
    blk_chunk.append(len(chunk_desc))
    return dict(pt_slot=slot, pt_start=pt_start, ocam=ocam, opt=opt, ouv=ouv.ravel(), cam_start=cam_start, cam_pt=cam_pt,
                cam_uv=cam_uv.ravel(), blk_crange=crange.ravel(), blk_cam=np.asarray(blk_cam, np.int64).ravel(),
                blk_chunk=np.asarray(blk_chunk), chunk_desc=np.asarray(chunk_desc, np.int64).ravel(), items=items.ravel())


def check(ctx, n_cam, n_pt, oc, op, uv, pts=None, fix0=1):
    rng = np.random.default_rng(1)
    pts = rng.normal(size=(n_pt, 3)) if pts is None else pts
    K0 = np.array([1000.0, 1000.0, 500.0, 400.0]); ext = np.zeros((n_cam, 6)); ext[:, 5] = 5.0
    opts = ctx.ba_options(fix_first_camera=fix0)
    pb = ctx.ba_create(K0, ext, pts, oc, op, uv, opts)
    ref = host_tables(n_cam, n_pt, oc, op, uv, fix0)
    for name, want in ref.items():
        got = pb.debug_table(name)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        assert np.array_equal(got, want), (name, np.flatnonzero(got != want)[:8])
    _, _, back = pb.params()          # the caller's point order comes back
    assert np.array_equal(back, pts)
    pb.close()


def test_tables_match_host_construction_synthetic_scene(ctx):
    sc = synth.ba_scene(24, 3000)
    check(ctx, 24, 3000, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["pts0"])


def test_tables_random_graph_shuffled_duplicates_and_gaps(ctx):
    # arbitrary camera sets (not a chain), shuffled observation order, a camera seen twice by one point, points
    # without observations, cameras without observations, track lengths beyond one packing group (64 / b positions)
    rng = np.random.default_rng(7)
    n_cam, n_pt = 37, 2500
    oc, op = [], []
    for p in range(n_pt):
        if p % 97 == 0:
            continue                                    # no observations at all
        L = int(rng.integers(1, 9)) if p % 50 else int(rng.integers(12, 30))
        cams = rng.choice(np.arange(n_cam - 2), size=min(L, n_cam - 2), replace=False)     # the last two cameras stay empty
        if p % 13 == 0:
            cams = np.concatenate([cams, cams[:1]])     # one camera twice
        oc += list(cams); op += [p] * len(cams)
    oc = np.asarray(oc, np.int32); op = np.asarray(op, np.int32)
    sh = rng.permutation(len(oc)); oc, op = oc[sh], op[sh]
    uv = rng.uniform(0, 1000, size=(len(oc), 2))
    check(ctx, n_cam, n_pt, oc, op, uv)
    check(ctx, n_cam, n_pt, oc, op, uv, fix0=0)


def test_tables_degenerate_sizes(ctx):
    rng = np.random.default_rng(3)
    # one point, two observations; and a problem whose only pairs involve the constant camera (no pair lists at all)
    check(ctx, 2, 1, np.array([1, 0], np.int32), np.array([0, 0], np.int32), rng.uniform(size=(2, 2)))
    check(ctx, 3, 4, np.array([0, 1, 0, 2, 0, 1, 0, 2], np.int32), np.array([0, 0, 1, 1, 2, 2, 3, 3], np.int32), rng.uniform(size=(8, 2)))
    # more than one radix tile per sort (4096 elements) with few distinct keys
    n_pt = 9000
    oc = np.tile(np.array([1, 2, 3], np.int32), n_pt); op = np.repeat(np.arange(n_pt, dtype=np.int32), 3)
    check(ctx, 4, n_pt, oc, op, rng.uniform(size=(len(oc), 2)))


def test_out_of_range_observation_is_rejected(ctx):
    from sfm_opencv_amd._lib import SfmHipError
    K0 = np.array([1000.0, 1000.0, 500.0, 400.0]); ext = np.zeros((2, 6)); pts = np.zeros((2, 3))
    for oc, op in (([0, 2], [0, 1]), ([0, 1], [0, 2]), ([0, -1], [0, 1])):
        with pytest.raises(SfmHipError):
            ctx.ba_create(K0, ext, pts, np.array(oc, np.int32), np.array(op, np.int32), np.zeros((2, 2)))
