"""GPU parity of the matching path vs the CPU oracle (bit-exact index sets, bit-exact float distances).
Everything goes through the C-ABI (sfm_opencv_amd.api -> libsfmhip.so)."""
import numpy as np
import pytest

import oracle as orc
from sfm_opencv_amd import synth
from sfm_opencv_amd import api

pytestmark = pytest.mark.gpu


def _check_knn(ctx, q, t, hamming=False):
    if hamming:
        gi, gd = ctx.knn2_hamming2(q, t); oi, od = orc.knn2_hamming2(q, t)
    else:
        gi, gd = ctx.knn2_l2(q, t); oi, od = orc.knn2_l2(q, t)
    assert np.array_equal(gi, oi), f"index mismatch at rows {np.nonzero((gi != oi).any(1))[0][:10]}"
    assert np.array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("nq,nt", [(1, 2), (5, 3), (128, 128), (129, 255), (300, 1000), (2000, 2000), (513, 4500)])
def test_knn2_l2_sift_like_int8_path(ctx, nq, nt):
    d = synth.sift_descriptor_chain(2, max(nq, nt), seed=11 + nq)
    q, t = d[0][:nq], d[1][:nt]
    s = ctx.descset_l2(q)
    assert s.info()["exact_u8"]
    _check_knn(ctx, q, t)


def test_knn2_l2_many_exact_ties(ctx):
    # few distinct values -> many equal distances: lowest train index must win, as cv::batchDistance does
    rng = np.random.default_rng(3)
    q = rng.integers(0, 3, (257, 128)).astype(np.float32)
    t = np.repeat(rng.integers(0, 3, (40, 128)), 8, axis=0).astype(np.float32)   # every train row 8 times
    _check_knn(ctx, q, t)


def _three_squares(n):
    for a in range(int(n ** 0.5), -1, -1):
        for b in range(int((n - a * a) ** 0.5), -1, -1):
            c2 = n - a * a - b * b
            c = int(round(c2 ** 0.5))
            if c * c == c2 and max(a, b, c) <= 255:
                return a, b, c
    return None


def test_knn2_l2_sqrt_collision_rescore(ctx):
    # d^2 >= 2^22: distinct integers can share one float32 sqrt; cv::batchDistance compares the float32 distances,
    # so the LOWER train index wins although its integer distance is larger.  Forces the flagged-row re-score branch.
    base = 125 * 200 * 200                      # 5,000,000 >= 2^22
    k = next(k for k in range(1, 5000)
             if np.sqrt(np.float32(base + k)) == np.sqrt(np.float32(base + k + 1))
             and _three_squares(k) and _three_squares(k + 1))
    rng = np.random.default_rng(5)
    t = rng.integers(230, 256, (300, 128)).astype(np.float32)     # far away from the zero query rows
    t[0, :125] = 200; t[0, 125:] = _three_squares(k + 1)          # index 0: d^2 = base + k + 1 (larger integer)
    t[1, :125] = 200; t[1, 125:] = _three_squares(k)              # index 1: d^2 = base + k     (smaller integer)
    q = np.zeros((130, 128), np.float32)
    q[1:] = rng.integers(0, 3, (129, 128))
    oi, od = orc.knn2_l2(q, t)
    assert oi[0, 0] == 0 and oi[0, 1] == 1 and od[0, 0] == od[0, 1]   # float tie -> lower index first
    d2 = ((q[0].astype(np.int64) - t.astype(np.int64)) ** 2).sum(-1)
    assert d2[1] < d2[0] and d2[1] >= 2 ** 22                          # integer order would say index 1
    _check_knn(ctx, q, t)


@pytest.mark.parametrize("dim", [128, 64, 40, 20])
def test_knn2_l2_general_float_exact_path(ctx, dim):
    rng = np.random.default_rng(7 + dim)
    q = rng.standard_normal((70, dim)).astype(np.float32) * 50
    t = rng.standard_normal((333, dim)).astype(np.float32) * 50
    assert not ctx.descset_l2(q).info()["exact_u8"]
    _check_knn(ctx, q, t)


@pytest.mark.parametrize("dim", [32, 64, 100])
def test_knn2_l2_small_dims_int8_path(ctx, dim):
    rng = np.random.default_rng(dim)
    q = rng.integers(0, 256, (200, dim)).astype(np.float32)
    t = rng.integers(0, 256, (300, dim)).astype(np.float32)
    _check_knn(ctx, q, t)


def test_knn2_l2_fewer_than_two_trains(ctx):
    rng = np.random.default_rng(9)
    q = rng.integers(0, 256, (10, 128)).astype(np.float32)
    t = rng.integers(0, 256, (1, 128)).astype(np.float32)
    gi, gd = ctx.knn2_l2(q, t)
    oi, od = orc.knn2_l2(q, t)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert (gi[:, 1] == -1).all()


@pytest.mark.parametrize("nq,nt,nb", [(3, 2, 61), (256, 257, 61), (1000, 3000, 61), (100, 100, 32), (77, 500, 64)])
def test_knn2_hamming2(ctx, nq, nt, nb):
    d = synth.akaze_descriptor_chain(2, max(nq, nt), nbytes=nb, seed=nq)
    _check_knn(ctx, d[0][:nq], d[1][:nt], hamming=True)


def _knn2_hamming_dev(ctx, q, t, path):
    import torch
    qs = ctx.descset_hamming2(torch.from_numpy(np.ascontiguousarray(q)).cuda()); ts = ctx.descset_hamming2(torch.from_numpy(np.ascontiguousarray(t)).cuda())
    idx = torch.empty((q.shape[0], 2), dtype=torch.int32, device="cuda"); dist = torch.empty((q.shape[0], 2), dtype=torch.float32, device="cuda")
    ctx.knn2_dev(qs, ts, idx, dist, force_path=path)
    ctx.synchronize()
    return idx.cpu().numpy(), dist.cpu().numpy()


@pytest.mark.parametrize("nq,nt,nb", [(1, 1, 61), (2, 1, 61), (3, 2, 61), (70, 33, 61), (256, 257, 61), (1000, 3000, 61), (300, 4097, 61), (257, 9000, 61),
                                      (100, 100, 32), (500, 700, 1), (64, 5000, 17)])
def test_knn2_hamming2_matrix_core_kernel_equals_valu_kernel_and_oracle(ctx, nq, nt, nb):
    """Round 3: Hamming2 on the FP4 matrix cores (rows of up to 61 bytes: knn2_hamming2_fp4_kernel) against the VALU popcount kernel
    (force_path 3) and the CPU oracle: same index pairs, same distances, on one chunk and on several (4097 / 9000 trains: windows of
    4096), on sets smaller than one tile, on fewer than two trains (the runner-up stays missing: pad rows are never selected)."""
    d = synth.akaze_descriptor_chain(2, max(nq, nt), nbytes=nb, seed=nq + nt)
    q, t = d[0][:nq], d[1][:nt]
    i4, d4 = _knn2_hamming_dev(ctx, q, t, 4)
    i3, d3 = _knn2_hamming_dev(ctx, q, t, 3)
    oi, od = orc.knn2_hamming2(q, t)
    assert np.array_equal(i4, oi), f"index mismatch at rows {np.nonzero((i4 != oi).any(1))[0][:10]}"
    assert np.array_equal(d4.view(np.uint32), od.view(np.uint32))
    assert np.array_equal(i3, oi) and np.array_equal(d3.view(np.uint32), od.view(np.uint32))


def test_knn2_hamming2_matrix_core_kernel_ties_and_extremes(ctx):
    """Ties go to the lower train index across lanes, tiles, stages and chunks (rows repeated all over a 5000-row set); distance 0
    (identical rows) and the maximum 244 (complement of every cell) both come out exactly."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (7, 61), dtype=np.uint8)
    t = base[rng.integers(0, 7, 5000)]
    q = np.concatenate([base, base ^ np.uint8(0x55), base ^ np.uint8(0xFF), rng.integers(0, 256, (300, 61), dtype=np.uint8)])
    i4, d4 = _knn2_hamming_dev(ctx, q, t, 4)
    oi, od = orc.knn2_hamming2(q, t)
    assert np.array_equal(i4, oi) and np.array_equal(d4.view(np.uint32), od.view(np.uint32))
    assert d4[:7, 0].max() == 0.0
    allq = np.array([[0x00] * 61, [0xFF] * 61], np.uint8); allt = np.array([[0x55] * 61, [0x00] * 61, [0xAA] * 61], np.uint8)
    i4, d4 = _knn2_hamming_dev(ctx, allq, allt, 4)
    oi, od = orc.knn2_hamming2(allq, allt)
    assert np.array_equal(i4, oi) and np.array_equal(d4, od) and d4.max() == 244.0


@pytest.mark.parametrize("nt", [1, 2, 31, 33, 45, 100, 257])
def test_knn2_hamming2_matrix_core_kernel_pad_rows_never_beat_the_worst_real_row(ctx, nt):
    """The margin between real rows and the rows that pad a set to whole tiles (advisor, round 3): train counts that are not
    multiples of 32, EVERY real distance at the maximum 4 * 61 = 244 -- the neighbours must still be real rows 0 and 1 (lowest
    indices on the tie), on the matrix-core kernel as on the VALU one."""
    q = np.zeros((70, 61), np.uint8); t = np.full((nt, 61), 0x55, np.uint8)
    for path in (4, 3):
        i, d = _knn2_hamming_dev(ctx, q, t, path)
        oi, od = orc.knn2_hamming2(q, t)
        assert np.array_equal(i, oi) and np.array_equal(d.view(np.uint32), od.view(np.uint32)), path
        assert (i[:, 0] == 0).all() and (d[:, 0] == 244.0).all() and (i[:, 1] == (1 if nt > 1 else -1)).all()


def test_knn2_hamming2_forced_paths_are_checked(ctx):
    import torch
    t64 = ctx.descset_hamming2(torch.zeros((10, 64), dtype=torch.uint8, device="cuda"))
    idx = torch.empty((10, 2), dtype=torch.int32, device="cuda"); dist = torch.empty((10, 2), dtype=torch.float32, device="cuda")
    with pytest.raises(api.SfmHipError):
        ctx.knn2_dev(t64, t64, idx, dist, force_path=4)         # 64-byte rows do not fit the 768-value encoding
    ctx.knn2_dev(t64, t64, idx, dist, force_path=0)             # ... and go to the VALU kernel by themselves
    ctx.synchronize()
    assert idx.cpu().numpy()[:, 0].tolist() == [0] * 10


def test_match_features_l2_and_chain(ctx):
    descs = synth.sift_descriptor_chain(4, 700, seed=21)
    got = api.match_features_for_all(descs, ctx=ctx)
    assert len(got) == 3
    for i, g in enumerate(got):
        ref = orc.match_features_l2(descs[i], descs[i + 1])
        assert len(ref) > 100
        assert np.array_equal(g, ref)          # structured compare: ids and float distance bits
    one = api.match_features(descs[0], descs[1], ctx=ctx)
    assert np.array_equal(one, orc.match_features_l2(descs[0], descs[1]))


def test_match_features_hamming2_chain_with_ragged_sizes(ctx):
    descs = synth.akaze_descriptor_chain(3, 900, seed=2)
    descs = [descs[0][:500], descs[1], descs[2][:130]]
    got = api.match_features_for_all(descs, ctx=ctx)
    for i, g in enumerate(got):
        ref = orc.match_features_hamming2(descs[i], descs[i + 1])
        assert np.array_equal(g, ref)


def test_ratio_tail_device_equals_host(ctx):
    descs = synth.sift_descriptor_chain(2, 1500, seed=33)
    idx, dist = ctx.knn2_l2(descs[0], descs[1])
    host = api.ratio_filter(idx, dist)
    dev = api.match_features(descs[0], descs[1], ctx=ctx)
    assert np.array_equal(host, dev)
    assert np.array_equal(host, orc.ratio_filter(idx, dist))


def test_distance_matrix(ctx):
    import torch
    descs = synth.sift_descriptor_chain(2, 600, seed=44)
    q, t = descs[0][:333], descs[1][:590]
    qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
    for ld in (592, 591):     # vector and scalar store paths
        out = torch.full((333, ld), -1.0, device="cuda", dtype=torch.float32)
        ctx.l2_distance_matrix_dev(qs, ts, out)
        ctx.synchronize()
        ref = orc.l2_distance_matrix(q, t)
        got = out.cpu().numpy()
        assert np.array_equal(got[:, :590].view(np.uint32), ref.view(np.uint32))
        assert (got[:, 590:] == -1.0).all()
    # row stride = 16 mod 32 floats (the reference's dense 10000-column matrix): rows alternate between line-aligned and half a
    # line off, and the kernel tiles by row parity with the odd rows' train window shifted by 16 columns.  592 above is such a
    # stride; here more than one 256-row group, a train count that ends inside a shifted window, and a base address that defeats it
    descs = synth.sift_descriptor_chain(2, 1100, seed=45)
    q, t = descs[0][:700], descs[1][:1008]
    qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
    ref = orc.l2_distance_matrix(q, t)
    for nt_use, base_off in ((1008, 0), (1001, 0), (1008, 16)):
        tsu = ctx.descset_l2(t[:nt_use])
        buf = torch.full((700 * 1008 + 64,), -1.0, device="cuda", dtype=torch.float32)
        out = buf[base_off:base_off + 700 * 1008].view(700, 1008)
        ctx.l2_distance_matrix_dev(qs, tsu, out[:, :nt_use] if nt_use == 1008 else out)
        ctx.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got[:, :nt_use].view(np.uint32), ref[:, :nt_use].view(np.uint32))
        assert (got[:, nt_use:] == -1.0).all() and (buf[:base_off].cpu().numpy() == -1.0).all() and (buf[base_off + 700 * 1008:].cpu().numpy() == -1.0).all()
    # general float inputs: exact path
    rng = np.random.default_rng(1)
    qf = rng.standard_normal((50, 128)).astype(np.float32); tf = rng.standard_normal((90, 128)).astype(np.float32)
    out = torch.zeros((50, 90), device="cuda", dtype=torch.float32)
    ctx.l2_distance_matrix_dev(ctx.descset_l2(qf), ctx.descset_l2(tf), out)
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), orc.l2_distance_matrix(qf, tf).view(np.uint32))


def test_device_resident_knn_and_forced_paths_agree(ctx):
    import torch
    descs = synth.sift_descriptor_chain(2, 1100, seed=55)
    q = torch.from_numpy(descs[0]).cuda(); t = torch.from_numpy(descs[1]).cuda()
    qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
    res = []
    for path in (2, 1):
        idx = torch.empty((1100, 2), dtype=torch.int32, device="cuda")
        dist = torch.empty((1100, 2), dtype=torch.float32, device="cuda")
        ctx.knn2_dev(qs, ts, idx, dist, force_path=path)
        ctx.synchronize()
        res.append((idx.cpu().numpy(), dist.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    oi, od = orc.knn2_l2(descs[0], descs[1])
    assert np.array_equal(res[0][0], oi) and np.array_equal(res[0][1], od)


def test_exact_sqrt_all_integers(ctx):
    # the distance-matrix epilogue uses a short correctly-rounded sqrt valid for integer-valued floats < 2^24:
    # exhaustive device-side comparison with sqrtf
    import ctypes as C
    n = C.c_int(-1)
    assert ctx.lib.sfmhip_selftest_exact_sqrt(ctx.h, C.byref(n)) == 0
    assert n.value == 0


def test_match_lists_written_straight_to_pinned_host_memory(ctx):
    """sfmhip_match_pairs_dev takes any device-accessible address: with pinned host tensors the ratio-tail kernel delivers the
    match lists without a copy (what bench.py does).  Same lists as through device buffers."""
    import torch
    descs = synth.sift_descriptor_chain(4, 700, seed=46)
    sets = [ctx.descset_l2(torch.from_numpy(d).cuda()) for d in descs]
    pairs = np.array([[0, 1], [1, 2], [2, 3]], np.int32)
    d_m = torch.zeros((3, 700, 4), dtype=torch.int32, device="cuda"); d_c = torch.zeros((3,), dtype=torch.int32, device="cuda")
    h_m = torch.zeros((3, 700, 4), dtype=torch.int32).pin_memory(); h_c = torch.zeros((3,), dtype=torch.int32).pin_memory()
    ctx.match_pairs_dev(sets, pairs, d_m, 700, d_c)
    ctx.match_pairs_dev(sets, pairs, h_m, 700, h_c)
    ctx.synchronize()
    cnt = d_c.cpu().numpy()
    assert (cnt > 100).all() and np.array_equal(cnt, h_c.numpy())
    for p in range(3):
        assert np.array_equal(d_m[p, :cnt[p]].cpu().numpy(), h_m[p, :cnt[p]].numpy())
        m = orc.match_features_l2(descs[p], descs[p + 1])
        assert np.array_equal(h_m[p, :cnt[p], 0].numpy(), m["queryIdx"]) and np.array_equal(h_m[p, :cnt[p], 1].numpy(), m["trainIdx"])


def test_host_entry_points_strided_rows_large_uploads_and_block_reuse(ctx):
    """Round 3: host rows go to HBM through a pinned staging ring (contiguous matrices) or a 2-D copy (row stride > row length, an
    OpenCV ROI), the sets' device blocks come from the context's cache and go back to it, and the 'all values integers in [0, 255]'
    flag of a batch of sets is resolved with one round trip.  None of that may change a match list."""
    import ctypes as C
    descs = synth.sift_descriptor_chain(3, 9000, seed=77)           # 4.6 MB per image: the staged (multi-threaded) upload path
    want = [orc.match_features_l2(descs[i], descs[i + 1]) for i in range(2)]
    for rep in range(3):                                            # blocks released by one round are reused by the next
        got = api.match_features_for_all(descs, ctx=ctx)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        if rep == 1:
            ctx.trim()                                              # give the idle blocks back to the driver in between
    # strided rows through the C entry point: the matrices embedded in wider buffers
    wide = [np.full((9000, 160), -7.0, np.float32) for _ in range(2)]
    for k in range(2):
        wide[k][:, 16:144] = descs[k]
    out = np.zeros(9000, api.DMATCH); n = C.c_int()
    rc = ctx.lib.sfmhip_match_features_l2(ctx.h, wide[0][:, 16:].ctypes.data, 9000, wide[1][:, 16:].ctypes.data, 9000, 128, 160, 160,
                                          out.ctypes.data, C.byref(n))
    assert rc == 0 and np.array_equal(out[:n.value], want[0])
    # a set of general floats among integer-valued ones: the batch falls back to the exact path as a whole, same lists as the oracle
    mixed = [descs[0][:700].copy(), descs[1][:700].copy()]
    mixed[1][3, 5] += 0.25
    g = api.match_features_for_all(mixed, ctx=ctx)[0]
    assert np.array_equal(g, orc.match_features_l2(mixed[0], mixed[1]))
    b = synth.akaze_descriptor_chain(2, 5000, seed=5)
    assert np.array_equal(api.match_features_for_all(b, ctx=ctx)[0], orc.match_features_hamming2(b[0], b[1]))


# ------------------------------------------------------------------------------------------------
# many images from host matrices in one call (sfmhip_descsets_create_{l2,hamming2}_host): integer-valued L2 rows cross
# PCIe as bytes, every other image as floats; the sets must be the ones the per-image calls build
# ------------------------------------------------------------------------------------------------
def test_batched_host_sets_match_like_per_image_sets(ctx):
    rng = np.random.default_rng(11)
    chain = synth.sift_descriptor_chain(6, 700, seed=3)
    chain = [c.copy() for c in chain]
    chain[2] = chain[2][:333]                                     # ragged
    chain[3][5, 7] += 0.25                                        # one non-integer value: this image goes the float way (exact kernels)
    chain[4] = np.ascontiguousarray(np.pad(chain[4], ((0, 0), (0, 16))))[:, :128]        # strided rows (ld = 144)
    assert chain[4].strides[0] == 144 * 4
    pairs = np.stack([np.arange(5), np.arange(1, 6)], 1)
    batch = ctx.descsets_host(chain)
    infos = [s.info() for s in batch]
    assert [i["exact_u8"] for i in infos] == [True, True, True, False, True, True] and infos[2]["rows"] == 333
    single = [ctx.descset_l2(np.ascontiguousarray(c)) for c in chain]
    got = ctx.match_pairs(batch, pairs); want = ctx.match_pairs(single, pairs)
    for i, (g, w) in enumerate(zip(got, want)):
        ref = orc.match_features_l2(np.ascontiguousarray(chain[i]), np.ascontiguousarray(chain[i + 1]))
        assert len(ref) > 20 and np.array_equal(g, w) and np.array_equal(g, ref), i
    # the float rows re-created on the device are the caller's: the exact kernel on them gives the same neighbours
    import torch
    idx = torch.empty((700, 2), dtype=torch.int32, device="cuda"); dist = torch.empty((700, 2), dtype=torch.float32, device="cuda")
    ctx.knn2_dev(batch[0], batch[1], idx, dist, force_path=2)
    oi, od = orc.knn2_l2(chain[0], chain[1])
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(dist.cpu().numpy().view(np.uint32), od.view(np.uint32))
    # values outside [0, 255], NaN, negative zero
    odd = [rng.integers(0, 256, (40, 128)).astype(np.float32) for _ in range(4)]
    odd[0][3, 3] = 256.0; odd[1][0, 0] = -1.0; odd[2][39, 127] = np.nan; odd[3][1, 1] = -0.0
    sets = ctx.descsets_host(odd)
    assert [s.info()["exact_u8"] for s in sets] == [False, False, False, True]
    # Hamming2
    hchain = [rng.integers(0, 256, (r, 61), dtype=np.uint8) for r in (300, 517, 64, 1)]
    hb = ctx.descsets_host(hchain); hs = [ctx.descset_hamming2(c) for c in hchain]
    hp = np.array([[0, 1], [1, 2], [2, 3], [3, 0]])
    for g, w in zip(ctx.match_pairs(hb, hp, ratio=0.97), ctx.match_pairs(hs, hp, ratio=0.97)):
        assert np.array_equal(g, w)
    assert ctx.descsets_host([]) == []


@pytest.mark.parametrize("n_ctx", [1, 2, 3])
def test_match_pairs_over_several_contexts(ctx, n_ctx):
    """sfmhip_match_pairs_multi: the chain's pairs in contiguous blocks over n contexts (all on the box's one card here: what differs
    from a node is only which PCIe link a block's images cross), a block's images + its halo image uploaded to its context only;
    the lists are the single-context ones, in pair order.  Ragged image sizes, both norms, a pair list that is not a chain."""
    rng = np.random.default_rng(5)
    chain = [c.copy() for c in synth.sift_descriptor_chain(7, 600, seed=9)]
    chain[1] = chain[1][:411]; chain[5] = chain[5][:64]
    pairs = np.stack([np.arange(6), np.arange(1, 7)], 1)
    ctxs = [api.Context(0, use_torch_stream=False) for _ in range(n_ctx)]
    want = api.match_features_for_all(chain, ctx=ctx)
    got = api.match_pairs_multi(ctxs, chain, pairs)
    assert len(got) == 6 and sum(len(g) for g in got) > 200
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    odd_pairs = np.array([[6, 0], [2, 5], [3, 3], [0, 1], [4, 2]])
    sets = ctx.descsets_host(chain)
    for g, w in zip(api.match_pairs_multi(ctxs, chain, odd_pairs), ctx.match_pairs(sets, odd_pairs)):
        assert np.array_equal(g, w)
    ham = [rng.integers(0, 256, (r, 61), dtype=np.uint8) for r in (300, 211, 64, 500, 77)]
    hp = np.stack([np.arange(4), np.arange(1, 5)], 1)
    hs = ctx.descsets_host(ham)
    for g, w in zip(api.match_pairs_multi(ctxs, ham, hp, ratio=0.97), ctx.match_pairs(hs, hp, ratio=0.97)):
        assert np.array_equal(g, w)
    for c in ctxs:
        c.close()
