"""Random-shape parity sweep of the matching entry points against the oracle (bit-exact indices and float32 distance bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
from sfm_opencv_amd import api, synth
ctx = api.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
sizes = [2, 3, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1000, 1023, 1025, 2047, 4097]
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    nq = int(rng.choice(sizes + [1])); nt = int(rng.choice(sizes))
    kind = trial % 6
    if kind == 0:      # SIFT-like integers
        q = synth._sift_like(rng, nq); t = synth._sift_like(rng, nt)
        if nq > 3 and nt > 3: t[: min(nq, nt) // 2] = np.clip(q[: min(nq, nt) // 2] + rng.integers(-2, 3, (min(nq, nt) // 2, 128)), 0, 255)
        gi, gd = ctx.knn2_l2(q, t); oi, od = orc.knn2_l2(q, t)
    elif kind == 1:    # few distinct values: many ties
        q = rng.integers(0, 3, (nq, 128)).astype(np.float32); t = rng.integers(0, 3, (nt, 128)).astype(np.float32)
        gi, gd = ctx.knn2_l2(q, t); oi, od = orc.knn2_l2(q, t)
    elif kind == 2:    # general floats
        dim = int(rng.choice([7, 64, 128, 130]))
        q = rng.standard_normal((nq, dim)).astype(np.float32); t = rng.standard_normal((nt, dim)).astype(np.float32)
        gi, gd = ctx.knn2_l2(q, t); oi, od = orc.knn2_l2(q, t)
    elif kind == 3:    # binary rows (<= 61 bytes: the FP4 matrix-core kernel; 62..64: the VALU kernel), up to three chunk windows
        nb = int(rng.choice([61, 61, 64, 62, 32, 17, 1, 60]))
        nt = int(rng.choice(sizes + [5000, 8191, 8193, 9000, 16500]))
        q = rng.integers(0, 256, (nq, nb), dtype=np.uint8); t = rng.integers(0, 256, (nt, nb), dtype=np.uint8)
        gi, gd = ctx.knn2_hamming2(q, t); oi, od = orc.knn2_hamming2(q, t)
    elif kind == 4:    # binary rows drawn from a handful of patterns + single-cell edits: ties everywhere, distances 0..3
        nb = int(rng.choice([61, 33, 8]))
        nt = int(rng.choice(sizes + [5000, 9000]))
        base = rng.integers(0, 256, (5, nb), dtype=np.uint8)
        q = base[rng.integers(0, 5, nq)].copy(); t = base[rng.integers(0, 5, nt)].copy()
        t[rng.integers(0, nt, nt // 3), rng.integers(0, nb, nt // 3)] ^= np.uint8(1) << rng.integers(0, 8, nt // 3).astype(np.uint8)
        gi, gd = ctx.knn2_hamming2(q, t); oi, od = orc.knn2_hamming2(q, t)
    else:              # a ragged chain through the batched entry point (several pairs of different sizes in one launch)
        nb = 61
        ns = [int(rng.choice([1, 2, 40, 300, 700, 1500, 2600])) for _ in range(4)]
        descs = [rng.integers(0, 256, (n, nb), dtype=np.uint8) for n in ns]
        for i in range(1, 4):      # plant true matches
            k = min(ns[i - 1], ns[i]) // 2
            descs[i][:k] = descs[i - 1][:k]
            if k: descs[i][:k, 0] ^= np.uint8(3)
        got = api.match_features_for_all(descs, ctx=ctx)
        ok = all(np.array_equal(g, orc.match_features_hamming2(descs[i], descs[i + 1])) for i, g in enumerate(got))
        if not ok:
            bad += 1
            print("MISMATCH chain", ns, flush=True)
        continue
    ok = np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH kind", kind, "nq", nq, "nt", nt, "idx diff rows", int((gi != oi).any(1).sum()), "dist diff rows", int((gd.view(np.uint32) != od.view(np.uint32)).any(1).sum()), flush=True)
print("trials done, mismatches:", bad)
