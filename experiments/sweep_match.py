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
    kind = trial % 4
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
    else:              # binary rows
        nb = int(rng.choice([61, 64, 32, 17]))
        q = rng.integers(0, 256, (nq, nb), dtype=np.uint8); t = rng.integers(0, 256, (nt, nb), dtype=np.uint8)
        gi, gd = ctx.knn2_hamming2(q, t); oi, od = orc.knn2_hamming2(q, t)
    ok = np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH kind", kind, "nq", nq, "nt", nt, "idx diff rows", int((gi != oi).any(1).sum()), "dist diff rows", int((gd.view(np.uint32) != od.view(np.uint32)).any(1).sum()), flush=True)
print("trials done, mismatches:", bad)
