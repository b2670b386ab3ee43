"""Where the from-host chain (match_features_for_all on host matrices) spends its time: descset creation loop / match call / list building."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth
import ctypes as C
n_img, n_desc = 200, 5000
ctx = api.Context(0)
for name, chain in (("hamming2", synth.akaze_descriptor_chain_mt(n_img, n_desc) if hasattr(synth, "akaze_descriptor_chain_mt") else synth.akaze_descriptor_chain(n_img, n_desc)),
                    ("l2", synth.sift_descriptor_chain_mt(n_img, n_desc) if hasattr(synth, "sift_descriptor_chain_mt") else synth.sift_descriptor_chain(n_img, n_desc))):
    mk = ctx.descset_hamming2 if name == "hamming2" else ctx.descset_l2
    pairs = np.stack([np.arange(n_img - 1), np.arange(1, n_img)], 1).astype(np.int32)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sets = [mk(d) for d in chain]
        t1 = time.perf_counter()
        ctx.synchronize()
        t2 = time.perf_counter()
        mpp = n_desc
        out = np.zeros((n_img - 1, mpp), api.DMATCH); counts = np.zeros(n_img - 1, np.int32)
        arr = (C.c_void_p * len(sets))(*[s.handle for s in sets])
        t3 = time.perf_counter()
        ctx._check(ctx.lib.sfmhip_match_pairs(ctx.h, arr, len(sets), pairs.ctypes.data, n_img - 1, 0.6, 10.0, 5.0, out.ctypes.data, mpp, counts.ctypes.data))
        t4 = time.perf_counter()
        res = [out[p, :counts[p]].copy() for p in range(n_img - 1)]
        t5 = time.perf_counter()
        del sets
        print(f"{name}: create loop {1e3*(t1-t0):.2f} ms (+ drain {1e3*(t2-t1):.2f}), output alloc {1e3*(t3-t2):.2f}, sfmhip_match_pairs {1e3*(t4-t3):.2f}, list building {1e3*(t5-t4):.2f}; total {1e3*(t5-t0):.2f} ms", flush=True)
