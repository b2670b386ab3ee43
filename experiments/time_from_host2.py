"""match_features_for_all from host matrices at C4 (200 x 5000): wall clock of the whole call, and of its parts"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth
n_img, n_desc = 200, 5000
ctx = api.Context(0)
pairs = np.stack([np.arange(n_img - 1), np.arange(1, n_img)], 1).astype(np.int32)
for name, chain in (("l2", synth.sift_descriptor_chain_mt(n_img, n_desc)), ("hamming2", synth.akaze_descriptor_chain_mt(n_img, n_desc))):
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sets = ctx.descsets_host(chain)
        t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
        out = ctx.match_pairs(sets, pairs)
        t3 = time.perf_counter()
        del sets
        t4 = time.perf_counter()
        got = api.match_features_for_all(chain, ctx=ctx)
        t5 = time.perf_counter()
        print(f"{name}: batched create {1e3*(t1-t0):.2f} ms (+ drain {1e3*(t2-t1):.2f}), match_pairs {1e3*(t3-t2):.2f}; match_features_for_all as one call {1e3*(t5-t4):.2f} ms, {sum(len(g) for g in got)} matches", flush=True)
