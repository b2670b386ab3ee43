"""Host-side time of the two calls of one C4 L2 matching pass (refresh + match_pairs_dev) against the pass time: is the host keeping ahead of the GPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth
n_img, n_desc = 200, 5000
ctx = api.Context(0, use_torch_stream=True)
chain = synth.sift_descriptor_chain_device(n_img, n_desc)
sets = [ctx.descset_l2(t) for t in chain]
pairs = np.stack([np.arange(n_img - 1), np.arange(1, n_img)], 1).astype(np.int32)
h_m = torch.zeros((n_img - 1, n_desc, 4), dtype=torch.int32).pin_memory(); h_c = torch.zeros((n_img - 1,), dtype=torch.int32).pin_memory()
for rep in range(3):
    for _ in range(5):
        ctx.refresh_descsets(sets); ctx.match_pairs_dev(sets, pairs, h_m, n_desc, h_c)
    torch.cuda.synchronize()
    tr = tm = 0.0
    t0 = time.perf_counter()
    for _ in range(20):
        a = time.perf_counter(); ctx.refresh_descsets(sets); b = time.perf_counter(); ctx.match_pairs_dev(sets, pairs, h_m, n_desc, h_c); c = time.perf_counter()
        tr += b - a; tm += c - b
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("pass %.3f ms (enqueue loop %.3f ms per pass: refresh call %.3f, match call %.3f; final drain %.3f ms)" % ((t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3, tr / 20 * 1e3, tm / 20 * 1e3, (t2 - t1) * 1e3), flush=True)
