"""sfmhip_match_pairs_multi at C4 from host matrices: 1, 2, 3, 4 contexts on the box's one card (two host threads and two streams already
overlap one block's upload with the other's kernels; on a node every context also has its own PCIe link)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth
n_img, n_desc = 200, 5000
pairs = np.stack([np.arange(n_img - 1), np.arange(1, n_img)], 1).astype(np.int32)
for name, chain in (("l2", synth.sift_descriptor_chain_mt(n_img, n_desc)), ("hamming2", synth.akaze_descriptor_chain_mt(n_img, n_desc))):
    for n_ctx in (1, 2, 3, 4):
        ctxs = [api.Context(0, use_torch_stream=False) for _ in range(n_ctx)]
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            got = api.match_pairs_multi(ctxs, chain, pairs)
            ts.append(1e3 * (time.perf_counter() - t0))
        for c in ctxs:
            c.close()
        print(f"{name}: {n_ctx} context(s): {min(ts[1:]):.2f} ms (calls: {' '.join('%.2f' % t for t in ts)}), {sum(len(g) for g in got)} matches", flush=True)
