"""Run-to-run determinism of the matching passes: the C4 L2 and Hamming2 chains N times, every pass's match lists (counts + records) must reproduce bit for bit.
usage: python experiments/soak_match.py [passes]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_img, n_desc = 200, 5000
ctx = api.Context(0, use_torch_stream=True)
pairs = np.stack([np.arange(n_img - 1), np.arange(1, n_img)], 1).astype(np.int32)
for name, chain, mk in (("L2", synth.sift_descriptor_chain_device(n_img, n_desc), ctx.descset_l2), ("Hamming2", synth.akaze_descriptor_chain_device(n_img, n_desc), ctx.descset_hamming2)):
    sets = [mk(t) for t in chain]
    d_m = torch.zeros((n_img - 1, n_desc, 4), dtype=torch.int32, device="cuda"); d_c = torch.zeros((n_img - 1,), dtype=torch.int32, device="cuda")
    ref = None; bad = 0
    for it in range(passes):
        d_m.zero_()
        if name == "L2": ctx.refresh_descsets(sets)
        ctx.match_pairs_dev(sets, pairs, d_m, n_desc, d_c)
        torch.cuda.synchronize()
        h = hashlib.sha256(d_c.cpu().numpy().tobytes() + d_m.cpu().numpy().tobytes()).hexdigest()
        if ref is None: ref = h
        elif h != ref: bad += 1
    print(f"{name}: {passes} passes of {n_img - 1} pairs, {int(d_c.sum().item())} matches per pass, digest {ref[:16]}, mismatching passes: {bad}", flush=True)
    del sets
