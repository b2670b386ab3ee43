"""AKAZE-like (61-byte, NORM_HAMMING2) matching pass timing: 40 images x 5000 descriptors, 39 chain pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth
n_img, n_desc = 40, 5000
ctx = api.Context(0, use_torch_stream=True)
chain = synth.akaze_descriptor_chain(n_img, n_desc)
keep = [torch.from_numpy(chain[i]).cuda() for i in range(n_img)]
sets = [ctx.descset_hamming2(t) for t in keep]
pairs = np.stack([np.arange(n_img - 1), np.arange(1, n_img)], 1).astype(np.int32)
d_matches = torch.zeros((n_img - 1, n_desc, 4), dtype=torch.int32, device="cuda")
d_counts = torch.zeros((n_img - 1,), dtype=torch.int32, device="cuda")
ctx.set_kernel_timing(True)
for _ in range(3):
    ctx.match_pairs_dev(sets, pairs, d_matches, n_desc, d_counts)
torch.cuda.synchronize()
km = ctx.match_kernel_ms()
print("hamming2 kNN kernel %.3f ms per 39-pair pass (%.1f us per pair), merge %.3f ms, matches %d" % (km[0], 1e3 * km[0] / 39, km[1], int(d_counts.sum().item())))
