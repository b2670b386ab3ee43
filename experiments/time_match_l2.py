"""kNN kernel time of the C4 SIFT / L2 chain with the library named by SFMHIP_LIB: python3 experiments/time_match_l2.py [n_img] [n_desc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_desc = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ctx = api.Context(0, use_torch_stream=True)
chain = synth.sift_descriptor_chain_device(n_img, n_desc)
sets = [ctx.descset_l2(t) for t in chain]
pairs = np.stack([np.arange(n_img - 1), np.arange(1, n_img)], 1).astype(np.int32)
d_matches = torch.zeros((n_img - 1, n_desc, 4), dtype=torch.int32, device="cuda")
d_counts = torch.zeros((n_img - 1,), dtype=torch.int32, device="cuda")
for _ in range(5):
    ctx.match_pairs_dev(sets, pairs, d_matches, n_desc, d_counts)
torch.cuda.synchronize()
ctx.set_kernel_timing(True)
for _ in range(8):
    ctx.match_pairs_dev(sets, pairs, d_matches, n_desc, d_counts)
torch.cuda.synchronize()
k, m, calls, _ = ctx.match_kernel_ms()
print(os.environ.get("SFMHIP_LIB", "default"), "knn kernel %.4f ms  merge+rescore %.4f ms  (%d launches)  matches %d" % (k, m, calls, int(d_counts.sum().item())))
