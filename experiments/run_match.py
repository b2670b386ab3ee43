"""C4 matching passes only (for rocprofv3 --pmc runs): python3 experiments/run_match.py [passes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth
n_img, n_desc = 40, 5000          # 39 chain pairs: same per-workgroup work as C4, shorter run
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = api.Context(0, use_torch_stream=True)
chain = synth.sift_descriptor_chain(n_img, n_desc)
keep = [torch.from_numpy(chain[i]).cuda() for i in range(n_img)]
sets = [ctx.descset_l2(t) for t in keep]
pairs = np.stack([np.arange(n_img - 1), np.arange(1, n_img)], 1).astype(np.int32)
d_matches = torch.zeros((n_img - 1, n_desc, 4), dtype=torch.int32, device="cuda")
d_counts = torch.zeros((n_img - 1,), dtype=torch.int32, device="cuda")
for _ in range(passes):
    ctx.match_pairs_dev(sets, pairs, d_matches, n_desc, d_counts)
torch.cuda.synchronize()
print("matches", int(d_counts.sum().item()))
