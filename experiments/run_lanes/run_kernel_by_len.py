"""Linearisation phase by track length: scenes whose points all have L observations, per-observation kernels (linearizer 0) against
run lanes (3) and run tiles (2).  usage: python experiments/run_kernel_by_len.py [n_cam n_pt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_opencv_amd import api, synth
ctx = api.Context(0, use_torch_stream=True)
n_cam, n_pt = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 300000)
for L in (2, 3, 4, 5, 6):
    sc = synth.ba_scene(n_cam, n_pt, min_len=L, max_len=L)
    row = []
    for lin in (0, 3, 2):
        pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], ctx.ba_options(linearizer=lin))
        pb.iterate(3)
        ctx.set_kernel_timing(True); pb.iterate(10); ph = pb.phase_ms(); ctx.set_kernel_timing(False)
        row.append((lin, ph[0], ph[4]))
        pb.close()
    print(f"L={L} ({sc['n_obs']} obs): " + "  ".join(f"lin{l}: phase {a:.4f} kernel {k:.4f}" for l, a, k in row), flush=True)
