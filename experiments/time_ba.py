"""LM iteration time by linearizer and configuration: host clock over forced iterations, then the library's phase events.
usage: python experiments/time_ba.py [C4 C5 ...] [--lin=0,2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_opencv_amd import api, synth
names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["C4"]
lins = [0, 2]
for a in sys.argv[1:]:
    if a.startswith("--lin="):
        lins = [int(x) for x in a[6:].split(",")]
ctx = api.Context(0, use_torch_stream=True)
for name in names:
    cfg = synth.CONFIGS[name]
    sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
    for lin in lins:
        pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], ctx.ba_options(linearizer=lin))
        pb.iterate(3)
        best = 1e9
        for rep in range(3):
            pb.reset(); pb.iterate(3); ctx.synchronize()
            t = time.perf_counter(); s = pb.iterate(20); ctx.synchronize(); best = min(best, (time.perf_counter() - t) / 20)
        ctx.set_kernel_timing(True)
        pb.reset(); pb.iterate(3); pb.iterate(20); ph = pb.phase_ms()
        ctx.set_kernel_timing(False)
        print(f"{name} linearizer={lin}: {best*1e3:.4f} ms/iteration (host clock, best of 3 x 20); phases: linearise {ph[0]:.4f} solve {ph[1]:.4f} back {ph[2]:.4f} total {ph[3]:.4f}; "
              f"kernels [4]={ph[4]:.4f} [5]={ph[5]:.4f} leaf level {ph[6]:.4f}; cost {s['final_cost']:.6e}", flush=True)
        pb.close()
