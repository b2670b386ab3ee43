"""Wall time of the drop-in call sfmhip_ba_solve (create + LM to convergence + parameters back) and of sfmhip_ba_create alone,
with the create's own phase clock.  usage: python experiments/time_create.py [C3 C4 C5 ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_opencv_amd import api, synth

ctx = api.Context(0)
for name in (sys.argv[1:] or ["C3", "C4"]):
    cfg = synth.CONFIGS[name]
    t = time.perf_counter(); sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"]); tg = time.perf_counter() - t
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    for rep in range(3):
        t = time.perf_counter(); pb = ctx.ba_create(*args); tc = time.perf_counter() - t
        ms = pb.debug_table("setup_ms")
        t = time.perf_counter(); s = pb.run(); tr = time.perf_counter() - t
        t = time.perf_counter(); pb.params(); tp = time.perf_counter() - t
        t = time.perf_counter(); pb.close(); td = time.perf_counter() - t
        print(f"{name} rep {rep}: scene gen {tg:.2f}s | create {tc*1e3:.1f} ms (clock: {np.round(ms, 2)}) | run {tr*1e3:.1f} ms ({s['iterations']} it, pre {s['preprocessor_time_s']*1e3:.1f} ms) | params {tp*1e3:.1f} | destroy {td*1e3:.1f}", flush=True)
    K, e, p = sc["K0"].copy(), sc["ext0"].copy(), sc["pts0"].copy()
    t = time.perf_counter(); s = ctx.ba_solve(K, e, p, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])[3]; ts = time.perf_counter() - t
    print(f"{name} sfmhip_ba_solve: {ts*1e3:.1f} ms wall; summary total {s['total_time_s']*1e3:.1f} pre {s['preprocessor_time_s']*1e3:.1f} min {s['minimizer_time_s']*1e3:.1f} post {s['postprocessor_time_s']*1e3:.1f}; {s['iterations']} it, cost {s['initial_cost']:.4e} -> {s['final_cost']:.4e}", flush=True)
