"""sfmhip_ba_solve_multi against sfmhip_ba_solve on the same host arrays (contexts share the box's one card: what is measured is the
host side -- sharding, shard construction, teardown -- not the exchange, which is host-staged here): the time split of both calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_opencv_amd import api, synth
ctx = api.Context(0)
for name, (nc, npt) in (("C4", (200, 300000)), ("C5", (1000, 2000000))):
    sc = synth.ba_scene_mt(nc, npt)
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    o = ctx.ba_options(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    for rep in range(3):
        t0 = time.perf_counter(); K, e, p, s1 = ctx.ba_solve(*args, opts=o); w1 = time.perf_counter() - t0
    for n_ctx in (2, 4):
        ctxs = [api.Context(0, use_torch_stream=False) for _ in range(n_ctx)]
        for rep in range(3):
            ov = ctx.ba_options(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0, verbose=1 if rep == 2 else 0)
            t0 = time.perf_counter(); K2, e2, p2, s2 = api.ba_solve_multi(ctxs, *args, opts=ov); w2 = time.perf_counter() - t0
        for c in ctxs:
            c.close()
        print(f"{name}: single: wall {1e3*w1:.1f} ms, preprocessor {1e3*s1['preprocessor_time_s']:.2f}, minimizer {1e3*s1['minimizer_time_s']:.2f}, post {1e3*s1['postprocessor_time_s']:.2f} | "
              f"{n_ctx} contexts on one card: wall {1e3*w2:.1f} ms, preprocessor {1e3*s2['preprocessor_time_s']:.2f} (sharding + slowest shard's construction + plan), "
              f"minimizer {1e3*s2['minimizer_time_s']:.2f}, post {1e3*s2['postprocessor_time_s']:.2f}; cost rel diff {abs(s2['final_cost']-s1['final_cost'])/s1['final_cost']:.2e}", flush=True)
