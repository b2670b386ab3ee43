import torch, numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from sfm_opencv_amd import api, synth
ctx = api.Context(0, use_torch_stream=True); st = ctx.torch_stream
dd = synth.sift_descriptor_chain(2, 10000, seed=1)
q = torch.from_numpy(dd[0]).cuda(); t = torch.from_numpy(dd[1]).cuda()
qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
outs = {ld: torch.empty((10000, ld), dtype=torch.float32, device="cuda") for ld in (10000, 10112)}
def run(ld, env):
    for k in ("SFMHIP_EXP_DISTMAT", "SFMHIP_EXP_BPW"): os.environ.pop(k, None)
    os.environ.update(env)
    out = outs[ld]
    for _ in range(3): ctx.l2_distance_matrix_dev(qs, ts, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(st)
    for _ in range(20): ctx.l2_distance_matrix_dev(qs, ts, out)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
variants = [("plain stores", {}), ("nt stores", {"SFMHIP_EXP_DISTMAT": "8"})]
res = {v[0]: {10000: [], 10112: []} for v in variants}
for rnd in range(3):
    for name, env in variants:
        for ld in (10000, 10112):
            res[name][ld].append(run(ld, env))
for name, _ in variants:
    print("%-16s ld10000 %.1f us (%.2f TB/s)   ld10112 %.1f us (%.2f TB/s)" % (name, 1e3 * min(res[name][10000]), 410.24e6 / min(res[name][10000]) / 1e9, 1e3 * min(res[name][10112]), 410.24e6 / min(res[name][10112]) / 1e9))
