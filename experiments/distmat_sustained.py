"""Distance-matrix kernel: time per block of 10 back-to-back launches over 150 launches (does the rate hold when sustained?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_opencv_amd import api, synth
ctx = api.Context(0, use_torch_stream=True)
nq = nt = 10000
dd = synth.sift_descriptor_chain(2, nq, seed=synth.SEED + 100000)
q = torch.from_numpy(dd[0]).cuda(); t = torch.from_numpy(dd[1]).cuda()
qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
stream = torch.cuda.current_stream()
for ld in (10000, 10016):
    buf = torch.empty((nq, ld), dtype=torch.float32, device="cuda"); out = buf[:, :nt]
    blocks = []
    for b in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(stream)
        for _ in range(10): ctx.l2_distance_matrix_dev(qs, ts, out)
        e1.record(stream); torch.cuda.synchronize()
        blocks.append(e0.elapsed_time(e1) / 10 * 1e3)
    print("stride", ld, "us per launch, blocks of 10:", " ".join("%.0f" % x for x in blocks), flush=True)
    time.sleep(0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(stream)
    for _ in range(10): ctx.l2_distance_matrix_dev(qs, ts, out)
    e1.record(stream); torch.cuda.synchronize()
    print("   after 0.5 s idle: %.0f" % (e0.elapsed_time(e1) / 10 * 1e3), flush=True)
    del out, buf
