import torch, numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from sfm_opencv_amd import api, synth
ctx = api.Context(0, use_torch_stream=True); st = ctx.torch_stream
dd = synth.sift_descriptor_chain(2, 10000, seed=1)
q = torch.from_numpy(dd[0]).cuda(); t = torch.from_numpy(dd[1]).cuda()
qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
def run(ld, tag):
    out = torch.empty((10000, ld), dtype=torch.float32, device="cuda")
    for _ in range(3): ctx.l2_distance_matrix_dev(qs, ts, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(st)
    for _ in range(20): ctx.l2_distance_matrix_dev(qs, ts, out)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print(tag, "ld", ld, "ms %.4f  TB/s %.2f" % (ms, 410.24e6/ms/1e9))
run(10000, os.environ.get("SFMHIP_EXP_DISTMAT","0")+" bpw"+os.environ.get("SFMHIP_EXP_BPW","4"))
run(10112, os.environ.get("SFMHIP_EXP_DISTMAT","0")+" bpw"+os.environ.get("SFMHIP_EXP_BPW","4"))
# plain torch copy for reference HBM rate
a = torch.empty(100_000_000, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(st)
for _ in range(10): b.copy_(a)
e1.record(st); torch.cuda.synchronize()
print("torch copy 400MB->400MB ms %.4f (r+w TB/s %.2f)" % (e0.elapsed_time(e1)/10, 800e6/(e0.elapsed_time(e1)/10)/1e9))
e0.record(st)
for _ in range(10): b.fill_(1.0)
e1.record(st); torch.cuda.synchronize()
print("torch fill 400MB ms %.4f (w TB/s %.2f)" % (e0.elapsed_time(e1)/10, 400e6/(e0.elapsed_time(e1)/10)/1e9))
