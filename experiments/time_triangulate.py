"""triangulate2_dev timing: all scene points through cameras 0 and 1 vs only the points both cameras observe (tiled to the same count)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth
ctx = api.Context(0, use_torch_stream=True)
sc = synth.ba_scene(200, 300000)
Kc = np.array([[sc["K0"][0], 0, sc["K0"][2]], [0, sc["K0"][1], sc["K0"][3]], [0, 0, 1]], np.float64)
def proj(ext6):
    w = np.asarray(ext6[:3], np.float64); th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) if th < 1e-12 else np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
    return (Kc @ np.concatenate([R, np.asarray(ext6[3:6], np.float64).reshape(3, 1)], 1)).astype(np.float32)
ext = sc["ext_true"].reshape(-1, 6); P1, P2 = proj(ext[0]), proj(ext[1])
pts = np.asarray(sc["pts_true"], np.float64).reshape(-1, 3)
seen0 = set(sc["obs_pt"][sc["obs_cam"] == 0].tolist()); seen1 = set(sc["obs_pt"][sc["obs_cam"] == 1].tolist())
both = np.array(sorted(seen0 & seen1), np.int64)
print("points seen by both cameras:", both.size)
def run(X, label):
    n = X.shape[0]
    Xh = np.concatenate([X, np.ones((n, 1))], 1)
    def pix(P):
        h = Xh @ P.astype(np.float64).T
        return (h[:, :2] / h[:, 2:3]).astype(np.float32)
    a = torch.from_numpy(pix(P1)).cuda(); b = torch.from_numpy(pix(P2)).cuda()
    w = torch.empty((4, n), dtype=torch.float32, device="cuda"); x = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    for _ in range(5): ctx.triangulate2_dev(P1, P2, a, b, w, x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ctx.triangulate2_dev(P1, P2, a, b, w, x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    err = np.abs(x.cpu().numpy() - X).max()
    print(f"{label}: {n} points, {dt*1e3:.3f} ms per call = {n/dt/1e6:.0f} M points/s; max |xyz - truth| {err:.2e}")
run(pts, "all scene points")
run(np.tile(pts[both], (300000 // both.size + 1, 1))[:300000], "co-visible points tiled")
