"""Seeded synthetic inputs for the matching and bundle-adjustment paths.

Descriptor sets and track sets are generated independently, as BASELINE.json's "synthetic
descriptor/track sets" wording allows.  numpy's PCG64 is used (not mt19937_64): the streams are
defined by this file, seed = 20240607 (+ image index for descriptors).

CONFIGS: the sizes named in BASELINE.json `configs`.
"""
import numpy as np

SEED = 20240607
# fx, fy, cx, cy of the reference's hard-coded K (NViewReconstuct.cpp:1353-1356)
K_REF = np.array([2826.561, 2826.519, 1835.259, 1370.103])

CONFIGS = {
    "C3": dict(n_img=50, n_desc=2000, n_pt=80_000),
    "C4": dict(n_img=200, n_desc=5000, n_pt=300_000),
    "C5": dict(n_img=1000, n_desc=10000, n_pt=2_000_000),
}


def _sift_like(rng, n, dim=128):
    """OpenCV-SIFT-shaped rows: |N(0,1)|, L2-normalise, clip 0.2, renormalise, min(255, floor(512 v))."""
    v = np.abs(rng.standard_normal((n, dim)))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v = np.minimum(v, 0.2)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return np.minimum(255.0, np.floor(512.0 * v)).astype(np.float32)


def sift_descriptor_chain(n_img, n_desc, dim=128, seed=SEED, overlap=0.6):
    """List of n_img integer-valued float32 (n_desc x dim) matrices; image i+1 is a random permutation
    of `overlap` rows copied from image i with U{-2..2} integer noise (clipped to [0,255]) plus fresh rows."""
    out = []
    prev = None
    for i in range(n_img):
        rng = np.random.default_rng(seed + i)
        if prev is None:
            cur = _sift_like(rng, n_desc, dim)
        else:
            n_copy = int(round(overlap * n_desc))
            src = rng.permutation(n_desc)[:n_copy]
            noise = rng.integers(-2, 3, size=(n_copy, dim)).astype(np.float32)
            copied = np.clip(prev[src] + noise, 0.0, 255.0)
            fresh = _sift_like(rng, n_desc - n_copy, dim)
            cur = np.concatenate([copied, fresh], axis=0)[rng.permutation(n_desc)]
        out.append(np.ascontiguousarray(cur, np.float32))
        prev = cur
    return out


def sift_descriptor_chain_device(n_img, n_desc, dim=128, seed=SEED, overlap=0.6, device="cuda"):
    """The construction of sift_descriptor_chain with torch on the device (its own random streams: same distribution, not the same
    numbers) -- for chains too large to generate on the host in bench time (C5: 1000 x 10000 x 128 floats = 5.1 GB)."""
    import torch
    g = torch.Generator(device=device)

    def sift_like(n):
        v = torch.randn((n, dim), generator=g, device=device, dtype=torch.float32).abs_()
        v /= torch.linalg.vector_norm(v, dim=1, keepdim=True)
        v.clamp_(max=0.2)
        v /= torch.linalg.vector_norm(v, dim=1, keepdim=True)
        return torch.clamp(torch.floor(512.0 * v), max=255.0)

    out, prev = [], None
    n_copy = int(round(overlap * n_desc))
    for i in range(n_img):
        g.manual_seed(seed + i)
        if prev is None:
            cur = sift_like(n_desc)
        else:
            src = torch.randperm(n_desc, generator=g, device=device)[:n_copy]
            noise = torch.randint(-2, 3, (n_copy, dim), generator=g, device=device).to(torch.float32)
            copied = torch.clamp(prev[src] + noise, 0.0, 255.0)
            cur = torch.cat([copied, sift_like(n_desc - n_copy)], 0)[torch.randperm(n_desc, generator=g, device=device)]
        out.append(cur.contiguous())
        prev = cur
    return out


def akaze_descriptor_chain_device(n_img, n_desc, nbytes=61, seed=SEED, overlap=0.6, flip=0.03, device="cuda"):
    """akaze_descriptor_chain's construction with torch on the device (own random streams)."""
    import torch
    g = torch.Generator(device=device)
    out, prev = [], None
    n_copy = int(round(overlap * n_desc))
    w = (2 ** torch.arange(8, device=device, dtype=torch.int32)).view(1, 1, 8)
    for i in range(n_img):
        g.manual_seed(seed + 7919 + i)
        if prev is None:
            cur = torch.randint(0, 256, (n_desc, nbytes), generator=g, device=device, dtype=torch.int32).to(torch.uint8)
        else:
            src = torch.randperm(n_desc, generator=g, device=device)[:n_copy]
            bits = (torch.rand((n_copy, nbytes, 8), generator=g, device=device) < flip).to(torch.int32)
            mask = (bits * w).sum(2).to(torch.uint8)
            copied = prev[src] ^ mask
            fresh = torch.randint(0, 256, (n_desc - n_copy, nbytes), generator=g, device=device, dtype=torch.int32).to(torch.uint8)
            cur = torch.cat([copied, fresh], 0)[torch.randperm(n_desc, generator=g, device=device)]
        out.append(cur.contiguous())
        prev = cur
    return out


def akaze_descriptor_chain(n_img, n_desc, nbytes=61, seed=SEED, overlap=0.6, flip=0.03):
    """Binary (AKAZE MLDB-486-like) rows: 61 random bytes; copies get 3 % of their bits flipped."""
    out = []
    prev = None
    for i in range(n_img):
        rng = np.random.default_rng(seed + 7919 + i)
        if prev is None:
            cur = rng.integers(0, 256, size=(n_desc, nbytes), dtype=np.uint8)
        else:
            n_copy = int(round(overlap * n_desc))
            src = rng.permutation(n_desc)[:n_copy]
            bits = (rng.random((n_copy, nbytes, 8)) < flip)
            mask = np.packbits(bits, axis=2).reshape(n_copy, nbytes)
            copied = prev[src] ^ mask
            fresh = rng.integers(0, 256, size=(n_desc - n_copy, nbytes), dtype=np.uint8)
            cur = np.concatenate([copied, fresh], axis=0)[rng.permutation(n_desc)]
        out.append(np.ascontiguousarray(cur, np.uint8))
        prev = cur
    return out


def rotmat_to_angle_axis(R):
    """Rodrigues log (cv::Rodrigues 3x3 -> 3x1, NViewReconstuct.cpp:1480)."""
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-12:
        return 0.5 * w
    if np.pi - th < 1e-6:
        A = (R + np.eye(3)) / 2.0
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(ax))
        ax = A[k] / ax[k]
        ax /= np.linalg.norm(ax)
        return th * ax
    return th * w / (2.0 * np.sin(th))


def angle_axis_to_rotmat(aa):
    aa = np.asarray(aa, np.float64)
    th = np.linalg.norm(aa)
    if th < 1e-15:
        return np.eye(3)
    w = aa / th
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def _cross(a, b):
    """np.cross for (m,3) arrays, same arithmetic (one multiply pair and a subtract per component), without its overhead"""
    out = np.empty_like(b)
    out[:, 0] = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    out[:, 1] = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    out[:, 2] = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    return out


def project(K4, ext6, X):
    """ReprojectCost's forward model (NViewReconstuct.cpp:151-177) for arrays: ext6 (m,6), X (m,3) -> (m,2)."""
    aa = ext6[:, :3]; t = ext6[:, 3:]
    th = np.sqrt((aa[:, 0] * aa[:, 0] + aa[:, 1] * aa[:, 1] + aa[:, 2] * aa[:, 2]))[:, None]
    safe = np.where(th > 0, th, 1.0)
    w = aa / safe
    c = np.cos(th); s = np.sin(th)
    wx = (w[:, 0] * X[:, 0] + w[:, 1] * X[:, 1] + w[:, 2] * X[:, 2])[:, None]
    p = X * c + _cross(w, X) * s + w * (wx * (1 - c))
    p = np.where(th * th > np.finfo(np.float64).eps, p, X + _cross(aa, X)) + t
    return np.stack([K4[0] * p[:, 0] / p[:, 2] + K4[2], K4[1] * p[:, 1] / p[:, 2] + K4[3]], axis=1)


def ba_scene(n_cam, n_pt, seed=SEED, noise_px=0.5, outlier_frac=0.02, min_len=2, max_len=6,
             perturb=True):
    """Ring scene of SURVEY 8d: cameras on a ring of radius 10 (+-1 radial and +-2 height modulation,
    added to remove the planar-turntable degeneracy that leaves fy unobservable) looking at the origin,
    points uniform in a radius-3 ball, each seen by L ~ U{2..6} consecutive cameras (mean 4 => n_obs = 4 n_pt),
    0.5 px noise, 2 % gross outliers, perturbed start (cam 0 exact).  Returns dict with truth and initial parameters and the observation list
    sorted by (camera, point) -- the order bundle_adjustment adds residual blocks (NView:1187-1197)."""
    rng = np.random.default_rng(seed)
    ext_true = np.zeros((n_cam, 6))
    for c in range(n_cam):
        phi = 2.0 * np.pi * c / n_cam
        rad = 10.0 + np.cos(5.0 * phi)
        C = np.array([rad * np.sin(phi), 2.0 * np.sin(3.0 * phi), -rad * np.cos(phi)])
        z = -C / np.linalg.norm(C)
        x = np.cross(np.array([0.0, 1.0, 0.0]), z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        ext_true[c, :3] = rotmat_to_angle_axis(R)
        ext_true[c, 3:] = -R @ C
    d = rng.standard_normal((n_pt, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts_true = d * (3.0 * rng.random((n_pt, 1)) ** (1.0 / 3.0))
    max_len = min(max_len, n_cam); min_len = min(min_len, max_len)
    L = rng.integers(min_len, max_len + 1, size=n_pt)
    start = (rng.random(n_pt) * (n_cam - L + 1)).astype(np.int64)
    obs_pt = np.repeat(np.arange(n_pt), L)
    offs = np.arange(L.sum()) - np.repeat(np.cumsum(L) - L, L)
    obs_cam = np.repeat(start, L) + offs
    uv = project(K_REF, ext_true[obs_cam], pts_true[obs_pt])
    uv += noise_px * rng.standard_normal(uv.shape)
    n_obs = obs_pt.shape[0]
    out = rng.random(n_obs) < outlier_frac
    uv[out] += rng.uniform(-50.0, 50.0, size=(int(out.sum()), 2))
    order = np.lexsort((obs_pt, obs_cam))
    obs_cam = obs_cam[order].astype(np.int32); obs_pt = obs_pt[order].astype(np.int32); uv = uv[order]
    K0 = K_REF.copy(); ext0 = ext_true.copy(); pts0 = pts_true.copy()
    if perturb:
        K0 = K_REF * 1.01
        ext0[1:, :3] += 0.01 * rng.standard_normal((n_cam - 1, 3))
        ext0[1:, 3:] += 0.05 * rng.standard_normal((n_cam - 1, 3))
        pts0 = pts_true + 0.05 * rng.standard_normal((n_pt, 3))
    return dict(K_true=K_REF.copy(), ext_true=ext_true, pts_true=pts_true,
                K0=K0, ext0=ext0, pts0=pts0,
                obs_cam=obs_cam, obs_pt=obs_pt, obs_uv=np.ascontiguousarray(uv),
                n_cam=n_cam, n_pt=n_pt, n_obs=int(n_obs))


def two_view_scene(n, seed=SEED, noise_px=0.3):
    """Two cameras of the reference's K, baseline along x, n points in front: float32 P1,P2 and float32 pixels."""
    rng = np.random.default_rng(seed + 31)
    Kfull = np.array([[K_REF[0], 0, K_REF[2]], [0, K_REF[1], K_REF[3]], [0, 0, 1.0]])
    R1 = np.eye(3); T1 = np.zeros(3)
    R2 = angle_axis_to_rotmat(np.array([0.02, -0.15, 0.01])); T2 = np.array([-1.0, 0.05, 0.1])
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(6, 14, n)], axis=1)
    def proj(R, T):
        p = X @ R.T + T
        return np.stack([K_REF[0] * p[:, 0] / p[:, 2] + K_REF[2], K_REF[1] * p[:, 1] / p[:, 2] + K_REF[3]], 1)
    xy1 = (proj(R1, T1) + noise_px * rng.standard_normal((n, 2))).astype(np.float32)
    xy2 = (proj(R2, T2) + noise_px * rng.standard_normal((n, 2))).astype(np.float32)
    return dict(K=Kfull, R1=R1, T1=T1, R2=R2, T2=T2, X=X, xy1=xy1, xy2=xy2)


# ------------------------------------------------------------------------------------------------
# SURVEY 8d names std::mt19937_64 (seed 20240607 + i): the same constructions in C++ (host/sfm_synth.cpp -> libsfmsynth.so).
# bench.py uses these; the tests keep the numpy streams above, whose scenes their fixed expectations were written for.
# ------------------------------------------------------------------------------------------------
_synth_lib = None


def _mt_lib():
    global _synth_lib
    if _synth_lib is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsfmsynth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: make -C sfm_opencv_amd/host (or __graft_entry__.build())")
        lib = C.CDLL(path)
        lib.sfmsynth_sift_chain.restype = C.c_int
        lib.sfmsynth_sift_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_void_p]
        lib.sfmsynth_akaze_chain.restype = C.c_int
        lib.sfmsynth_akaze_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_void_p]
        lib.sfmsynth_ba_scene.restype = C.c_longlong
        lib.sfmsynth_ba_scene.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 9
        _synth_lib = lib
    return _synth_lib


def sift_descriptor_chain_mt(n_img, n_desc, dim=128, seed=SEED, overlap=0.6):
    """sift_descriptor_chain on std::mt19937_64 (seed + i per image), generated by libsfmsynth.so"""
    out = np.empty((n_img, n_desc, dim), np.float32)
    if _mt_lib().sfmsynth_sift_chain(n_img, n_desc, dim, seed, overlap, out.ctypes.data) != 0:
        raise RuntimeError("sfmsynth_sift_chain failed")
    return [out[i] for i in range(n_img)]


def akaze_descriptor_chain_mt(n_img, n_desc, nbytes=61, seed=SEED, overlap=0.6, flip=0.03):
    out = np.empty((n_img, n_desc, nbytes), np.uint8)
    if _mt_lib().sfmsynth_akaze_chain(n_img, n_desc, nbytes, seed, overlap, flip, out.ctypes.data) != 0:
        raise RuntimeError("sfmsynth_akaze_chain failed")
    return [out[i] for i in range(n_img)]


def ba_scene_mt(n_cam, n_pt, seed=SEED, noise_px=0.5, outlier_frac=0.02, min_len=2, max_len=6):
    """ba_scene on std::mt19937_64: same construction and the same dict"""
    lib = _mt_lib()
    null = None
    n_obs = lib.sfmsynth_ba_scene(n_cam, n_pt, seed, noise_px, outlier_frac, min_len, max_len, *([null] * 9))
    if n_obs < 0:
        raise RuntimeError("sfmsynth_ba_scene failed")
    Kt = np.empty(4); et = np.empty((n_cam, 6)); pt = np.empty((n_pt, 3)); K0 = np.empty(4); e0 = np.empty((n_cam, 6)); p0 = np.empty((n_pt, 3))
    oc = np.empty(n_obs, np.int32); op = np.empty(n_obs, np.int32); uv = np.empty((n_obs, 2))
    got = lib.sfmsynth_ba_scene(n_cam, n_pt, seed, noise_px, outlier_frac, min_len, max_len, Kt.ctypes.data, et.ctypes.data, pt.ctypes.data,
                                K0.ctypes.data, e0.ctypes.data, p0.ctypes.data, oc.ctypes.data, op.ctypes.data, uv.ctypes.data)
    assert got == n_obs
    return dict(K_true=Kt, ext_true=et, pts_true=pt, K0=K0, ext0=e0, pts0=p0, obs_cam=oc, obs_pt=op, obs_uv=uv, n_cam=n_cam, n_pt=n_pt, n_obs=int(n_obs))
