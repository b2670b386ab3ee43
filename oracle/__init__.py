"""oracle -- numpy-facing ctypes binding of oracle/liboracle.so (CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY (see oracle/orc.h): imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under sfm_opencv_amd/ may import this package.
Parity status: unpinned for matching / triangulation / BA (the reference ships no vectors and its
OpenCV/Ceres dependencies are absent); pinned for normals + writers by Viewer/structure_ba.{yml,ply}.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


class BAOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("huber_delta", C.c_double),
                ("jacobi_scaling", C.c_int), ("fix_first_camera", C.c_int), ("fix_intrinsics", C.c_int),
                ("verbose", C.c_int)]


class BASummary(C.Structure):
    _fields_ = [("termination", C.c_int), ("iterations", C.c_int), ("successful_steps", C.c_int),
                ("num_residuals", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double), ("final_gradient_max_norm", C.c_double),
                ("total_time_s", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        # OpenMP would start one thread per hardware thread of the HOST (hundreds on a GPU box whose container owns 16 cores):
        # spinning barriers oversubscribed like that take minutes for a one-second solve.  Default to the cores we may run on.
        try:
            _lib.orc_set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
        except (AttributeError, OSError):
            _lib.orc_set_num_threads(8)
        _lib.orc_ratio_filter.restype = C.c_int
        _lib.orc_ba_solve.restype = C.c_int
        _lib.orc_ba_reduced_system.restype = C.c_int
        _lib.orc_get_max_threads.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


DMATCH = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def max_threads():
    return lib().orc_get_max_threads()


def knn2_l2(q, t):
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    nq, dim = q.shape; nt = t.shape[0]
    idx = np.empty((nq, 2), np.int32); dist = np.empty((nq, 2), np.float32)
    lib().orc_knn2_l2_f32(_p(q, C.c_float), nq, _p(t, C.c_float), nt, dim, C.c_size_t(dim), C.c_size_t(dim),
                          _p(idx, C.c_int32), _p(dist, C.c_float))
    return idx, dist


def knn2_hamming2(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    nq, nb = q.shape; nt = t.shape[0]
    idx = np.empty((nq, 2), np.int32); dist = np.empty((nq, 2), np.float32)
    lib().orc_knn2_hamming2_u8(_p(q, C.c_uint8), nq, _p(t, C.c_uint8), nt, nb, C.c_size_t(nb), C.c_size_t(nb),
                               _p(idx, C.c_int32), _p(dist, C.c_float))
    return idx, dist


def l2_distance_matrix(q, t):
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    nq, dim = q.shape; nt = t.shape[0]
    d = np.empty((nq, nt), np.float32)
    lib().orc_l2_distance_matrix_f32(_p(q, C.c_float), nq, _p(t, C.c_float), nt, dim, C.c_size_t(dim),
                                     C.c_size_t(dim), _p(d, C.c_float), C.c_size_t(nt))
    return d


def ratio_filter(idx2, dist2, ratio=0.6, floor_=10.0, mult=5.0):
    idx2 = np.ascontiguousarray(idx2, np.int32); dist2 = np.ascontiguousarray(dist2, np.float32)
    nq = idx2.shape[0]
    out = np.zeros(max(nq, 1), DMATCH)
    n = lib().orc_ratio_filter(_p(idx2, C.c_int32), _p(dist2, C.c_float), nq, C.c_double(ratio),
                               C.c_float(floor_), C.c_float(mult), out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()


def match_features_l2(q, t):
    return ratio_filter(*knn2_l2(q, t))


def match_features_hamming2(q, t):
    return ratio_filter(*knn2_hamming2(q, t))


def projection_matrix(K, R, T):
    K = np.ascontiguousarray(K, np.float64).reshape(9); R = np.ascontiguousarray(R, np.float64).reshape(9)
    T = np.ascontiguousarray(T, np.float64).reshape(3)
    P = np.empty(12, np.float32)
    lib().orc_projection_matrix(_p(K, C.c_double), _p(R, C.c_double), _p(T, C.c_double), _p(P, C.c_float))
    return P.reshape(3, 4)


def triangulate2(P1, P2, xy1, xy2):
    P1 = np.ascontiguousarray(P1, np.float32).reshape(12); P2 = np.ascontiguousarray(P2, np.float32).reshape(12)
    xy1 = np.ascontiguousarray(xy1, np.float32); xy2 = np.ascontiguousarray(xy2, np.float32)
    n = xy1.shape[0]
    xyzw = np.empty((4, n), np.float32); xyz = np.empty((n, 3), np.float64)
    lib().orc_triangulate2(_p(P1, C.c_float), _p(P2, C.c_float), _p(xy1, C.c_float), _p(xy2, C.c_float), n,
                           _p(xyzw, C.c_float), _p(xyz, C.c_double))
    return xyzw, xyz


def ba_default_options(**kw):
    o = BAOptions()
    lib().orc_ba_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def triangulate_tracks(K4, ext, obs_cam, obs_pt, obs_uv, n_pt):
    K4 = np.ascontiguousarray(K4, np.float64); ext = np.ascontiguousarray(ext, np.float64)
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32); uv = np.ascontiguousarray(obs_uv, np.float64)
    pts = np.empty((n_pt, 3), np.float64); nv = np.empty(n_pt, np.int32)
    lib().orc_triangulate_tracks(_p(K4, C.c_double), _p(ext, C.c_double), ext.shape[0], _p(oc, C.c_int32), _p(op, C.c_int32),
                                 _p(uv, C.c_double), oc.shape[0], n_pt, _p(pts, C.c_double), _p(nv, C.c_int32))
    return pts, nv


def reprojection_errors(K4, ext, pts, obs_cam, obs_pt, obs_uv):
    K4 = np.ascontiguousarray(K4, np.float64); ext = np.ascontiguousarray(ext, np.float64); pts = np.ascontiguousarray(pts, np.float64)
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32); uv = np.ascontiguousarray(obs_uv, np.float64)
    err = np.empty(oc.shape[0], np.float64)
    lib().orc_reprojection_errors(_p(K4, C.c_double), _p(ext, C.c_double), ext.shape[0], _p(pts, C.c_double), _p(oc, C.c_int32),
                                  _p(op, C.c_int32), _p(uv, C.c_double), oc.shape[0], _p(err, C.c_double))
    return err


def reproject(K4, ext6, X, uv):
    K4 = np.ascontiguousarray(K4, np.float64); ext6 = np.ascontiguousarray(ext6, np.float64)
    X = np.ascontiguousarray(X, np.float64); uv = np.ascontiguousarray(uv, np.float64)
    r = np.empty(2); J = np.empty((2, 13))
    lib().orc_reproject(_p(K4, C.c_double), _p(ext6, C.c_double), _p(X, C.c_double), _p(uv, C.c_double),
                        _p(r, C.c_double), _p(J, C.c_double))
    return r, J


def _ba_args(K4, ext, pts, obs_cam, obs_pt, obs_uv):
    K4 = np.array(K4, np.float64).reshape(4).copy()
    ext = np.array(ext, np.float64).reshape(-1, 6).copy()
    pts = np.array(pts, np.float64).reshape(-1, 3).copy()
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
    uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2)
    return K4, ext, pts, oc, op, uv


def ba_solve(K4, ext, pts, obs_cam, obs_pt, obs_uv, opts=None, force_iterations=0, trace_cap=256):
    """Returns (K4, ext, pts, summary dict, trace dict). Inputs are not modified."""
    K4, ext, pts, oc, op, uv = _ba_args(K4, ext, pts, obs_cam, obs_pt, obs_uv)
    o = opts if opts is not None else ba_default_options()
    s = BASummary()
    tc = np.zeros(trace_cap); tr = np.zeros(trace_cap); tk = np.zeros(trace_cap, np.int32)
    rc = lib().orc_ba_solve(_p(K4, C.c_double), _p(ext, C.c_double), ext.shape[0], _p(pts, C.c_double), pts.shape[0],
                            _p(oc, C.c_int32), _p(op, C.c_int32), _p(uv, C.c_double), oc.shape[0],
                            C.byref(o), C.byref(s), int(force_iterations),
                            _p(tc, C.c_double), _p(tr, C.c_double), _p(tk, C.c_int32), trace_cap)
    if rc != 0:
        raise RuntimeError("orc_ba_solve failed")
    k = min(s.iterations, trace_cap)
    return K4, ext, pts, s.asdict(), {"cost": tc[:k].copy(), "radius": tr[:k].copy(), "accepted": tk[:k].copy()}


def ba_reduced_system(K4, ext, pts, obs_cam, obs_pt, obs_uv, radius, opts=None):
    K4, ext, pts, oc, op, uv = _ba_args(K4, ext, pts, obs_cam, obs_pt, obs_uv)
    o = opts if opts is not None else ba_default_options()
    args = (_p(K4, C.c_double), _p(ext, C.c_double), ext.shape[0], _p(pts, C.c_double), pts.shape[0],
            _p(oc, C.c_int32), _p(op, C.c_int32), _p(uv, C.c_double), oc.shape[0], C.byref(o), C.c_double(radius))
    n = lib().orc_ba_reduced_system(*args, None, None, None)
    if n < 0:
        raise RuntimeError("orc_ba_reduced_system failed")
    S = np.zeros((n, n)); rhs = np.zeros(n); cost = C.c_double(0)
    rc = lib().orc_ba_reduced_system(*args, _p(S, C.c_double), _p(rhs, C.c_double), C.byref(cost))
    if rc < 0:
        raise RuntimeError("orc_ba_reduced_system failed")
    return S, rhs, cost.value


def estimate_normals(pts, K=10):
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    out = np.empty_like(pts)
    lib().orc_estimate_normals(_p(pts, C.c_double), pts.shape[0], int(K), _p(out, C.c_double))
    return out
