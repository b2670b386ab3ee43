"""Python binding of the C-ABI (include/sfmhip.h) + a mirror of the reference's hot-path functions.

Names and argument meaning follow OpenCV_SFM/NViewReconstuct.cpp: match_features (873), match_features_for_all (850),
get_matched_points (989), reconstruct (1117), bundle_adjustment (1162).  cv::Mat / std::vector arguments become numpy
arrays (host entry points) or torch CUDA tensors (device entry points); outputs are returned instead of filled.
Everything here calls libsfmhip.so; nothing falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BAOptions, BASummary, SfmHipError

DMATCH = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


def _ptr(a):
    """address of a numpy array or a torch tensor"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()


class DescSet:
    def __init__(self, ctx, handle, rows, keepalive=None):
        self.ctx, self.handle, self.rows, self._keep = ctx, handle, rows, keepalive

    def info(self):
        kind, rows, dim, ex = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.ctx._check(self.ctx.lib.sfmhip_descset_info(self.handle, kind, rows, dim, ex))
        return dict(kind=kind.value, rows=rows.value, dim=dim.value, exact_u8=bool(ex.value))

    def refresh(self):
        self.ctx._check(self.ctx.lib.sfmhip_descset_refresh(self.handle))

    def close(self):
        if self.handle and self.ctx.h:      # the context owns the stream: never touch a set after its context died
            self.ctx.lib.sfmhip_descset_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One per process / GPU (sfmhip_create)."""

    def __init__(self, device=0, use_torch_stream=False):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.sfmhip_create(int(device), C.byref(h))
        if rc != 0:
            raise SfmHipError(f"sfmhip_create(device={device}) failed with {rc} "
                              "(no usable gfx950 device; there is no CPU fallback)")
        self.h = h
        self.device = device
        self.torch_stream = None
        if use_torch_stream:
            import torch
            st = torch.cuda.current_stream(device)
            if st.cuda_stream == 0:
                # the legacy default stream has handle 0, which the C-ABI reads as "use the context's own stream":
                # make a real stream current so that torch events / copies / collectives and the library share it
                st = torch.cuda.Stream(device=device)
                torch.cuda.set_stream(st)
            self.torch_stream = st
            self.set_stream(st.cuda_stream)

    def _check(self, rc):
        if rc != 0:
            raise SfmHipError(f"libsfmhip error {rc}: {self.lib.sfmhip_last_error(self.h).decode()}")

    def set_stream(self, stream_ptr):
        self._check(self.lib.sfmhip_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def synchronize(self):
        self._check(self.lib.sfmhip_synchronize(self.h))

    # ---------------------------------------------------------------- RCCL (multi-GPU)
    def rccl_available(self):
        return bool(self.lib.sfmhip_rccl_available())

    def rccl_unique_id(self):
        """128 opaque bytes from ncclGetUniqueId (rank 0; ship them to the other ranks)"""
        buf = (C.c_char * 128)()
        self._check(self.lib.sfmhip_rccl_get_unique_id(buf))
        return bytes(buf)

    def rccl_comm_create(self, unique_id, rank, world):
        comm = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._check(self.lib.sfmhip_rccl_comm_create(self.h, buf, int(rank), int(world), C.byref(comm)))
        return comm

    def rccl_comm_destroy(self, comm):
        self._check(self.lib.sfmhip_rccl_comm_destroy(comm))

    def rccl_allreduce_f64(self, comm, dev_ptr, count):
        self._check(self.lib.sfmhip_rccl_allreduce_f64(self.h, comm, C.c_void_p(int(dev_ptr)), int(count)))

    def trim(self):
        """release the device blocks the context keeps from destroyed BA problems (sfmhip_trim)"""
        self._check(self.lib.sfmhip_trim(self.h))

    def set_kernel_timing(self, enable=True):
        """bracket every kNN launch sequence with HIP events (sfmhip_set_kernel_timing)"""
        self._check(self.lib.sfmhip_set_kernel_timing(self.h, 1 if enable else 0))

    def match_kernel_ms(self):
        """[kNN kernel ms, merge + re-score ms, calls averaged, 0] since the last query (synchronises)"""
        out = (C.c_double * 4)()
        self._check(self.lib.sfmhip_match_kernel_ms(self.h, out))
        return list(out)

    def close(self):
        if self.h:
            self.lib.sfmhip_destroy(self.h)
            self.h = None

    # ---------------------------------------------------------------- matching
    def descset_l2(self, desc):
        """desc: numpy float32 (rows, dim) on the host, or a torch float32 CUDA tensor (borrowed, must stay alive)."""
        out = C.c_void_p()
        if isinstance(desc, np.ndarray):
            d = np.ascontiguousarray(desc, np.float32)
            rows, dim = d.shape
            self._check(self.lib.sfmhip_descset_create_l2_host(self.h, d.ctypes.data, rows, dim, dim, C.byref(out)))
            return DescSet(self, out, rows)
        assert desc.is_cuda and desc.dim() == 2 and desc.stride(1) == 1
        rows, dim = desc.shape
        self._check(self.lib.sfmhip_descset_create_l2_dev(self.h, desc.data_ptr(), rows, dim, desc.stride(0), C.byref(out)))
        return DescSet(self, out, rows, keepalive=desc)

    def descset_hamming2(self, desc):
        out = C.c_void_p()
        if isinstance(desc, np.ndarray):
            d = np.ascontiguousarray(desc, np.uint8)
            rows, nb = d.shape
            self._check(self.lib.sfmhip_descset_create_hamming2_host(self.h, d.ctypes.data, rows, nb, nb, C.byref(out)))
            return DescSet(self, out, rows)
        assert desc.is_cuda and desc.dim() == 2 and desc.stride(1) == 1
        rows, nb = desc.shape
        self._check(self.lib.sfmhip_descset_create_hamming2_dev(self.h, desc.data_ptr(), rows, nb, desc.stride(0), C.byref(out)))
        return DescSet(self, out, rows, keepalive=desc)

    def descsets_host(self, mats):
        """Many host matrices (float32: L2 / uint8: Hamming2; one row length) in ONE call: sfmhip_descsets_create_{l2,hamming2}_host."""
        if len(mats) == 0:
            return []
        ham = np.asarray(mats[0]).dtype == np.uint8
        dt = np.uint8 if ham else np.float32
        arrs = [np.asarray(m, dt) for m in mats]
        arrs = [a if a.ndim == 2 and a.strides[1] == a.itemsize and a.strides[0] % a.itemsize == 0 and a.strides[0] >= a.shape[1] * a.itemsize
                else np.ascontiguousarray(a) for a in arrs]
        dim = arrs[0].shape[1]
        assert all(a.shape[1] == dim for a in arrs)
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        rows = np.array([a.shape[0] for a in arrs], np.int32)
        ld = (C.c_size_t * n)(*[a.strides[0] // a.itemsize for a in arrs])
        out = (C.c_void_p * n)()
        fn = self.lib.sfmhip_descsets_create_hamming2_host if ham else self.lib.sfmhip_descsets_create_l2_host
        self._check(fn(self.h, ptrs, rows.ctypes.data, dim, ld, n, out))
        return [DescSet(self, C.c_void_p(out[i]), int(rows[i])) for i in range(n)]

    def refresh_descsets(self, sets):
        """re-run the preparation pass of many sets in one launch (enqueues only)"""
        arr = (C.c_void_p * len(sets))(*[s.handle for s in sets])
        self._check(self.lib.sfmhip_descsets_refresh(self.h, arr, len(sets)))

    def knn2_dev(self, qset, tset, idx2, dist2, force_path=0):
        """idx2 (nq,2) int32 / dist2 (nq,2) float32 torch CUDA tensors; enqueues only."""
        self._check(self.lib.sfmhip_knn2_dev(self.h, qset.handle, tset.handle, idx2.data_ptr(), dist2.data_ptr(), force_path))

    def knn2_l2(self, q, t):
        q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
        nq, dim = q.shape; nt = t.shape[0]
        idx = np.empty((nq, 2), np.int32); dist = np.empty((nq, 2), np.float32)
        self._check(self.lib.sfmhip_knn2_l2_f32(self.h, q.ctypes.data, nq, t.ctypes.data, nt, dim, dim, dim,
                                                idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def knn2_hamming2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        nq, nb = q.shape; nt = t.shape[0]
        idx = np.empty((nq, 2), np.int32); dist = np.empty((nq, 2), np.float32)
        self._check(self.lib.sfmhip_knn2_hamming2_u8(self.h, q.ctypes.data, nq, t.ctypes.data, nt, nb, nb, nb,
                                                     idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def match_pairs(self, sets, pairs, ratio=0.6, floor_=10.0, mult=5.0):
        """sets: list of DescSet; pairs: (n_pairs, 2) int.  Returns list of DMATCH arrays (host)."""
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        n_pairs = pairs.shape[0]
        if n_pairs == 0:
            return []
        mpp = max(1, max(sets[a].rows for a in pairs[:, 0]))
        out = np.empty((n_pairs, mpp), DMATCH)        # (only out[p, :counts[p]] is ever handed out: zero-filling 16 MB cost ~1 ms of a 4 ms chain)
        counts = np.zeros(n_pairs, np.int32)
        arr = (C.c_void_p * len(sets))(*[s.handle for s in sets])
        self._check(self.lib.sfmhip_match_pairs(self.h, arr, len(sets), pairs.ctypes.data, n_pairs,
                                                ratio, floor_, mult, out.ctypes.data, mpp, counts.ctypes.data))
        # views into the one result buffer (a copy per pair cost 3.8 ms of a 13 ms C4 chain from host rows)
        return [out[p, :c] for p, c in enumerate(counts.tolist())]

    def match_pairs_dev(self, sets, pairs, d_matches, max_per_pair, d_counts, ratio=0.6, floor_=10.0, mult=5.0):
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        arr = (C.c_void_p * len(sets))(*[s.handle for s in sets])
        self._check(self.lib.sfmhip_match_pairs_dev(self.h, arr, len(sets), pairs.ctypes.data, pairs.shape[0],
                                                    ratio, floor_, mult, d_matches.data_ptr(), max_per_pair,
                                                    d_counts.data_ptr()))

    def l2_distance_matrix_dev(self, qset, tset, dist, force_path=0):
        """dist: torch float32 CUDA tensor (nq, >= nt), row-contiguous; enqueues only."""
        self._check(self.lib.sfmhip_l2_distance_matrix_dev(self.h, qset.handle, tset.handle, dist.data_ptr(),
                                                           dist.stride(0), force_path))

    # ---------------------------------------------------------------- triangulation
    def triangulate2(self, P1, P2, xy1, xy2):
        P1 = np.ascontiguousarray(P1, np.float32).reshape(12); P2 = np.ascontiguousarray(P2, np.float32).reshape(12)
        xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2); xy2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
        n = xy1.shape[0]
        xyzw = np.empty((4, n), np.float32); xyz = np.empty((n, 3), np.float64)
        self._check(self.lib.sfmhip_triangulate2_f32(self.h, P1.ctypes.data, P2.ctypes.data, xy1.ctypes.data,
                                                     xy2.ctypes.data, n, xyzw.ctypes.data, xyz.ctypes.data))
        return xyzw, xyz

    def triangulate2_dev(self, P1, P2, xy1, xy2, xyzw, xyz):
        P1 = np.ascontiguousarray(P1, np.float32).reshape(12); P2 = np.ascontiguousarray(P2, np.float32).reshape(12)
        self._check(self.lib.sfmhip_triangulate2_f32_dev(self.h, P1.ctypes.data, P2.ctypes.data, xy1.data_ptr(),
                                                         xy2.data_ptr(), xy1.shape[0], _ptr(xyzw), _ptr(xyz)))

    def triangulate2_matches_dev(self, P1, P2, kp1, kp2, matches, n, xyzw, xyz):
        P1 = np.ascontiguousarray(P1, np.float32).reshape(12); P2 = np.ascontiguousarray(P2, np.float32).reshape(12)
        self._check(self.lib.sfmhip_triangulate2_matches_dev(self.h, P1.ctypes.data, P2.ctypes.data, _ptr(kp1), _ptr(kp2),
                                                             _ptr(matches), n, _ptr(xyzw), _ptr(xyz)))

    # ---------------------------------------------------------------- bundle adjustment
    def triangulate_tracks(self, K4, ext, obs_cam, obs_pt, obs_uv, n_pt):
        """N-view DLT of every track (extension, sfmhip_triangulate_tracks): returns (pts (n_pt,3) float64, n_views int32)."""
        K4 = np.ascontiguousarray(K4, np.float64); ext = np.ascontiguousarray(ext, np.float64)
        oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32); uv = np.ascontiguousarray(obs_uv, np.float64)
        pts = np.empty((n_pt, 3), np.float64); nv = np.empty(n_pt, np.int32)
        self._check(self.lib.sfmhip_triangulate_tracks(self.h, _ptr(K4), _ptr(ext), ext.shape[0], _ptr(oc), _ptr(op), _ptr(uv),
                                                       oc.shape[0], int(n_pt), _ptr(pts), _ptr(nv)))
        return pts, nv

    def reprojection_errors(self, K4, ext, pts, obs_cam, obs_pt, obs_uv):
        """pixel error of every observation (sfmhip_reprojection_errors)"""
        K4 = np.ascontiguousarray(K4, np.float64); ext = np.ascontiguousarray(ext, np.float64); pts = np.ascontiguousarray(pts, np.float64)
        oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32); uv = np.ascontiguousarray(obs_uv, np.float64)
        err = np.empty(oc.shape[0], np.float64)
        self._check(self.lib.sfmhip_reprojection_errors(self.h, _ptr(K4), _ptr(ext), ext.shape[0], _ptr(pts), pts.shape[0],
                                                        _ptr(oc), _ptr(op), _ptr(uv), oc.shape[0], _ptr(err)))
        return err

    def ba_options(self, **kw):
        o = BAOptions()
        self.lib.sfmhip_ba_default_options(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def ba_create(self, K4, ext, pts, obs_cam, obs_pt, obs_uv, opts=None):
        return BAProblem(self, K4, ext, pts, obs_cam, obs_pt, obs_uv, opts)

    def ba_solve(self, K4, ext, pts, obs_cam, obs_pt, obs_uv, opts=None):
        """One-shot sfmhip_ba_solve on copies; returns (K4, ext, pts, summary dict)."""
        K4 = np.array(K4, np.float64).reshape(4).copy(); ext = np.array(ext, np.float64).reshape(-1, 6).copy()
        pts = np.array(pts, np.float64).reshape(-1, 3).copy()
        oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
        uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2)
        o = opts if opts is not None else self.ba_options()
        s = BASummary()
        self._check(self.lib.sfmhip_ba_solve(self.h, K4.ctypes.data, ext.ctypes.data, ext.shape[0], pts.ctypes.data,
                                             pts.shape[0], oc.ctypes.data, op.ctypes.data, uv.ctypes.data, oc.shape[0],
                                             C.byref(o), C.byref(s)))
        return K4, ext, pts, s.asdict()

    # ---------------------------------------------------------------- normals
    def estimate_normals(self, pts, K=10):
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty_like(pts)
        self._check(self.lib.sfmhip_estimate_normals(self.h, pts.ctypes.data, pts.shape[0], int(K), out.ctypes.data))
        return out


class BAProblem:
    """HBM-resident BA problem (sfmhip_ba_create)."""

    def __init__(self, ctx, K4, ext, pts, obs_cam, obs_pt, obs_uv, opts=None):
        self.ctx = ctx
        K4 = np.ascontiguousarray(K4, np.float64).reshape(4); ext = np.ascontiguousarray(ext, np.float64).reshape(-1, 6)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
        uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2)
        self.n_cam, self.n_pt, self.n_obs = ext.shape[0], pts.shape[0], oc.shape[0]
        o = opts if opts is not None else ctx.ba_options()
        self.opts = o
        h = C.c_void_p()
        ctx._check(ctx.lib.sfmhip_ba_create(ctx.h, K4.ctypes.data, ext.ctypes.data, self.n_cam, pts.ctypes.data, self.n_pt,
                                            oc.ctypes.data, op.ctypes.data, uv.ctypes.data, self.n_obs, C.byref(o), C.byref(h)))
        self.h = h
        self._cb = None

    def set_allreduce(self, fn, rank, world):
        """fn(dev_ptr:int, count:int, stream:int) -> 0 on success; sums `count` doubles in place over all ranks."""
        def _tramp(user, buf, count, stream):
            try:
                return int(fn(buf, count, stream) or 0)
            except Exception:   # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return -1
        self._cb = _lib.ALLREDUCE_FN(_tramp)
        self.ctx._check(self.ctx.lib.sfmhip_ba_set_allreduce(self.h, self._cb, None, int(rank), int(world)))

    def set_rccl(self, comm, rank, world):
        """install the in-library RCCL hook (sfmhip_ba_set_rccl); comm from Context.rccl_comm_create"""
        self._cb = None
        self.ctx._check(self.ctx.lib.sfmhip_ba_set_rccl(self.h, comm, int(rank), int(world)))

    def run(self):
        s = BASummary()
        self.ctx._check(self.ctx.lib.sfmhip_ba_run(self.h, C.byref(s)))
        return s.asdict()

    def iterate(self, n):
        s = BASummary()
        self.ctx._check(self.ctx.lib.sfmhip_ba_iterate(self.h, int(n), C.byref(s)))
        return s.asdict()

    def reset(self):
        self.ctx._check(self.ctx.lib.sfmhip_ba_reset(self.h))

    def params(self):
        K4 = np.empty(4); ext = np.empty((self.n_cam, 6)); pts = np.empty((self.n_pt, 3))
        self.ctx._check(self.ctx.lib.sfmhip_ba_get_params(self.h, K4.ctypes.data, ext.ctypes.data, pts.ctypes.data))
        return K4, ext, pts

    def reduced_system(self, radius):
        n = C.c_int(); cost = C.c_double()
        self.ctx._check(self.ctx.lib.sfmhip_ba_reduced_system(self.h, radius, None, None, C.byref(n), C.byref(cost)))
        S = np.zeros((n.value, n.value)); rhs = np.zeros(n.value)
        self.ctx._check(self.ctx.lib.sfmhip_ba_reduced_system(self.h, radius, S.ctypes.data, rhs.ctypes.data, C.byref(n), C.byref(cost)))
        return S, rhs, cost.value

    _TABLE_DTYPES = {"ouv": np.float64, "cam_uv": np.float64, "setup_ms": np.float64}

    def debug_table(self, name):
        """One of the tables sfmhip_ba_create built on the device (sfmhip_ba_debug_table), as a numpy array."""
        nb = C.c_size_t()
        self.ctx._check(self.ctx.lib.sfmhip_ba_debug_table(self.h, name.encode(), None, 0, C.byref(nb)))
        dt = np.dtype(self._TABLE_DTYPES.get(name, np.int32))
        out = np.empty(nb.value // dt.itemsize, dt)
        self.ctx._check(self.ctx.lib.sfmhip_ba_debug_table(self.h, name.encode(), out.ctypes.data, out.nbytes, C.byref(nb)))
        return out

    def phase_ms(self):
        out = (C.c_double * 8)()
        self.ctx._check(self.ctx.lib.sfmhip_ba_phase_ms(self.h, out))
        return list(out)

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.lib.sfmhip_ba_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ba_solve_multi(ctxs, K4, ext, pts, obs_cam, obs_pt, obs_uv, opts=None):
    """sfmhip_ba_solve_multi on copies: one context per GPU of this process (two contexts on one device: host-staged rehearsal).
    Returns (K4, ext, pts, summary dict)."""
    K4 = np.array(K4, np.float64).reshape(4).copy(); ext = np.array(ext, np.float64).reshape(-1, 6).copy()
    pts = np.array(pts, np.float64).reshape(-1, 3).copy()
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
    uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2)
    o = opts if opts is not None else ctxs[0].ba_options()
    s = BASummary()
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    ctxs[0]._check(ctxs[0].lib.sfmhip_ba_solve_multi(arr, len(ctxs), K4.ctypes.data, ext.ctypes.data, ext.shape[0], pts.ctypes.data, pts.shape[0],
                                                      oc.ctypes.data, op.ctypes.data, uv.ctypes.data, oc.shape[0], C.byref(o), C.byref(s)))
    return K4, ext, pts, s.asdict()


def match_pairs_multi(ctxs, mats, pairs, ratio=0.6, floor_=10.0, mult=5.0):
    """sfmhip_match_pairs_multi: host matrices (float32: L2, uint8: Hamming2) matched over several contexts of this process, the pairs in
    contiguous blocks; list of DMATCH arrays in pair order."""
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    n_pairs, n = pairs.shape[0], len(mats)
    if n_pairs == 0:
        return []
    ham = np.asarray(mats[0]).dtype == np.uint8
    arrs = [np.ascontiguousarray(m, np.uint8 if ham else np.float32) for m in mats]
    dim = arrs[0].shape[1]
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    rows = np.array([a.shape[0] for a in arrs], np.int32)
    mpp = max(1, int(rows[pairs[:, 0]].max()))
    out = np.empty((n_pairs, mpp), DMATCH); counts = np.zeros(n_pairs, np.int32)
    carr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    ctxs[0]._check(ctxs[0].lib.sfmhip_match_pairs_multi(carr, len(ctxs), 2 if ham else 1, ptrs, rows.ctypes.data, dim, None, n, pairs.ctypes.data, n_pairs,
                                                         ratio, floor_, mult, out.ctypes.data, mpp, counts.ctypes.data))
    return [out[p, :c] for p, c in enumerate(counts.tolist())]


def ratio_filter(idx2, dist2, ratio=0.6, floor_=10.0, mult=5.0):
    """sfmhip_ratio_filter (host C, needs no GPU): the tail of match_features, NViewReconstuct.cpp:880-908."""
    lib = _lib.load()
    idx2 = np.ascontiguousarray(idx2, np.int32); dist2 = np.ascontiguousarray(dist2, np.float32)
    nq = idx2.shape[0]
    out = np.zeros(max(nq, 1), DMATCH); n = C.c_int()
    rc = lib.sfmhip_ratio_filter(idx2.ctypes.data, dist2.ctypes.data, nq, ratio, floor_, mult, out.ctypes.data, C.byref(n))
    if rc != 0:
        raise SfmHipError(f"sfmhip_ratio_filter failed with {rc}")
    return out[:n.value].copy()


# ------------------------------------------------------------------------------------------------
# mirror of the reference's free functions (same names, same argument meaning)
# ------------------------------------------------------------------------------------------------
_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def match_features(query, train, ctx=None):
    """NViewReconstuct.cpp:873 (uint8 rows -> NORM_HAMMING2) / TwoViewReconstruct.cpp:156 (float32 rows -> NORM_L2).
    Returns the DMatch array in query order."""
    ctx = ctx or default_context()
    query = np.asarray(query); train = np.asarray(train)
    if query.dtype == np.uint8:
        sets = [ctx.descset_hamming2(query), ctx.descset_hamming2(train)]
    else:
        sets = [ctx.descset_l2(query), ctx.descset_l2(train)]
    if query.shape[0] == 0:
        return np.zeros(0, DMATCH)
    return ctx.match_pairs(sets, [[0, 1]])[0]


def match_features_for_all(descriptor_for_all, ctx=None):
    """NViewReconstuct.cpp:850-871: consecutive pairs (i, i+1); one batched launch sequence for the whole chain."""
    ctx = ctx or default_context()
    n = len(descriptor_for_all)
    if n < 2:
        return []
    pairs = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1)
    mats = [np.asarray(d) for d in descriptor_for_all]
    if len({(m.dtype, m.shape[1]) for m in mats}) == 1 and mats[0].dtype in (np.float32, np.uint8):
        # ONE C call for the whole chain (sfmhip_match_pairs_multi on this context: sets created from the host matrices in one pass of
        # the staging threads, one preparation launch, one batched kNN-2 + ratio tail, sets released)
        out = match_pairs_multi([ctx], mats, pairs)
    else:
        sets = ctx.descsets_host(mats)
        out = ctx.match_pairs(sets, pairs)
    for i, m in enumerate(out):
        if len(m) == 0:
            print("[Warning]: zero matches between %d and %d." % (i, i + 1))
    return out


def get_matched_points(p1, p2, matches):
    """NViewReconstuct.cpp:989-1003 on KEYPOINT arrays."""
    return (np.stack([p1["x"][matches["queryIdx"]], p1["y"][matches["queryIdx"]]], 1).astype(np.float32),
            np.stack([p2["x"][matches["trainIdx"]], p2["y"][matches["trainIdx"]]], 1).astype(np.float32))


def projection_matrix(K, R, T):
    """NViewReconstuct.cpp:1129-1143: float32(K) @ [float32(R) | float32(T)] (float32 cv::Mat product:
    each dot product accumulated in double and rounded once [3P])."""
    fK = np.asarray(K, np.float64).reshape(3, 3).astype(np.float32)
    RT = np.concatenate([np.asarray(R, np.float64).reshape(3, 3), np.asarray(T, np.float64).reshape(3, 1)], 1).astype(np.float32)
    return (fK.astype(np.float64) @ RT.astype(np.float64)).astype(np.float32)


def reconstruct(K, R1, T1, R2, T2, p1, p2, ctx=None):
    """NViewReconstuct.cpp:1117-1159.  Returns (ret, structure): ret = -1 and an "[Err]" line on empty input."""
    p1 = np.asarray(p1, np.float32).reshape(-1, 2); p2 = np.asarray(p2, np.float32).reshape(-1, 2)
    if p1.shape[0] == 0 or p2.shape[0] == 0:
        print("[Err]: empty 2d points.")
        return -1, np.zeros((0, 3))
    ctx = ctx or default_context()
    _, xyz = ctx.triangulate2(projection_matrix(K, R1, T1), projection_matrix(K, R2, T2), p1, p2)
    return 0, xyz


def bundle_adjustment(intrinsic, extrinsics, correspond_struct_idx, key_points_for_all, structure, ctx=None, opts=None):
    """NViewReconstuct.cpp:1162-1244.  intrinsic (4,), extrinsics (n_cam,6), structure (n_pt,3) are updated IN PLACE
    (numpy float64 arrays); correspond_struct_idx[img][kp] = point id or -1; key_points_for_all[img] = KEYPOINT array
    or (n,2) float array.  Returns the summary dict and prints the reference's statistics block."""
    ctx = ctx or default_context()
    oc, op, uv = [], [], []
    for img, ids in enumerate(correspond_struct_idx):
        ids = np.asarray(ids)
        kp = key_points_for_all[img]
        xy = np.stack([kp["x"], kp["y"]], 1) if getattr(kp, "dtype", None) is not None and kp.dtype.names else np.asarray(kp)
        sel = np.nonzero(ids >= 0)[0]
        oc.append(np.full(sel.shape[0], img, np.int32)); op.append(ids[sel].astype(np.int32))
        uv.append(xy[sel].astype(np.float32).astype(np.float64))      # Point2d observed = key_points[pt_id].pt (1199)
    oc = np.concatenate(oc); op = np.concatenate(op); uv = np.concatenate(uv)
    K4, ext, pts, s = ctx.ba_solve(intrinsic, extrinsics, structure, oc, op, uv, opts)
    np.copyto(intrinsic, K4.reshape(np.shape(intrinsic))); np.copyto(extrinsics, ext.reshape(np.shape(extrinsics)))
    np.copyto(structure, pts.reshape(np.shape(structure)))
    if s["termination"] == 2:
        print("Bundle Adjustment failed.")
    else:
        print("\nBundle Adjustment statistics (approximated RMSE):\n #views: %d\n #residuals: %d\n"
              " Initial RMSE(pixel): %g\n Final   RMSE(pixel): %g\n Time (s): %g\n"
              % (len(extrinsics), s["num_residuals"], np.sqrt(s["initial_cost"] / max(1, s["num_residuals"])),
                 np.sqrt(s["final_cost"] / max(1, s["num_residuals"])), s["total_time_s"]))
    return s


def refine_structure(intrinsic, extrinsics, obs_cam, obs_pt, obs_uv, structure, max_px=4.0, ctx=None, opts=None):
    """Extension (not reference behaviour, which never filters or re-triangulates: NViewReconstuct.cpp:1428-1453):
    drop observations whose reprojection error exceeds max_px, re-triangulate every track from all its remaining observations
    (sfmhip_triangulate_tracks), drop tracks left with fewer than two views, bundle-adjust again.
    Returns (K4, ext, pts (kept), kept point ids, kept observation mask, summary)."""
    ctx = ctx or default_context()
    K4 = np.array(intrinsic, np.float64).reshape(4); ext = np.array(extrinsics, np.float64).reshape(-1, 6)
    pts = np.array(structure, np.float64).reshape(-1, 3)
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32); uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2)
    err = ctx.reprojection_errors(K4, ext, pts, oc, op, uv)
    keep = err <= max_px
    new_pts, nv = ctx.triangulate_tracks(K4, ext, oc[keep], op[keep], uv[keep], len(pts))
    ok = (nv >= 2) & np.isfinite(new_pts).all(1)
    ids = np.nonzero(ok)[0]
    remap = np.full(len(pts), -1, np.int64); remap[ids] = np.arange(len(ids))
    keep &= ok[op]
    K2, ext2, pts2, s = ctx.ba_solve(K4, ext, new_pts[ids], oc[keep], remap[op[keep]].astype(np.int32), uv[keep], opts)
    return K2, ext2, pts2, ids, keep, s
