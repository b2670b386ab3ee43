"""Features file of the driver programs (sfm_opencv_amd/host/sfm_pipeline.hpp): what the reference's extract_features()
(NViewReconstuct.cpp:785-848) leaves behind -- key points, descriptor matrix, BGR colours per image -- plus K and,
optionally, one world->camera pose per image.  Little-endian; layout documented in sfm_pipeline.hpp."""
import struct

import numpy as np

from .api import KEYPOINT

MAGIC = b"SFMFEAT1"


def write_features(path, K, key_points, descriptors, colors=None, poses=None):
    """K 3x3; key_points: list of KEYPOINT arrays or (n,2) float arrays; descriptors: list of uint8 / float32 matrices;
    colors: list of (n,3) uint8 BGR (default zeros); poses: list of (R 3x3, T 3) or None."""
    n_img = len(key_points)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<i", n_img))
        f.write(np.asarray(K, "<f8").reshape(9).tobytes())
        f.write(struct.pack("<i", 1 if poses is not None else 0))
        for i in range(n_img):
            kp = key_points[i]
            if getattr(kp, "dtype", None) is None or kp.dtype.names is None:
                xy = np.asarray(kp, np.float32).reshape(-1, 2)
                kp = np.zeros(len(xy), KEYPOINT)
                kp["x"], kp["y"], kp["angle"], kp["class_id"] = xy[:, 0], xy[:, 1], -1.0, -1
            d = np.ascontiguousarray(descriptors[i])
            assert d.dtype in (np.uint8, np.float32) and d.shape[0] == len(kp)
            f.write(struct.pack("<iii", len(kp), 0 if d.dtype == np.uint8 else 5, d.shape[1]))
            f.write(np.ascontiguousarray(kp).tobytes())
            f.write(d.tobytes())
            c = np.zeros((len(kp), 3), np.uint8) if colors is None else np.ascontiguousarray(colors[i], np.uint8).reshape(-1, 3)
            f.write(c.tobytes())
            if poses is not None:
                R, T = poses[i]
                f.write(np.asarray(R, "<f8").reshape(9).tobytes()); f.write(np.asarray(T, "<f8").reshape(3).tobytes())


def read_features(path):
    """-> dict(K 3x3, key_points [KEYPOINT arrays], descriptors [uint8 / float32 matrices], colors [(n,3) uint8], poses or None)"""
    raw = open(path, "rb").read()
    assert raw[:8] == MAGIC
    n_img = struct.unpack_from("<i", raw, 8)[0]
    K = np.frombuffer(raw, "<f8", 9, 12).reshape(3, 3).copy()
    has_poses = struct.unpack_from("<i", raw, 84)[0]
    off = 88
    kps, descs, cols, poses = [], [], [], []
    for _ in range(n_img):
        n_kp, typ, ncol = struct.unpack_from("<iii", raw, off); off += 12
        kps.append(np.frombuffer(raw, KEYPOINT, n_kp, off).copy()); off += 28 * n_kp
        if typ == 5:
            descs.append(np.frombuffer(raw, "<f4", n_kp * ncol, off).reshape(n_kp, ncol).copy()); off += 4 * n_kp * ncol
        else:
            d = np.frombuffer(raw, np.uint8, n_kp * ncol, off).reshape(n_kp, ncol).copy(); off += n_kp * ncol
            descs.append(d.astype(np.float32) if typ == 100 else d)
        cols.append(np.frombuffer(raw, np.uint8, 3 * n_kp, off).reshape(n_kp, 3).copy()); off += 3 * n_kp
        if has_poses:
            R = np.frombuffer(raw, "<f8", 9, off).reshape(3, 3).copy(); T = np.frombuffer(raw, "<f8", 3, off + 72).copy(); off += 96
            poses.append((R, T))
    return dict(K=K, key_points=kps, descriptors=descs, colors=cols, poses=poses if has_poses else None)
