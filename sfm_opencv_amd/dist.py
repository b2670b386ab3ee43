"""One process per GPU: sharding of the hot path over ranks and the all-reduce hook.

* matching / triangulation: image pairs are independent (NViewReconstuct.cpp:857-862) -> contiguous blocks of the
  pair chain per rank, one halo image at each boundary, NO collective.
* bundle adjustment: points (with all their observations) are partitioned over ranks, cameras + intrinsics are
  replicated; per LM iteration ONE sum all-reduce of the reduced-camera-system message and one of 4 step scalars.
  The library calls back into `make_allreduce_hook`'s function, which runs torch.distributed.all_reduce (RCCL over
  xGMI with the nccl backend; gloo in the CPU tests) on a zero-copy view of the library's buffer.
"""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None):
    """torch.distributed over RANK/WORLD_SIZE/MASTER_* (nccl == RCCL on ROCm when a GPU is present, else gloo)."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    # rehearsal knobs (several ranks sharing one card, where RCCL refuses duplicate devices):
    # SFM_DIST_BACKEND=gloo, SFM_LOCAL_DEVICE=0
    backend = backend or os.environ.get("SFM_DIST_BACKEND") or None
    if "SFM_LOCAL_DEVICE" in os.environ:
        local = int(os.environ["SFM_LOCAL_DEVICE"])
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n units for this rank (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_pairs(n_img, rank, world):
    """Chain pairs (i, i+1) of match_features_for_all for this rank, the images it must hold, and the pair list
    re-indexed into that local image list."""
    lo, hi = shard_range(max(n_img - 1, 0), rank, world)
    images = list(range(lo, hi + 1)) if hi > lo else []
    pairs_global = np.stack([np.arange(lo, hi), np.arange(lo + 1, hi + 1)], 1).astype(np.int32) if hi > lo else np.zeros((0, 2), np.int32)
    pairs_local = pairs_global - lo
    return pairs_global, images, pairs_local


def shard_points(obs_cam, obs_pt, obs_uv, pts, rank, world, by="first_camera"):
    """Partition the points (with all their observations) over the ranks, (nearly) equal observation counts per rank.
    by="first_camera" (default): points ordered by the lowest camera that sees them and cut into contiguous runs -- "by the image
    pair that created the point" (NViewReconstuct.cpp:1297-1299 appends a point when its first pair is fused), so a
    rank works on a window of the camera chain: its partial reduced system touches only that window's band, and its kernels
    gather from a fraction of the camera blocks.  by="id": contiguous point-id ranges (round 1-2 behaviour).
    Returns this rank's (pts_local, obs_cam_local, obs_pt_local (re-indexed), obs_uv_local, point_ids)."""
    obs_pt = np.asarray(obs_pt); obs_cam = np.asarray(obs_cam); n_pt = len(pts)
    cnt = np.bincount(obs_pt, minlength=n_pt)
    if by == "first_camera":
        first = np.full(n_pt, np.iinfo(np.int64).max, np.int64)
        np.minimum.at(first, obs_pt, obs_cam.astype(np.int64))
        order = np.argsort(first, kind="stable")            # points without observations go last
    elif by == "id":
        order = np.arange(n_pt)
    else:
        raise ValueError(by)
    cum = np.concatenate([[0], np.cumsum(cnt[order])])
    total = cum[-1]
    bounds = [int(np.searchsorted(cum, total * r / world, side="left")) for r in range(world)] + [n_pt]
    bounds[0] = 0
    ids = np.sort(order[bounds[rank]:bounds[rank + 1]])     # ascending ids: the local order is the caller's order restricted to the shard
    local = np.full(n_pt, -1, np.int64); local[ids] = np.arange(len(ids))
    sel = local[obs_pt] >= 0
    return (np.ascontiguousarray(np.asarray(pts)[ids]), np.ascontiguousarray(obs_cam[sel]),
            np.ascontiguousarray(local[obs_pt[sel]]).astype(np.int32), np.ascontiguousarray(np.asarray(obs_uv)[sel]), ids)


class _CudaView:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def make_allreduce_hook(group=None, device="cuda"):
    """fn(ptr, count, stream) -> 0: in-place sum of `count` float64 at address `ptr` over the process group.
    device="cuda": ptr is a device pointer; the collective is enqueued on torch's current stream, which must be the
    stream the sfmhip context uses (Context(use_torch_stream=True)).  device="cpu": ptr is a host address (tests)."""
    import ctypes

    import torch
    import torch.distributed as dist
    # (address, count) -> tensor view.  The views own nothing (raw address + length): one that outlives the buffer it was
    # made for is harmless, and if a later buffer lands on the same address with the same length the view IS that buffer.
    # The library reduces at most three distinct buffers per problem; bounded anyway.
    cache = {}

    def hook(ptr, count, stream):
        key = (ptr, count)
        t = cache.get(key)
        if t is None:
            if len(cache) >= 64:
                cache.clear()
            if device == "cpu":
                buf = (ctypes.c_double * count).from_address(ptr)
                t = torch.from_numpy(np.frombuffer(buf, np.float64, count))
            else:
                t = torch.as_tensor(_CudaView(ptr, count), device="cuda")
            cache[key] = t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return 0

    return hook


def make_native_rccl(ctx, group=None):
    """Communicator for the in-library RCCL hook (sfmhip_ba_set_rccl): rank 0 draws the unique id, torch.distributed carries its 128
    bytes to the other ranks, every rank creates its communicator on the context's device.  Returns the opaque communicator
    (BAProblem.set_rccl(comm, rank, world)); the LM loop then calls ncclAllReduce itself, no Python in between."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    uid = ctx.rccl_unique_id() if rank == 0 else bytes(128)
    if world > 1:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0, group=group)
        uid = bytes(t.cpu().tolist())
    return ctx.rccl_comm_create(uid, rank, world)
