"""Multi-GPU sharding of the observation list: one process per GPU, torch.distributed (nccl = RCCL
over xGMI on the GPU box, gloo in the CPU tests).

The path shards by contiguous camera ranges: every rank owns the observations
of its cameras, points are replicated, outputs stay sharded.  The only exchange step is the
1-element sum all-reduce behind BAProblem::total_reprojection_error (src/baproblem.rs:265-279):
    total = (sum over ranks of sum_obs |du|^norm + |dv|^norm) ^ (1/norm).
"""
import ctypes as C

import numpy as np

from . import _lib as L


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    import os
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def camera_count_bounds(n_cam, world):
    """Equal camera counts per rank: bounds[world+1]."""
    return [n_cam * k // world for k in range(world + 1)]


def partition_by_observations(row_ptr, world):
    """Contiguous camera ranges holding ~equal numbers of observations (c2b_partition_cameras)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    bounds = np.zeros(world + 1, dtype=np.int64)
    L.check(L.lib().c2b_partition_cameras(row_ptr.ctypes.data_as(C.c_void_p), len(row_ptr) - 1, int(world),
                                          bounds.ctypes.data_as(C.c_void_p)))
    return bounds


def shard_csr(row_ptr, pt_idx, uv, lo, hi):
    """Slice the CSR graph to cameras [lo, hi): local row_ptr (starting at 0), pt_idx, uv and the
    global index of the shard's first observation (the obs_base of the noise / expand kernels)."""
    row_ptr = np.asarray(row_ptr)
    a, b = int(row_ptr[lo]), int(row_ptr[hi])
    local = (row_ptr[lo:hi + 1] - row_ptr[lo]).astype(np.uint64)
    return local, pt_idx[a:b], uv[a:b], a


def exclusive_offset(count, group=None):
    """Global index of this rank's first element given every rank's count (all_gather of one int)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0, int(count)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([int(count)], dtype=torch.int64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    counts = [int(t.item()) for t in every]
    return sum(counts[:rank]), sum(counts)


def all_reduce_sum_(t, group=None):
    """In-place sum all-reduce of a (1-element) tensor; a no-op without a process group.  With a group of one rank
    the collective is still issued (that is how the RCCL path is exercised on a 1-GPU box)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if t.is_cuda and dist.get_backend(group) != "nccl":
            # gloo (single-GPU rehearsal of the multi-rank path): stage the scalar through the host
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def combine_stats_partials(parts, n_entities):
    """parts [world][20] = every rank's c2b_stats_partial_pass1 record, in rank order -> (mean3, min3, max3,
    origin3, origin_global_index).  mean: the shares summed in rank order; origin: smallest distance, ties to the
    larger global index (fold1 with strict <, src/noise.rs:80-86).  The host half of the collective, computed by the
    C ABI (c2b_stats_combine_shares) so that the Python and the C++ callers (c2b_stats_sharded) share one
    implementation."""
    parts = np.ascontiguousarray(parts, dtype=np.float64).reshape(-1, 20)
    st = np.zeros(20)
    L.check(L.lib().c2b_stats_combine_shares(parts.ctypes.data_as(C.c_void_p), len(parts), st.ctypes.data_as(C.c_void_p)))
    return st[0:3].copy(), st[6:9].copy(), st[9:12].copy(), st[15:18].copy(), int(st[18])


def finish_stats(mean, mn, mx, origin, origin_index, sumsq, n_entities):
    """the 20-double statistics record of c2b_stats from the combined pieces (std = sqrt(sum of squares / n));
    sumsq = [world][3] per-rank sums in rank order, or one already summed row (c2b_stats_finish_shares)"""
    st = np.zeros(20)
    st[0:3], st[6:9], st[9:12], st[12:15] = mean, mn, mx, np.asarray(mx) - np.asarray(mn)
    st[15:18], st[18] = origin, float(origin_index)
    sq = np.ascontiguousarray(sumsq, dtype=np.float64).reshape(-1, 3)
    L.check(L.lib().c2b_stats_finish_shares(sq.ctypes.data_as(C.c_void_p), len(sq), int(n_entities),
                                            st.ctypes.data_as(C.c_void_p)))
    return st


def _all_gather_rows(row, group=None):
    """every rank's 1-D float64 numpy row, in rank order, as a [world, len] array"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(row, dtype=np.float64)[None, :]
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.from_numpy(np.ascontiguousarray(row, dtype=np.float64)).to(dev)
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    return np.stack([t.cpu().numpy() for t in every])


def stats_sharded(camblk, cam_base, n_cam_global, pts4, ws, group=None, centers=None):
    """BAProblem::mean/std/extent/dimensions + add_drift's origin (src/baproblem.rs:282-337, src/noise.rs:75-87) when
    cameras are sharded: this rank's camblk holds cameras [cam_base, cam_base + len) of n_cam_global, pts4 is the whole
    replicated table and rank r of W reduces its r-th slice.  Two small all-gathers (20 and 3 doubles per rank), no
    other traffic.  Returns the statistics record as a device tensor, identical on every rank (sums in rank order)."""
    import torch
    import torch.distributed as dist
    from . import device as D
    on = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if on else (1, 0)
    n_pts = pts4.shape[0]
    lo, hi = n_pts * rank // world, n_pts * (rank + 1) // world
    n_ent = int(n_cam_global) + int(n_pts)
    part = D.stats_partial_pass1(camblk, cam_base, n_cam_global, pts4[lo:hi], lo, n_ent, ws, centers=centers)
    parts = _all_gather_rows(part.cpu().numpy(), group)
    mean, mn, mx, origin, oidx = combine_stats_partials(parts, n_ent)
    mean_d = torch.from_numpy(mean).to(camblk.device)
    sq = D.stats_partial_pass2(camblk, pts4[lo:hi], mean_d, ws, centers=centers)
    rows = _all_gather_rows(sq.cpu().numpy(), group)            # summed in rank order: identical on every rank
    return torch.from_numpy(finish_stats(mean, mn, mx, origin, oidx, rows, n_ent)).to(camblk.device)


def finish_error(total_sum, norm):
    """.powf(1. / norm), src/baproblem.rs:278"""
    return float(total_sum) ** (1.0 / float(norm))
