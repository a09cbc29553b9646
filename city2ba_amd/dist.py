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


def finish_error(total_sum, norm):
    """.powf(1. / norm), src/baproblem.rs:278"""
    return float(total_sum) ** (1.0 / float(norm))
