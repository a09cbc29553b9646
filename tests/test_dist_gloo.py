"""World-size-2 gloo tests (CPU) of the multi-GPU host logic: camera-range sharding, global observation
offsets, and the single sum all-reduce behind total_reprojection_error.  The per-shard sums come from the
CPU oracle here (no GPU in this container); on the GPU box the same code path runs with nccl/RCCL in bench.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle as O
    from _problems import random_problem
    from city2ba_amd import dist as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert D.env_world() == (rank, rank, world)
        P = random_problem(90, 2000, 14, seed=77, noise=1e-2, empty_every=6)
        bounds = D.partition_by_observations(P["row_ptr"], world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        row_ptr, pt_idx, uv, base = D.shard_csr(P["row_ptr"], P["pt_idx"], P["uv"], lo, hi)
        # global offsets agree with the CSR prefix
        off, total = D.exclusive_offset(len(pt_idx))
        assert off == base and total == len(P["pt_idx"])
        res = {}
        for norm in (1.0, 2.0, 3.0):
            part = O.reprojection_error_sum(P["cams15"][lo:hi], P["pts"], row_ptr, pt_idx, uv, norm)
            t = torch.tensor([part], dtype=torch.float64)
            D.all_reduce_sum_(t)
            res[norm] = D.finish_error(t.item(), norm)
            want = O.total_reprojection_error(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"], norm)
            assert abs(res[norm] - want) / want < 1e-12, (norm, res[norm], want)
        # shard-independent observation noise: noising shards with obs_base == noising the whole
        _, _, uv_all = O.add_noise(P["cams15"], P["pts"], P["uv"], 0, 0, 0, 0.1, seed=5)
        _, _, uv_part = O.add_noise(P["cams15"][lo:hi], P["pts"], uv, 0, 0, 0, 0.1, seed=5, obs_offset=base)
        assert np.array_equal(uv_part, uv_all[base:base + len(uv)])
        out[rank] = (lo, hi, len(pt_idx), res[2.0])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_sharded_error_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == 2
    (lo0, hi0, n0, e0), (lo1, hi1, n1, e1) = out[0], out[1]
    assert lo0 == 0 and hi0 == lo1 and hi1 == 90          # contiguous camera ranges covering everything
    assert abs(n0 - n1) <= 20                              # balanced on observation counts
    assert e0 == e1                                        # every rank ends with the same total


def test_single_process_helpers():
    sys.path.insert(0, ROOT)
    from city2ba_amd import dist as D
    assert D.camera_count_bounds(10, 4) == [0, 2, 5, 7, 10]
    assert D.exclusive_offset(7) == (0, 7)
    t = torch.tensor([3.0], dtype=torch.float64)
    assert D.all_reduce_sum_(t).item() == 3.0
    assert D.finish_error(9.0, 2.0) == 3.0
