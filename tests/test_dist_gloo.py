"""World-size-2 gloo tests (CPU) of the multi-GPU host logic: camera-range sharding, global observation
offsets, and the single sum all-reduce behind total_reprojection_error.  The per-shard sums come from the
CPU oracle here (no GPU in this container); on the GPU box the same code path runs with nccl/RCCL in bench.py."""
import os
import subprocess
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
        # sharded statistics: every rank's share of the sums / extrema / closest entity, combined
        # by dist.combine_stats_partials after an all-gather, equals the oracle's mean / std / extent / drift origin.
        # (The shares come from numpy here -- no GPU in this container; on the GPU they come from
        # c2b_stats_partial_pass1/2, tests/test_gpu_level0.py.)
        centers = O.centers(P["cams15"])
        n_ent = len(centers) + len(P["pts"])
        plo, phi = len(P["pts"]) * rank // world, len(P["pts"]) * (rank + 1) // world
        mine = np.vstack([centers[lo:hi], P["pts"][plo:phi]])
        gidx = np.concatenate([np.arange(lo, hi), len(centers) + np.arange(plo, phi)])
        part = np.zeros(20)
        for row in mine:
            part[0:3] += row / n_ent
        part[6:9], part[9:12] = mine.min(axis=0), mine.max(axis=0)
        d = np.sqrt((mine[:, 0] ** 2 + mine[:, 1] ** 2) + mine[:, 2] ** 2)
        k = max(np.nonzero(d == d.min())[0])                    # ties -> the later element
        part[15:18], part[18], part[19] = mine[k], gidx[k], d[k]
        parts = D._all_gather_rows(part)
        assert parts.shape == (world, 20) and np.array_equal(parts[rank], part)
        mean, mn, mx, origin, oidx = D.combine_stats_partials(parts, n_ent)
        sq = ((mine - mean) ** 2).sum(axis=0)
        sumsq = D._all_gather_rows(sq).sum(axis=0)
        st = D.finish_stats(mean, mn, mx, origin, oidx, sumsq, n_ent)
        assert np.allclose(st[0:3], O.mean(P["cams15"], P["pts"]), rtol=1e-12, atol=1e-12)
        assert np.allclose(st[3:6], O.std(P["cams15"], P["pts"]), rtol=1e-12)
        e_mn, e_mx = O.extent(P["cams15"], P["pts"])
        assert np.array_equal(st[6:9], e_mn) and np.array_equal(st[9:12], e_mx)
        o_xyz, o_idx = O.drift_origin(P["cams15"], P["pts"])
        assert int(st[18]) == o_idx and np.array_equal(st[15:18], o_xyz)
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
    # origin tie-break: equal distances -> the larger global index (fold1 with strict <); empty shards are skipped
    a, b, c = np.zeros(20), np.zeros(20), np.zeros(20)
    a[15:18], a[18], a[19] = (1, 0, 0), 4, 1.0
    b[15:18], b[18], b[19] = (0, 1, 0), 9, 1.0
    c[18], c[19] = -1, np.inf
    for x in (a, b, c):
        x[6:9], x[9:12] = (0, 0, 0), (1, 1, 1)
    mean, mn, mx, origin, oidx = D.combine_stats_partials([a, c, b], 10)
    assert oidx == 9 and list(origin) == [0, 1, 0]
    st = D.finish_stats(mean, mn, mx, origin, oidx, [10.0, 40.0, 90.0], 10)
    assert np.allclose(st[3:6], [1, 2, 3]) and abs(st[19] - 14 ** 0.5) < 1e-15 and list(st[12:15]) == [1, 1, 1]


def test_watchdog_ends_a_rank_that_stops_making_progress():
    """bench.Watchdog: no beat for longer than the limit -> the rank says where it was and exits with status 3 (a fresh
    exit); a rank that keeps beating, or a stopped watchdog, is left alone"""
    code = ("import bench, time\n"
            "d = bench.Watchdog(5, 1.0)\n"
            "for k in range(3):\n"
            "    d.beat('loop %d' % k); time.sleep(0.6)\n"
            "d.beat('all-reduce of step 17'); time.sleep(30)\n")
    p = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert p.returncode == 3, (p.returncode, p.stderr)
    assert "rank 5 made no progress" in p.stderr and "all-reduce of step 17" in p.stderr
    code = "import bench, time\nd = bench.Watchdog(0, 1.0); d.stop(); time.sleep(2.5); print('alive')\n"
    p = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert p.returncode == 0 and "alive" in p.stdout
