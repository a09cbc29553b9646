#!/usr/bin/env python3
"""Where does a multi-rank step's time go on the host?  One rank at 1/8 of the headline problem (the shard each of 8
ranks holds), the bench's dist step (kernel on the main stream, RCCL all_reduce on a side stream) against the plain
step, with the host's own issue time per step measured apart from the wall time.
   RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 python tools/probe_host_overhead.py"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29612")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sh = bench.build_shard(argparse.Namespace(blocks=45), 0, 1, dev)
n = sh["n_obs"]
r = torch.empty((n, 2), dtype=torch.float64, device=dev)
Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
ws = D.workspace(n, dev)
err2 = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(2)]
main = torch.cuda.current_stream()
side = torch.cuda.Stream(device=dev)
pe = [torch.cuda.Event(), torch.cuda.Event()]
fe = [torch.cuda.Event(), torch.cuda.Event()]
fd = [None, None]
args = (sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws)


def kernel_only(i):
    D.residual_jacobian_sum(*args, err2[i & 1])


def dist_step(i):
    k = i & 1
    if fd[k] is not None:
        main.wait_event(fd[k])
    D.residual_jacobian_sum(*args, err2[k])
    pe[k].record(main)
    with torch.cuda.stream(side):
        side.wait_event(pe[k])
        dist.all_reduce(err2[k])
        fe[k].record(side)
        fd[k] = fe[k]


def same_stream_step(i):
    D.residual_jacobian_sum(*args, err2[i & 1])
    dist.all_reduce(err2[i & 1])


def measure(name, fn, K=300):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        fn(i)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%-28s host issue %6.1f us/step   wall %6.1f us/step" % (name, t_issue / K * 1e6, t_all / K * 1e6))


print("n_obs", n)
measure("kernel only", kernel_only)
measure("kernel + side-stream reduce", dist_step)
measure("kernel + same-stream reduce", same_stream_step)
dist.destroy_process_group()
