#!/usr/bin/env python3
"""Does an initialised RCCL communicator (or an all_reduce per step) slow the residual+Jacobian kernel itself?
One process, one GPU, --blocks B: kernel time (HIP events) before init_process_group, after it, and with an
all_reduce of the error scalar queued behind every kernel."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=128)
a = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29613")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)
n = sh["n_obs"]
r = torch.empty((n, 2), dtype=torch.float64, device=dev)
Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)
args = (sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws, err)


def kernel_us(with_reduce, K=40):
    for _ in range(5):
        D.residual_jacobian_sum(*args)
        if with_reduce:
            dist.all_reduce(err)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for a_, b_ in ev:
        a_.record()
        D.residual_jacobian_sum(*args)
        b_.record()
        if with_reduce:
            dist.all_reduce(err)
    e.record()
    torch.cuda.synchronize()
    ks = sorted(x.elapsed_time(y) for x, y in ev)
    return ks[len(ks) // 2] * 1e3, s.elapsed_time(e) / K * 1e3


print("n_obs", n)
print("before init_process_group : kernel %.1f us, step %.1f us" % kernel_us(False))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("after init, no collective : kernel %.1f us, step %.1f us" % kernel_us(False))
dist.all_reduce(err)
torch.cuda.synchronize()
print("after a first all_reduce  : kernel %.1f us, step %.1f us" % kernel_us(False))
print("all_reduce every step     : kernel %.1f us, step %.1f us" % kernel_us(True))
print("again without             : kernel %.1f us, step %.1f us" % kernel_us(False))
dist.destroy_process_group()
print("after destroy             : kernel %.1f us, step %.1f us" % kernel_us(False))
