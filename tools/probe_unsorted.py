#!/usr/bin/env python3
"""Cost of the slow path: the per-observation kernels on a randomly permuted observation list (every lane of a wave
names a different camera, so every camera is restaged one at a time) against the camera-major list."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

dev = torch.device("cuda", 0)
sh = bench.build_shard(argparse.Namespace(blocks=32), 0, 1, dev)
n = sh["n_obs"]
g = torch.Generator(device=dev)
g.manual_seed(1)
perm = torch.randperm(n, device=dev, generator=g)
r, Jc, Jp = torch.empty((n, 2), dtype=torch.float64, device=dev), torch.empty((n, 18), dtype=torch.float64, device=dev), torch.empty((n, 6), dtype=torch.float64, device=dev)
uv_out = torch.empty((n, 2), dtype=torch.float64, device=dev)
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for name, ci, pi, uv in (("camera-major", sh["cam_idx"], sh["pt_idx"], sh["uv"]),
                         ("random order", sh["cam_idx"][perm].contiguous(), sh["pt_idx"][perm].contiguous(), sh["uv"][perm].contiguous())):
    tp = timed(lambda: D.project(sh["camblk"], sh["pts4"], ci, pi, uv_out))
    tj = timed(lambda: D.residual_jacobian_sum(sh["camblk"], sh["pts4"], ci, pi, uv, r, Jc, Jp, 2.0, ws, err))
    print("%-13s project %8.1f us   residual+Jacobian %8.1f us   (%d observations)" % (name, tp, tj, n))
