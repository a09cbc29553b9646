#!/usr/bin/env python3
"""What bounds the light per-observation kernels?  Times project / error / Jacobian on the bench workload with the point
indices replaced by (a) the real ones, (b) all zero (every gather hits one line), (c) a coalesced stream (i mod n_pts),
(d) uniformly random ones.   python tools/probe_gather.py [--blocks 128]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=128)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)
n, n_pts = sh["n_obs"], sh["n_pts"]
camblk, pts4, ci, pi, uv = sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"]
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)
uv_out = torch.empty_like(uv)
r = torch.empty((n, 2), dtype=torch.float64, device=dev)
Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(1)
variants = {
    "real": pi,
    "all_zero": torch.zeros_like(pi),
    "stream": (torch.arange(n, device=dev, dtype=torch.int64) % n_pts).to(torch.int32),
    "random": torch.randint(0, n_pts, (n,), dtype=torch.int32, device=dev, generator=g),
}


def timed(fn):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / a.reps * 1e3, 1)


out = {}
for name, idx in variants.items():
    out[name] = {
        "project_us": timed(lambda: D.project(camblk, pts4, ci, idx, uv_out)),
        "error_us": timed(lambda: D.reprojection_error_sum(camblk, pts4, ci, idx, uv, 2.0, ws, err)),
        "jacobian_us": timed(lambda: D.residual_jacobian(camblk, pts4, ci, idx, uv, r, Jc, Jp, 2.0, ws)),
    }
# and the camera side: every observation on camera 0 (one LDS tile row, no camera traffic)
ci0 = torch.zeros_like(ci)
out["real_points_camera0"] = {"project_us": timed(lambda: D.project(camblk, pts4, ci0, pi, uv_out))}
out["all_zero_camera0"] = {"project_us": timed(lambda: D.project(camblk, pts4, ci0, variants["all_zero"], uv_out))}
print(json.dumps(out))
