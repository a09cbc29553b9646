#!/usr/bin/env python3
"""Runs each per-observation kernel a few times on the bench workload so that a `rocprofv3 --pmc ...` pass over this
script yields per-kernel counters (tools/profile_light.sh).   python tools/probe_light.py [--blocks 128] [--reps 5]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=128)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)
n = sh["n_obs"]
camblk, pts4, ci, pi, uv = sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"]
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)
uv_out = torch.empty_like(uv)
keep = torch.empty(n, dtype=torch.uint8, device=dev)
r = torch.empty((n, 2), dtype=torch.float64, device=dev)
Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
for _ in range(a.reps):
    D.project(camblk, pts4, ci, pi, uv_out)
    D.reprojection_error_sum(camblk, pts4, ci, pi, uv, 2.0, ws, err)
    D.visibility_pairs(camblk, pts4, ci, pi, 10.0, uv_out, keep)
    D.residual_jacobian(camblk, pts4, ci, pi, uv, r, Jc, Jp, 2.0, ws)
    D.add_noise_observations(uv_out, 0, 0.0, 1)
    # the row-structure forms (separate template instances: they show up under their own kernel names)
    D.project_rows(camblk, pts4, sh["rows"], pi, uv_out)
    D.reprojection_error_sum_rows(camblk, pts4, sh["rows"], pi, uv, 2.0, ws, err)
    D.visibility_rows(camblk, pts4, sh["rows"], pi, 10.0, uv_out, keep)
    D.residual_jacobian_rows(camblk, pts4, sh["rows"], pi, uv, r, Jc, Jp, 2.0, ws)
torch.cuda.synchronize()
print("n_obs", n)
