#!/usr/bin/env python3
"""After the outputs are placed (device.alloc_jacobian_outputs), does re-placing the INPUT arrays help?  Greedy: clone
one input into a new allocation, time the kernel, keep the clone if it is faster."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
n = sh["n_obs"]
torch.cuda.empty_cache()
(r, Jc, Jp), log = D.alloc_jacobian_outputs(n, dev)
print("output placement", log)
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)


def kernel_us(reps=8):
    for _ in range(2):
        D.residual_jacobian_sum(sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws, err)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        D.residual_jacobian_sum(sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws, err)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


best = kernel_us()
print("start: kernel %.1f us" % best)
held = []
for name in ("uv", "camblk", "pt_idx", "cam_idx", "pts4", "uv", "camblk"):
    for attempt in range(3):
        old = sh[name]
        sh[name] = old.clone()
        t = kernel_us()
        if t < best * 0.995:
            print("  %-8s attempt %d: %.1f -> %.1f us  (kept)" % (name, attempt, best, t))
            best = t
            held.append(old)
        else:
            print("  %-8s attempt %d: %.1f us (no gain)" % (name, attempt, t))
            held.append(sh[name])
            sh[name] = old
print("end: kernel %.1f us  frac %.3f" % (best, bench.algorithmic_bytes(n, sh["n_cam"], sh["n_pts"]) / best / 1e3 / 8000))
