#!/usr/bin/env python3
"""r05: do the light passes depend on where their INPUT arrays lie?  --blocks 128, a fixed output buffer; at each of 32 positions 2 GiB
apart (fillers held in between) the inputs are cloned -- all of them, then one kind at a time -- and project_rows / the L1 + L2 error
pass are timed reading the clones, back to back.       python tools/probes/light_input_position_probe.py"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

dev = torch.device("cuda", 0)
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
n = sh["n_obs"]
ws = D.workspace(n, dev)
err = torch.zeros(2, dtype=torch.float64, device=dev)
out = torch.empty((n, 2), dtype=torch.float64, device=dev)


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


base = dict(camblk=sh["camblk"], pts4=sh["pts4"], pt_idx=sh["pt_idx"], uv=sh["uv"])
print("inputs where build_shard left them: project %d us, L1+L2 %d us" % (
    timed(lambda: D.project_rows(base["camblk"], base["pts4"], sh["rows"], base["pt_idx"], out)),
    timed(lambda: D.reprojection_error_sums2_rows(base["camblk"], base["pts4"], sh["rows"], base["pt_idx"], base["uv"], ws, err))), flush=True)
keep = []
rows = {k: [] for k in ("all", "camblk", "pts4", "pt_idx", "uv(L1+L2)")}
for k in range(32):
    try:
        c = {name: t.clone() for name, t in base.items()}
        used = sum(t.numel() * t.element_size() for t in c.values())
        fill = torch.empty(max((2 << 30) - used, 1 << 20), dtype=torch.uint8, device=dev)
    except Exception:
        break
    keep += [c, fill]
    rows["all"].append(round(timed(lambda: D.project_rows(c["camblk"], c["pts4"], sh["rows"], c["pt_idx"], out))))
    rows["camblk"].append(round(timed(lambda: D.project_rows(c["camblk"], base["pts4"], sh["rows"], base["pt_idx"], out))))
    rows["pts4"].append(round(timed(lambda: D.project_rows(base["camblk"], c["pts4"], sh["rows"], base["pt_idx"], out))))
    rows["pt_idx"].append(round(timed(lambda: D.project_rows(base["camblk"], base["pts4"], sh["rows"], c["pt_idx"], out))))
    rows["uv(L1+L2)"].append(round(timed(lambda: D.reprojection_error_sums2_rows(c["camblk"], c["pts4"], sh["rows"], c["pt_idx"], c["uv"], ws, err))))
for name, v in rows.items():
    print("%-12s clones at position k (x 2 GiB): %s" % (name, " ".join(map(str, v))), flush=True)
