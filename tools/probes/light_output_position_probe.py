#!/usr/bin/env python3
"""r05: do the light passes that STORE depend on where their output buffer lies, like the step kernel (vram_store_map.py)?
--blocks 128; 48 output buffers (uv_out, 309 MB) allocated 2 GiB apart (fillers held in between), project_rows timed into each, back to
back; the same for the fused noise pass's in-place uv array (a copy of the observations at each position).
    python tools/probes/light_output_position_probe.py"""
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
a = (sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"])


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


keep, proj, noise = [], [], []
for k in range(48):
    try:
        out = torch.empty((n, 2), dtype=torch.float64, device=dev)
        fill = torch.empty((2 << 30) - out.numel() * 8, dtype=torch.uint8, device=dev)
    except Exception:
        break
    keep += [out, fill]
    proj.append(round(timed(lambda: D.project_rows(*a, out))))
    out.copy_(sh["uv"])
    noise.append(round(timed(lambda: D.add_noise_observations_error_sums2_rows(*a, out, 0, 1e-9, 7, ws, err), 6)))
print("project_rows us, output 2 GiB further each time    :", " ".join(map(str, proj)))
print("noise + L1 + L2 us, its uv array at those positions  :", " ".join(map(str, noise)), flush=True)
