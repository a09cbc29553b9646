#!/usr/bin/env python3
"""project_rows / error_sum_rows / visibility_rows on the bench grid, 12 launches each back to back and 12 launches each
with the caches swept by a 1-GiB read before every launch (run under rocprofv3 --pmc: the counter CSV then holds cold and
warm dispatches of the same kernels; the cold ones follow a reduce kernel of torch's)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
import bench                                                   # noqa: E402
from city2ba_amd import device as D                            # noqa: E402

dev = torch.device("cuda", 0)
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
n = sh["n_obs"]
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)
uv_out = torch.empty_like(sh["uv"])
keep = torch.empty(n, dtype=torch.uint8, device=dev)
sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
fns = [lambda: D.project_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], uv_out),
       lambda: D.reprojection_error_sum_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], 2.0, ws, err),
       lambda: D.visibility_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], 10.0, uv_out, keep)]
for fn in fns:
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
for fn in fns:
    for _ in range(12):
        sweep.sum()
        fn()
    torch.cuda.synchronize()
