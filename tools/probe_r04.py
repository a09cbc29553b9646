#!/usr/bin/env python3
"""Round 4's passes on the bench grid (--blocks 128), 12 launches each back to back: the one-norm error, the L1 + L2
error in one pass, observation noise alone, observation noise + both errors in one pass; then the generators' visibility
loop once through Level 1 (c2b_problem_visibility_within_distance).  A target for
    rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 tools/probe_r04.py
(tools/profile_r04.sh): per-kernel time and HBM bytes next to the algorithmic bytes."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                             # noqa: E402
import torch                                                   # noqa: E402
import bench                                                   # noqa: E402
import city2ba_amd as c2b                                      # noqa: E402
from city2ba_amd import device as D                            # noqa: E402
from city2ba_amd import synthetic as S                         # noqa: E402

dev = torch.device("cuda", 0)
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
n = sh["n_obs"]
ws = D.workspace(n, dev)
err = torch.zeros(2, dtype=torch.float64, device=dev)
uv = sh["uv"].clone()
fns = [lambda: D.reprojection_error_sum_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], 2.0, ws, err),
       lambda: D.reprojection_error_sums2_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], ws, err),
       lambda: D.add_noise_observations(uv, 0, 1e-9, 7),
       lambda: D.add_noise_observations_error_sums2_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], uv, 0, 1e-9, 7, ws, err)]
for fn in fns:
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
del sh, uv
torch.cuda.empty_cache()
ba = c2b.BAProblem(0)
c2b._lib.check(c2b._lib.lib().c2b_problem_synthetic_grid_layout(ba._h, 10, 10, 128, 20.0, 1.0, 1.0, 1.0))
row = ba.visibility_within_distance(10.0, True, 20.0, 1.0, fetch=False)
assert int(row[-1]) == 19_302_494
ba.adopt_visibility()
ba.cull()
print("ok", ba.num_observations())
