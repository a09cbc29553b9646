#!/usr/bin/env python3
"""Every light pass of the hot path on the bench grid (--blocks 128), the row-structure forms, 12 launches each back to
back: project, one-norm error, L1 + L2 in one pass, observation noise + both errors, observation noise alone, the
visibility predicate (byte mask and ballot words), the statistics (compact centre table), the step kernel twice.
A target for rocprofv3 --kernel-trace --stats / --pmc ... -- python3 tools/probe_light.py (tools/profile_light.sh)."""
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
err = torch.zeros(2, dtype=torch.float64, device=dev)
st = torch.empty(20, dtype=torch.float64, device=dev)
uv = sh["uv"].clone()
uv_out = torch.empty_like(uv)
keep = torch.empty(n, dtype=torch.uint8, device=dev)
keep_bits = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
outs = D.JacobianOutputs(n, dev, max_attempts=1)
a = (sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"])
fns = [lambda: D.project_rows(*a, uv_out),
       lambda: D.reprojection_error_sum_rows(*a, sh["uv"], 2.0, ws, err),
       lambda: D.reprojection_error_sums2_rows(*a, sh["uv"], ws, err),
       lambda: D.add_noise_observations_error_sums2_rows(*a, uv, 0, 1e-9, 7, ws, err),
       lambda: D.add_noise_observations(uv, 0, 1e-9, 7),
       lambda: D.visibility_rows(*a, 10.0, uv_out, keep),
       lambda: D.visibility_rows_bits(*a, 10.0, uv_out, keep_bits),
       lambda: D.stats(sh["camblk"], sh["pts4"], ws, st, centers=sh["cen4"]),
       lambda: D.cameras_prepare_state(sh["cam15"], sh["camblk"], centers=sh["cen4"])]
for fn in fns:
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
for _ in range(2):
    D.residual_jacobian_rows_placed(*a, sh["uv"], outs, 2.0, ws, err)
torch.cuda.synchronize()
print("ok", n)
