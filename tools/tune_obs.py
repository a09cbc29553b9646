#!/usr/bin/env python3
"""A/B the light per-observation kernels (project / error / visibility) over (observations per lane, waves per
workgroup) in ONE process, interleaved rounds, on the bench workload; outputs checked bit-for-bit against the first
variant.   python tools/tune_obs.py [--blocks 128] [--variants 208,108,204,216,308,408,404]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402
from city2ba_amd import _lib as L  # noqa: E402

L.LIB_PATH = entry.build_tune()        # the tuning library (kernel variants + selectors), never the product one
import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=128)
ap.add_argument("--variants", default="308,208,408,2008")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
variants = [int(v) for v in a.variants.split(",")]

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)
n = sh["n_obs"]
raw = C.CDLL(L.LIB_PATH)
raw.c2b_tune_set_observation_variant.argtypes = [C.c_int]
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)
camblk, pts4, ci, pi, uv = sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"]
uv_out = torch.empty_like(uv)
# the row-structure forms (camera from row_ptr + tile records instead of a 4-byte index per observation): extra modes,
# not variants -- they are product entry points
n_cam = camblk.shape[0]
row_ptr = torch.zeros(n_cam + 1, dtype=torch.int64, device=dev)
row_ptr[1:] = torch.cumsum(torch.bincount(ci.long(), minlength=n_cam), 0)
rows = D.Rows(row_ptr)
uv_rows = torch.empty_like(uv)
keep = torch.empty(n, dtype=torch.uint8, device=dev)
keep_rows = torch.empty(n, dtype=torch.uint8, device=dev)
err_rows = torch.zeros(1, dtype=torch.float64, device=dev)
print("tiles %d, with an empty list inside %d" % (rows.tiles.shape[0], int((rows.tiles[:, 2] < 0).sum())))

modes = {
    "project": lambda: D.project(camblk, pts4, ci, pi, uv_out),
    "error_L2": lambda: D.reprojection_error_sum(camblk, pts4, ci, pi, uv, 2.0, ws, err),
    "visibility": lambda: D.visibility_pairs(camblk, pts4, ci, pi, 10.0, uv_out, keep),
}
r_o = torch.empty((n, 2), dtype=torch.float64, device=dev)
Jc_o = torch.empty((n, 18), dtype=torch.float64, device=dev)
Jp_o = torch.empty((n, 6), dtype=torch.float64, device=dev)
modes["jacobian_sum"] = lambda: D.residual_jacobian_sum(camblk, pts4, ci, pi, uv, r_o, Jc_o, Jp_o, 2.0, ws, err)
row_modes = {
    "jacobian_sum_rows": lambda: D.residual_jacobian_rows(camblk, pts4, rows, pi, uv, r_o, Jc_o, Jp_o, 2.0, ws, err_rows),
    "project_rows": lambda: D.project_rows(camblk, pts4, rows, pi, uv_rows),
    "project_rows_same_output": lambda: D.project_rows(camblk, pts4, rows, pi, uv_out),
    "project_rows_shard_rows": lambda: D.project_rows(camblk, pts4, sh["rows"], pi, uv_out),
    "error_L2_rows": lambda: D.reprojection_error_sum_rows(camblk, pts4, rows, pi, uv, 2.0, ws, err_rows),
    "visibility_rows": lambda: D.visibility_rows(camblk, pts4, rows, pi, 10.0, uv_rows, keep_rows),
}

ref = {}
for v in variants:
    raw.c2b_tune_set_observation_variant(v)
    uv_out.fill_(float("nan"))
    modes["project"]()
    torch.cuda.synchronize()
    p = uv_out.clone()
    modes["error_L2"]()
    torch.cuda.synchronize()
    e = err.item()
    modes["visibility"]()
    torch.cuda.synchronize()
    k = keep.clone()
    if not ref:
        ref = {"p": p, "e": e, "k": k}
    else:
        print("variant %d: project bit-equal %s, keep equal %s, error rel diff %.1e" %
              (v, torch.equal(p, ref["p"]), torch.equal(k, ref["k"]), abs(e - ref["e"]) / max(ref["e"], 1e-300)))

raw.c2b_tune_set_observation_variant(variants[0])
modes["project"](); row_modes["project_rows"]()
torch.cuda.synchronize()
eq_p = torch.equal(uv_out, uv_rows)
modes["visibility"](); row_modes["visibility_rows"](); modes["error_L2"](); row_modes["error_L2_rows"]()
torch.cuda.synchronize()
print("rows forms: project bit-equal %s, keep equal %s, uv equal %s, error equal %s" %
      (eq_p, torch.equal(keep, keep_rows), torch.equal(uv_out.view(torch.int64), uv_rows.view(torch.int64)), err.item() == err_rows.item()))

modes["jacobian_sum"]()
torch.cuda.synchronize()
ja = (r_o.clone(), Jc_o[::97].clone(), Jp_o[::89].clone(), err.item())
row_modes["jacobian_sum_rows"]()
torch.cuda.synchronize()
print("rows Jacobian: r equal %s, Jc sample equal %s, Jp sample equal %s, sum equal %s" %
      (torch.equal(ja[0], r_o), torch.equal(ja[1], Jc_o[::97]), torch.equal(ja[2], Jp_o[::89]), ja[3] == err_rows.item()))

times = {(m, v): [] for m in modes for v in variants}
times.update({(m, 0): [] for m in row_modes})
for _ in range(a.rounds):
    for m, fn in row_modes.items():
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        times[(m, 0)].append(s.elapsed_time(e) / a.reps * 1e3)
    for m, fn in modes.items():
        for v in variants:
            raw.c2b_tune_set_observation_variant(v)
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            times[(m, v)].append(s.elapsed_time(e) / a.reps * 1e3)
print("n_obs=%d" % n)
for m in modes:
    for v in variants:
        t = sorted(times[(m, v)])
        print("%-10s OPL=%d WPB=%-2d: median %.1f us  min %.1f us  %.1f Gobs/s" % (m, v // 100, v % 100, t[len(t) // 2], t[0], n / t[len(t) // 2] / 1e3))
for m in row_modes:
    t = sorted(times[(m, 0)])
    print("%-16s      : median %.1f us  min %.1f us  %.1f Gobs/s" % (m, t[len(t) // 2], t[0], n / t[len(t) // 2] / 1e3))
