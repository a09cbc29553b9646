#!/usr/bin/env python3
"""Per-kernel timings of the whole hot path on the bench workload (one process, HIP events).
   python tools/bench_kernels.py [--blocks 128] [--reps 20]
Prints one JSON object; algorithmic bytes follow SURVEY section 8(d)."""
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
n, n_cam, n_pts = sh["n_obs"], sh["n_cam"], sh["n_pts"]
camblk, pts4, cam_idx, pt_idx, uv, cam15 = sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], sh["cam15"]
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)
uv_out = torch.empty_like(uv)
entity_bytes = n_cam * 72 + n_pts * 24


def timed(fn, reps=a.reps, rounds=5):
    """median over `rounds` of the mean of `reps` back-to-back launches; the first round is a warm-up and is dropped
    (a kernel timed cold right after set-up reads 5-8 % slow: clocks and caches)"""
    ts = []
    for k in range(rounds + 1):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        if k:
            ts.append(s.elapsed_time(e) / reps * 1e-3)
    return sorted(ts)[len(ts) // 2]


for _ in range(300):                                   # bring the device to its working clocks before anything is timed
    D.project(camblk, pts4, cam_idx, pt_idx, uv_out)
torch.cuda.synchronize()
flush_src = torch.zeros(1 << 27, dtype=torch.float64, device=dev)          # 1 GiB: four times the Infinity Cache


def timed_cold(fn, reps=7):
    """one launch at a time, the caches swept by a 1-GiB read before each (a read, not a copy: a copy would leave the
    caches full of dirty lines and charge their write-back to the kernel): what a single call on a problem nobody has
    touched since costs (the back-to-back figure finds the previous launch's inputs in the 256 MB Infinity Cache when
    results leave non-temporally)"""
    ts = []
    for _ in range(reps):
        flush_src.sum()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    return sorted(ts)[len(ts) // 2]


out = {"blocks": a.blocks, "n_obs": n, "n_cameras": n_cam, "n_points": n_pts, "kernels": {}}


def report(name, secs, units, alg_bytes, unit_name):
    out["kernels"][name] = {"us": round(secs * 1e6, 1), "G%s/s" % unit_name: round(units / secs / 1e9, 3),
                            "alg_GB/s": round(alg_bytes / secs / 1e9, 1), "frac_hbm_peak": round(alg_bytes / secs / 8e12, 4)}


t = timed(lambda: D.project(camblk, pts4, cam_idx, pt_idx, uv_out))
report("project", t, n, n * (4 + 16) + entity_bytes, "obs")
rows = sh["rows"]                      # the row-structure forms: camera from row_ptr tile records, no cam_idx stream
t = timed(lambda: D.project_rows(camblk, pts4, rows, pt_idx, uv_out))
report("project_rows", t, n, n * (4 + 16) + entity_bytes, "obs")
t = timed_cold(lambda: D.project_rows(camblk, pts4, rows, pt_idx, uv_out))
report("project_rows, caches flushed before every launch", t, n, n * (4 + 16) + entity_bytes, "obs")
for norm in (2.0, 1.0, 1.5):
    t = timed(lambda: D.reprojection_error_sum(camblk, pts4, cam_idx, pt_idx, uv, norm, ws, err))
    report("error_sum(norm=%g)" % norm, t, n, n * (4 + 16) + entity_bytes, "obs")
    t = timed(lambda: D.reprojection_error_sum_rows(camblk, pts4, rows, pt_idx, uv, norm, ws, err))
    report("error_sum_rows(norm=%g)" % norm, t, n, n * (4 + 16) + entity_bytes, "obs")
    if norm == 2.0:
        t = timed_cold(lambda: D.reprojection_error_sum_rows(camblk, pts4, rows, pt_idx, uv, norm, ws, err))
        report("error_sum_rows(norm=2), caches flushed before every launch", t, n, n * (4 + 16) + entity_bytes, "obs")
# the pair run_noise evaluates back to back (src/bin/city2ba.rs:283-287, 350-354), as ONE pass
err2 = torch.zeros(2, dtype=torch.float64, device=dev)
t = timed(lambda: D.reprojection_error_sums2_rows(camblk, pts4, rows, pt_idx, uv, ws, err2))
report("error_sums2_rows(L1+L2, one launch)", t, n, n * (4 + 16) + entity_bytes, "obs")
t = timed_cold(lambda: D.reprojection_error_sums2_rows(camblk, pts4, rows, pt_idx, uv, ws, err2))
report("error_sums2_rows(L1+L2, one launch), caches flushed before every launch", t, n, n * (4 + 16) + entity_bytes, "obs")
# add_noise's observation pass fused with that pair (src/noise.rs:152-170 -> src/bin/city2ba.rs:350-354): uv read + written once
uv3 = uv.clone()
t = timed(lambda: D.add_noise_observations_error_sums2_rows(camblk, pts4, rows, pt_idx, uv3, 0, 1e-6, 7, ws, err2))
report("add_noise_observations+error_sums2_rows (one launch)", t, n, n * (4 + 32) + entity_bytes, "obs")
t = timed_cold(lambda: D.add_noise_observations_error_sums2_rows(camblk, pts4, rows, pt_idx, uv3, 0, 1e-6, 7, ws, err2))
report("add_noise_observations+error_sums2_rows, caches flushed before every launch", t, n, n * (4 + 32) + entity_bytes, "obs")
del uv3
r = torch.empty((n, 2), dtype=torch.float64, device=dev)
Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
t = timed(lambda: D.residual_jacobian(camblk, pts4, cam_idx, pt_idx, uv, r, Jc, Jp, 2.0, ws))
report("residual_jacobian+err", t, n, bench.algorithmic_bytes(n, n_cam, n_pts), "obs")
t = timed(lambda: D.residual_jacobian_sum(camblk, pts4, cam_idx, pt_idx, uv, r, Jc, Jp, 2.0, ws, err))
report("residual_jacobian_sum (one launch)", t, n, bench.algorithmic_bytes(n, n_cam, n_pts), "obs")
t = timed(lambda: D.residual_jacobian_rows(camblk, pts4, rows, pt_idx, uv, r, Jc, Jp, 2.0, ws, err))
report("residual_jacobian_rows (one launch, bench step)", t, n, bench.algorithmic_bytes(n, n_cam, n_pts), "obs")
t = timed_cold(lambda: D.residual_jacobian_rows(camblk, pts4, rows, pt_idx, uv, r, Jc, Jp, 2.0, ws, err))
report("residual_jacobian_rows, caches flushed before every launch", t, n, bench.algorithmic_bytes(n, n_cam, n_pts), "obs")
del r, Jc, Jp

# visibility predicate on a candidate list of the same size class as the generator's
keep = torch.empty(n, dtype=torch.uint8, device=dev)
t = timed(lambda: D.visibility_pairs(camblk, pts4, cam_idx, pt_idx, 10.0, uv_out, keep))
report("visibility_pairs", t, n, n * (8 + 16 + 1) + entity_bytes, "pairs")
t = timed(lambda: D.visibility_rows(camblk, pts4, rows, pt_idx, 10.0, uv_out, keep))
report("visibility_rows", t, n, n * (4 + 16 + 1) + entity_bytes, "pairs")
t = timed_cold(lambda: D.visibility_rows(camblk, pts4, rows, pt_idx, 10.0, uv_out, keep))
report("visibility_rows, caches flushed before every launch", t, n, n * (4 + 16 + 1) + entity_bytes, "pairs")
t = timed(lambda: D.Rows(rows.row_ptr, n), 3, 2)
report("rows_pack (once per list)", t, n, n_cam * 8 + n // 4, "obs")

uv2 = uv.clone()
t = timed(lambda: D.add_noise_observations(uv2, 0, 1e-6, 7))
report("add_noise_observations", t, n, n * 32, "obs")

st = torch.empty(20, dtype=torch.float64, device=dev)
t = timed(lambda: D.stats(camblk, pts4, ws, st, centers=sh["cen4"]))
report("stats(mean,std,extent,origin) [compact centre table]", t, n_cam + n_pts, n_cam * 24 + n_pts * 24, "entities")
t = timed(lambda: D.stats(camblk, pts4, ws, st))
report("stats(mean,std,extent,origin) [centres read from camblk]", t, n_cam + n_pts, n_cam * 24 + n_pts * 24, "entities")
t = timed(lambda: D.cameras_prepare_state(cam15, camblk, centers=sh["cen4"]))
report("cameras_prepare_state (+ centre table)", t, n_cam, n_cam * (120 + 256 + 32), "cams")
bal9 = D.cameras_to_bal(cam15)
t = timed(lambda: D.cameras_to_bal(cam15))
report("cameras_to_bal", t, n_cam, n_cam * (120 + 72), "cams")
t = timed(lambda: D.cameras_from_bal(bal9))
report("cameras_from_bal", t, n_cam, n_cam * (120 + 72), "cams")
c2, p2 = cam15.clone(), pts4.clone()
t = timed(lambda: D.add_drift_normalized(c2, p2, st, 1e-9, 1e-9, 0.1, 3))
report("add_drift_normalized", t, n_cam + n_pts, 2 * (n_cam * 120 + n_pts * 24), "entities")
t = timed(lambda: D.add_noise_entities(c2, p2, st, 1e-9, 1e-9, 1e-9, 4))
report("add_noise_entities", t, n_cam + n_pts, 2 * (n_cam * 120 + n_pts * 24), "entities")
print(json.dumps(out, indent=1))
