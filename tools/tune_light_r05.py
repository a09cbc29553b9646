#!/usr/bin/env python3
"""r05 experiments on the light per-observation passes (--blocks 128, row-structure forms, tuning library):
  * stagger: wave w of every workgroup sleeps w * K * 64 cycles before its first load (do waves started in lockstep
    serialise on each other's phases?);
  * the gather's share: the same launches with every point index 0 (one line for all gathers) and with consecutive
    point indices (perfectly coalesced gathers) -- timing only, the outputs are of another problem.
Each case back to back and with the caches swept.   python tools/tune_light_r05.py"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402
from city2ba_amd import _lib as L  # noqa: E402

L.LIB_PATH = entry.build_tune()
import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

dev = torch.device("cuda", 0)
raw = C.CDLL(L.LIB_PATH)
raw.c2b_tune_set_stagger.argtypes = [C.c_int]
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
n = sh["n_obs"]
ws = D.workspace(n, dev)
err = torch.zeros(2, dtype=torch.float64, device=dev)
uv_out = torch.empty_like(sh["uv"])
uv2 = sh["uv"].clone()
keep = torch.empty(n, dtype=torch.uint8, device=dev)
sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
idx = {"real": sh["pt_idx"], "all zero": torch.zeros_like(sh["pt_idx"]),
       "consecutive": (torch.arange(n, dtype=torch.int64, device=dev) % sh["n_pts"]).to(torch.int32)}


def cases(pi):
    a = (sh["camblk"], sh["pts4"], sh["rows"], pi)
    return {"project_rows": lambda: D.project_rows(*a, uv_out),
            "error_sums2_rows": lambda: D.reprojection_error_sums2_rows(*a, sh["uv"], ws, err),
            "noise+error_sums2_rows": lambda: D.add_noise_observations_error_sums2_rows(*a, uv2, 0, 1e-9, 7, ws, err),
            "visibility_rows": lambda: D.visibility_rows(*a, 10.0, uv_out, keep)}


def measure(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    warm = s.elapsed_time(e) / 20 * 1e3
    cold = []
    for _ in range(5):
        sweep.sum()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        cold.append(s.elapsed_time(e) * 1e3)
    return warm, sorted(cold)[2]


for name in ("project_rows", "error_sums2_rows", "noise+error_sums2_rows", "visibility_rows"):
    for K in (0, 1, 2, 4, 8, 16, 32):
        raw.c2b_tune_set_stagger(K)
        w, c = measure(cases(idx["real"])[name])
        print("%-24s stagger %2d x 64 cycles per wave index: warm %6.1f us  cold %6.1f us" % (name, K, w, c), flush=True)
    for K in (-12, -25, -50, -100, -200):                      # workgroup-slot stagger of the first generation (r05)
        raw.c2b_tune_set_stagger(K)
        w, c = measure(cases(idx["real"])[name])
        print("%-24s first-generation workgroups delayed by slot x %3d x 64 cycles: warm %6.1f us  cold %6.1f us" % (name, -K, w, c), flush=True)
    raw.c2b_tune_set_stagger(0)
    for kind in ("all zero", "consecutive"):
        w, c = measure(cases(idx[kind])[name])
        print("%-24s point indices %-12s          : warm %6.1f us  cold %6.1f us" % (name, kind, w, c), flush=True)
