#!/usr/bin/env python3
"""A/B of the one-launch statistics pass (k_stats_pass1): workgroup size x entities per thread and trip, reading the
compact centre table (cen4) or the camblk records, back to back and with the caches swept.  Tuning library only.
    python tools/tune_stats.py [--blocks 128]"""
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

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda", 0)
raw = C.CDLL(L.LIB_PATH)
raw.c2b_tune_set_stats_variant.argtypes = [C.c_int]
sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)
ws = D.workspace(sh["n_obs"], dev)
st = torch.empty(20, dtype=torch.float64, device=dev)
sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
names = {0: "512 x 4 (shipped)", 1: "256 x 4 (r03/r04 shape)", 2: "512 x 5", 3: "512 x 8", 4: "512 x 2", 5: "256 x 8", 6: "512 x 3"}
ref = None
for cen in (sh["cen4"], None):
    for v, name in names.items():
        raw.c2b_tune_set_stats_variant(v)
        fn = lambda: D.stats(sh["camblk"], sh["pts4"], ws, st, centers=cen)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        got = st.clone()
        if ref is None:
            ref = got
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            fn()
        e.record()
        torch.cuda.synchronize()
        warm = s.elapsed_time(e) / 50 * 1e3
        cold = []
        for _ in range(7):
            sweep.sum()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            cold.append(s.elapsed_time(e) * 1e3)
        exact = bool(torch.equal(got[6:19], ref[6:19]))
        close = float(((got[:6] - ref[:6]).abs() / ref[:6].abs().clamp_min(1e-300)).max())
        print("%-8s %-26s warm %6.1f us  cold %6.1f us   min/max/origin equal: %s   mean/std rel diff %.1e" % (
            "cen4" if cen is not None else "camblk", name, warm, sorted(cold)[3], exact, close), flush=True)
