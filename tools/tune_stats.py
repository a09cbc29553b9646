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
import numpy as np  # noqa: E402
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
names = {0: "shipped", 1: "256 x 4 (r03/r04 shape)", 2: "512 x 4", 3: "256 x 4 pipe", 5: "256 x 2 pipe",
         7: "256 x 4 chunk", 12: "256 x 8 chunk",
         101: "256 x 4, 768 workgroups", 201: "256 x 4, 1024 workgroups", 205: "256 x 2 pipe, 1024 workgroups", 203: "256 x 4 pipe, 1024 workgroups",
         202: "512 x 4, 1024 workgroups", 207: "256 x 4 chunk, 1024 workgroups"}
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

# ---- where the time goes: the constant-rate wall clock (100 MHz) at seven points of every workgroup --------------------
raw.c2b_tune_set_probe.argtypes = [C.c_void_p]
probe = torch.zeros((1024, 8), dtype=torch.int64, device=dev)
for v in (1, 201):
    raw.c2b_tune_set_stats_variant(v)
    fn = lambda: D.stats(sh["camblk"], sh["pts4"], ws, st, centers=sh["cen4"])
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    raw.c2b_tune_set_probe(C.c_void_p(probe.data_ptr()))
    probe.zero_()
    fn()
    torch.cuda.synchronize()
    raw.c2b_tune_set_probe(None)
    t = probe.cpu().numpy().astype("float64") * 10e-3          # -> microseconds
    t = t[t[:, 0] > 0]                                        # the workgroups this launch had
    t0 = t[:, 0].min()
    last = int((t[:, 6] > 0).nonzero()[0][0]) if (t[:, 6] > 0).any() else -1
    print("probe %-24s starts spread %.2f us | loop: median %.2f max %.2f (ends at %.2f .. %.2f after the first start) | block reduce "
          "median %.2f | publish + ticket median %.2f | last arrival at %.2f | last workgroup: records %.2f, reduce %.2f, "
          "finish %.2f | end at %.2f us" % (
              names[v], t[:, 0].max() - t0, float(np.median((t[:, 1] - t[:, 0])[t[:, 0] > 0])), (t[:, 1] - t[:, 0]).max(), t[:, 1].min() - t0,
              t[:, 1].max() - t0, float(np.median((t[:, 2] - t[:, 1])[t[:, 0] > 0])), float(np.median((t[:, 3] - t[:, 2])[t[:, 0] > 0])), t[:, 3].max() - t0,
              t[last, 4] - t[last, 3], t[last, 5] - t[last, 4], t[last, 6] - t[last, 5], t[last, 6] - t0), flush=True)
