#!/usr/bin/env python3
"""r05: a rank's-eighth output arrays (2.4 M observations: 38 + 347 + 116 MB) never reach the 7 TB/s store class when they
are allocations of their own (docs/log_r01_r03.md: "large allocations only").  Do they when they are a WINDOW of a larger
allocation?  For several capacities: six sets each, the store rate of the whole set and of three windows of 2.4 M
observations (first, middle, last; 64-observation aligned)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402
entry.build_hip()
from city2ba_amd import device as D  # noqa: E402

dev = torch.device("cuda", 0)
M = 2_412_824


def rate(r, Jc, Jp, lo, n):
    a = (r[lo:lo + n], Jc[lo:lo + n], Jp[lo:lo + n])
    for _ in range(2):
        D.calib_store_pattern(*a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(8):
        D.calib_store_pattern(*a)
    e.record()
    torch.cuda.synchronize()
    return n * 208 / (s.elapsed_time(e) / 8 * 1e-3) / 1e9


for cap in (M, 2 * M, 4 * M, 6 * M, 8 * M):
    sets = [D.JacobianOutputs(cap, dev, max_attempts=1) for _ in range(6)]
    rows = []
    for o in sets:
        wins = [0, ((cap - M) // 2) // 64 * 64, (cap - M) // 64 * 64] if cap > M else [0]
        rows.append("%4.0f whole | windows %s" % (o.store_GBs / 1e0, " ".join("%4.0f" % rate(o.r, o.Jc, o.Jp, w, M) for w in wins)))
    print("capacity %9d observations (Jc %.2f GB):  %s" % (cap, cap * 144 / 1e9, "   ".join(rows)), flush=True)
    del sets, o
    torch.cuda.empty_cache()
