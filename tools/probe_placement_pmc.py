#!/usr/bin/env python3
"""One bounded diagnosis of the fast / slow output allocations: in ONE process allocate up to eight
r/Jc/Jp sets, time the store pattern (c2b_calib_store_pattern) in each, then launch it three times into the SLOWEST and
three times into the FASTEST set with different grid sizes so that a rocprofv3 --pmc pass of this script can tell the
dispatches apart (slow: n - 16384 observations; fast: n - 8192).  Run plain it prints the rates; run under rocprofv3 the
counter CSV holds per-dispatch values (tools/probe_placement_pmc.sh drives the passes and tools/summarize_placement_pmc.py
reads them)."""
import json
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
from city2ba_amd import device as D                            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19_302_494
sets_wanted = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)


def rate(bufs, n_obs):
    for _ in range(2):
        D.calib_store_pattern(*[b[:n_obs] for b in bufs])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(4):
        D.calib_store_pattern(*[b[:n_obs] for b in bufs])
    e.record()
    torch.cuda.synchronize()
    return n_obs * 208 / (s.elapsed_time(e) / 4 * 1e-3) / 1e9


sets = []
for _ in range(sets_wanted):
    try:
        bufs = tuple(torch.empty((n, k), dtype=torch.float64, device=dev) for k in (2, 18, 6))
    except RuntimeError:
        break
    sets.append((rate(bufs, n), bufs))
rates = [round(x[0], 1) for x in sets]
slow = min(range(len(sets)), key=lambda i: sets[i][0])
fast = max(range(len(sets)), key=lambda i: sets[i][0])
n_fast, n_slow = n - 8192, n - 16384
tagged = {"slow": [], "fast": []}
for _ in range(3):
    tagged["slow"].append(round(rate(sets[slow][1], n_slow), 1))  # grid of n - 16384 observations = the slow allocation
    tagged["fast"].append(round(rate(sets[fast][1], n_fast), 1))  # grid of n - 8192 observations  = the fast allocation


def grid_threads(k):
    return ((k + 63) // 64 + 7) // 8 * 512


print(json.dumps({"n_obs": n, "store_GBs_per_set": rates, "slow_set": slow, "fast_set": fast, "n_slow_tag": n_slow,
                  "n_fast_tag": n_fast, "grid_slow": grid_threads(n_slow), "grid_fast": grid_threads(n_fast),
                  "rates_tagged": tagged,
                  "data_ptrs": {"slow": [hex(b.data_ptr()) for b in sets[slow][1]],
                                "fast": [hex(b.data_ptr()) for b in sets[fast][1]]}}), flush=True)
