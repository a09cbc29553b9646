#!/usr/bin/env python3
"""r05 experiment (tuning library): does any cache-policy spelling of a gfx950 global store stream the step kernel's store
geometry (208 B per observation as whole 1-KiB lines; kernels.hpp: k_store_pattern_pol) faster than the shipped `nt` --
in particular into output sets of the slow class (5.6-5.9 TB/s)?  Eight output-sized sets, the pattern under the eight
spellings in the slowest and in the fastest of them, interleaved rounds.
    python tools/probe_store_policy.py [--blocks 128] [--rounds 4]"""
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

POL = ["plain", "nt (shipped)", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"]
ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=128)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--sets", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
raw = C.CDLL(L.LIB_PATH)
raw.c2b_tune_store_pattern_policy.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
n = {128: 19302494, 32: 1225066}.get(a.blocks) or bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)["n_obs"]


def timed(fn, reps=6):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def pat(o, pol):
    rc = raw.c2b_tune_store_pattern_policy(n, o.r.data_ptr(), o.Jc.data_ptr(), o.Jp.data_ptr(), pol, None)
    assert rc == 0, rc


sets = []
for k in range(a.sets):
    o = D.JacobianOutputs(n, dev, max_attempts=1)
    sets.append((n * 208 / timed(lambda: D.calib_store_pattern(o.r, o.Jc, o.Jp)) / 1e3, o))
rates = [round(r, 1) for r, _ in sets]
print("store GB/s of %d output sets: %s -> device class %s" % (a.sets, rates, bench.store_class(rates)), flush=True)
order = sorted(range(len(sets)), key=lambda k: sets[k][0])
picks = [("slowest set", order[0])] + ([("fastest set", order[-1])] if sets[order[-1]][0] > 1.05 * sets[order[0]][0] else [])
for label, k in picks:
    o = sets[k][1]
    t = {p: [] for p in range(8)}
    for _ in range(a.rounds):
        for p in range(8):
            t[p].append(timed(lambda: pat(o, p)))
    base = sorted(t[1])[len(t[1]) // 2]
    print("\n%s (%.0f GB/s by the product's pattern kernel)" % (label, sets[k][0]))
    for p in range(8):
        m = sorted(t[p])[len(t[p]) // 2]
        print("  %-14s %7.1f us  %6.0f GB/s  %+5.1f %% vs nt" % (POL[p], m, n * 208 / m / 1e3, (m / base - 1) * 100), flush=True)
