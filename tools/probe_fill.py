#!/usr/bin/env python3
"""Is a device whose Jacobian-shaped stores stream at 5.7 TB/s slow for EVERY pure write stream?  Times, in one process:
the library's store pattern (non-temporal 16-B stores, 1 KiB per wave instruction, three output arrays), torch's fill_
(its vectorised elementwise kernel) and zero_ (a memset) over the same byte count, and a 16-B copy of half that count.
   python tools/probe_fill.py"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=19302494)
a = ap.parse_args()
dev = torch.device("cuda", 0)
n = a.n
r = torch.empty((n, 2), dtype=torch.float64, device=dev)
Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
flat = torch.empty(n * 26, dtype=torch.float64, device=dev)
half = torch.empty(n * 13, dtype=torch.float64, device=dev)
half2 = torch.empty_like(half)
bytes_out = n * 208


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


for rnd in range(3):
    t = timed(lambda: D.calib_store_pattern(r, Jc, Jp))
    print("store pattern (3 arrays, nt)   %7.1f us  %6.0f GB/s" % (t * 1e6, bytes_out / t / 1e9))
    t = timed(lambda: flat.fill_(1.5))
    print("torch fill_ (one array)        %7.1f us  %6.0f GB/s" % (t * 1e6, bytes_out / t / 1e9))
    t = timed(lambda: flat.zero_())
    print("torch zero_ (memset)           %7.1f us  %6.0f GB/s" % (t * 1e6, bytes_out / t / 1e9))
    t = timed(lambda: (r.fill_(1.5), Jc.fill_(1.5), Jp.fill_(1.5)))
    print("torch fill_ (the three arrays) %7.1f us  %6.0f GB/s" % (t * 1e6, bytes_out / t / 1e9))
    t = timed(lambda: D.calib_copy(half, half2))
    print("16-B copy of half the bytes    %7.1f us  %6.0f GB/s (read + write)" % (t * 1e6, bytes_out / t / 1e9))
