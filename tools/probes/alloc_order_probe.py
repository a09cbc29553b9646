#!/usr/bin/env python3
"""r05: does the ORDER of a process's allocations decide which output set is a fast one?  (The device lottery of this round saw,
on 8 of 19 devices, the first 4-GB set of a fresh process stream at 6.1-7.1 TB/s and every later one at 5.7.)
Two fresh processes on the same device: (A) ballast first -- --ballast-gb of other allocations, like bench.py's problem set-up --
then eight output sets; (B) eight output sets first.      python tools/probes/alloc_order_probe.py --ballast-gb 2"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ballast-gb", type=float, default=0.0)
ap.add_argument("--free-ballast", action="store_true", help="free the ballast (back to the driver) before the output sets are allocated")
a = ap.parse_args()
dev = torch.device("cuda", 0)
ballast = [torch.empty(256 << 20, dtype=torch.uint8, device=dev) for _ in range(int(a.ballast_gb * 4))]
for b in ballast:
    b.zero_()
torch.cuda.synchronize()
if a.free_ballast:
    del ballast
    torch.cuda.empty_cache()
keep, rates = [], []
for _ in range(8):
    o = D.JacobianOutputs(19_302_494, dev, max_attempts=1)
    keep.append(o)
    rates.append(round(o.store_GBs))
print("ballast %.1f GB%s -> store GB/s of 8 output sets: %s" % (a.ballast_gb, " (freed)" if a.free_ballast else "", rates), flush=True)
