#!/usr/bin/env python3
"""r05: the store rate of output-sized allocations across the WHOLE device memory.  A fresh process allocates --sets output sets
of the --blocks 128 step (4.0 GB each) one after the other, keeps them all, and times the step kernel's store pattern into each
(c2b_jacobian_outputs_alloc with one attempt): a map of where this device takes streaming stores fast.
    python tools/probes/vram_store_map.py --sets 56"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sets", type=int, default=56)
ap.add_argument("--n-obs", type=int, default=19_302_494, help="observations per set (2 412 812 = a rank's eighth of --blocks 128: 0.5 GB)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
keep, rates = [], []
for k in range(a.sets):
    try:
        o = D.JacobianOutputs(a.n_obs, dev, max_attempts=1)
    except Exception as exc:                                    # out of memory: the map ends here
        print("set %d: %s" % (k, str(exc)[:80]))
        break
    keep.append(o)
    rates.append(int(round(o.store_GBs / 100.0)))
print("store rate (x 100 GB/s) of %d consecutive %.1f-GB output sets: %s" % (len(rates), a.n_obs * 208 / 1e9, " ".join("%d" % r for r in rates)))
print("fast (>= 6.9 TB/s): %d, in between: %d, slow (< 6.0): %d; first fast set: #%s" % (
    sum(r >= 69 for r in rates), sum(60 <= r < 69 for r in rates), sum(r < 60 for r in rates),
    next((i for i, r in enumerate(rates) if r >= 69), None)), flush=True)
