#!/usr/bin/env python3
"""Do the RELATIVE positions of r, Jc and Jp decide the store rate?  One 12-GiB block; Jc at its start; Jp and r placed
behind it at a sweep of gaps (bytes, all multiples of 256) -- and, for comparison, the same three sizes as separate
allocations.  Prints store GB/s per layout."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
from city2ba_amd import device as D                            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19_302_494
dev = torch.device("cuda", 0)
KiB, MiB, GiB = 1 << 10, 1 << 20, 1 << 30


def rate(bufs, reps=4):
    for _ in range(2):
        D.calib_store_pattern(*bufs)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        D.calib_store_pattern(*bufs)
    e.record()
    torch.cuda.synchronize()
    return round(n * 208 / (s.elapsed_time(e) / reps * 1e-3) / 1e9, 1)


def up(x, a):
    return (x + a - 1) // a * a


block = torch.empty(12 * GiB // 8, dtype=torch.float64, device=dev)
szJc, szJp, szr = n * 144, n * 48, n * 16


def view(off, k):
    assert off % 16 == 0 and off + n * k * 8 <= 12 * GiB
    return block[off // 8: off // 8 + n * k].view(n, k)


out = {"separate": [], "one_block": {}}
for _ in range(4):
    bufs = (torch.empty((n, 2), dtype=torch.float64, device=dev), torch.empty((n, 18), dtype=torch.float64, device=dev),
            torch.empty((n, 6), dtype=torch.float64, device=dev))
    out["separate"].append(rate(bufs))
    del bufs
gaps = [0, 256, 4 * KiB, 64 * KiB, 1 * MiB, 2 * MiB + 4 * KiB, 16 * MiB, 96 * MiB, 160 * MiB, 1 * GiB, 1 * GiB + 96 * MiB, 2 * GiB + 37 * MiB]
for g1 in gaps:
    for g2 in (0, 64 * KiB, 96 * MiB):
        oJp = up(szJc, 256) + g1
        orr = up(oJp + szJp, 256) + g2
        if orr + szr > 12 * GiB:
            continue
        out["one_block"]["%d/%d" % (g1, g2)] = rate((view(orr, 2), view(0, 18), view(oJp, 6)))
# Jc not at the start of the block
for base in (0, 1 * MiB, 512 * MiB + 64 * KiB, 3 * GiB):
    oJp = base + up(szJc, 256) + 96 * MiB
    orr = up(oJp + szJp, 256) + 96 * MiB
    out["one_block"]["base%d" % base] = rate((view(orr, 2), view(base, 18), view(oJp, 6)))
print(json.dumps(out))
vals = sorted(out["one_block"].values())
print("separate allocations:", out["separate"], file=sys.stderr)
print("one block, %d layouts: min %.0f median %.0f max %.0f" % (len(vals), vals[0], vals[len(vals) // 2], vals[-1]), file=sys.stderr)
print({k: v for k, v in out["one_block"].items() if v > vals[len(vals) // 2] * 1.05}, file=sys.stderr)
