#!/usr/bin/env python3
"""Is the slow / fast store effect a property of WHICH memory an allocation got, at what granularity?  Allocates
`count` separate chunks of `mib` MiB each, runs the Jacobian kernel's store pattern inside every chunk (r / Jc / Jp carved
out of it) and prints the store rate per chunk, then repeats the measurement to show which differences are stable."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
from city2ba_amd import device as D                            # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
count = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = torch.device("cuda", 0)
n = (mib << 20) // 208 // 64 * 64


def carve(chunk):
    a = chunk[: 2 * n].view(n, 2)
    b = chunk[2 * n: 20 * n].view(n, 18)
    c = chunk[20 * n: 26 * n].view(n, 6)
    return a, b, c


def rate(bufs, reps=8):
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


chunks = []
for _ in range(count):
    try:
        chunks.append(torch.empty((mib << 20) // 8, dtype=torch.float64, device=dev))
    except RuntimeError:
        break
passes = [[rate(carve(c)) for c in chunks] for _ in range(3)]
print(json.dumps({"chunk_MiB": mib, "chunks": len(chunks), "ptrs": [hex(c.data_ptr()) for c in chunks], "store_GBs_passes": passes}))
best = [max(p[i] for p in passes) for i in range(len(chunks))]
srt = sorted(best)
print("chunks %d x %d MiB: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f GB/s" % (
    len(chunks), mib, srt[0], srt[len(srt) // 10], srt[len(srt) // 2], srt[len(srt) * 9 // 10], srt[-1]), file=sys.stderr)
print(" ".join("%4.0f" % (x / 10) for x in best), file=sys.stderr)
