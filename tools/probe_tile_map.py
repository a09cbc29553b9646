#!/usr/bin/env python3
"""Does the slow / fast allocation effect depend on how the chip's concurrent write fronts are laid
over the address space?  Eight output-sized allocation sets in one process; in each, the store pattern under seven
workgroup -> tile maps (c2b_calib_store_pattern_map): XCD eighths (the kernels' map: 8 fronts per array), launch order
(one moving window), chunked K = 4 ... 4096.  Prints store GB/s per (set, map)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
from city2ba_amd import device as D                            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19_302_494
dev = torch.device("cuda", 0)
maps = [0, 1, 4, 16, 64, 256, 1024, 4096]


def rate(bufs, m):
    for _ in range(2):
        D.calib_store_pattern_map(*bufs, m)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(4):
        D.calib_store_pattern_map(*bufs, m)
    e.record()
    torch.cuda.synchronize()
    return round(n * 208 / (s.elapsed_time(e) / 4 * 1e-3) / 1e9, 1)


sets = []
for _ in range(8):
    sets.append(tuple(torch.empty((n, k), dtype=torch.float64, device=dev) for k in (2, 18, 6)))
table = []
for bufs in sets:
    row = {str(m): rate(bufs, m) for m in maps}
    row["ptr_r"] = hex(bufs[0].data_ptr())
    table.append(row)
print(json.dumps({"n_obs": n, "maps": maps, "store_GBs": table}))
for row in table:
    print("  ".join("%7.1f" % row[str(m)] for m in maps), file=sys.stderr)
