#!/usr/bin/env python3
"""r05: is the READ rate of a buffer also a matter of where in the device memory it lies, like the store rate
(tools/probes/vram_store_map.py)?  A fresh process allocates --bufs consecutive buffers of --gb GB, keeps them all, and times into /
out of each: torch's fill (plain streaming stores), the library's store pattern geometry is not involved; a sum (streaming reads);
a copy from the buffer into ONE fixed destination buffer allocated first (reads from here, writes there).
    python tools/probes/vram_rw_map.py --bufs 56 --gb 4"""
import argparse
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--bufs", type=int, default=56)
ap.add_argument("--gb", type=float, default=4.0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
n = int(a.gb * 1e9 / 8)
dst = torch.empty(n, dtype=torch.float64, device=dev)


def timed(fn, reps=3):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


keep, w, r, c = [], [], [], []
for k in range(a.bufs):
    try:
        t = torch.empty(n, dtype=torch.float64, device=dev)
    except Exception:
        break
    keep.append(t)
    w.append(int(round(n * 8 / timed(lambda: t.fill_(1.0)) / 1e11)))
    r.append(int(round(n * 8 / timed(lambda: t.sum()) / 1e11)))
    c.append(int(round(2 * n * 8 / timed(lambda: dst.copy_(t)) / 1e11)))
print("%d consecutive %.1f-GB buffers (x 100 GB/s)" % (len(keep), a.gb))
print("fill  (stores)        :", " ".join("%d" % x for x in w))
print("sum   (reads)         :", " ".join("%d" % x for x in r))
print("copy out (r here + w) :", " ".join("%d" % x for x in c), flush=True)
