#!/usr/bin/env python3
"""Does the residual+Jacobian kernel's time depend on WHERE its three output arrays sit relative to each other?
One process, one box: r / Jc / Jp are views into one large allocation at controlled byte offsets; for each
placement the store-pattern floor and the kernel are timed."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
n = sh["n_obs"]
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)
GB = 1 << 30
big = torch.empty(6 * GB // 8, dtype=torch.float64, device=dev)


def views(off_r, off_jc, off_jp):
    def v(off, cols):
        assert off % 16 == 0
        return big[off // 8: off // 8 + n * cols].view(n, cols)
    return v(off_r, 2), v(off_jc, 18), v(off_jp, 6)


def timed(fn, reps=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


print("n_obs", n, "base address %#x" % big.data_ptr())
r_bytes, jc_bytes, jp_bytes = n * 16, n * 144, n * 48
up = lambda x, a: (x + a - 1) // a * a  # noqa: E731
for name, pad in (("packed, 256-B aligned", 256), ("4 KiB aligned", 4096), ("2 MiB aligned", 2 << 20), ("1 GiB aligned", GB),
                  ("2 MiB + 4 KiB skew", None), ("2 MiB + 64 KiB skew", None), ("2 MiB + 1 MiB skew", None)):
    if pad is not None:
        o_r = 0
        o_jc = up(o_r + r_bytes, pad)
        o_jp = up(o_jc + jc_bytes, pad)
    else:
        skew = {"2 MiB + 4 KiB skew": 4096, "2 MiB + 64 KiB skew": 65536, "2 MiB + 1 MiB skew": 1 << 20}[name]
        o_r = 0
        o_jc = up(o_r + r_bytes, 2 << 20) + skew
        o_jp = up(o_jc + jc_bytes, 2 << 20) + 2 * skew
    r, Jc, Jp = views(o_r, o_jc, o_jp)
    t_store = timed(lambda: D.calib_store_pattern(r, Jc, Jp))
    t_k = timed(lambda: D.residual_jacobian_sum(sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws, err))
    print("%-24s store floor %6.1f us   kernel %6.1f us" % (name, t_store, t_k))
# separate allocations (what bench.py does)
del big
torch.cuda.empty_cache()
for k in range(3):
    junk = torch.empty((k * 12345 + 1) * 1024, dtype=torch.uint8, device=dev)
    r = torch.empty((n, 2), dtype=torch.float64, device=dev)
    Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
    Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
    t_store = timed(lambda: D.calib_store_pattern(r, Jc, Jp))
    t_k = timed(lambda: D.residual_jacobian_sum(sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws, err))
    print("separate allocations #%d (%#x %#x %#x)  store floor %6.1f us   kernel %6.1f us" % (k, r.data_ptr(), Jc.data_ptr(), Jp.data_ptr(), t_store, t_k))
    del r, Jc, Jp, junk
    torch.cuda.empty_cache()
