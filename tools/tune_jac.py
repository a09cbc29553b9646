#!/usr/bin/env python3
"""A/B the residual+Jacobian kernel variants in ONE process, interleaved rounds (guide rule 24), on the
bench workload.  Also checks every variant's outputs bit-for-bit against the first variant.
   python tools/tune_jac.py [--blocks 128] [--variants 0,1,2,...]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402
from city2ba_amd import _lib as L  # noqa: E402

L.LIB_PATH = entry.build_tune()        # the tuning library (kernel variants + selectors), never the product one
import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=128)
ap.add_argument("--variants", default="0,40,14,104,-1,-2")   # 0 = the shipped kernel (index form), 40 = the same with cached loads   # -1 = store pattern only, -2 = 16-B copy of the same bytes
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--no-err", action="store_true", help="launch without the fused error reduce")
a = ap.parse_args()
variants = [int(v) for v in a.variants.split(",")]

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)
n = sh["n_obs"]
alg = bench.algorithmic_bytes(n, sh["n_cam_local"], sh["n_pts"])
raw = C.CDLL(L.LIB_PATH)
raw.c2b_tune_set_jacobian_variant.argtypes = [C.c_int]
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)


copy_src = torch.empty(int(alg) // 16 // 2 * 2, dtype=torch.float64, device=dev).normal_()
copy_dst = torch.empty_like(copy_src)          # copy moves 2 x 8 x numel = alg bytes (read + write)


def run(v, r, Jc, Jp):
    if v == -1:
        return D.calib_store_pattern(r, Jc, Jp)
    if v == -2:
        return D.calib_copy(copy_src, copy_dst)
    raw.c2b_tune_set_jacobian_variant(v)
    D.residual_jacobian(sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, None if a.no_err else ws)


bufs = [torch.empty((n, 2), dtype=torch.float64, device=dev), torch.empty((n, 18), dtype=torch.float64, device=dev),
        torch.empty((n, 6), dtype=torch.float64, device=dev)]
ref = [torch.empty_like(b) for b in bufs]
run(variants[0], *ref)
D.error_sum_finish(ws, n, err)
torch.cuda.synchronize()
e_ref = err.item()
for v in variants[1:]:
    for b in bufs:
        b.fill_(float("nan"))
    run(v, *bufs)
    D.error_sum_finish(ws, n, err)
    torch.cuda.synchronize()
    same = (20 <= v < 100) or v < 0 or all(torch.equal(x, y) for x, y in zip(bufs, ref))
    print("variant %d: outputs bit-equal to variant %d: %s; err rel diff %.2e" %
          (v, variants[0], same, abs(err.item() - e_ref) / e_ref))

times = {v: [] for v in variants}
for _ in range(a.rounds):
    for v in variants:
        run(v, *bufs)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.reps):
            run(v, *bufs)
        e.record()
        torch.cuda.synchronize()
        times[v].append(s.elapsed_time(e) / a.reps * 1e3)
print("n_obs=%d alg=%.3f GB" % (n, alg / 1e9))
for v in variants:
    t = sorted(times[v])
    med = t[len(t) // 2]
    print("variant %d: median %.1f us  min %.1f us  -> %.0f GB/s  frac %.3f  %.2f Gobs/s" %
          (v, med, t[0], alg / med / 1e3, alg / med / 1e3 / 8000.0, n / med / 1e3))
