#!/usr/bin/env python3
"""Allocation ORDER vs kernel time, one fresh process per recipe (tools/probe_placement.py showed 742 vs 863 us for the
same kernel in one process depending on which allocation the outputs live in).
   python tools/probe_placement2.py            # runs every recipe in its own subprocess
   python tools/probe_placement2.py A          # one recipe"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N128 = 19302494

if len(sys.argv) == 1 or sys.argv[1] == "all":
    for rep in range(2):
        for recipe in (sys.argv[2:] or ["A", "C", "F8", "F40", "F120", "G"]):
            out = subprocess.run([sys.executable, __file__, recipe], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
            print(out.strip().splitlines()[-1] if out.strip() else "%s: no output" % recipe, flush=True)
    sys.exit(0)

recipe = sys.argv[1]
import torch  # noqa: E402

import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def outputs(n):
    return (torch.empty((n, 2), dtype=torch.float64, device=dev), torch.empty((n, 18), dtype=torch.float64, device=dev),
            torch.empty((n, 6), dtype=torch.float64, device=dev))


if recipe == "C":                       # outputs first, in a pristine address space
    r, Jc, Jp = outputs(N128)
if recipe.startswith("F"):              # outputs first, but behind a dummy allocation of so many GB that stays alive
    dummy = torch.empty(int(recipe[1:]) << 30, dtype=torch.uint8, device=dev)
    r, Jc, Jp = outputs(N128)
if recipe == "G":                       # outputs first, allocated, freed to the driver and allocated again
    r, Jc, Jp = outputs(N128)
    del r, Jc, Jp
    torch.cuda.empty_cache()
    r, Jc, Jp = outputs(N128)
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
n = sh["n_obs"]
assert n == N128
if recipe == "A":                       # what bench.py did: straight after build_shard (allocator cache full of its temporaries)
    r, Jc, Jp = outputs(n)
elif recipe == "B":                     # return the cached temporaries to the driver first
    torch.cuda.empty_cache()
    r, Jc, Jp = outputs(n)
elif recipe == "D":                     # one block, carved
    torch.cuda.empty_cache()
    big = torch.empty(n * 26, dtype=torch.float64, device=dev)
    r, Jc, Jp = big[: 2 * n].view(n, 2), big[2 * n: 20 * n].view(n, 18), big[20 * n:].view(n, 6)
elif recipe == "E":                     # inputs re-made after the cache is emptied too
    torch.cuda.empty_cache()
    for k in ("camblk", "pts4", "cam_idx", "pt_idx", "uv"):
        sh[k] = sh[k].clone()
    torch.cuda.empty_cache()
    r, Jc, Jp = outputs(n)
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)


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
    return s.elapsed_time(e) / reps * 1e3


t_store = timed(lambda: D.calib_store_pattern(r, Jc, Jp))
t_k = timed(lambda: D.residual_jacobian_sum(sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws, err))
print("recipe %s: store floor %6.1f us  kernel %6.1f us   r %#x Jc %#x Jp %#x" % (recipe, t_store, t_k, r.data_ptr(), Jc.data_ptr(), Jp.data_ptr()))
