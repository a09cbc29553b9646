#!/usr/bin/env python3
"""A/B of two BUILDS of the product library in one process, on whatever device the call landed on: every light pass of the
hot path (row-structure forms, --blocks 128) and the step kernel, interleaved rounds, back to back and from swept caches,
outputs compared bit for bit.  How a source change is judged when devices differ by more than the change is worth.

    git stash; hipcc ... -o city2ba_amd/csrc/libcity2ba_hip_old.so city2ba_amd/csrc/capi.hip; git stash pop     (CPU box)
    python tools/ab_libs.py --old city2ba_amd/csrc/libcity2ba_hip_old.so [--new <the product library>] [--rounds 5]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from city2ba_amd import _lib as L  # noqa: E402
import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--old", required=True)
ap.add_argument("--new", default=L.LIB_PATH)
ap.add_argument("--blocks", type=int, default=128)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--k2", type=float, default=0.0, help="give every camera this k2 (and k1 = -k2): the |p|^4 path of the projection")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def load(path):
    lb = C.CDLL(os.path.abspath(path))
    for name, (res, args) in L.SIGNATURES.items():
        try:
            f = getattr(lb, name)
        except AttributeError:                     # an older build: entries that did not exist yet are not called through it here
            continue
        f.restype, f.argtypes = res, args
    return lb


L.lib()                                            # the product library first (shares torch's HIP runtime)
libs = {"old": load(a.old), "new": load(a.new)}
sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)
n = sh["n_obs"]
if a.k2 != 0.0:                                    # the blocked table: a group of 8 cameras = 256 doubles, light line k at [16 k, 16 k + 16)
    flat = sh["camblk"].flat
    c = torch.arange(sh["camblk"].shape[0], device=dev)
    at = (c // 8) * 256 + (c % 8) * 16
    flat[at + 13] = -a.k2
    flat[at + 14] = a.k2
    print("every camera: k1 = %g, k2 = %g (outputs of the two builds may differ by the rounding of |p|^4)" % (-a.k2, a.k2))
ws = D.workspace(n, dev)
err = torch.zeros(2, dtype=torch.float64, device=dev)
uv_out = torch.empty_like(sh["uv"])
uv_noise = sh["uv"].clone()
keep = torch.empty(n, dtype=torch.uint8, device=dev)
outs = D.JacobianOutputs(n, dev, max_attempts=1)
st = torch.empty(20, dtype=torch.float64, device=dev)
sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
aa = (sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"])
PASSES = [
    ("project_rows", lambda: D.project_rows(*aa, uv_out), lambda: uv_out),
    ("error_sum_rows L2", lambda: D.reprojection_error_sum_rows(*aa, sh["uv"], 2.0, ws, err), lambda: err),
    ("error_sums2_rows L1+L2", lambda: D.reprojection_error_sums2_rows(*aa, sh["uv"], ws, err), lambda: err),
    ("noise + error_sums2_rows", lambda: D.add_noise_observations_error_sums2_rows(*aa, uv_noise, 0, 1e-9, 7, ws, err), lambda: err),
    ("visibility_rows", lambda: D.visibility_rows(*aa, 10.0, uv_out, keep), lambda: keep),
    ("residual_jacobian_rows (512 x 2)", lambda: D.residual_jacobian_rows(*aa, sh["uv"], outs.r, outs.Jc, outs.Jp, 2.0, ws, err), lambda: outs.Jc[:n]),
]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def cold(fn):
    sweep.sum()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3


med = lambda v: sorted(v)[len(v) // 2]                                                          # noqa: E731
print("%-34s %22s %22s %9s %9s  bits" % ("pass", "old warm / cold us", "new warm / cold us", "warm", "cold"))
for name, fn, out in PASSES:
    res, bits = {}, {}
    w = {"old": [], "new": []}
    c = {"old": [], "new": []}
    for k in ("old", "new"):
        L._lib = libs[k]
        uv_noise.copy_(sh["uv"])
        fn()
        torch.cuda.synchronize()
        bits[k] = out().clone()
    for _ in range(a.rounds):
        for k in ("old", "new"):
            L._lib = libs[k]
            w[k].append(timed(fn, a.reps))
            c[k].append(med([cold(fn) for _ in range(3)]))
    same = bool(torch.equal(bits["old"], bits["new"]))
    print("%-34s %10.1f / %9.1f %10.1f / %9.1f %+8.1f%% %+8.1f%%  %s" % (
        name, med(w["old"]), med(c["old"]), med(w["new"]), med(c["new"]), (med(w["new"]) / med(w["old"]) - 1) * 100,
        (med(c["new"]) / med(c["old"]) - 1) * 100, "equal" if same else "DIFFER"), flush=True)
