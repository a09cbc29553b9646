#!/usr/bin/env python3
"""r05 experiment (tuning library): does any cache-policy spelling of a gfx950 global store stream the step kernel's store
geometry (208 B per observation as whole 1-KiB lines; kernels.hpp: k_store_pattern_pol) faster than the shipped `nt` --
in particular into output sets of the slow class (5.6-5.9 TB/s)?  Eight output-sized sets, the pattern under the eight
spellings in the slowest and in the fastest of them, interleaved rounds.
    python tools/probe_store_policy.py [--blocks 128] [--rounds 4]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402
from city2ba_amd import _lib as L  # noqa: E402

L.LIB_PATH = entry.build_tune()
import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

POL = ["plain", "nt (shipped)", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"]
ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=128)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--kernel", default="", help="comma-separated spellings (0-7) to A/B in the real step kernel, e.g. 1,0,2,3,4")
ap.add_argument("--build-only", action="store_true", help="(CPU box) compile the --kernel libraries and stop")
a = ap.parse_args()
dev = torch.device("cuda", 0)
if a.build_only:
    import subprocess
    for p in [int(p) for p in a.kernel.split(",")]:
        path = os.path.join(entry.CSRC, "libcity2ba_hip_pol%d.so" % p)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-DC2B_STORE_POL=%d" % p] + entry.HIPCC_FLAGS + ["-o", path, os.path.join(entry.CSRC, "capi.hip")])
        print("built", path)
    sys.exit(0)
torch.cuda.set_device(0)
raw = C.CDLL(L.LIB_PATH)
raw.c2b_tune_store_pattern_policy.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
n = {128: 19302494, 32: 1225066}.get(a.blocks) or bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)["n_obs"]


def timed(fn, reps=6):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def pat(o, pol):
    rc = raw.c2b_tune_store_pattern_policy(n, o.r.data_ptr(), o.Jc.data_ptr(), o.Jp.data_ptr(), pol, None)
    assert rc == 0, rc


sets = []
for k in range(a.sets):
    o = D.JacobianOutputs(n, dev, max_attempts=1)
    sets.append((n * 208 / timed(lambda: D.calib_store_pattern(o.r, o.Jc, o.Jp)) / 1e3, o))
rates = [round(r, 1) for r, _ in sets]
print("store GB/s of %d output sets: %s -> device class %s" % (a.sets, rates, bench.store_class(rates)), flush=True)
order = sorted(range(len(sets)), key=lambda k: sets[k][0])
picks = [("slowest set", order[0])] + ([("fastest set", order[-1])] if sets[order[-1]][0] > 1.05 * sets[order[0]][0] else [])
for label, k in picks:
    o = sets[k][1]
    t = {p: [] for p in range(8)}
    for _ in range(a.rounds):
        for p in range(8):
            t[p].append(timed(lambda: pat(o, p)))
    base = sorted(t[1])[len(t[1]) // 2]
    print("\n%s (%.0f GB/s by the product's pattern kernel)" % (label, sets[k][0]))
    for p in range(8):
        m = sorted(t[p])[len(t[p]) // 2]
        print("  %-14s %7.1f us  %6.0f GB/s  %+5.1f %% vs nt" % (POL[p], m, n * 208 / m / 1e3, (m / base - 1) * 100), flush=True)


# ---- the real step kernel under the same spellings (--kernel): builds of the PRODUCT sources with -DC2B_STORE_POL=<p>, whose every
# ---- non-temporal store is spelled p; the raw-pointer launch (512 threads x 2 tiles) through each library in turn, same tensors
def kernel_ab():
    import subprocess
    pols = [int(p) for p in a.kernel.split(",")]
    libs = {}
    for p in pols:
        path = os.path.join(entry.CSRC, "libcity2ba_hip_pol%d.so" % p)
        if not os.path.exists(path):
            subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-DC2B_STORE_POL=%d" % p] + entry.HIPCC_FLAGS + ["-o", path, os.path.join(entry.CSRC, "capi.hip")])
        lb = C.CDLL(path)
        res, args = L.SIGNATURES["c2b_residual_jacobian_rows"]
        lb.c2b_residual_jacobian_rows.restype, lb.c2b_residual_jacobian_rows.argtypes = res, args
        libs[p] = lb
    if a.build_only:
        return
    sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)
    ws = D.workspace(sh["n_obs"], dev)
    err = torch.zeros(1, dtype=torch.float64, device=dev)
    rows = sh["rows"]
    P = lambda t: C.c_void_p(t.data_ptr())                                                      # noqa: E731

    def step(lb, o):
        rc = lb.c2b_residual_jacobian_rows(P(sh["camblk"]), P(sh["pts4"]), sh["pts4"].shape[0], P(rows.row_ptr), rows.n_cam, P(rows.tiles), 0,
                                           P(sh["pt_idx"]), P(sh["uv"]), rows.n_obs, P(o.r), P(o.Jc), P(o.Jp), 2.0, P(ws), P(err), None)
        assert rc == 0, rc

    for label, k in picks:
        o = sets[k][1]
        D.residual_jacobian_rows(sh["camblk"], sh["pts4"], rows, sh["pt_idx"], sh["uv"], o.r, o.Jc, o.Jp, 2.0, ws, err)
        torch.cuda.synchronize()
        nn = rows.n_obs                                                                         # (the arrays are padded past n)
        ref = (o.r[:nn].clone(), o.Jc[:nn].clone(), o.Jp[:nn].clone())
        t = {p: [] for p in ["product"] + pols}
        same = {}
        for p in pols:
            o.Jc.fill_(float("nan"))
            step(libs[p], o)
            torch.cuda.synchronize()
            same[p] = bool(torch.equal(o.r[:nn], ref[0]) and torch.equal(o.Jc[:nn], ref[1]) and torch.equal(o.Jp[:nn], ref[2]))
        del ref
        for _ in range(a.rounds):
            t["product"].append(timed(lambda: D.residual_jacobian_rows(sh["camblk"], sh["pts4"], rows, sh["pt_idx"], sh["uv"], o.r, o.Jc, o.Jp, 2.0, ws, err), 10))
            for p in pols:
                t[p].append(timed(lambda: step(libs[p], o), 10))
        base = sorted(t["product"])[len(t["product"]) // 2]
        print("\nstep kernel (512 x 2, raw-pointer launch) in the %s (%.0f GB/s)" % (label, sets[k][0]))
        print("  %-26s %7.1f us" % ("product library (nt)", base))
        for p in pols:
            m = sorted(t[p])[len(t[p]) // 2]
            print("  %-26s %7.1f us  %+5.1f %%  bits %s" % ("stores spelled '%s'" % POL[p], m, (m / base - 1) * 100, same[p]), flush=True)


if a.kernel:
    kernel_ab()
