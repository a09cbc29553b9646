#!/usr/bin/env python3
"""A/B the light per-observation kernels (project / error / visibility) and the residual + Jacobian step in ONE
process, interleaved rounds, on the bench workload, every mode writing the SAME output arrays (where an output is
allocated moves these kernels by several us):
  * index forms (a 4-byte camera index per observation) over the tuning library's variants (--variants: observations
    per lane x waves per workgroup, cache policies, ablations);
  * the row-structure forms (c2b_*_rows) under every cache policy of their streams.
Outputs are checked bit-for-bit against the first index variant.
   python tools/tune_obs.py [--blocks 128] [--variants 308,20308,208]"""
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
ap.add_argument("--variants", default="308,20308")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="", help="time only the modes whose name contains this")
ap.add_argument("--cold", action="store_true", help="also time every mode with the caches swept by a 1-GiB read before each launch")
ap.add_argument("--place", action="store_true", help="place the Jacobian outputs by measured store rate, as bench.py does")
a = ap.parse_args()
variants = [int(v) for v in a.variants.split(",")]

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)
n = sh["n_obs"]
raw = C.CDLL(L.LIB_PATH)
raw.c2b_tune_set_observation_variant.argtypes = [C.c_int]
raw.c2b_tune_set_jacobian_variant.argtypes = [C.c_int]
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)
camblk, pts4, ci, pi, uv, rows = sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], sh["rows"]
uv_out = torch.empty_like(uv)
uv_noise = uv.clone()
err2 = torch.zeros(2, dtype=torch.float64, device=dev)
keep = torch.empty(n, dtype=torch.uint8, device=dev)
if a.place:
    (r_o, Jc_o, Jp_o), log = D.alloc_jacobian_outputs(n, dev)
    print("output placement:", log)
else:
    r_o = torch.empty((n, 2), dtype=torch.float64, device=dev)
    Jc_o = torch.empty((n, 18), dtype=torch.float64, device=dev)
    Jp_o = torch.empty((n, 6), dtype=torch.float64, device=dev)
print("n_obs=%d  tiles %d, with an empty list inside %d" % (n, rows.tiles.shape[0], int((rows.tiles[:, 2] < 0).sum())))


def with_obs(v, fn):
    def run():
        raw.c2b_tune_set_observation_variant(v)
        fn()
        raw.c2b_tune_set_observation_variant(308)
    return run


def with_jac(v, fn):
    def run():
        raw.c2b_tune_set_jacobian_variant(v)
        fn()
        raw.c2b_tune_set_jacobian_variant(0)
    return run


kinds = {
    "project": (lambda: D.project(camblk, pts4, ci, pi, uv_out), lambda: D.project_rows(camblk, pts4, rows, pi, uv_out)),
    "error_L2": (lambda: D.reprojection_error_sum(camblk, pts4, ci, pi, uv, 2.0, ws, err),
                 lambda: D.reprojection_error_sum_rows(camblk, pts4, rows, pi, uv, 2.0, ws, err)),
    "visibility": (lambda: D.visibility_pairs(camblk, pts4, ci, pi, 10.0, uv_out, keep),
                   lambda: D.visibility_rows(camblk, pts4, rows, pi, 10.0, uv_out, keep)),
    # r05: observation noise + L1 + L2 in one pass (rows form only; its sums depend on the grid, so no bit comparison)
    "noise_L1L2": (lambda: D.add_noise_observations_error_sums2_rows(camblk, pts4, rows, pi, uv_noise, 0, 1e-9, 7, ws, err2),
                   lambda: D.add_noise_observations_error_sums2_rows(camblk, pts4, rows, pi, uv_noise, 0, 1e-9, 7, ws, err2)),
}
jac = (lambda: D.residual_jacobian_sum(camblk, pts4, ci, pi, uv, r_o, Jc_o, Jp_o, 2.0, ws, err),
       lambda: D.residual_jacobian_rows(camblk, pts4, rows, pi, uv, r_o, Jc_o, Jp_o, 2.0, ws, err))
# cache policies of the row-structure forms (capi.hip: launch_obs / launch_jacobian)
POLICIES = (("shipped", 308), ("all cached", 20308), ("nt stores", 21308), ("nt stores+uv", 22308), ("nt everything", 23308),
            ("shipped policy, 1 tile per wave", 30108), ("shipped policy, 2 tiles", 30208), ("shipped policy, 4 tiles", 30408),
            ("shipped policy, 3 tiles, 4 waves per workgroup", 30304), ("shipped policy, 5 tiles", 30508), ("shipped policy, 6 tiles", 30608),
            ("shipped policy, 6 tiles, 4 waves per workgroup", 30604), ("shipped policy, 8 tiles", 30808),
            # r05 experiment (obs_split.hpp; projection only, the other kinds run the shipped kernel under these numbers):
            # one loader wave + seven compute waves per workgroup, G tiles per loader batch x K tiles per compute wave
            ("split: loader + 7 compute waves, G 6 K 6", 4066), ("split, G 4 K 6", 4046), ("split, G 6 K 12", 4612), ("split, G 8 K 8", 4088))
modes = {}                                        # name -> (kind, callable)
for k, (idx_fn, rows_fn) in kinds.items():
    for v in variants:
        modes["%s idx v%d" % (k, v)] = (k, with_obs(v, idx_fn))
    for tag, v in POLICIES:
        modes["%s rows [%s]" % (k, tag)] = (k, with_obs(v, rows_fn))
modes["jacobian idx [shipped]"] = ("jacobian", jac[0])
modes["jacobian rows [shipped: policy %d by size]" % D.jacobian_stream_policy(n, rows.n_cam, pts4.shape[0])] = ("jacobian", jac[1])
modes["jacobian rows [one tile per wave]"] = ("jacobian", with_jac(61, jac[1]))
modes["jacobian rows [two tiles per wave]"] = ("jacobian", with_jac(64, jac[1]))
modes["jacobian rows [every load cached]"] = ("jacobian", with_jac(51, jac[1]))
modes["jacobian rows [nt uv]"] = ("jacobian", with_jac(52, jac[1]))
modes["jacobian rows [nt uv + point index]"] = ("jacobian", with_jac(53, jac[1]))
if a.only:
    modes = {k: v for k, v in modes.items() if a.only in k}


def snapshot(kind):
    if kind == "project":
        return (uv_out.clone().view(torch.int64),)
    if kind == "visibility":
        return (uv_out.clone().view(torch.int64), keep.clone())
    if kind == "error_L2":
        return (err.clone(),)
    if kind == "noise_L1L2":
        return ()
    return (r_o.clone(), Jc_o[::97].clone(), Jp_o[::89].clone(), err.clone())


ref, bad = {}, 0
for name, (kind, fn) in modes.items():
    uv_out.fill_(float("nan")); keep.fill_(9); err.fill_(-1.0); r_o.fill_(float("nan"))
    fn()
    torch.cuda.synchronize()
    got = snapshot(kind)
    if kind not in ref:
        ref[kind] = got
    elif not all(torch.equal(x, y) for x, y in zip(got, ref[kind])):
        if "v9" in name:
            continue                                 # the ablations' outputs are wrong by construction
        bad += 1
        print("MISMATCH against the first %s mode: %s" % (kind, name))
print("%d modes checked bit-for-bit against the first of their kind, %d mismatches" % (len(modes), bad))

times = {m: [] for m in modes}
for _ in range(a.rounds):
    for m, (kind, fn) in modes.items():
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        times[m].append(s.elapsed_time(e) / a.reps * 1e3)
cold = {m: [] for m in modes}
if a.cold:
    sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
    for _ in range(max(a.rounds, 7)):
        for m, (kind, fn) in modes.items():
            sweep.sum()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            cold[m].append(s.elapsed_time(e) * 1e3)
for m in modes:
    t = sorted(times[m])
    c = sorted(cold[m])
    print("%-58s: median %7.1f us  min %7.1f us  %6.1f Gobs/s%s" % (m, t[len(t) // 2], t[0], n / t[len(t) // 2] / 1e3,
                                                                  ("   cold median %7.1f us" % c[len(c) // 2]) if c else ""))
