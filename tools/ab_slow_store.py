#!/usr/bin/env python3
"""the residual+Jacobian step kernel A/B'd on the device class the driver drew.

One process on whatever MI355X this call landed on:
  1. eight output-sized allocation sets, the kernel's own store pattern timed into each -> the device's store class
     ("slow": every set < 6.0 TB/s; "fast": every set >= 6.8; "mixed");
  2. in the SLOWEST and in the FASTEST set: the rows-form kernel instances of the tuning library (tiles per wave 1/2/3,
     256- vs 512-thread workgroups, every cache policy of the once-read streams, observed uv up front or per tile, plain
     vs non-temporal stores, chunked XCD maps), interleaved rounds, outputs compared bit for bit with the shipped
     instance, each as microseconds and as a multiple of that set's own algorithmic floor (the launch's algorithmic bytes
     at the store rate the set sustains);
  3. one JSON line + a table.

    python tools/ab_slow_store.py [--blocks 128] [--rounds 4] [--reps 10] [--require slow|mixed|fast|any]
exit status 7 when --require names a class and the device is another one (nothing but the 4-second look is spent).
"""
import argparse
import ctypes as C
import json
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

VARIANTS = {
    700: "shipped: 512 thr, 2 tiles/wave, uv+idx nt loads, nt stores",
    701: "1 tile/wave", 702: "3 tiles/wave", 703: "256 thr, 2 tiles/wave", 704: "256 thr, 4 tiles/wave",
    713: "256 thr, 1 tile/wave", 705: "uv requested per tile", 706: "plain stores", 712: "1 tile/wave, plain stores",
    707: "every load cached", 708: "idx nt only", 709: "uv nt only", 714: "XCD map in chunks of 4", 715: "XCD map in chunks of 64",
    716: "256 thr, 1 tile, every load cached", 717: "256 thr, 1 tile, uv nt only", 718: "256 thr, 1 tile, uv per tile",
    719: "128 thr, 2 tiles/wave", 720: "1024 thr, 1 tile/wave", 721: "256 thr, 1 tile, XCD chunks of 64", 722: "256 thr, 1 tile, plain stores",
    730: "1024 thr, 2 tiles/wave",
}

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=128)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--require", default="any")
ap.add_argument("--slowest-only", action="store_true", help="A/B in the slowest set only")
ap.add_argument("--variants", default=",".join(str(v) for v in VARIANTS))
a = ap.parse_args()
variants = [int(v) for v in a.variants.split(",")]

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
raw = C.CDLL(L.LIB_PATH)
raw.c2b_tune_set_jacobian_variant.argtypes = [C.c_int]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3          # us


sh = bench.build_shard(argparse.Namespace(blocks=a.blocks), 0, 1, dev)        # 0.1 s on the device route
n_probe = sh["n_obs"]
sets = []
for k in range(a.sets):
    o = D.JacobianOutputs(n_probe, dev, max_attempts=1)
    for _ in range(2):
        D.calib_store_pattern(o.r, o.Jc, o.Jp)
    t = timed(lambda: D.calib_store_pattern(o.r, o.Jc, o.Jp), 6)
    sets.append((n_probe * 208 / t / 1e3, o))       # GB/s
rates = [round(r, 1) for r, _ in sets]
cls = bench.store_class(rates)
print("store GB/s of %d output sets: %s -> device class %s" % (a.sets, rates, cls), flush=True)
if a.require != "any" and cls != a.require:
    print(json.dumps({"device_store_class": cls, "store_GBs_per_set": rates, "skipped": "wanted a %s device" % a.require}))
    sys.exit(7)

n = sh["n_obs"]
alg = bench.algorithmic_bytes(n, sh["n_cam_local"], sh["n_pts"])
ws = D.workspace(n, dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)


def run(v, r, Jc, Jp):
    raw.c2b_tune_set_jacobian_variant(v)
    D.residual_jacobian_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws, err)


order = sorted(range(len(sets)), key=lambda k: sets[k][0])
picks = [("slowest set", order[0])] + ([("fastest set", order[-1])] if order[-1] != order[0] else [])
mid = [k for k in order[1:-1] if 6100.0 <= sets[k][0] <= 6700.0]
if mid:
    picks.insert(1, ("a set in between", mid[len(mid) // 2]))
if a.slowest_only:
    picks = picks[:1]
keep = {k for _, k in picks}
for k in range(len(sets)):                         # the others go back to the allocator
    if k not in keep:
        sets[k] = (sets[k][0], None)
torch.cuda.empty_cache()

# bits: every variant against the shipped instance, in the first picked set
_, o = sets[picks[0][1]]
run(variants[0], o.r, o.Jc, o.Jp)
torch.cuda.synchronize()
ref = (o.r.clone(), o.Jc.clone(), o.Jp.clone())
e_ref = err.item()
bits = {}
for v in variants[1:]:
    o.r.fill_(float("nan")); o.Jc.fill_(float("nan")); o.Jp.fill_(float("nan"))
    run(v, o.r, o.Jc, o.Jp)
    torch.cuda.synchronize()
    bits[v] = bool(torch.equal(o.r, ref[0]) and torch.equal(o.Jc, ref[1]) and torch.equal(o.Jp, ref[2]))
    # the folded sum depends on the grid (tiles per workgroup): equal bits for equal shapes, rounding otherwise
    bits[str(v) + "_err_rel"] = abs(err.item() - e_ref) / e_ref
del ref

result = {"device_store_class": cls, "store_GBs_per_set": rates, "n_obs": n, "algorithmic_bytes": alg, "tables": {}}
for label, k in picks:
    rate, o = sets[k]
    store_us = timed(lambda: D.calib_store_pattern(o.r, o.Jc, o.Jp), 10)
    rate = n * 208 / store_us / 1e3
    floor_us = alg / rate / 1e3
    times = {v: [] for v in variants}
    for _ in range(a.rounds):
        for v in variants:
            times[v].append(timed(lambda: run(v, o.r, o.Jc, o.Jp), a.reps))
    tab = {}
    print("\n%s: store pattern %.1f us = %.0f GB/s; algorithmic floor at that rate %.1f us" % (label, store_us, rate, floor_us))
    base = sorted(times[variants[0]])[len(times[variants[0]]) // 2]
    for v in variants:
        t = sorted(times[v])
        med = t[len(t) // 2]
        tab[v] = {"median_us": round(med, 1), "min_us": round(t[0], 1), "over_floor": round(med / floor_us, 4),
                  "frac_of_8TBs": round(alg / med / 1e3 / 8000.0, 4), "vs_shipped": round(med / base, 4)}
        print("  %4d  %-52s median %7.1f us  min %7.1f  x%.3f of the floor  frac %.3f  %+5.1f %% vs shipped  bits %s" % (
            v, VARIANTS.get(v, ""), med, t[0], med / floor_us, alg / med / 1e3 / 8000.0, (med / base - 1) * 100,
            "ref" if v == variants[0] else bits.get(v)))
    result["tables"][label] = {"store_GBs": round(rate, 1), "store_floor_us": round(store_us, 1),
                               "algorithmic_floor_us": round(floor_us, 1), "variants": tab}
result["bits_equal_to_shipped"] = {str(k): v for k, v in bits.items()}
print(json.dumps(result))
