#!/usr/bin/env python3
"""Config 5 stress case: the camera x point visibility sweep of the mesh generator at 10k cameras x 1M points
(1e10 pair evaluations, src/generate.rs:446-469 without Embree) on one MI355X, plus the noise kernels on the
same entity counts.  python tools/bench_dense.py [--cams 10000 --points 1000000 --max-dist 10]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from city2ba_amd import _lib as L  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cams", type=int, default=10000)
ap.add_argument("--points", type=int, default=1000000)
ap.add_argument("--max-dist", type=float, default=10.0)
ap.add_argument("--extent", type=float, default=300.0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--cpu-pairs", type=int, default=20_000_000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
rng = np.random.default_rng(20245)
# cameras: seeded, w ~ U(-pi,pi)^3 scaled, centres in a square; points on the ground +- a few metres
w = rng.uniform(-np.pi, np.pi, (a.cams, 3)) * rng.uniform(0, 1, (a.cams, 1))
centre = np.column_stack([rng.uniform(0, a.extent, a.cams), rng.uniform(1, 3, a.cams), rng.uniform(0, a.extent, a.cams)])
bal9 = np.column_stack([w, np.zeros((a.cams, 3)), rng.uniform(0.8, 1.2, a.cams), rng.uniform(-1e-2, 1e-2, (a.cams, 2))])
bal_d = torch.from_numpy(np.ascontiguousarray(bal9)).to(dev)
cam15 = D.cameras_from_bal(bal_d)
# t = -R c  (set on the host from the device's R)
R = cam15[:, :9].cpu().numpy().reshape(-1, 3, 3).transpose(0, 2, 1)
t = -np.einsum("nij,nj->ni", R, centre)
cam15[:, 9:12] = torch.from_numpy(t).to(dev)
camblk = D.cameras_prepare_state(cam15)
pts = np.column_stack([rng.uniform(0, a.extent, a.points), rng.uniform(0, 6, a.points), rng.uniform(0, a.extent, a.points)])
pts4 = D.points_pad(torch.from_numpy(np.ascontiguousarray(pts)).to(dev))
lib = L.lib()
n_tiles = lib.c2b_visibility_dense_tiles(a.points)
counts = torch.empty(a.cams * n_tiles, dtype=torch.int32, device=dev)
tot = torch.empty(a.cams + 1, dtype=torch.int64, device=dev)
row = torch.empty(a.cams + 1, dtype=torch.int64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731


def count():
    L.check(lib.c2b_visibility_dense_count(p(camblk), a.cams, p(pts4), a.points, a.max_dist, p(counts), p(tot), p(row), st))


count()
torch.cuda.synchronize()
n_obs = int(row[-1].item())
pt_idx = torch.empty(max(n_obs, 1), dtype=torch.int32, device=dev)
uv = torch.empty((max(n_obs, 1), 2), dtype=torch.float64, device=dev)


def fill():
    L.check(lib.c2b_visibility_dense_fill(p(camblk), a.cams, p(pts4), a.points, a.max_dist, p(counts), p(row), p(pt_idx), p(uv), st))


def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)


t_count = timed(count)
t_fill = timed(fill)
pairs = a.cams * a.points
out = {"cams": a.cams, "points": a.points, "pairs": pairs, "max_dist": a.max_dist, "observations_kept": n_obs,
       "count_ms": round(t_count * 1e3, 3), "fill_ms": round(t_fill * 1e3, 3),
       "sweep_Gpairs_per_s": round(pairs / (t_count + t_fill) / 1e9, 2),
       "reference_published_lower_bound_pairs_per_s": 2.8e7}
# noise kernels at the same entity counts (config 5)
ws = D.workspace(0, dev)
stt = D.stats(camblk, pts4, ws)
c2, p2 = cam15.clone(), pts4.clone()
for name, fn in (("stats", lambda: D.stats(camblk, pts4, ws, stt)),
                 ("add_drift_normalized", lambda: D.add_drift_normalized(c2, p2, stt, 1e-9, 1e-9, 0.1, 3)),
                 ("add_noise_entities", lambda: D.add_noise_entities(c2, p2, stt, 1e-9, 1e-9, 1e-9, 4))):
    out[name + "_us"] = round(timed(fn) * 1e6, 1)
# the f32 path of the same entity kernels (configs[4]): cameras / points stored as float, draws and statistics in f64
c32, p32 = D.to_f32(cam15), D.to_f32(pts4)
st32 = D.stats_f32(c32, p32, ws)
for name, fn in (("stats_f32", lambda: D.stats_f32(c32, p32, ws, st32)),
                 ("add_drift_normalized_f32", lambda: D.add_drift_normalized_f32(c32, p32, st32, 1e-9, 1e-9, 0.1, 3)),
                 ("add_noise_entities_f32", lambda: D.add_noise_entities_f32(c32, p32, st32, 1e-9, 1e-9, 1e-9, 4))):
    out[name + "_us"] = round(timed(fn) * 1e6, 1)
# CPU oracle on a sample of the same pairs (1 thread)
if a.cpu_pairs > 0:
    sys.path.insert(0, ROOT)
    import oracle as O
    nc = max(1, a.cpu_pairs // a.points)
    ci = np.repeat(np.arange(nc, dtype=np.uint32), a.points)
    pi = np.tile(np.arange(a.points, dtype=np.uint32), nc)
    cams_h = cam15[:nc].cpu().numpy()
    t0 = time.perf_counter()
    uv_c, keep_c = O.visibility_pairs(cams_h, pts, ci, pi, a.max_dist)
    dt = time.perf_counter() - t0
    out["cpu_oracle_Mpairs_per_s_1thread"] = round(len(ci) / dt / 1e6, 2)
    # parity on the sample
    r = row[: nc + 1].cpu().numpy()
    got = pt_idx[: r[-1]].cpu().numpy()
    want = pi[keep_c == 1]
    out["sample_indices_equal"] = bool(np.array_equal(got.astype(np.uint32), want))
print(json.dumps(out))
