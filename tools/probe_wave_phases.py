#!/usr/bin/env python3
"""r05 (tuning library): where does a wave of the light passes spend its life?  Wave 0 of every workgroup of
k_observations stamps the 100 MHz wall clock at: 0 entry, 1 tile records decoded (camera ids known), 2 point indices + camera
chunks arrived, 3 points arrived and cameras in LDS (arithmetic starts), 4 / 5 / 6 after tile 0 / 1 / 2 (stores issued),
7 every store acknowledged.  --blocks 128, project / L1+L2 / noise+L1+L2, back to back (3rd launch) and from swept caches.
    python tools/probe_wave_phases.py"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402
from city2ba_amd import _lib as L  # noqa: E402

L.LIB_PATH = entry.build_tune()
import bench  # noqa: E402
from city2ba_amd import device as D  # noqa: E402

dev = torch.device("cuda", 0)
raw = C.CDLL(L.LIB_PATH)
raw.c2b_tune_set_probe.argtypes = [C.c_void_p]
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
n = sh["n_obs"]
ws = D.workspace(n, dev)
err = torch.zeros(2, dtype=torch.float64, device=dev)
uv_out = torch.empty_like(sh["uv"])
uv2 = sh["uv"].clone()
sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
n_wg = (n + 1535) // 1536
probe = torch.zeros((n_wg, 8), dtype=torch.int64, device=dev)
a = (sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"])
cases = {"project_rows": lambda: D.project_rows(*a, uv_out),
         "error_sums2_rows": lambda: D.reprojection_error_sums2_rows(*a, sh["uv"], ws, err),
         "noise+error_sums2_rows": lambda: D.add_noise_observations_error_sums2_rows(*a, uv2, 0, 1e-9, 7, ws, err)}
NAMES = ["entry -> records decoded", "-> indices + camera chunks here", "-> points here, cameras in LDS", "-> tile 0 out", "-> tile 1 out",
         "-> tile 2 out", "-> stores acknowledged"]
for name, fn in cases.items():
    for mode in ("back to back", "caches swept"):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        if mode == "caches swept":
            sweep.sum()
        raw.c2b_tune_set_probe(C.c_void_p(probe.data_ptr()))
        probe.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        raw.c2b_tune_set_probe(None)
        t = probe.cpu().numpy().astype(np.float64) * 10e-3            # microseconds
        t = t[(t[:, 0] > 0) & (t[:, 7] > 0)]
        t0 = t[:, 0].min()
        d = np.diff(t, axis=1)
        life = t[:, 7] - t[:, 0]
        print("\n%s, %s: launch %.1f us (with the probe), %d workgroups; wave 0's life: median %.2f us (10 %% %.2f, 90 %% %.2f); starts from %.1f to %.1f us" % (
            name, mode, s.elapsed_time(e) * 1e3, len(t), np.median(life), np.percentile(life, 10), np.percentile(life, 90), 0.0, t[:, 0].max() - t0))
        for k in range(7):
            print("    %-34s median %5.2f us   10 %% %5.2f   90 %% %5.2f" % (NAMES[k], np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
        # workgroups in flight over time: a slot is busy from a workgroup's entry to its last acknowledgement
        dur = t[:, 7].max() - t0
        print("    sum of wave-0 lives / (launch span %.1f us x 1 024 slots) = %.2f" % (dur, life.sum() / (dur * 1024)), flush=True)
