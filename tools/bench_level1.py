#!/usr/bin/env python3
"""PCIe-inclusive rate of the Level-1 (host buffers in / out) boundary on `synthetic --blocks 32`:
upload once, then residual+Jacobian with results copied back to pageable host memory.  Never the headline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from city2ba_amd import synthetic as S
t0 = time.perf_counter()
ba = S.synthetic_grid(10, 10, 32, 20.0, 1.0, 1.0, 1.0, 10.0, False)
t_gen = time.perf_counter() - t0
n = ba.num_observations()
ba.residual_jacobian()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); r, Jc, Jp = ba.residual_jacobian(); ts.append(time.perf_counter() - t0)
te = []
for _ in range(5):
    t0 = time.perf_counter(); e = ba.total_reprojection_error(2.0); te.append(time.perf_counter() - t0)
print({"n_obs": n, "generate_s": round(t_gen, 2), "level1_residual_jacobian_ms": round(min(ts) * 1e3, 1),
       "Mobs/s_pcie_inclusive": round(n / min(ts) / 1e6, 1), "bytes_back": int(r.nbytes + Jc.nbytes + Jp.nbytes),
       "GB/s_back": round((r.nbytes + Jc.nbytes + Jp.nbytes) / min(ts) / 1e9, 1),
       "level1_total_reprojection_error_ms": round(min(te) * 1e3, 2)})
