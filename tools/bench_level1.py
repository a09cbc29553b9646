#!/usr/bin/env python3
"""PCIe-inclusive rate of the Level-1 (host buffers in / out) boundary on `synthetic --blocks 32`: upload once, then
residual + Jacobian with the 255 MB of results copied back, into pageable numpy arrays and into page-locked ones
(pinned_empty / c2b_host_alloc) that are allocated once and reused.  Never the headline."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from city2ba_amd import synthetic as S  # noqa: E402

t0 = time.perf_counter()
ba = S.synthetic_grid(10, 10, 32, 20.0, 1.0, 1.0, 1.0, 10.0, False)
t_gen = time.perf_counter() - t0
n = ba.num_observations()


def best(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


out_pageable = ba.residual_jacobian()
out_pinned = ba.residual_jacobian(pinned=True)
assert all((a == b).all() for a, b in zip(out_pageable, out_pinned))
nbytes = sum(a.nbytes for a in out_pinned)
t_new = best(lambda: ba.residual_jacobian())                       # fresh pageable arrays every call (page faults included)
t_page = best(lambda: ba.residual_jacobian(out=out_pageable))
t_pin = best(lambda: ba.residual_jacobian(out=out_pinned))
t_err = best(lambda: ba.total_reprojection_error(2.0))
print(json.dumps({
    "n_obs": n, "generate_s": round(t_gen, 2), "bytes_back": nbytes,
    "fresh_pageable_ms": round(t_new * 1e3, 2), "fresh_pageable_Mobs/s": round(n / t_new / 1e6, 1),
    "reused_pageable_ms": round(t_page * 1e3, 2), "reused_pageable_Mobs/s": round(n / t_page / 1e6, 1),
    "reused_pageable_GB/s": round(nbytes / t_page / 1e9, 1),
    "pinned_ms": round(t_pin * 1e3, 2), "pinned_Mobs/s": round(n / t_pin / 1e6, 1), "pinned_GB/s": round(nbytes / t_pin / 1e9, 1),
    "round1_pageable_Mobs/s": 84.0,
    "level1_total_reprojection_error_ms": round(t_err * 1e3, 3)}))
