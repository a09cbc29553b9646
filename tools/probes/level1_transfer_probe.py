#!/usr/bin/env python3
"""What Level 1's host-buffer boundary costs on the headline grid: c2b_problem_upload_bal (validation, narrowing, H2D) and
c2b_problem_download + _download_graph + _download_bal (D2H, widening) for 19.3 M observations, from / into reused numpy
arrays (no page faults) and fresh ones."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import city2ba_amd as c2b                                     # noqa: E402
from city2ba_amd import _lib as L                             # noqa: E402
from city2ba_amd import synthetic as S                        # noqa: E402

g = S.synthetic_grid(10, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 20.0, 1.0, 1.0, 1.0, 10.0, False)
bal9, pts, row_ptr, pt_idx, uv = g.cameras_bal(), g.points(), g.row_ptr.copy(), g.pt_idx.copy(), g.observations()
n_cam, n_pts, n_obs = len(bal9), len(pts), len(pt_idx)
g.close()
p = lambda a: a.ctypes.data_as(C.c_void_p)
ba = c2b.BAProblem(0)
for rep in range(3):
    t = time.perf_counter()
    L.check(L.lib().c2b_problem_upload_bal(ba._h, n_cam, p(bal9), n_pts, p(pts), p(row_ptr), p(pt_idx), p(uv)))
    up = time.perf_counter() - t
    cams15, pts_o, uv_o = np.empty((n_cam, 15)), np.empty((n_pts, 3)), np.empty((n_obs, 2))
    rp_o, pi_o, b9_o = np.empty(n_cam + 1, np.uint64), np.empty(n_obs, np.uint64), np.empty((n_cam, 9))
    t = time.perf_counter()
    L.check(L.lib().c2b_problem_download(ba._h, p(cams15), p(pts_o), p(uv_o)))
    L.check(L.lib().c2b_problem_download_graph(ba._h, p(rp_o), p(pi_o)))
    L.check(L.lib().c2b_problem_download_bal(ba._h, p(b9_o)))
    down_fresh = time.perf_counter() - t
    t = time.perf_counter()
    L.check(L.lib().c2b_problem_download(ba._h, p(cams15), p(pts_o), p(uv_o)))
    L.check(L.lib().c2b_problem_download_graph(ba._h, p(rp_o), p(pi_o)))
    L.check(L.lib().c2b_problem_download_bal(ba._h, p(b9_o)))
    down_reused = time.perf_counter() - t
    assert np.array_equal(uv_o, uv) and np.array_equal(pi_o, pt_idx) and np.array_equal(b9_o, bal9)
    mb = (bal9.nbytes + pts.nbytes + row_ptr.nbytes + pt_idx.nbytes + uv.nbytes) / 1e6
    print("%d observations, %.0f MB of host arrays: upload %.1f ms | download into fresh arrays %.1f ms, into reused arrays %.1f ms"
          % (n_obs, mb, up * 1e3, down_fresh * 1e3, down_reused * 1e3))
