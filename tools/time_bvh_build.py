#!/usr/bin/env python3
"""host-side hierarchy build time on the large city mesh (tools/make_city_obj.py --blocks 96 --detail 8)"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from city2ba_amd import _lib as L  # noqa: E402
from city2ba_amd import generate as G  # noqa: E402

with tempfile.TemporaryDirectory() as t:
    path = os.path.join(t, "city.obj")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_city_obj.py"), path, "--blocks", "96", "--detail", "8"])
    t0 = time.perf_counter()
    o = G.ObjFile(path)
    t1 = time.perf_counter()
    tri = o.triangles(o.index("street"))
    t2 = time.perf_counter()
    print("load %.1f ms, triangles %.1f ms, n_tri %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(tri)))
    for rep in range(3):
        h = C.c_void_p()
        t0 = time.perf_counter()
        L.check(L.lib().c2b_bvh_build(tri.ctypes.data_as(C.c_void_p), len(tri), C.byref(h)))
        print("bvh build %.1f ms" % ((time.perf_counter() - t0) * 1e3))
        L.lib().c2b_bvh_free(h)
print("cores", os.cpu_count())
