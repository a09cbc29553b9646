#!/usr/bin/env python3
"""Device-memory leak check of the Level-1 object: many create / upload / sweep / occlude / adopt / cull / noise /
destroy cycles must leave the free-memory reading where it started.
python tools/check_leaks.py [--cycles 60] [--what all|grid|noise|index|jac|generate|dense|files|each]"""
import argparse
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import city2ba_amd as c2b  # noqa: E402,F401
from city2ba_amd import generate as G  # noqa: E402
from city2ba_amd import noise as N  # noqa: E402
from city2ba_amd import synthetic as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cycles", type=int, default=60)
ap.add_argument("--what", default="all")
a = ap.parse_args()
torch.cuda.init()
scene = os.path.join(ROOT, "tests", "golden", "test_scene.obj")


def cycle(k, what):
    if what in ("all", "grid", "noise", "index", "jac"):
        ba = S.synthetic_grid(10, 20, 3, 5.0, 1.0, 1.0, 1.0, 10.0, False)
        if what in ("all", "noise"):
            ba = N.add_drift_normalized(ba, 1e-3, 1e-3, 0.1, seed=k)
            ba = N.add_noise(ba, 1e-3, 1e-3, 1e-3, 1e-3, seed=k)
        if what in ("all", "index"):
            ba = N.drop_features(ba, 0.9, seed=k).cull()
        if what in ("all", "jac"):
            ba.total_reprojection_error(2.0)
            ba.residual_jacobian()
        ba.close()
    if what in ("all", "files"):
        # both file forms written and read back on the device (r04): .bbal words, .bal decimal text, and a text file whose
        # observations are not camera-major (the device's stable sort)
        import tempfile
        import numpy as np
        os.environ["C2B_TEXT_DEVICE_MIN_BYTES"] = "0"
        ba = S.synthetic_grid(10, 20, 3, 5.0, 1.0, 1.0, 1.0, 10.0, False)
        with tempfile.TemporaryDirectory() as d:
            for ext in ("bbal", "bal"):
                ba.write(os.path.join(d, "p." + ext))
                c2b.BAProblem.from_file(os.path.join(d, "p." + ext)).close()
            lines = open(os.path.join(d, "p.bal")).read().split("\n")
            n = ba.num_observations()
            obs = [lines[1 + i] for i in np.random.default_rng(k).permutation(n)]
            open(os.path.join(d, "m.bal"), "w").write("\n".join(lines[:1] + obs + lines[1 + n:]))
            c2b.BAProblem.from_file(os.path.join(d, "m.bal")).close()
        ba.close()
    if what in ("all", "generate", "dense"):
        g = G.generate(scene, num_cameras=60, num_world_points=400, path_name="path", seed=k)
        if what in ("all", "dense"):
            g.visibility_graph(50.0)
        g.close()
    gc.collect()


def free_mib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


worst = 0.0
for what in (["grid", "noise", "index", "jac", "generate", "dense", "files"] if a.what == "each" else [a.what]):
    cycle(0, what)
    f0 = free_mib()
    for k in range(a.cycles):
        cycle(k + 1, what)
    f1 = free_mib()
    print("%-9s free before %.1f MiB, after %d cycles %.1f MiB, delta %.2f MiB" % (what, f0, a.cycles, f1, f0 - f1))
    worst = max(worst, f0 - f1)
sys.exit(0 if worst < 64 else 1)
