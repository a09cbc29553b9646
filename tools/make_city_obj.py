#!/usr/bin/env python3
"""Writes a city-block mesh as a Wavefront .obj for `city2ba generate`: a ground plane, one box per block (random
heights, optional roof detail to raise the triangle count) and a polyline `street` that snakes through the street grid.
python tools/make_city_obj.py OUT.obj [--blocks 32 --block-length 20 --inset 3 --detail 0 --seed 1]"""
import argparse

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--block-length", type=float, default=20.0)
    ap.add_argument("--inset", type=float, default=3.0)
    ap.add_argument("--detail", type=int, default=0, help="extra roof boxes per building")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    B, L, ins = a.blocks, a.block_length, a.inset
    lines = ["# city blocks for city2ba generate"]
    nv = 0

    def box(x0, x1, y0, y1, z0, z1):
        nonlocal nv
        c = [(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), (x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1)]
        for v in c:
            lines.append("v %.6f %.6f %.6f" % v)
        for q in ((0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7), (4, 5, 6, 7)):
            lines.append("f %d %d %d %d" % tuple(nv + 1 + k for k in q))
        nv += 8

    lines.append("o Ground")
    for v in ((0, 0, 0), (B * L, 0, 0), (B * L, 0, B * L), (0, 0, B * L)):
        lines.append("v %.6f %.6f %.6f" % v)
    lines.append("f 1 4 3 2")
    nv = 4
    lines.append("o Buildings")
    for bx in range(B):
        for bz in range(B):
            h = float(rng.uniform(6, 30))
            x0, x1, z0, z1 = L * bx + ins, L * (bx + 1) - ins, L * bz + ins, L * (bz + 1) - ins
            box(x0, x1, 0.0, h, z0, z1)
            for _ in range(a.detail):
                w = rng.uniform(0.5, 2.0)
                px, pz = rng.uniform(x0, x1 - w), rng.uniform(z0, z1 - w)
                box(px, px + w, h, h + float(rng.uniform(0.3, 2.0)), pz, pz + w)
    # street polyline at eye height: along z = L * k for k = 0..B, alternating direction, joined at the ends
    lines.append("o street")
    first = nv + 1
    n = 0
    for k in range(B + 1):
        xs = (0.0, B * L) if k % 2 == 0 else (B * L, 0.0)
        for x in xs:
            lines.append("v %.6f 1.700000 %.6f" % (x, L * k))
            n += 1
    for i in range(n - 1):
        lines.append("l %d %d" % (first + i, first + i + 1))
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote %s: %d blocks, %d vertices" % (a.out, B * B, nv + n))


if __name__ == "__main__":
    main()
