#!/usr/bin/env python3
"""Occlusion rays of the mesh generator (src/generate.rs:455-476) at city scale: the synthetic grid's cameras and
points against a triangle mesh of its buildings (one box per block, 12 triangles each).  Rays = the pairs that pass the
visibility predicate.  Reports rays, triangles, kernel time and ray-triangle tests per second for the brute-force
kernel, and how the 3-D mesh test compares with the generator's own 2-D hits_building filter.
python tools/bench_occlusion.py [--blocks 16] [--reps 3]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from city2ba_amd import device as D  # noqa: E402
from city2ba_amd import synthetic as S  # noqa: E402


def city_mesh(B, L, inset, height, shrink=0.01):
    """[12 * B * B, 9] f32: one axis-aligned box per block, footprint = the square hits_building tests
    (src/synthetic.rs:52-124) pulled in by `shrink` so facade points are not their own occluders"""
    bx, bz = np.meshgrid(np.arange(B), np.arange(B), indexing="ij")
    x0 = (L * bx + inset + shrink).ravel()
    x1 = (L * (bx + 1) - inset - shrink).ravel()
    z0 = (L * bz + inset + shrink).ravel()
    z1 = (L * (bz + 1) - inset - shrink).ravel()
    y0, y1 = np.full_like(x0, -1.0), np.full_like(x0, height)
    c = [np.stack(v, axis=1) for v in ((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1),
                                       (x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1))]
    quads = [(0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7), (4, 5, 6, 7), (0, 3, 2, 1)]
    tris = []
    for a, b, cc, d in quads:
        tris.append(np.concatenate([c[a], c[b], c[cc]], axis=1))
        tris.append(np.concatenate([c[a], c[cc], c[d]], axis=1))
    return np.ascontiguousarray(np.stack(tris, axis=1).reshape(-1, 9), dtype=np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--max-dist", type=float, default=10.0)
    ap.add_argument("--bvh", type=int, default=1)
    ap.add_argument("--brute-max", type=float, default=4e11, help="skip the all-triangles kernel above this many tests")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, L, inset = a.blocks, 20.0, 1.0
    pos, dirs, pts = S.grid_layout(B, 10, 10, L, inset, 1.0, 1.0)
    cam15 = D.cameras_from_position_direction(torch.from_numpy(pos).to(dev), torch.from_numpy(dirs).to(dev))
    cen4 = D.centers_table(cam15.shape[0], dev)
    camblk = D.cameras_prepare_state(cam15, centers=cen4)
    pts4 = D.points_pad(torch.from_numpy(pts).to(dev))
    centers = cen4[:, :3].cpu().numpy().copy()
    out = {"blocks": B, "cameras": len(pos), "points": len(pts)}
    for name, occl in (("all", False), ("hits_building", True)):
        ci, pi = S.candidate_pairs(centers, pts, a.max_dist, occlusion=occl, block_length=L, block_inset=inset)
        ci_d, pi_d = torch.from_numpy(ci.astype(np.int32)).to(dev), torch.from_numpy(pi.astype(np.int32)).to(dev)
        uv = torch.empty((len(ci), 2), dtype=torch.float64, device=dev)
        keep = torch.empty(len(ci), dtype=torch.uint8, device=dev)
        D.visibility_pairs(camblk, pts4, ci_d, pi_d, a.max_dist, uv, keep)
        k = keep.bool()
        if not occl:
            rays_c, rays_p = ci_d[k].contiguous(), pi_d[k].contiguous()
        else:
            n_2d = int(k.sum().item())
            kept_2d = set(zip(ci_d[k].cpu().numpy().tolist(), pi_d[k].cpu().numpy().tolist())) if n_2d <= 8_000_000 else None
        out["candidates_" + name] = int(len(ci))
        out["visible_" + name] = int(k.sum().item())
    tri = city_mesh(B, L, inset, 10.0)
    tri_d = torch.from_numpy(tri).to(dev)
    n_rays = rays_c.shape[0]
    keep3 = torch.empty(n_rays, dtype=torch.uint8, device=dev)
    out.update(rays=int(n_rays), triangles=int(len(tri)))

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    brute = None
    if n_rays * len(tri) <= a.brute_max:
        ms = timed(lambda: D.occlusion_filter(camblk, pts4, rays_c, rays_p, tri_d, keep3))
        brute = keep3.clone()
        out["brute_ms"] = round(ms, 3)
        out["brute_Gtests_per_s"] = round(n_rays * len(tri) / ms / 1e6, 1)
    if a.bvh:
        import time
        t0 = time.perf_counter()
        bvh = D.OcclusionBVH(tri, dev)
        out["bvh_build_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        out["bvh_nodes"] = bvh.n_nodes
        ms = timed(lambda: bvh.filter(camblk, pts4, rays_c, rays_p, keep3))
        out["bvh_ms"] = round(ms, 3)
        out["bvh_Mrays_per_s"] = round(n_rays / ms / 1e3, 1)
        if brute is not None:
            out["bvh_mismatch_vs_brute"] = int((keep3 != brute).sum().item())
        else:
            brute = keep3.clone()
    out["kept_after_mesh"] = int(brute.sum().item())
    # CPU side: the oracle's all-triangles loop (1 thread) on a bounded sample of the same rays, checked against the device
    import time
    import oracle as O
    n_s = int(min(n_rays, max(1000, 2e8 // max(len(tri), 1))))
    cams_h = cam15.cpu().numpy()
    ci_s, pi_s = rays_c[:n_s].cpu().numpy().astype(np.uint32), rays_p[:n_s].cpu().numpy().astype(np.uint32)
    t0 = time.perf_counter()
    keep_cpu = O.occlusion_filter(cams_h, pts, ci_s, pi_s, tri)
    dt = time.perf_counter() - t0
    out["cpu_all_triangles"] = {"rays": n_s, "seconds": round(dt, 3), "Mrays_per_s": round(n_s / dt / 1e6, 4),
                                "Gtests_per_s_upper": round(n_s * len(tri) / dt / 1e9, 3), "threads": 1,
                                "mismatch_vs_device": int((keep_cpu != brute[:n_s].cpu().numpy()).sum())}
    # 3-D mesh occlusion against the generator's 2-D segment test on the same pairs
    out["kept_2d_hits_building"] = n_2d
    if kept_2d is not None and n_rays <= 8_000_000:
        kept_3d = set(zip(rays_c[brute.bool()].cpu().numpy().tolist(), rays_p[brute.bool()].cpu().numpy().tolist()))
        out["kept_both"] = len(kept_2d & kept_3d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
