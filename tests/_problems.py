"""Seeded synthetic BA problems shared by the CPU and GPU tests (inputs only)."""
import numpy as np

import oracle as O


def random_problem(n_cam, n_pts, obs_per_cam, seed, noise=0.0, k_scale=1e-2, empty_every=0):
    """Cameras with w in (-pi,pi)^3-ish, every observed point in front of its camera.

    Returns dict(bal9, cams15, pts, row_ptr(u64), pt_idx(u64), uv) with uv = exact oracle
    projection (+ optional gaussian noise)."""
    rng = np.random.default_rng(seed)
    w = rng.uniform(-np.pi, np.pi, size=(n_cam, 3)) * rng.uniform(0.0, 1.0, size=(n_cam, 1))
    t = rng.uniform(-50, 50, size=(n_cam, 3))
    intr = np.column_stack([rng.uniform(0.8, 1.2, n_cam), rng.uniform(-k_scale, k_scale, n_cam),
                            rng.uniform(-k_scale, k_scale, n_cam)])
    bal9 = np.ascontiguousarray(np.column_stack([w, t, intr]))
    cams15 = O.camera_from_bal(bal9)
    pts = np.empty((n_pts, 3))
    row_ptr = [0]
    pt_idx = []
    owner = rng.integers(0, n_cam, size=n_pts)          # each point placed in front of one camera
    for j in range(n_pts):
        z = -rng.uniform(1.0, 10.0)
        q = np.array([rng.uniform(-0.9, 0.9) * -z, rng.uniform(-0.9, 0.9) * -z, z])
        pts[j] = O.to_world(cams15[owner[j]], q)
    by_owner = [np.nonzero(owner == c)[0] for c in range(n_cam)]
    for c in range(n_cam):
        if empty_every and c % empty_every == 0:
            row_ptr.append(len(pt_idx)); continue
        mine = by_owner[c]
        k = min(obs_per_cam, len(mine))
        pick = rng.choice(mine, size=k, replace=False) if k else np.empty(0, dtype=int)
        pt_idx.extend(int(x) for x in pick)
        row_ptr.append(len(pt_idx))
    row_ptr = np.asarray(row_ptr, dtype=np.uint64)
    pt_idx = np.asarray(pt_idx, dtype=np.uint64)
    uv = O.project_observations(cams15, pts, row_ptr, pt_idx)
    if noise:
        uv = uv + rng.normal(scale=noise, size=uv.shape)
    return dict(bal9=bal9, cams15=cams15, pts=pts, row_ptr=row_ptr, pt_idx=pt_idx, uv=uv)


def grid_cameras_points(num_blocks, cpb=10, ppb=10, L=20.0, inset=1.0, cam_h=1.0, pt_h=1.0):
    """Camera / point layout of src/synthetic.rs:178-258 (layout only; visibility is computed by
    the caller).  Returns cams15 (via the oracle's from_position_direction) and pts."""
    dirs = {"-90": O.basis_from_angle_y_deg(-90.0), "90": O.basis_from_angle_y_deg(90.0),
            "180": O.basis_from_angle_y_deg(180.0), "one": np.array([1, 0, 0, 0, 1, 0, 0, 0, 1.0])}
    cams, pts = [], []
    for bx in range(num_blocks + 1):
        ox = L * bx
        for by in range(num_blocks + 1):
            oz = L * by
            for i in range(cpb):
                if bx != num_blocks:
                    loc = [ox + i / cpb * L, cam_h, oz]
                    cams.append(O.from_position_direction(loc, dirs["-90"]))
                    cams.append(O.from_position_direction(loc, dirs["90"]))
                if by != num_blocks:
                    loc = [ox, cam_h, oz + i / cpb * L]
                    cams.append(O.from_position_direction(loc, dirs["180"]))
                    cams.append(O.from_position_direction(loc, dirs["one"]))
    for bx in range(num_blocks + 1):
        ox = L * bx
        for by in range(num_blocks + 1):
            oz = L * by
            for i in range(ppb):
                step = (L - inset * 2.0) / ppb
                if bx != num_blocks:
                    lx = ox + inset + i * step
                    pts += [[lx, pt_h, oz - inset], [lx, pt_h, oz + inset],
                            [lx + step / 2.0, 0.0, oz - inset], [lx + step / 2.0, 0.0, oz + inset],
                            [lx + step / 2.0, 0.0, oz - inset / 2.0], [lx + step / 2.0, 0.0, oz + inset / 2.0]]
                if by != num_blocks:
                    lz = oz + inset + i * step
                    pts += [[ox - inset, pt_h, lz], [ox + inset, pt_h, lz],
                            [ox - inset, 0.0, lz + step / 2.0], [ox + inset, 0.0, lz + step / 2.0],
                            [ox - inset / 2.0, 0.0, lz + step / 2.0], [ox + inset / 2.0, 0.0, lz + step / 2.0]]
    return np.asarray(cams), np.asarray(pts, dtype=np.float64)


def grid_candidate_pairs(cams15, pts, max_dist):
    """All (camera, point) pairs with |center - p|^2 <= max_dist^2, camera-major, point index
    ascending (a canonical order; rstar's traversal order is not reproducible)."""
    ctr = O.centers(cams15)
    cam_idx, pt_idx = [], []
    for c in range(len(cams15)):
        d2 = ((pts - ctr[c]) ** 2).sum(axis=1)
        sel = np.nonzero(d2 <= max_dist * max_dist)[0]
        cam_idx.append(np.full(len(sel), c, dtype=np.uint32))
        pt_idx.append(sel.astype(np.uint32))
    return np.concatenate(cam_idx), np.concatenate(pt_idx)
