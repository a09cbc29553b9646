"""Seeded synthetic BA problems shared by the CPU and GPU tests (inputs only)."""
import math

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


# ---- numpy restatement of the grid layout / candidate search (independent of the C++ host code) ----
def _basis_y(deg):
    """Basis3::from_angle_y(Deg(deg)) as col-major 9 (cgmath: Deg -> Rad is deg * (PI/180));
    libm sin/cos like Rust's f64::sin_cos."""
    th = deg * (math.pi / 180.0)
    s, c = math.sin(th), math.cos(th)
    return np.array([c, 0.0, -s, 0.0, 1.0, 0.0, s, 0.0, c])


def np_grid_layout(num_blocks, cpb=10, ppb=10, block_length=20.0, block_inset=1.0, camera_height=1.0,
                point_height=1.0):
    """cams15 [4*cpb*B*(B+1), 15] and pts [12*ppb*B*(B+1), 3] in the reference's push order."""
    B, L, ins = int(num_blocks), float(block_length), float(block_inset)
    assert ins * 2.0 < L, "Block inset must be less than half the block length"
    bx, by, i, k = np.meshgrid(np.arange(B + 1), np.arange(B + 1), np.arange(cpb), np.arange(4), indexing="ij")
    bx, by, i, k = bx.ravel(), by.ravel(), i.ravel(), k.ravel()
    horiz = k < 2
    ok = np.where(horiz, bx != B, by != B)
    bx, by, i, k, horiz = bx[ok], by[ok], i[ok], k[ok], horiz[ok]
    off_x, off_z = L * bx.astype(np.float64), L * by.astype(np.float64)
    along = i.astype(np.float64) / float(cpb) * L
    px = np.where(horiz, off_x + along, off_x)
    pz = np.where(horiz, off_z, off_z + along)
    py = np.full_like(px, float(camera_height))
    dirs = np.stack([_basis_y(-90.0), _basis_y(90.0), _basis_y(180.0), np.array([1.0, 0, 0, 0, 1, 0, 0, 0, 1])])
    R = dirs[k]                                               # [n, 9] col-major
    # from_position_direction (src/baproblem.rs:153-159): loc = -1.0 * (dir . pos), dot = (a+b)+c
    t = np.stack([-1.0 * ((R[:, 0] * px + R[:, 3] * py) + R[:, 6] * pz),
                  -1.0 * ((R[:, 1] * px + R[:, 4] * py) + R[:, 7] * pz),
                  -1.0 * ((R[:, 2] * px + R[:, 5] * py) + R[:, 8] * pz)], axis=1)
    intr = np.tile(np.array([1.0, 0.0, 0.0]), (len(px), 1))
    cams15 = np.ascontiguousarray(np.concatenate([R, t, intr], axis=1))

    step = (L - ins * 2.0) / float(ppb)
    bx, by, i, k = np.meshgrid(np.arange(B + 1), np.arange(B + 1), np.arange(ppb), np.arange(12), indexing="ij")
    bx, by, i, k = bx.ravel(), by.ravel(), i.ravel(), k.ravel()
    horiz = k < 6
    ok = np.where(horiz, bx != B, by != B)
    bx, by, i, k, horiz = bx[ok], by[ok], i[ok], k[ok], horiz[ok]
    off_x, off_z = L * bx.astype(np.float64), L * by.astype(np.float64)
    kk = k % 6
    base = np.where(horiz, off_x, off_z) + ins + i.astype(np.float64) * step     # loc_x / loc_z
    along = np.where(kk < 2, base, base + step / 2.0)
    other0 = np.where(horiz, off_z, off_x)
    lateral = np.select([kk == 0, kk == 1, kk == 2, kk == 3, kk == 4, kk == 5],
                        [other0 - ins, other0 + ins, other0 - ins, other0 + ins,
                         other0 - ins / 2.0, other0 + ins / 2.0])
    y = np.where(kk < 2, float(point_height), 0.0)
    x = np.where(horiz, along, lateral)
    z = np.where(horiz, lateral, along)
    pts = np.ascontiguousarray(np.stack([x, y, z], axis=1))
    return cams15, pts


def np_candidate_pairs(centers, pts, max_dist, cam_lo=0, cam_hi=None, chunk=100_000):
    """(cam, point) pairs with |center - p|^2 <= max_dist^2 (rstar's locate_within_distance takes the
    squared radius, src/synthetic.rs:277-280), camera-major, ascending point index per camera.
    Uniform-cell binning over (x, z)."""
    cam_hi = len(centers) if cam_hi is None else cam_hi
    cs = float(max_dist)
    x0 = min(pts[:, 0].min(), centers[:, 0].min()) - cs
    z0 = min(pts[:, 2].min(), centers[:, 2].min()) - cs
    pcx = np.floor((pts[:, 0] - x0) / cs).astype(np.int64)
    pcz = np.floor((pts[:, 2] - z0) / cs).astype(np.int64)
    ncz = int(pcz.max()) + 3
    ncx = int(pcx.max()) + 3
    cell = pcx * ncz + pcz
    order = np.argsort(cell, kind="stable")
    sorted_cell = cell[order]
    starts = np.searchsorted(sorted_cell, np.arange(ncx * ncz + 1))
    out_c, out_p = [], []
    r2 = max_dist * max_dist
    for lo in range(cam_lo, cam_hi, chunk):
        hi = min(lo + chunk, cam_hi)
        c = centers[lo:hi]
        ccx = np.floor((c[:, 0] - x0) / cs).astype(np.int64)
        ccz = np.floor((c[:, 2] - z0) / cs).astype(np.int64)
        cams, pidx = [], []
        for dx in (-1, 0, 1):
            for dz in (-1, 0, 1):
                cid = (ccx + dx) * ncz + (ccz + dz)
                s, e = starts[cid], starts[cid + 1]
                ln = e - s
                tot = int(ln.sum())
                if tot == 0:
                    continue
                rep = np.repeat(np.arange(lo, hi), ln)
                first = np.repeat(np.cumsum(ln) - ln, ln)
                pos = np.arange(tot) - first + np.repeat(s, ln)
                cams.append(rep)
                pidx.append(order[pos])
        cams = np.concatenate(cams)
        pidx = np.concatenate(pidx)
        d = pts[pidx] - centers[cams]
        keep = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2] <= r2
        cams, pidx = cams[keep], pidx[keep]
        o = np.lexsort((pidx, cams))
        out_c.append(cams[o].astype(np.int32))
        out_p.append(pidx[o].astype(np.int32))
    if not out_c:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    return np.concatenate(out_c), np.concatenate(out_p)
