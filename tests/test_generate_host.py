"""Host rows of `generate` (src/generate.rs, src/bin/city2ba.rs:480-573) through the C ABI, no GPU: the .obj loader on the
reference's own tests/box.obj (kept as a data fixture under tests/golden/), the camera samplers, modify_intrinsics, the
world-point sampler and the PLY writer, each against an independent numpy statement of the reference's definition."""
import os

import numpy as np
import pytest

BOX = os.path.join(os.path.dirname(__file__), "golden", "box.obj")


@pytest.fixture(scope="module")
def G():
    import __graft_entry__ as entry
    entry.build_hip()
    from city2ba_amd import generate
    return generate


def parse_obj(path):
    """tiny independent .obj reader: global vertices, per-object faces (v index only) and polylines"""
    verts, objs = [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            verts.append([float(x) for x in t[1:4]])
        elif t[0] == "o":
            objs.append({"name": t[1], "faces": [], "lines": []})
        elif t[0] == "f":
            objs[-1]["faces"].append([int(x.split("/")[0]) - 1 for x in t[1:]])
        elif t[0] == "l":
            objs[-1]["lines"].append([int(x) - 1 for x in t[1:]])
    return np.array(verts, dtype=np.float32), objs


def test_obj_loader_matches_file(G):
    verts, objs = parse_obj(BOX)
    o = G.ObjFile(BOX)
    assert o.names() == [m["name"] for m in objs] == ["Cube", "BezierCurve", "Plane"]
    for m, ref in enumerate(objs):
        pos, idx, is_lines = o.model(m)
        assert is_lines == (len(ref["lines"]) > 0)
        if is_lines:
            want = np.array([verts[i] for seg in ref["lines"] for i in seg])           # segment end points
            assert np.array_equal(pos[idx], want)
        else:
            # fan triangulation of each polygon (tobj): (0,1,2), (0,2,3), ...
            want = [verts[[f[0], f[k], f[k + 1]]] for f in ref["faces"] for k in range(1, len(f) - 1)]
            assert np.array_equal(pos[idx].reshape(-1, 3, 3), np.array(want))
    tri = o.triangles(-1)
    assert tri.shape == (14, 9)                              # 6 quads + 1 quad, polyline skipped
    assert np.array_equal(o.triangles(o.index("BezierCurve")), tri)
    assert o.index("nope") == -1
    with pytest.raises(Exception, match="Could not open file"):
        G.ObjFile(BOX + ".missing")


def test_move_to_origin(G):
    o = G.ObjFile(BOX)
    pm = o.index("BezierCurve")
    before = [o.model(m)[0] for m in range(3)]
    o.move_to_origin(pm)                                     # run_generate: the path model is not in the list
    mn = np.minimum(before[0].min(0), before[2].min(0))
    for m in (0, 2):
        assert np.array_equal(o.model(m)[0], before[m] - mn)
    assert np.array_equal(o.model(pm)[0], before[pm])
    o2 = G.ObjFile(BOX)
    o2.move_to_origin()                                      # library call over all models (src/generate.rs:484)
    mn = np.min([b.min(0) for b in before], axis=0)
    for m in range(3):
        assert np.array_equal(o2.model(m)[0], before[m] - mn)


def basis_between(a, b):
    """cgmath Basis3::between_vectors = Quaternion::from_arc(a, b, None) as a matrix, column-major"""
    mag_avg = np.sqrt(np.dot(a, a) * np.dot(b, b))
    dot = np.dot(a, b)
    if abs(dot - mag_avg) <= 1e-12 * max(abs(dot), abs(mag_avg), 1.0):
        q = np.array([1.0, 0, 0, 0])
    else:
        q = np.concatenate([[mag_avg + dot], np.cross(a, b)])
        q = q / np.linalg.norm(q)
    s, x, y, z = q
    return np.array([1 - 2 * (y * y + z * z), 2 * (x * y + s * z), 2 * (x * z - s * y),
                     2 * (x * y - s * z), 1 - 2 * (x * x + z * z), 2 * (y * z + s * x),
                     2 * (x * z + s * y), 2 * (y * z - s * x), 1 - 2 * (x * x + y * y)])


def path_segments(G):
    o = G.ObjFile(BOX)
    pm = o.index("BezierCurve")
    pos, idx, _ = o.model(pm)
    seg = pos[idx].astype(np.float64).reshape(-1, 2, 3)
    return o, pm, seg


def test_cameras_path_random(G):
    o, pm, seg = path_segments(G)
    pos, dirs = G.generate_cameras_path(o, pm, 400, seed=7)
    d = seg[:, 1] - seg[:, 0]
    hit = np.zeros(len(seg), dtype=int)
    for p, m in zip(pos, dirs):
        # every camera sits on exactly one segment x + t (y - x), t in [0, 1), and looks along it
        t = ((p - seg[:, 0]) * d).sum(1) / (d * d).sum(1)
        off = np.linalg.norm(seg[:, 0] + t[:, None] * d - p, axis=1)
        k = int(np.argmin(np.where((t >= 0) & (t < 1), off, np.inf)))
        assert off[k] < 1e-12
        hit[k] += 1
        want = basis_between(d[k] / np.linalg.norm(d[k]), np.array([0.0, 0.0, -1.0]))
        assert np.allclose(m, want, rtol=0, atol=1e-12)
        R = m.reshape(3, 3).T                                 # column-major storage
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert np.allclose(R @ (d[k] / np.linalg.norm(d[k])), [0, 0, -1], atol=1e-12)
    # length-weighted choice of segment: long segments are hit more often than short ones
    ln = np.linalg.norm(d, axis=1)
    order = np.argsort(ln)
    assert hit[order[len(order) // 2:]].sum() > hit[order[:len(order) // 2]].sum()
    again = G.generate_cameras_path(o, pm, 400, seed=7)
    assert np.array_equal(again[0], pos) and np.array_equal(again[1], dirs)      # seeded
    assert not np.array_equal(G.generate_cameras_path(o, pm, 400, seed=8)[0], pos)


def test_cameras_path_step(G):
    o, pm, seg = path_segments(G)
    step = 0.1
    pos, dirs = G.generate_cameras_path_step(o, pm, 100, step)
    # restatement of src/generate.rs:183-210: walk the segments, carrying the remainder over
    d = seg[:, 1] - seg[:, 0]
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    si, dist, want = 0, 0.0, []
    for _ in range(100):
        want.append(seg[si, 0] + (dist / ln[si]) * d[si])
        dist += step
        while dist >= ln[si]:
            dist -= ln[si]
            si += 1
    assert np.array_equal(pos, np.array(want))
    assert np.array_equal(pos[0], seg[0, 0])
    gaps = np.linalg.norm(np.diff(pos, axis=0), axis=1)
    assert np.all(gaps <= step + 1e-12) and np.all(gaps > 0.5 * step)    # chord <= arc
    with pytest.raises(Exception, match="less than the number of cameras"):
        G.generate_cameras_path_step(o, pm, 100, 1.0)                      # assert! at src/generate.rs:172
    with pytest.raises(Exception, match="not a polyline"):
        G.generate_cameras_path(o, o.index("Cube"), 3)


def test_cameras_poisson(G):
    o = G.ObjFile(BOX)
    tri = o.triangles(o.index("BezierCurve"))
    lo, hi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
    height = 1.0
    pos, dirs = G.generate_cameras_poisson(tri, 100, height, 5.0, seed=11)
    assert 80 <= len(pos) <= 140                             # ~0.6 x (2 x num_points): with_samples' radius rule
    # straight down from (x, top + 0.1, z): first surface is the cube top (y = 1) over [-1,1]^2, else the plane
    on_cube = (np.abs(pos[:, 0]) < 1) & (np.abs(pos[:, 2]) < 1)
    surf = np.where(on_cube, 1.0, float(np.float32(-0.876138)))
    edge = (np.abs(np.abs(pos[:, 0]) - 1) < 1e-6) | (np.abs(np.abs(pos[:, 2]) - 1) < 1e-6)
    assert np.allclose(pos[~edge, 1], surf[~edge] + height, atol=2e-6)
    assert on_cube.any() and (~on_cube).any()
    assert np.all(pos[:, 0] >= lo[0]) and np.all(pos[:, 0] <= hi[0]) and np.all(pos[:, 2] >= lo[2]) and np.all(pos[:, 2] <= hi[2])
    # blue noise: no two x-z samples closer than the disk radius of the dart throwing, scaled by the extent
    dx = (pos[:, None, 0] - pos[None, :, 0]) / (hi[0] - lo[0])
    dz = (pos[:, None, 2] - pos[None, :, 2]) / (hi[2] - lo[2])
    dist = np.sqrt(dx * dx + dz * dz) + np.eye(len(pos))
    assert dist.min() >= np.sqrt(2 / np.sqrt(3) / 200) - 1e-12
    # the filter of src/generate.rs:264 compares pt[2] (the z coordinate) with lower_y + ground
    assert np.all(pos[:, 2] < lo[1] + 5.0)
    few, _ = G.generate_cameras_poisson(tri, 100, height, -1.0, seed=11)
    assert 0 < len(few) < len(pos) and np.all(few[:, 2] < lo[1] - 1.0)
    # yaw-only orientation (Basis3::from_angle_y)
    for m in dirs:
        R = m.reshape(3, 3).T
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and R[1, 1] == 1.0 and R[0, 1] == 0.0 and R[2, 1] == 0.0


def test_cameras_poisson_through_hierarchy(G):
    """test_scene.obj has 200 triangles, so the downward rays go through the host hierarchy: every camera must sit
    `height` above the FIRST surface under it (independent float64 vertical ray against all triangles)"""
    o = G.ObjFile(os.path.join(os.path.dirname(BOX), "test_scene.obj"))
    tri = o.triangles(o.index("path")).astype(np.float64).reshape(-1, 3, 3)
    height = 0.75
    pos, _ = G.generate_cameras_poisson(tri.reshape(-1, 9), 300, height, 1e9, seed=4)
    assert len(pos) > 100
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    checked = 0
    for p in pos:
        # barycentric coordinates of (x, z) in each triangle's x-z footprint
        d = (b[:, 0] - a[:, 0]) * (c[:, 2] - a[:, 2]) - (c[:, 0] - a[:, 0]) * (b[:, 2] - a[:, 2])
        with np.errstate(all="ignore"):
            u = ((p[0] - a[:, 0]) * (c[:, 2] - a[:, 2]) - (c[:, 0] - a[:, 0]) * (p[2] - a[:, 2])) / d
            w = ((b[:, 0] - a[:, 0]) * (p[2] - a[:, 2]) - (p[0] - a[:, 0]) * (b[:, 2] - a[:, 2])) / d
            inside = (np.abs(d) > 1e-12) & (u >= 0) & (w >= 0) & (u + w <= 1)
            margin = np.minimum(np.minimum(u, w), 1 - u - w)
        if not inside.any() or np.any(inside & (margin < 1e-4)):
            continue                                              # on an edge: either neighbour may win
        y = a[inside, 1] + u[inside] * (b[inside, 1] - a[inside, 1]) + w[inside] * (c[inside, 1] - a[inside, 1])
        assert abs(p[1] - (y.max() + height)) < 1e-4
        checked += 1
    assert checked > 0.8 * len(pos)


def test_modify_intrinsics(G):
    cams = np.arange(15 * 50, dtype=np.float64).reshape(50, 15)
    out = G.modify_intrinsics(cams, [1.0, -0.1, 0.0], [2.0, 0.1, 0.0], seed=5)
    assert np.array_equal(out[:, :12], cams[:, :12])
    assert np.all(out[:, 12] >= 1.0) and np.all(out[:, 12] < 2.0) and out[:, 12].std() > 0.1
    assert np.all(out[:, 13] >= -0.1) and np.all(out[:, 13] < 0.1)
    assert np.all(out[:, 14] == 0.0)
    same = G.modify_intrinsics(cams, [1.0, 0.0, 0.0], [1.0, 0.0, 0.0], seed=5)      # the CLI default: f=1, k=0
    assert np.all(same[:, 12] == 1.0) and np.all(same[:, 13:] == 0.0)


def test_world_points_on_mesh_near_cameras(G):
    o = G.ObjFile(BOX)
    tri = o.triangles(o.index("BezierCurve"))
    centers = np.array([[3.0, 0.0, 3.0], [-3.0, 0.5, 3.0]])
    max_dist = 2.5
    pts = G.generate_world_points_uniform(tri, centers, 300, max_dist, seed=3)
    assert len(pts) == 300
    # candidates draw from counter-based streams and are accepted in candidate order: a shorter request is a prefix
    assert np.array_equal(G.generate_world_points_uniform(tri, centers, 120, max_dist, seed=3), pts[:120])
    assert not np.array_equal(G.generate_world_points_uniform(tri, centers, 120, max_dist, seed=4), pts[:120])
    d = np.linalg.norm(pts[:, None, :] - centers[None], axis=2).min(1)
    assert np.all(d <= max_dist)
    # every point lies on a triangle of the mesh: barycentric coordinates in [0,1] and zero plane distance
    T = tri.astype(np.float64).reshape(-1, 3, 3)
    ok = np.zeros(len(pts), dtype=bool)
    for a, b, c in T:
        n = np.cross(b - a, c - a)
        n = n / np.linalg.norm(n)
        M = np.stack([b - a, c - a, n], axis=1)
        uvw = np.linalg.solve(M, (pts - a).T).T
        ok |= (np.abs(uvw[:, 2]) < 1e-9) & (uvw[:, 0] >= -1e-9) & (uvw[:, 1] >= -1e-9) & (uvw[:, 0] + uvw[:, 1] <= 1 + 1e-9)
    assert ok.all()
    with pytest.raises(Exception, match="0 cameras"):
        G.generate_world_points_uniform(tri, np.zeros((0, 3)), 10, 1.0)
    with pytest.raises(Exception):
        G.generate_world_points_uniform(tri, np.array([[100.0, 100.0, 100.0]]), 10, 1.0)    # nothing in range


def test_ply_writer(tmp_path):
    import ctypes as C
    import __graft_entry__ as entry
    entry.build_hip()
    from city2ba_amd import _lib as L
    centers = np.array([[0.5, 1.0, -2.0], [1e-3, 1234567.0, 0.1]])
    pts = np.array([[1.0, 2.0, 3.0], [0.25, -0.125, 1e10], [7.0, 8.0, 9.5]])
    row_ptr = np.array([0, 2, 3], dtype=np.uint64)
    pt_idx = np.array([0, 2, 1], dtype=np.uint64)
    out = tmp_path / "c.ply"
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.check(L.lib().c2b_ply_write(os.fsencode(str(out)), 2, p(centers), 3, p(pts), p(row_ptr), p(pt_idx)))
    lines = out.read_text().split("\n")
    assert lines[:3] == ["ply", "format ascii 1.0", "element vertex 5"]
    assert lines[3:9] == ["property float x", "property float y", "property float z", "property uchar red",
                          "property uchar green", "property uchar blue"]
    assert lines[9:13] == ["element edge 3", "property int vertex1", "property int vertex2", "end_header"]
    body = lines[13:]
    assert body[0] == "0.5 1 -2 255 0 0"                      # Rust Display of f32: shortest round-trip digits
    assert body[1] == "0.001 1234567 0.1 255 0 0"
    assert body[2] == "1 2 3 0 255 0"
    assert body[3] == "0.25 -0.125 10000000000 0 255 0"
    assert body[5:8] == ["0 2", "0 4", "1 3"] and body[8] == ""
