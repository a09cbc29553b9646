"""GPU tests of the `generate` row (src/generate.rs:424-481 visibility_graph; src/bin/city2ba.rs:480-573 run_generate):
the occlusion-ray kernel against an independent float32 numpy Moeller-Trumbore on top of the oracle's visibility
predicate, and the reference's own generate CLI tests (tests/main.rs:65-128) on its tests/box.obj."""
import os
import subprocess

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

BOX = os.path.join(os.path.dirname(__file__), "golden", "box.obj")
f32 = np.float32


@pytest.fixture(scope="module")
def c2b():
    import __graft_entry__ as entry
    entry.build()
    import city2ba_amd
    assert city2ba_amd.device_count() > 0
    return city2ba_amd


@pytest.fixture(scope="module")
def cli():
    import __graft_entry__ as entry
    return entry.build_cli()


def occluded_numpy(centers, pts, tri):
    """float32 rays as embree_rs::Ray::new(center as f32, dir.normalize() as f32), tfar = |dir| as f32 - 1e-6
    (src/generate.rs:456-464); occluded iff some triangle is hit with 0 < t <= tfar.  Same operation order as the
    kernel, numpy float32 arithmetic (no fused multiply-add)."""
    e = pts - centers
    mag = np.sqrt((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2])
    inv = 1.0 / mag
    o = centers.astype(f32)
    d = (e * inv[:, None]).astype(f32)
    tfar = mag.astype(f32) - f32(1e-6)
    occ = np.zeros(len(pts), dtype=bool)
    with np.errstate(all="ignore"):
        for q in tri.astype(f32):
            e1, e2 = q[3:6] - q[0:3], q[6:9] - q[0:3]
            px = d[:, 1] * e2[2] - d[:, 2] * e2[1]
            py = d[:, 2] * e2[0] - d[:, 0] * e2[2]
            pz = d[:, 0] * e2[1] - d[:, 1] * e2[0]
            det = (e1[0] * px + e1[1] * py) + e1[2] * pz
            nz = det != 0
            idet = f32(1.0) / det
            tx, ty, tz = o[:, 0] - q[0], o[:, 1] - q[1], o[:, 2] - q[2]
            u = ((tx * px + ty * py) + tz * pz) * idet
            qx = ty * e1[2] - tz * e1[1]
            qy = tz * e1[0] - tx * e1[2]
            qz = tx * e1[1] - ty * e1[0]
            w = ((d[:, 0] * qx + d[:, 1] * qy) + d[:, 2] * qz) * idet
            th = ((e2[0] * qx + e2[1] * qy) + e2[2] * qz) * idet
            hit = nz & ~(u < 0) & ~(u > 1) & ~(w < 0) & ~(u + w > 1) & (th > 0) & (th <= tfar)
            occ |= hit
    return occ


def scene(seed, n_cam=40, n_pts=700, n_tri=90):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-6, 6, (n_cam, 3))
    # look at the origin-ish: -z axis of the camera towards a random target
    tgt = rng.uniform(-1, 1, (n_cam, 3))
    cams = np.zeros((n_cam, 15))
    for i in range(n_cam):
        fwd = tgt[i] - pos[i]
        fwd /= np.linalg.norm(fwd)
        up = np.array([0.0, 1.0, 0.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        R = np.stack([right, up, -fwd])                      # world -> camera, camera looks down -z
        cams[i, :9] = R.T.reshape(9)                         # column-major
        cams[i, 9:12] = -R @ pos[i]
        cams[i, 12:] = [rng.uniform(0.8, 1.5), rng.uniform(-0.05, 0.05), 0.0]
    pts = rng.uniform(-4, 4, (n_pts, 3))
    a = rng.uniform(-4, 4, (n_tri, 3))
    tri = np.concatenate([a, a + rng.normal(0, 1.2, (n_tri, 3)), a + rng.normal(0, 1.2, (n_tri, 3))], axis=1).astype(f32)
    return cams, pts, tri


@pytest.mark.parametrize("seed,n_tri", [(1, 90), (2, 300), (3, 1)])
def test_visibility_graph_with_occlusion(c2b, seed, n_tri):
    cams, pts, tri = scene(seed, n_tri=n_tri)
    n_cam, n_pts = len(cams), len(pts)
    max_dist = 9.0
    empty = np.zeros(n_cam + 1, dtype=np.uint64)
    ba = c2b.BAProblem.from_visibility(cams, pts, empty, [], np.zeros((0, 2)))
    row0, pi0, uv0 = ba.visibility_graph(max_dist)
    row_d, pi_d, uv_d = ba.visibility_graph(max_dist, dense=True)         # the brute-force sweep: the same lists, bit for bit
    assert np.array_equal(row_d, row0) and np.array_equal(pi_d, pi0) and np.array_equal(uv_d.view(np.uint64), uv0.view(np.uint64))
    row_t, pi_t, uv_t = ba.visibility_graph(max_dist, triangles=tri, dense=True)
    row, pi, uv = ba.visibility_graph(max_dist, triangles=tri)
    assert np.array_equal(row_t, row) and np.array_equal(pi_t, pi) and np.array_equal(uv_t.view(np.uint64), uv.view(np.uint64))
    # checker: the oracle's predicate over all pairs, then the numpy rays over the survivors
    ci_all = np.repeat(np.arange(n_cam, dtype=np.uint32), n_pts)
    pi_all = np.tile(np.arange(n_pts, dtype=np.uint32), n_cam)
    uv_all, keep = O.visibility_pairs(cams, pts, ci_all, pi_all, max_dist)
    k = keep == 1
    ci_k, pi_k, uv_k = ci_all[k], pi_all[k], uv_all[k]
    assert np.array_equal(pi0, pi_k.astype(np.uint64)) and int(row0[-1]) == int(k.sum())
    occ = occluded_numpy(O.centers(cams)[ci_k], pts[pi_k], tri)
    assert np.array_equal(O.occlusion_filter(cams, pts, ci_k, pi_k, tri) == 0, occ)      # C oracle == numpy statement
    assert 0.05 * len(occ) < occ.sum() < 0.98 * len(occ) or n_tri == 1
    want_row = np.concatenate([[0], np.cumsum(np.bincount(ci_k[~occ], minlength=n_cam))]).astype(np.uint64)
    assert np.array_equal(row, want_row)
    assert np.array_equal(pi, pi_k[~occ].astype(np.uint64))
    assert np.allclose(uv, uv_k[~occ], rtol=0, atol=1e-12)
    # the hierarchy built by the caller (c2b_problem_visibility_dense_occlude_bvh; also for meshes below the size at which
    # the library would build one itself): the same lists
    row_b, pi_b, uv_b = ba.visibility_graph(max_dist, triangles=tri, prebuilt_hierarchy=True)
    assert np.array_equal(row_b, row) and np.array_equal(pi_b, pi) and np.array_equal(uv_b, uv)
    # a second fetch after the filter returns the same lists; an empty mesh filters nothing
    row2, pi2, _ = ba.visibility_graph(max_dist, triangles=np.zeros((0, 9), f32))
    assert np.array_equal(row2, row0) and np.array_equal(pi2, pi0)


def test_bvh_filter_equals_all_triangles_filter(c2b):
    """Level 0: the hierarchy only prunes -- the mask equals the all-triangles kernel's on random soups of mixed
    scale (incl. axis-aligned triangles and rays) and on a city-block mesh with the synthetic grid's own rays"""
    import torch
    from city2ba_amd import device as D
    from city2ba_amd import synthetic as S
    dev = torch.device("cuda", 0)
    for seed, n_tri in ((4, 65), (5, 5000), (6, 60000)):
        cams, pts, _ = scene(seed, n_cam=64, n_pts=3000, n_tri=1)
        rng = np.random.default_rng(seed)
        a = rng.uniform(-5, 5, (n_tri, 3))
        size = rng.choice([0.05, 0.4, 3.0], n_tri, p=[0.6, 0.3, 0.1])[:, None] * min(1.0, (300.0 / n_tri) ** 0.5)
        tri = np.concatenate([a, a + rng.normal(0, 1, (n_tri, 3)) * size, a + rng.normal(0, 1, (n_tri, 3)) * size], axis=1).astype(f32)
        tri[: n_tri // 10, [1, 4, 7]] = f32(0.5)                              # horizontal triangles: zero-height boxes
        pts[:200, 1] = O.centers(cams)[np.arange(200) % len(cams), 1]          # rays with dy == 0 exactly
        camblk = D.cameras_prepare_state(torch.from_numpy(cams).to(dev))
        pts4 = D.points_pad(torch.from_numpy(pts).to(dev))
        ci = torch.from_numpy(np.repeat(np.arange(len(cams)), len(pts)).astype(np.int32)).to(dev)
        pi = torch.from_numpy(np.tile(np.arange(len(pts)), len(cams)).astype(np.int32)).to(dev)
        brute = torch.empty(len(ci), dtype=torch.uint8, device=dev)
        D.occlusion_filter(camblk, pts4, ci, pi, torch.from_numpy(tri).to(dev), brute)
        bvh = D.OcclusionBVH(tri, dev)
        got = torch.full((len(ci),), 7, dtype=torch.uint8, device=dev)
        bvh.filter(camblk, pts4, ci, pi, got)
        assert bvh.depth < 64 and sorted(bvh.order.tolist()) == list(range(n_tri))
        assert torch.equal(got, brute)
        frac = float(brute.float().mean())
        assert 0.0 < frac < 1.0
    # city blocks (one box per block) against the grid generator's own cameras and points
    B, L, inset = 6, 20.0, 1.0
    pos, dirs, gpts = S.grid_layout(B, 10, 10, L, inset, 1.0, 1.0)
    cam15 = D.cameras_from_position_direction(torch.from_numpy(pos).to(dev), torch.from_numpy(dirs).to(dev))
    cen4 = D.centers_table(cam15.shape[0], dev)
    camblk = D.cameras_prepare_state(cam15, centers=cen4)       # (camblk is blocked by groups of 8 cameras: the centres come from their table)
    pts4 = D.points_pad(torch.from_numpy(gpts).to(dev))
    ci, pi = S.candidate_pairs(cen4[:, :3].cpu().numpy().copy(), gpts, 10.0)
    ci_d, pi_d = torch.from_numpy(ci.astype(np.int32)).to(dev), torch.from_numpy(pi.astype(np.int32)).to(dev)
    boxes = []
    for bx in range(B):
        for bz in range(B):
            x0, x1, z0, z1 = L * bx + inset, L * (bx + 1) - inset, L * bz + inset, L * (bz + 1) - inset   # facades ON the points
            c = [(x0, -1, z0), (x1, -1, z0), (x1, -1, z1), (x0, -1, z1), (x0, 9, z0), (x1, 9, z0), (x1, 9, z1), (x0, 9, z1)]
            for q in ((0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7), (4, 5, 6, 7), (0, 3, 2, 1)):
                boxes.append(c[q[0]] + c[q[1]] + c[q[2]])
                boxes.append(c[q[0]] + c[q[2]] + c[q[3]])
    tri = np.array(boxes, dtype=f32)
    brute = torch.empty(len(ci), dtype=torch.uint8, device=dev)
    D.occlusion_filter(camblk, pts4, ci_d, pi_d, torch.from_numpy(tri).to(dev), brute)
    got = torch.empty_like(brute)
    D.OcclusionBVH(tri, dev).filter(camblk, pts4, ci_d, pi_d, got)
    assert torch.equal(got, brute) and 0.2 < float(brute.float().mean()) < 0.95


def test_occlusion_wall(c2b):
    """one camera at the origin looking down -z, a wall at z = -2 covering x < 0: points behind the wall on that
    side are dropped, points in front of it or on the open side are kept; a point ON the wall is its own occluder
    only if the hit falls inside tfar = |dir| - 1e-6 (the reference stops 'a little short of the point')."""
    cams = np.zeros((1, 15))
    cams[0, :9] = np.eye(3).reshape(9)
    cams[0, 12] = 1.0
    wall = np.array([[-50, -50, -2, 0, -50, -2, 0, 50, -2], [-50, -50, -2, 0, 50, -2, -50, 50, -2]], dtype=f32)
    pts = np.array([[-1.0, 0.2, -4.0],      # behind the wall: occluded
                    [1.0, 0.2, -4.0],       # open side: kept
                    [-0.4, 0.1, -1.0],      # in front of the wall: kept
                    [-0.5, -0.3, -3.0],     # behind: occluded
                    [0.3, 0.3, -1.5]])      # open side, in front
    ba = c2b.BAProblem.from_visibility(cams, pts, np.zeros(2, np.uint64), [], np.zeros((0, 2)))
    row, pi, uv = ba.visibility_graph(100.0, triangles=wall)
    assert list(pi) == [1, 2, 4] and list(row) == [0, 3]
    assert np.allclose(uv[0], [0.25, 0.05])                   # -f * x / z with f = 1: (1/4, .2/4)


# ---- the reference's generate CLI tests, tests/main.rs:65-128 ------------------------------------------------------
def _run(cli, *args):
    return subprocess.run([cli] + [str(a) for a in args], capture_output=True, text=True, timeout=300)


def _check_generated(c2b, r, path, no_lcc=False):
    assert r.returncode == 0, r.stderr
    for s in ("Generated ", "Modified intrinsics", " world points", "Computed visibility graph with ",
              "Computed LCC with ", "Total reprojection error"):
        assert s in r.stdout, r.stdout
    ba = c2b.BAProblem.from_file(path)
    assert ba.num_cameras() > 0 and ba.num_points() > 0
    assert ("Computed LCC with %d cameras, %d points, %d edges" % (ba.num_cameras(), ba.num_points(), ba.num_observations())) in r.stdout
    assert ba.total_reprojection_error(1.0) < 1e-9 * max(1, ba.num_observations())      # observations = projections
    if not no_lcc:
        assert np.all(np.diff(ba.row_ptr.astype(np.int64)) > 3)                          # cull post-conditions
        assert np.all(np.bincount(ba.pt_idx.astype(np.int64), minlength=ba.num_points()) > 1)
    return ba


def test_cli_from_box_path(c2b, cli, tmp_path):              # from_box_path, :65-84
    out = tmp_path / "box.bal"
    r = _run(cli, "generate", BOX, out, "--cameras", "100", "--points", "100", "--path", "BezierCurve", "--seed", "1")
    _check_generated(c2b, r, out)
    assert "Generated 100 cameras" in r.stdout and "Generated 100 world points" in r.stdout


def test_cli_from_box_path_step(c2b, cli, tmp_path):         # from_box_path_step, :86-107
    out = tmp_path / "box.bal"
    r = _run(cli, "generate", BOX, out, "--cameras", "100", "--points", "100", "--path", "BezierCurve",
             "--step-size", "0.1", "--seed", "1")
    _check_generated(c2b, r, out)
    assert "Generated 100 cameras" in r.stdout


def test_cli_from_box_path_ground(c2b, cli, tmp_path):       # from_box_path_ground, :109-128
    out = tmp_path / "box.bal"
    r = _run(cli, "generate", BOX, out, "--cameras", "100", "--points", "100", "--ground", "-1.0", "--seed", "1")
    _check_generated(c2b, r, out)


def test_cli_readme_scene(c2b, cli, tmp_path):
    """the reference's documented invocation (Readme.md:10, src/lib.rs:18) on its test_scene.obj, Poisson cameras,
    and the same scene along its `path` polyline"""
    scene_obj = os.path.join(os.path.dirname(BOX), "test_scene.obj")
    out = tmp_path / "problem.bal"
    r = _run(cli, "generate", scene_obj, out, "--cameras", "100", "--points", "200", "--seed", "1")
    _check_generated(c2b, r, out)
    r = _run(cli, "generate", scene_obj, tmp_path / "p.bbal", "--cameras", "100", "--points", "200", "--path", "path", "--seed", "1")
    _check_generated(c2b, r, tmp_path / "p.bbal")


def test_cli_generate_flags_and_errors(c2b, cli, tmp_path):
    out = tmp_path / "b.bbal"
    r = _run(cli, "generate", BOX, out, "--cameras", "60", "--points", "400", "--path", "BezierCurve", "--no-lcc",
             "--move-to-origin", "--intrinsics-start", "1,-0.1,0", "--intrinsics-end", "2,0.1,0", "--max-dist", "5", "--seed", "4")
    ba = _check_generated(c2b, r, out, no_lcc=True)
    assert ba.num_cameras() == 60 and ba.num_points() == 400          # --no-lcc keeps everything
    f = ba.cameras()[:, 12]
    assert np.all(f >= 1.0) and np.all(f < 2.0) and f.std() > 0.05
    r = _run(cli, "generate", BOX, out, "--path", "Nope")
    assert r.returncode != 0 and "Could not find a path named Nope. Available model names are Cube, BezierCurve, Plane" in r.stderr
    r = _run(cli, "generate", BOX, out, "--path", "BezierCurve", "--ground", "1")
    assert r.returncode != 0 and "cannot be used with" in r.stderr       # conflicts_with = "ground"
    r = _run(cli, "generate", BOX, out, "--path", "BezierCurve", "--cameras", "100", "--step-size", "1.0")
    assert r.returncode != 0 and "less than the number of cameras" in r.stderr
    r = _run(cli, "generate", BOX, out, "--cameras", "0", "--path", "BezierCurve")
    assert r.returncode != 0 and "0 cameras" in r.stderr                 # panic at src/generate.rs:364
    r = _run(cli, "generate", BOX, out, "--path", "BezierCurve", "--points", "30", "--max-dist", "0.3", "--seed", "2")
    assert r.returncode != 0                                             # nothing survives: rejection limit or EmptyProblem


def test_python_generate_equals_cli(c2b, cli, tmp_path):
    """two hosts over the same C ABI with the same seeds produce the same problem"""
    from city2ba_amd import generate as G
    out = tmp_path / "g.bbal"
    r = _run(cli, "generate", BOX, out, "--cameras", "80", "--points", "300", "--path", "BezierCurve", "--seed", "9")
    assert r.returncode == 0, r.stderr
    got = c2b.BAProblem.from_file(out)
    ba = G.generate(BOX, num_cameras=80, num_world_points=300, path_name="BezierCurve", seed=9)
    assert str(ba) == str(got)
    assert np.array_equal(ba.row_ptr, got.row_ptr) and np.array_equal(ba.pt_idx, got.pt_idx)
    assert np.array_equal(ba.points(), got.points()) and np.array_equal(ba.observations(), got.observations())
    assert np.max(np.abs(ba.cameras() - got.cameras())) < 1e-12           # file round trip of the rotation
    # Poisson placement, library call
    ba2 = G.generate(BOX, num_cameras=100, num_world_points=200, ground=-1.0, seed=1)
    assert ba2.num_cameras() > 3 and ba2.total_reprojection_error(2.0) < 1e-9


def test_cli_ply(c2b, cli, tmp_path):                       # run_ply, src/bin/city2ba.rs:441-445
    bal = tmp_path / "s.bal"
    assert _run(cli, "synthetic", bal, "--blocks", "2").returncode == 0
    ply = tmp_path / "s.ply"
    r = _run(cli, "ply", bal, ply)
    assert r.returncode == 0, r.stderr
    ba = c2b.BAProblem.from_file(bal)
    lines = ply.read_text().split("\n")
    nc, npt, no = ba.num_cameras(), ba.num_points(), ba.num_observations()
    assert lines[2] == "element vertex %d" % (nc + npt) and lines[9] == "element edge %d" % no
    body = lines[13:]
    assert len(body) == nc + npt + no + 1
    v = np.array([[float(x) for x in ln.split()] for ln in body[:nc + npt]])
    assert np.array_equal(v[:nc, :3].astype(f32), ba._camera_centers().astype(f32))
    assert np.array_equal(v[nc:, :3].astype(f32), ba.points().astype(f32))
    assert np.all(v[:nc, 3:] == [255, 0, 0]) and np.all(v[nc:, 3:] == [0, 255, 0])
    e = np.array([[int(x) for x in ln.split()] for ln in body[nc + npt:-1]])
    ci = np.repeat(np.arange(nc), np.diff(ba.row_ptr.astype(np.int64)))
    assert np.array_equal(e[:, 0], ci) and np.array_equal(e[:, 1], ba.pt_idx.astype(np.int64) + nc)


def test_bvh_stack_overflow_is_flagged_not_silent(c2b):
    """a hierarchy deeper than the 64-entry traversal stack (never one from c2b_bvh_build, which refuses
    them -- but Level 0 takes caller-made node arrays) used to lose subtrees silently.  A comb of 70 levels whose two
    children are both inner nodes with identical boxes makes the traversal push one entry per level: the kernel must
    raise the overflow flag, and the Python wrapper turns it into an error."""
    import torch
    from city2ba_amd import device as D
    from city2ba_amd import _lib as L
    dev = torch.device("cuda", 0)
    cams = O.camera_from_bal([0, 0, 0, 0, 0, 5.0, 1.0, 0, 0]).reshape(1, 15)       # centre (0, 0, -5)
    pts = np.array([[0.0, 0.0, 5.0], [0.1, 0.1, 4.0]])
    camblk = D.cameras_prepare_state(torch.from_numpy(cams).to(dev))
    pts4 = D.points_pad(torch.from_numpy(pts).to(dev))
    ci = torch.zeros(2, dtype=torch.int32, device=dev)
    pi = torch.arange(2, dtype=torch.int32, device=dev)
    tri = np.array([[100, 100, 100, 101, 100, 100, 100, 101, 100]], dtype=f32)      # far away: never hit
    good = D.OcclusionBVH(tri, dev)

    def comb(depth):
        empty = np.int32(-2 ** 31)
        nodes = np.zeros((2 * depth, 16), dtype=f32)
        ints = nodes.view(np.int32)
        box = [-10, -10, -10, 10, 10, 10]
        for k in range(depth):
            nodes[k, 0:6] = box
            nodes[k, 6:12] = box
            ints[k, 12] = k + 1 if k + 1 < depth else empty                        # child 0: the next comb node
            ints[k, 13] = depth + k                                                 # child 1: a dead end (pushed)
            nodes[depth + k, 0:12] = box + box
            ints[depth + k, 12] = empty
            ints[depth + k, 13] = empty
        return nodes

    for depth, expect_overflow in ((40, False), (70, True)):
        bvh = D.OcclusionBVH(tri, dev)
        nodes = comb(depth)
        bvh.nodes = torch.from_numpy(nodes).to(dev)
        bvh.n_nodes = len(nodes)
        keep = torch.full((2,), 7, dtype=torch.uint8, device=dev)
        if expect_overflow:
            with pytest.raises(L.City2baError) as ei:
                bvh.filter(camblk, pts4, ci, pi, keep)
            assert ei.value.status == L.ERR_INVALID_ARGUMENT and "traversal stack" in str(ei.value)
        else:
            bvh.filter(camblk, pts4, ci, pi, keep)
            assert keep.tolist() == [1, 1]                                          # nothing in the comb occludes
    keep = torch.full((2,), 7, dtype=torch.uint8, device=dev)
    good.filter(camblk, pts4, ci, pi, keep)
    assert keep.tolist() == [1, 1]


def test_world_points_sampled_on_the_device_are_the_host_samplers(c2b):
    """generate_world_points_uniform (src/generate.rs:356-420) on the device (c2b_problem_generate_world_points: candidate
    k from its own splitmix64 stream, the triangle by area, the point, `some camera within max_dist` through a cell
    list, acceptance in candidate order): the same points, bit for bit, as the host sampler -- with most candidates
    rejected, with a request that ends inside a chunk, with max_dist larger than the scene -- and its errors."""
    from city2ba_amd import generate as G
    rng = np.random.default_rng(7)
    n = 24
    h = rng.uniform(0, 0.5, (n + 1, n + 1)).astype(np.float32)
    tri = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = (i, h[i, j], j), (i + 1, h[i + 1, j], j), (i, h[i, j + 1], j + 1), (i + 1, h[i + 1, j + 1], j + 1)
            tri += [a + b + c, b + d + c]
    tri = np.array(tri, dtype=np.float32)
    pos = np.column_stack([rng.uniform(2, 9, 40), rng.uniform(1, 2, 40), rng.uniform(3, 20, 40)])
    dirs = np.tile(np.eye(3).ravel(), (40, 1))
    stage = c2b.BAProblem(0)
    cams = stage._cameras_from_position_direction(pos, dirs)
    empty = np.zeros(len(cams) + 1, dtype=np.uint64)
    for num, max_dist, seed in ((5000, 2.5, 3), (70_000, 1.6, 4), (300, 100.0, 5), (0, 3.0, 6)):
        ba = c2b.BAProblem.from_visibility(cams, np.zeros((0, 3)), empty, [], np.zeros((0, 2)))
        centers = ba._camera_centers()
        want = G.generate_world_points_uniform(tri, centers, num, max_dist, seed)
        got_n = ba.generate_world_points(tri, num, max_dist, seed)
        assert got_n == len(want) == num
        assert np.array_equal(ba.points().view(np.uint64), want.view(np.uint64))
        if num:
            d = np.sqrt(((want[:, None, :] - centers[None, :, :]) ** 2).sum(-1)).min(1) if num <= 5000 else None
            assert d is None or d.max() <= max_dist * (1 + 1e-12)
        ba.close()
    # the reference's errors: nothing near any camera (10 * num failures); no cameras; then a problem that has observations
    ba = c2b.BAProblem.from_visibility(cams, np.zeros((0, 3)), empty, [], np.zeros((0, 2)))
    far = tri + np.float32(1000.0)
    with pytest.raises(c2b.City2baError, match="Failed to generate enough points. 0 successes, 500 failures, 50 requested"):
        ba.generate_world_points(far, 50, 2.0, 1)
    with pytest.raises(c2b.City2baError, match="Failed to generate enough points"):
        G.generate_world_points_uniform(far, ba._camera_centers(), 50, 2.0, 1)
    ba.close()
    e = c2b.BAProblem.from_visibility(np.zeros((0, 15)), np.zeros((0, 3)), np.zeros(1, dtype=np.uint64), [], np.zeros((0, 2)))
    with pytest.raises(c2b.City2baError, match="0 cameras"):
        e.generate_world_points(tri, 10, 2.0, 1)
    e.close()


def test_cli_generate_device_and_host_routes_write_the_same_file(c2b, cli, tmp_path):
    """`city2ba generate` samples its world points on the device and takes its candidates from the cell list (r04); the
    host sampler (C2B_HOST_SAMPLER=1) and the all-pairs sweep (C2B_DENSE_SWEEP=1) are the rounds 1-3 routes: every
    combination writes the same file, along the scene's path and with Poisson cameras (tools/probes/generate_routes_probe.sh
    does the same on a 32 x 32 block city, where the occlusion hierarchy is built on the worker thread)."""
    scene_obj = os.path.join(os.path.dirname(BOX), "test_scene.obj")
    for extra in (["--path", "path"], []):
        files = []
        for k, env in enumerate(({}, {"C2B_HOST_SAMPLER": "1"}, {"C2B_DENSE_SWEEP": "1"}, {"C2B_HOST_SAMPLER": "1", "C2B_DENSE_SWEEP": "1"})):
            out = tmp_path / ("r%d%d.bbal" % (len(extra), k))
            r = subprocess.run([cli, "generate", scene_obj, str(out), "--cameras", "120", "--points", "900", "--seed", "4"] + extra,
                               capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr
            files.append(out.read_bytes())
        assert len(files[0]) > 1000 and all(f == files[0] for f in files[1:])
