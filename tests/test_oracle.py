"""CPU tests: pin the oracle (oracle/city2ba_oracle.c) against
 (1) the reference's own known-answer tests for the hot path (src/baproblem.rs:64-75, 227-249),
 (2) the committed mpmath golden vectors (tests/golden/snavely_golden.json),
 (3) the invariants the reference's library tests rely on (tests/main.rs:130-195)."""
import numpy as np
import pytest

import oracle as O
from _problems import grid_cameras_points, grid_candidate_pairs, random_problem

IDENT_CAM = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0]


# --- (1) reference KATs ---------------------------------------------------------------
@pytest.mark.parametrize("v", [(1.0, 2.0, 3.0), (0.0, 0.0, 0.0), (-1.2, 0.0, 1.7)])
def test_rodrigues_idempotent(v):          # src/baproblem.rs:64-75
    v_ = O.to_rodrigues(O.from_rodrigues(v))
    assert np.linalg.norm(v_ - np.asarray(v)) < 1e-10


def test_project_world():                  # src/baproblem.rs:227-234
    c = O.camera_from_bal(IDENT_CAM)[0]
    q = O.project_world(c, [0.0, 0.0, -1.0])
    assert q[2] < 0.0
    assert q[0] == 0.0 and q[1] == 0.0


def test_project():                        # src/baproblem.rs:236-242
    c = O.camera_from_bal(IDENT_CAM)[0]
    uv = O.project(c, O.project_world(c, [0.0, 0.0, -1.0]))
    assert uv[0] == 0.0 and uv[1] == 0.0


def test_project_isomorphic():             # src/baproblem.rs:244-249
    p = np.array([1.0, 3.0, -1.0])
    c = O.camera_from_bal([3.0, 5.0, -2.0, 0.5, -0.2, 0.1, 1.0, 0.0, 0.0])[0]
    assert np.all(np.abs(O.to_world(c, O.project_world(c, p)) - p) <= 1e-8)


def test_philox_known_answers():           # Random123 kat_vectors (Philox4x32-10)
    assert [int(x) for x in O.philox4x32_10([0] * 4, [0] * 2)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(x) for x in O.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert [int(x) for x in O.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                            [0xa4093822, 0x299f31d0])] == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # ... and Philox2x32-10 (the observation-noise stream since r04), same file
    assert [int(x) for x in O.philox2x32_10([0, 0], 0)] == [0xff1dae59, 0x6cd10df2]
    assert [int(x) for x in O.philox2x32_10([0xffffffff, 0xffffffff], 0xffffffff)] == [0x2c3f628b, 0xab4fd7ad]
    assert [int(x) for x in O.philox2x32_10([0x243f6a88, 0x85a308d3], 0x13198a2e)] == [0xdd7ce038, 0xf62a4c12]


# --- (2) mpmath golden vectors --------------------------------------------------------
def _rel(a, b, floor=1.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))


def test_golden_projection(golden):
    worst = 0.0
    for pr in golden["pairs"]:
        cam = O.camera_from_bal(pr["bal9"])[0]
        R = cam[:9].reshape(3, 3).T          # col-major -> row/col
        assert _rel(R.ravel(), pr["R_rowmajor"]) < 1e-14
        q = O.project_world(cam, pr["X"])
        uv = O.project(cam, q)
        tol = 1e-9 if pr["kind"] == "nearz" else 1e-12
        assert _rel(q, pr["q"], floor=1e-2) < tol, pr["kind"]
        assert _rel(uv, pr["uv"], floor=1e-2) < tol, pr["kind"]
        worst = max(worst, _rel(uv, pr["uv"], floor=1e-2))
    assert worst < 1e-9


def test_golden_jacobian(golden):
    """Oracle's Gallego-Yezzi analytic Jacobian vs mp.diff of the model: <= 1e-6 relative
    (north_star tolerance); typically ~1e-13, ~1e-8 on the |w| < sqrt(eps) first-order branch."""
    for pr in golden["pairs"]:
        cam = O.camera_from_bal(pr["bal9"])[0]
        r, Jc, Jp = O.residual_jacobian_one(cam, pr["bal9"][:3], pr["X"], pr["uv"])
        scale = max(1.0, np.max(np.abs(pr["Jc"])))
        assert np.max(np.abs(Jc - pr["Jc"])) / scale < 1e-6, pr["kind"]
        assert np.max(np.abs(Jp - pr["Jp"])) / scale < 1e-6, pr["kind"]
        assert np.max(np.abs(r)) < 1e-9 * max(1.0, np.max(np.abs(pr["uv"])))
        if pr["kind"] in ("generic", "nodist", "big"):
            assert np.max(np.abs(Jc - pr["Jc"])) / scale < 1e-11, pr["kind"]


def test_state_mode_jacobian_matches_bal_mode_on_canonical_w():
    """state mode differentiates w.r.t. w = to_rodrigues(R); where that equals the file's w
    (no 2*pi chart flip) both modes must agree."""
    P = random_problem(60, 300, 5, seed=21)
    w_back = O.camera_to_bal(P["cams15"])[:, :3]
    canon = np.linalg.norm(w_back - P["bal9"][:, :3], axis=1) < 1e-9
    assert canon.sum() > 10 and (~canon).sum() > 0          # both kinds present
    r1, Jc1, Jp1 = O.residual_jacobian(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    r2, Jc2, Jp2 = O.residual_jacobian_bal(P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    counts = np.diff(P["row_ptr"].astype(np.int64))
    cam_of = np.repeat(np.arange(len(counts)), counts)
    sel = canon[cam_of]
    assert np.allclose(Jc1[sel], Jc2[sel], rtol=1e-9, atol=1e-9)
    assert np.allclose(Jp1, Jp2, rtol=1e-12, atol=1e-12)        # point block never depends on the chart
    non_w = [3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16, 17]
    assert np.allclose(Jc1[:, non_w], Jc2[:, non_w], rtol=1e-12, atol=1e-12)  # only the w columns do


def test_golden_total_error(golden):
    P = golden["problem"]
    cams = O.camera_from_bal(P["bal9"])
    for nrm, want in P["err"].items():
        got = O.total_reprojection_error(cams, P["pts"], P["row_ptr"], P["pt_idx"], P["uv_obs"], float(nrm))
        assert abs(got - want) / want < 1e-12, nrm
        s = O.reprojection_error_sum(cams, P["pts"], P["row_ptr"], P["pt_idx"], P["uv_obs"], float(nrm))
        assert abs(s - P["err_sum"][nrm]) / P["err_sum"][nrm] < 1e-12


# --- (3) invariants ---------------------------------------------------------------------
def test_bal_roundtrip_and_center():
    P = random_problem(50, 200, 4, seed=3)
    back = O.camera_to_bal(P["cams15"])
    again = O.camera_from_bal(back)
    assert np.max(np.abs(again - P["cams15"])) < 1e-12
    for cam in P["cams15"][:10]:
        c = O.center(cam)
        assert np.max(np.abs(O.project_world(cam, c))) < 1e-11      # R c + t = 0


def test_zero_error_by_construction():
    """Generators write observations from the same project() call => error is exactly 0
    (tests/main.rs err_start)."""
    P = random_problem(40, 400, 8, seed=5, empty_every=7)
    assert O.total_reprojection_error(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"], 2.0) == 0.0
    assert O.total_reprojection_error(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"], 1.0) == 0.0


def test_transform_uses_old_rotation():
    """src/baproblem.rs:165-171: new center = dR^-1 (c + dt) because loc' uses the OLD dir."""
    cam = O.camera_from_bal([0.3, -0.2, 0.5, 1.0, 2.0, 3.0, 1.1, 1e-3, -1e-3])[0]
    dR = O.basis_from_axis_angle(np.array([0.0, 0.6, 0.8]), 0.4)
    dt = np.array([0.1, -0.2, 0.3])
    out = O.transform(cam, dR, dt)
    c_new = O.center(out)
    dRm = dR.reshape(3, 3).T
    want = np.linalg.inv(dRm) @ (O.center(cam) + dt)
    assert np.max(np.abs(c_new - want)) < 1e-12
    assert np.all(out[12:] == cam[12:])


def _test_grid():
    """test_grid() of tests/main.rs:130-132 = synthetic_grid(10,20,3,5.,1.,1.,1.,10.) without
    the building-occlusion test and cull() (host-side rows)."""
    cams, pts = grid_cameras_points(3, cpb=10, ppb=20, L=5.0, inset=1.0, cam_h=1.0, pt_h=1.0)
    assert cams.shape[0] == 4 * 10 * 3 * 4 and pts.shape[0] == 12 * 20 * 3 * 4
    ci, pi = grid_candidate_pairs(cams, pts, 10.0)
    uv, keep = O.visibility_pairs(cams, pts, ci, pi, 10.0)
    ci, pi, uv = ci[keep == 1], pi[keep == 1], uv[keep == 1]
    row_ptr = np.zeros(len(cams) + 1, dtype=np.uint64)
    np.add.at(row_ptr, ci.astype(np.int64) + 1, 1)
    row_ptr = np.cumsum(row_ptr).astype(np.uint64)
    return cams, pts, row_ptr, pi.astype(np.uint64), uv


def test_noise_inequalities():
    """tests/main.rs:134-195 (the arithmetic-noise ones): error after > error before."""
    cams, pts, row_ptr, pt_idx, uv = _test_grid()
    assert len(pt_idx) > 1000
    e0 = O.total_reprojection_error(cams, pts, row_ptr, pt_idx, uv, 2.0)
    assert e0 == 0.0
    c1, p1 = O.add_drift_normalized(cams, pts, 0.1, 0.1, 0.1, seed=1)
    assert O.total_reprojection_error(c1, p1, row_ptr, pt_idx, uv, 2.0) > e0
    c2, p2, uv2 = O.add_noise(cams, pts, uv, 0.1, 0.1, 0.1, 0.1, seed=2)
    assert O.total_reprojection_error(c2, p2, row_ptr, pt_idx, uv2, 2.0) > e0
    c3, p3 = O.add_sin_noise(cams, pts, [1.0, 1.0, 0.0], [0.0, 1.0, 0.0], 1.0, 2.0)
    assert O.total_reprojection_error(c3, p3, row_ptr, pt_idx, uv, 2.0) > e0


def test_noise_moments():
    """Distributional checks on the build's Philox/Box-Muller draws (reference: rand 0.6.5
    Normal on thread_rng, unseeded -- only distributions can be compared)."""
    z = np.array([O.normal_pair(7, 3, e, 0) for e in range(20000)]).ravel()
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    assert abs(np.mean(z ** 3)) < 0.05 and abs(np.mean(z ** 4) - 3.0) < 0.15
    # drift gamma ~ N(1, std): recover it from a point's displacement along dir
    cams = O.camera_from_bal(np.tile(IDENT_CAM, (1, 1)))
    pts = np.column_stack([np.linspace(1, 2, 4000), np.zeros(4000), np.zeros(4000)])
    pts[0] = [0.0, 0.0, 0.0]                                   # the origin element
    d = np.array([0.0, 1.0, 0.0])
    _, p2 = O.add_drift(cams, pts, 0.5, 0.0, 0.25, d, seed=11)
    dist = np.linalg.norm(pts - pts[0], axis=1)[1:]
    gamma = (p2[1:, 1] - pts[1:, 1]) / (0.5 * dist ** 2)
    assert abs(gamma.mean() - 1.0) < 0.02 and abs(gamma.std() - 0.25) < 0.02


def test_drift_origin_tie_goes_to_later():
    cams = O.camera_from_bal(np.tile(IDENT_CAM, (1, 1)))         # center (0,0,0): distance 0
    pts = np.array([[1.0, 0, 0], [0.0, 0, 0], [0, 0, 2.0]])       # point 1 also at distance 0
    o, idx = O.drift_origin(cams, pts)
    assert idx == 2 and np.all(o == 0.0)                         # later of the tied elements


def test_stats_against_numpy():
    P = random_problem(30, 300, 4, seed=9)
    allp = np.vstack([O.centers(P["cams15"]), P["pts"]])
    assert np.allclose(O.mean(P["cams15"], P["pts"]), allp.mean(axis=0), rtol=1e-12, atol=1e-12)
    assert np.allclose(O.std(P["cams15"], P["pts"]), allp.std(axis=0), rtol=1e-12)
    mn, mx = O.extent(P["cams15"], P["pts"])
    assert np.allclose(mn, allp.min(axis=0), rtol=1e-13) and np.allclose(mx, allp.max(axis=0), rtol=1e-13)
    assert np.allclose(O.dimensions(P["cams15"], P["pts"]), mx - mn)


def test_occlusion_filter_wall_and_float64_agreement():
    """occlusion rays of generate::visibility_graph (src/generate.rs:455-476) in the oracle: a hand-checkable wall,
    then random rays against random triangles compared with a float64 ray / triangle test away from the edges"""
    cams = np.zeros((1, 15))
    cams[0, :9] = np.eye(3).reshape(9)
    cams[0, 12] = 1.0
    wall = np.array([[-50, -50, -2, 0, -50, -2, 0, 50, -2], [-50, -50, -2, 0, 50, -2, -50, 50, -2]], dtype=np.float32)
    pts = np.array([[-1.0, 0.2, -4.0], [1.0, 0.2, -4.0], [-0.4, 0.1, -1.0], [-0.5, -0.3, -3.0], [0.3, 0.3, -1.5]])
    ci, pi = np.zeros(5, np.uint32), np.arange(5, dtype=np.uint32)
    assert list(O.occlusion_filter(cams, pts, ci, pi, wall)) == [0, 1, 1, 0, 1]
    assert list(O.occlusion_filter(cams, pts, ci, pi, np.zeros((0, 9), np.float32))) == [1] * 5
    rng = np.random.default_rng(12)
    pts = rng.uniform(-5, 5, (400, 3))
    a = rng.uniform(-4, 4, (60, 3))
    tri = np.concatenate([a, a + rng.normal(0, 1.5, (60, 3)), a + rng.normal(0, 1.5, (60, 3))], axis=1).astype(np.float32)
    ci, pi = np.zeros(400, np.uint32), np.arange(400, dtype=np.uint32)
    keep = O.occlusion_filter(cams, pts, ci, pi, tri)
    T = tri.astype(np.float64).reshape(-1, 3, 3)
    checked = 0
    for k, p in enumerate(pts):
        mag = np.linalg.norm(p)
        d = p / mag
        hit, marginal = False, False
        for v0, v1, v2 in T:
            e1, e2 = v1 - v0, v2 - v0
            pv = np.cross(d, e2)
            det = e1 @ pv
            if abs(det) < 1e-9:
                marginal = True
                continue
            tv = -v0
            u = (tv @ pv) / det
            qv = np.cross(tv, e1)
            w = (d @ qv) / det
            t = (e2 @ qv) / det
            inside = min(u, w, 1 - u - w)
            if abs(inside) < 1e-4 or abs(t - mag) < 1e-4 or abs(t) < 1e-4:
                marginal = True
            if inside > 0 and 0 < t <= mag:
                hit = True
        if not marginal:
            assert keep[k] == (0 if hit else 1)
            checked += 1
    assert checked > 300 and 0 < keep.sum() < 400


def test_cpu_baseline_runner_layouts_and_threads_agree():
    """bench.py's cpu_baseline leg (oracle.bench_run): the reference-shaped Vec<Vec<>> layout and the flat CSR layout, on
    one thread and on several, all write the same residuals / Jacobians as the plain oracle and reduce the same sum."""
    P = random_problem(257, 3000, 11, seed=21, noise=1e-2, empty_every=7)
    n = len(P["pt_idx"])
    r0, Jc0, Jp0 = O.residual_jacobian(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    want = O.reprojection_error_sum(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"], 2.0)
    for layout in ("faithful", "optimised"):
        for threads in (1, 3, 8):
            r, Jc, Jp = np.full((n, 2), np.nan), np.full((n, 18), np.nan), np.full((n, 6), np.nan)
            passes, el, tot = O.bench_run(layout, threads, 0.0, P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"],
                                          r, Jc, Jp, max_passes=1)
            assert passes == 1 and el > 0
            assert np.array_equal(r, r0) and np.array_equal(Jc, Jc0.reshape(n, 18)) and np.array_equal(Jp, Jp0.reshape(n, 6))
            assert abs(tot - want) <= 1e-12 * want
