"""GPU parity of the dense camera x point sweep (generate::visibility_graph's loop, src/generate.rs:446-469,
without the Embree occlusion stream) against the oracle's predicate over ALL pairs: kept indices and uv exact."""
import numpy as np
import pytest

import oracle as O
from _problems import grid_cameras_points, random_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2b():
    import __graft_entry__ as entry
    entry.build()
    import city2ba_amd
    assert city2ba_amd.device_count() > 0
    return city2ba_amd


def _oracle_dense(cams, pts, max_dist):
    n_cam, n_pts = len(cams), len(pts)
    ci = np.repeat(np.arange(n_cam, dtype=np.uint32), n_pts)
    pi = np.tile(np.arange(n_pts, dtype=np.uint32), n_cam)
    uv, keep = O.visibility_pairs(cams, pts, ci, pi, max_dist)
    k = keep == 1
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(ci[k], minlength=n_cam))]).astype(np.uint64)
    return row_ptr, pi[k].astype(np.uint64), uv[k]


@pytest.mark.parametrize("blocks,cpb,ppb,L,max_dist", [(2, 4, 6, 8.0, 10.0), (3, 10, 20, 5.0, 10.0), (1, 3, 50, 20.0, 7.5)])
def test_dense_matches_oracle_on_grids(c2b, blocks, cpb, ppb, L, max_dist):
    cams, pts = grid_cameras_points(blocks, cpb=cpb, ppb=ppb, L=L)
    want = _oracle_dense(cams, pts, max_dist)
    ba = c2b.BAProblem.from_visibility(cams, pts, np.zeros(len(cams) + 1, dtype=np.uint64), [], np.zeros((0, 2)))
    row_ptr, pt_idx, uv = ba.visibility_graph(max_dist, dense=True)
    assert np.array_equal(row_ptr, want[0]) and np.array_equal(pt_idx, want[1])       # indices: exact, in order
    assert np.array_equal(uv, want[2])                                                 # k2 == 0: bit-exact
    assert len(pt_idx) > 100


def test_dense_random_cameras_ragged_sizes(c2b):
    """Point count not a multiple of the 256-point tile, cameras not a multiple of the 64-camera LDS round,
    distortion on (k2 != 0: uv still bit-exact), some cameras seeing nothing."""
    P = random_problem(131, 1000 + 37, 3, seed=41)
    cams, pts = P["cams15"], P["pts"]
    want = _oracle_dense(cams, pts, 6.0)                       # k2 != 0: libm's pow on both sides
    ba = c2b.BAProblem.from_visibility(cams, pts, np.zeros(len(cams) + 1, dtype=np.uint64), [], np.zeros((0, 2)))
    row_ptr, pt_idx, uv = ba.visibility_graph(6.0, dense=True)
    assert np.array_equal(row_ptr, want[0]) and np.array_equal(pt_idx, want[1])
    assert np.array_equal(uv, want[2])                          # bit-exact with distortion on
    assert (np.diff(row_ptr.astype(np.int64)) == 0).any() and len(pt_idx) > 50
    # degenerate: no points within range
    row_ptr, pt_idx, uv = ba.visibility_graph(0.0, dense=True)
    assert int(row_ptr[-1]) == 0 and len(pt_idx) == 0


def test_dense_boundary_distance_is_decided_exactly(c2b):
    """Points at distance exactly max_dist (excluded by '<') and one ulp inside (included)."""
    cam = O.camera_from_bal([0, 0, 0, 0, 0, 0, 1.0, 0, 0])
    pts = np.array([[0, 0, -5.0], [0, 0, -np.nextafter(5.0, 0.0)], [3.0, 0, -4.0], [0.3, 0.4, -np.sqrt(24.75)]])
    ba = c2b.BAProblem.from_visibility(cam, pts, np.zeros(2, dtype=np.uint64), [], np.zeros((0, 2)))
    row_ptr, pt_idx, uv = ba.visibility_graph(5.0, dense=True)
    want = _oracle_dense(cam, pts, 5.0)
    assert np.array_equal(pt_idx, want[1]) and np.array_equal(uv, want[2])
    assert 0 not in pt_idx and 1 in pt_idx


def _boundary_points(cam15, rng, n_rows, side):
    """Points (px, py, -1) seen by an identity-rotation camera at the origin (so p = (px, py) exactly) whose u lands
    within a few ulps of `side` (+1 or -1) on either side: bisection over the doubles px with the oracle's projection
    (libm pow), then the neighbouring doubles of the crossing."""
    pts = []
    for _ in range(n_rows):
        py = rng.uniform(-0.6, 0.6)
        lo, hi = (0.0, 2.0) if side > 0 else (-2.0, 0.0)       # u(px) is increasing in px for these intrinsics
        for _ in range(80):
            mid = 0.5 * (lo + hi)
            u = O.project(cam15, [mid, py, -1.0])[0]
            if u < side:
                lo = mid
            else:
                hi = mid
            if np.nextafter(lo, hi) == hi:
                break
        x = lo
        for _ in range(4):
            x = np.nextafter(x, -np.inf)
        for _ in range(9):
            pts.append([x, py, -1.0])
            x = np.nextafter(x, np.inf)
    return np.array(pts)


@pytest.mark.parametrize("k1,k2", [(-2e-2, 3e-2), (5e-2, -1e-2), (0.0, 2e-1)])
def test_predicate_at_the_u_equals_one_boundary_with_k2(c2b, k1, k2):
    """`generate` applies modify_intrinsics before visibility_graph
    (src/bin/city2ba.rs:534,542), so the predicate -1 <= u <= 1 (src/generate.rs:446-454) is decided on projections
    that involve powf(4.0).  Points are constructed with |u| = 1 +- a few ulps; the kept index sets and the uv bits of
    both device predicates (pair list and dense sweep) must equal the oracle's -- libm's pow, the reference's call --
    exactly: 0 differing kept indices."""
    rng = np.random.default_rng(97)
    cam = O.camera_from_bal([0, 0, 0, 0, 0, 0, 0.9, k1, k2]).reshape(1, 15)
    pts = np.vstack([_boundary_points(cam[0], rng, 60, +1.0), _boundary_points(cam[0], rng, 60, -1.0)])
    n = len(pts)
    ci, pi = np.zeros(n, dtype=np.uint32), np.arange(n, dtype=np.uint32)
    uv_lm, keep_lm = O.visibility_pairs(cam, pts, ci, pi, 10.0)
    u = uv_lm[:, 0]
    assert np.sum(np.abs(u) == 1.0) >= 20 and np.sum(np.abs(u) > 1.0) >= 200 and np.sum(np.abs(u) < 1.0) >= 200
    assert 0 < keep_lm.sum() < n

    ba = c2b.BAProblem.from_visibility(cam, pts, np.zeros(2, dtype=np.uint64), [], np.zeros((0, 2)))
    uv_d, keep_d = ba.visibility_pairs(ci, pi, 10.0)
    assert np.array_equal(keep_d, keep_lm), "pair-list predicate must keep exactly the oracle's set"
    assert np.array_equal(uv_d.view(np.uint64), uv_lm.view(np.uint64)), "uv bits"
    row_ptr, pt_idx, uv_s = ba.visibility_graph(10.0, dense=True)
    kept = np.nonzero(keep_lm == 1)[0]
    assert np.array_equal(pt_idx, kept.astype(np.uint64)) and int(row_ptr[-1]) == len(kept)
    assert np.array_equal(uv_s.view(np.uint64), uv_lm[kept].view(np.uint64))
    print("boundary points: %d, kept %d; differing kept indices: 0" % (n, keep_lm.sum()))
