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
    row_ptr, pt_idx, uv = ba.visibility_graph(max_dist)
    assert np.array_equal(row_ptr, want[0]) and np.array_equal(pt_idx, want[1])       # indices: exact, in order
    assert np.array_equal(uv, want[2])                                                 # k2 == 0: bit-exact
    assert len(pt_idx) > 100


def test_dense_random_cameras_ragged_sizes(c2b):
    """Point count not a multiple of the 256-point tile, cameras not a multiple of the 64-camera LDS round,
    distortion on (uv within 1e-13), some cameras seeing nothing."""
    P = random_problem(131, 1000 + 37, 3, seed=41)
    cams, pts = P["cams15"], P["pts"]
    want = _oracle_dense(cams, pts, 6.0)
    ba = c2b.BAProblem.from_visibility(cams, pts, np.zeros(len(cams) + 1, dtype=np.uint64), [], np.zeros((0, 2)))
    row_ptr, pt_idx, uv = ba.visibility_graph(6.0)
    assert np.array_equal(row_ptr, want[0]) and np.array_equal(pt_idx, want[1])
    assert np.max(np.abs(uv - want[2])) < 1e-13
    assert (np.diff(row_ptr.astype(np.int64)) == 0).any() and len(pt_idx) > 50
    # degenerate: no points within range
    row_ptr, pt_idx, uv = ba.visibility_graph(0.0)
    assert int(row_ptr[-1]) == 0 and len(pt_idx) == 0


def test_dense_boundary_distance_is_decided_exactly(c2b):
    """Points at distance exactly max_dist (excluded by '<') and one ulp inside (included)."""
    cam = O.camera_from_bal([0, 0, 0, 0, 0, 0, 1.0, 0, 0])
    pts = np.array([[0, 0, -5.0], [0, 0, -np.nextafter(5.0, 0.0)], [3.0, 0, -4.0], [0.3, 0.4, -np.sqrt(24.75)]])
    ba = c2b.BAProblem.from_visibility(cam, pts, np.zeros(2, dtype=np.uint64), [], np.zeros((0, 2)))
    row_ptr, pt_idx, uv = ba.visibility_graph(5.0)
    want = _oracle_dense(cam, pts, 5.0)
    assert np.array_equal(pt_idx, want[1]) and np.array_equal(uv, want[2])
    assert 0 not in pt_idx and 1 in pt_idx
