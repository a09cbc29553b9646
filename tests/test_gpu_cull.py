"""BAProblem::cull on the device (c2b_problem_cull: union-find components, singleton counts, scan renumbering) against
the host implementation (c2b_cull, itself checked against a pure-Python restatement of src/baproblem.rs:392-550 in
tests/test_host_rows.py): identical cameras, points, graph and observations, for both settings of the reference's
observation-filter quirk."""
import numpy as np
import pytest

from _problems import grid_cameras_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2b():
    import __graft_entry__ as entry
    entry.build()
    import city2ba_amd
    assert city2ba_amd.device_count() > 0
    return city2ba_amd


def random_graph(seed, n_cam, n_pts, lo, hi, clusters, spread):
    """cameras see points near their own cluster: several components, isolated points, weak cameras"""
    rng = np.random.default_rng(seed)
    deg = rng.integers(lo, hi, n_cam)
    deg[rng.random(n_cam) < 0.1] = 0                                        # cameras without observations
    home = rng.integers(0, clusters, n_cam)
    rows, cols = [], []
    per = n_pts // clusters
    for c in range(n_cam):
        lo_p = home[c] * per
        span = min(per, spread)
        start = lo_p + rng.integers(0, max(1, per - span))
        cols.append(np.sort(rng.choice(np.arange(start, start + span), size=min(deg[c], span), replace=False)))
        rows.append(len(cols[-1]))
    row_ptr = np.concatenate([[0], np.cumsum(rows)]).astype(np.uint64)
    pt_idx = np.concatenate(cols).astype(np.uint64) if len(cols) else np.zeros(0, np.uint64)
    uv = rng.uniform(-1, 1, (len(pt_idx), 2))
    cams = rng.uniform(-1, 1, (n_cam, 15))
    cams[:, :9] = np.eye(3).reshape(9)
    pts = rng.uniform(-5, 5, (n_pts, 3))
    return cams, pts, row_ptr, pt_idx, uv


def check(c2b, cams, pts, row_ptr, pt_idx, uv, faithful):
    from city2ba_amd.baproblem import cull_arrays
    want = cull_arrays(cams, pts, row_ptr, pt_idx, uv, faithful)
    ba = c2b.BAProblem.from_visibility(cams, pts, row_ptr, pt_idx, uv)
    out = ba.cull(faithful)
    assert out is ba
    assert (ba.num_cameras(), ba.num_points(), ba.num_observations()) == (len(want[0]), len(want[1]), len(want[3]))
    assert np.array_equal(ba.row_ptr, want[2]) and np.array_equal(ba.pt_idx, want[3])
    assert np.array_equal(ba.cameras(), want[0]) and np.array_equal(ba.points(), want[1])
    assert np.array_equal(ba.observations(), want[4])
    return ba


@pytest.mark.parametrize("faithful", [True, False])
@pytest.mark.parametrize("seed,n_cam,n_pts,lo,hi,clusters,spread", [
    (1, 300, 2000, 0, 12, 1, 400),          # one loose component, many isolated points
    (2, 500, 3000, 2, 10, 5, 300),          # five components of similar size: the tie-break and the quirk matter
    (3, 2000, 20000, 0, 30, 3, 2000),
    (4, 50, 100, 0, 4, 2, 20),              # almost everything is a singleton: may cull to nothing
])
def test_device_cull_equals_host_cull(c2b, seed, n_cam, n_pts, lo, hi, clusters, spread, faithful):
    g = random_graph(seed, n_cam, n_pts, lo, hi, clusters, spread)
    check(c2b, *g, faithful)


def test_device_cull_on_the_grid_and_degenerate_graphs(c2b):
    from city2ba_amd import synthetic as S
    ba = S.synthetic_grid(10, 20, 3, 5.0, 1.0, 1.0, 1.0, 10.0, False, cull=False)
    g = (ba.cameras(), ba.points(), ba.row_ptr.copy(), ba.pt_idx.copy(), ba.observations())
    for faithful in (True, False):
        out = check(c2b, *g, faithful)
        assert out.num_cameras() > 100 and out.total_reprojection_error(2.0) == 0.0
        # a culled problem is a fixed point
        n = (out.num_cameras(), out.num_points(), out.num_observations())
        out.cull(faithful)
        assert (out.num_cameras(), out.num_points(), out.num_observations()) == n
    # no observations at all; no cameras at all
    cams, pts = grid_cameras_points(1, cpb=2, ppb=2, L=5.0)
    empty = c2b.BAProblem.from_visibility(cams, pts, np.zeros(len(cams) + 1, np.uint64), [], np.zeros((0, 2)))
    empty.cull()
    assert (empty.num_cameras(), empty.num_points(), empty.num_observations()) == (0, 0, 0)
    none = c2b.BAProblem.from_visibility(np.zeros((0, 15)), pts, np.zeros(1, np.uint64), [], np.zeros((0, 2)))
    none.cull()
    assert none.num_cameras() == 0
    # the 9-vector form survives the cull (columns of the Jacobian keep referring to the file's own w)
    bal = c2b.BAProblem.from_bal(np.tile([0.1, 0.2, -0.1, 0, 0, 0, 1.0, 0, 0], (len(g[0]), 1)), g[1], g[2], g[3], g[4])
    before = bal.cameras_bal()
    bal.cull()
    assert np.all(bal.cameras_bal() == before[0])


def test_device_cull_property_small_graphs(c2b):
    """hypothesis over small graphs (components, ties and the faithful filter's index aliasing are frequent there):
    device == host, element for element"""
    from hypothesis import given, settings
    from hypothesis import strategies as st
    from city2ba_amd.baproblem import cull_arrays

    graphs = st.integers(1, 9).flatmap(lambda nc: st.integers(1, 12).flatmap(lambda npt: st.tuples(
        st.just(nc), st.just(npt),
        st.lists(st.lists(st.integers(0, npt - 1), max_size=7, unique=True), min_size=nc, max_size=nc), st.booleans())))

    @settings(max_examples=150, deadline=None)
    @given(graphs)
    def run(g):
        n_cam, n_pts, obs, faithful = g
        row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in obs])]).astype(np.uint64)
        pt_idx = np.array([p for r in obs for p in r], dtype=np.uint64)
        uv = (np.arange(2 * len(pt_idx), dtype=np.float64).reshape(-1, 2) + 0.25)
        cams = np.arange(n_cam * 15, dtype=np.float64).reshape(n_cam, 15)
        pts = np.arange(n_pts * 3, dtype=np.float64).reshape(n_pts, 3) + 0.5
        want = cull_arrays(cams, pts, row_ptr, pt_idx, uv, faithful)
        ba = c2b.BAProblem.from_visibility(cams, pts, row_ptr, pt_idx, uv)
        ba.cull(faithful)
        assert np.array_equal(ba.row_ptr, want[2]) and np.array_equal(ba.pt_idx, want[3])
        assert np.array_equal(ba.cameras(), want[0].reshape(-1, 15)) and np.array_equal(ba.points(), want[1].reshape(-1, 3))
        assert np.array_equal(ba.observations(), want[4].reshape(-1, 2))
        ba.close()

    run()


def test_single_steps_subset_and_camera_accessors(c2b):
    """the reference's separate public methods: largest_connected_component, remove_singletons, subset
    (src/baproblem.rs:394-534) and SnavelyCamera's accessors (:178-225)"""
    from city2ba_amd.baproblem import cull_arrays
    from city2ba_amd import camera as K
    g = random_graph(7, 400, 2500, 0, 12, 4, 300)
    for faithful in (True, False):
        want = cull_arrays(*g, faithful, step="lcc")
        ba = c2b.BAProblem.from_visibility(*g).largest_connected_component(faithful)
        assert np.array_equal(ba.row_ptr, want[2]) and np.array_equal(ba.pt_idx, want[3])
        assert np.array_equal(ba.cameras(), want[0]) and np.array_equal(ba.points(), want[1])
        assert np.array_equal(ba.observations(), want[4])
    want = cull_arrays(*g, step="singletons")
    ba = c2b.BAProblem.from_visibility(*g).remove_singletons()
    assert np.array_equal(ba.row_ptr, want[2]) and np.array_equal(ba.pt_idx, want[3]) and np.array_equal(ba.cameras(), want[0])
    # subset: cameras / points in the GIVEN order, observations of dropped points disappear
    cams, pts, row_ptr, pt_idx, uv = g
    full = c2b.BAProblem.from_visibility(*g)
    ci, pi = np.array([5, 2, 300, 17]), np.array([9, 3, 4, 2000, 1999, 8])
    sub = full.subset(ci, pi)
    assert np.array_equal(sub.cameras(), cams[ci]) and np.array_equal(sub.points(), pts[pi])
    new_of = {int(p): k for k, p in enumerate(pi)}
    rows = []
    for c in ci:
        a, b = int(row_ptr[c]), int(row_ptr[c + 1])
        rows.append([(new_of[int(p)], tuple(v)) for p, v in zip(pt_idx[a:b], uv[a:b]) if int(p) in new_of])
    assert list(sub.row_ptr) == list(np.concatenate([[0], np.cumsum([len(r) for r in rows])]))
    assert list(sub.pt_idx) == [p for r in rows for (p, _) in r]
    assert [tuple(v) for v in sub.observations()] == [v for r in rows for (_, v) in r]
    with pytest.raises(c2b.City2baError):
        full.subset([len(cams)], [0])
    # camera records: from_vec / to_vec round trip and the accessors
    bal = np.array([[0.1, -0.2, 0.3, 1.0, 2.0, 3.0, 1.5, 0.01, -0.002], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0]])
    rec = K.from_vec(bal)
    assert rec.shape == (2, 15) and np.allclose(K.to_vec(rec), bal, atol=1e-12)
    R = K.rotation(rec)
    assert np.allclose(R @ np.swapaxes(R, -1, -2), np.eye(3), atol=1e-12) and np.allclose(R[1], np.eye(3))
    w = bal[0, :3]
    th = np.linalg.norm(w)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    assert np.allclose(R[0], np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx, atol=1e-12)     # Rodrigues
    assert np.array_equal(K.focal_length(rec), [1.5, 1.0]) and np.array_equal(K.distortion(rec)[0], [0.01, 0.0])
    mod = K.modify_intrin(rec, [0.5, 0.1, 0.2])
    assert np.array_equal(mod[:, :12], rec[:, :12]) and np.allclose(mod[:, 12:], rec[:, 12:] + [0.5, 0.1, 0.2])
