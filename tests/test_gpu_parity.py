"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(city2ba_amd -> ctypes -> libcity2ba_hip.so), against the CPU oracle on the same seeded inputs and
against the committed golden vectors.

Tolerances (north_star: indices bit-exact, f64 <= 1e-6 relative):
  * observation indices / keep masks: exact;
  * projection: BIT-EXACT for every k2 -- the same IEEE operations in the same order, and |p|^4 =
    p.magnitude().powf(4.0) by a restatement of glibc's pow (csrc/pow4_libm.hpp) that returns libm's bits, including the
    ~1e-3 of arguments where libm's pow is not correctly rounded;
  * Jacobian, error sums, stats: <= 1e-6 relative required, ~1e-12 asserted;
  * noise: same Philox draws; the device's log / sin / cos (fdlibm kernels on the draws' domains, camera_math.hpp)
    differ from glibc by ulps -> 1e-9 on results, 1e-13 on the raw observation draws.
"""
import numpy as np
import pytest

import oracle as O
from _problems import grid_cameras_points, grid_candidate_pairs, random_problem

pytestmark = pytest.mark.gpu

IDENT_CAM = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0]


@pytest.fixture(scope="module")
def c2b():
    import __graft_entry__ as entry
    entry.build()
    import city2ba_amd
    assert city2ba_amd.device_count() > 0, "no HIP device: the gpu tests must run on the GPU box"
    return city2ba_amd


def _upload(c2b, P, bal=False):
    if bal:
        return c2b.BAProblem.from_bal(P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    return c2b.BAProblem.from_visibility(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])


def _relerr(a, b, floor=1.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


# ---------------------------------------------------------------------------------------------
# the reference's own KATs, through the GPU (src/baproblem.rs:64-75, 227-249)
# ---------------------------------------------------------------------------------------------
def test_reference_kats_on_gpu(c2b):
    pts = np.array([[0.0, 0.0, -1.0], [1.0, 3.0, -1.0]])
    bal = np.array([IDENT_CAM, [3.0, 5.0, -2.0, 0.5, -0.2, 0.1, 1.0, 0.0, 0.0],
                    [1.0, 2.0, 3.0, 0, 0, 0, 1, 0, 0], [-1.2, 0.0, 1.7, 0, 0, 0, 1, 0, 0]])
    row_ptr = np.array([0, 1, 2, 2, 2], dtype=np.uint64)
    ba = c2b.BAProblem.from_bal(bal, pts, row_ptr, np.array([0, 1], dtype=np.uint64), np.zeros((2, 2)))
    uv = ba.project()
    assert uv[0, 0] == 0.0 and uv[0, 1] == 0.0                    # test_project: exact
    cams = ba.cameras()
    assert np.all(cams[0, :9] == np.eye(3).ravel())               # w = 0 -> identity exactly
    # rodrigues_idempotent: to_rodrigues(from_rodrigues(v)) ~ v to 1e-10 (device trig)
    back = ba_bal_roundtrip(c2b, bal)
    for i in (0, 2, 3):             # the three vectors of rodrigues_idempotent
        assert np.linalg.norm(back[i, :3] - bal[i, :3]) < 1e-10
    # test_project_isomorphic: to_world(project_world(p)) ~ p to 1e-8, with the device's R
    q = cams[1, :9].reshape(3, 3).T @ pts[1] + cams[1, 9:12]
    assert np.all(np.abs(O.to_world(cams[1], q) - pts[1]) <= 1e-8)


def ba_bal_roundtrip(c2b, bal9):
    """from_vec on the device, then to_vec on the device (state path, not the cached 9-vector)."""
    n = len(bal9)
    ba = c2b.BAProblem.from_bal(bal9, np.zeros((1, 3)), np.zeros(n + 1, dtype=np.uint64), [], np.zeros((0, 2)))
    cams = ba.cameras()
    ba2 = c2b.BAProblem.from_visibility(cams, np.zeros((1, 3)), np.zeros(n + 1, dtype=np.uint64), [], np.zeros((0, 2)))
    return ba2.cameras_bal()


# ---------------------------------------------------------------------------------------------
# golden vectors (mpmath)
# ---------------------------------------------------------------------------------------------
def _golden_problem(golden):
    bal = np.array([p["bal9"] for p in golden["pairs"]])
    pts = np.array([p["X"] for p in golden["pairs"]])
    n = len(bal)
    row_ptr = np.arange(n + 1, dtype=np.uint64)
    return bal, pts, row_ptr, np.arange(n, dtype=np.uint64), np.array([p["uv"] for p in golden["pairs"]])


def test_golden_projection_and_jacobian(c2b, golden):
    bal, pts, row_ptr, pt_idx, uv = _golden_problem(golden)
    ba = c2b.BAProblem.from_bal(bal, pts, row_ptr, pt_idx, uv)
    got_uv = ba.project()
    r, Jc, Jp = ba.residual_jacobian()          # bal mode: columns w.r.t. the file's own w
    Jc = Jc.reshape(-1, 18); Jp = Jp.reshape(-1, 6)
    for i, pr in enumerate(golden["pairs"]):
        tol = 1e-9 if pr["kind"] == "nearz" else 1e-12
        assert _relerr(got_uv[i], pr["uv"], floor=1e-2) < tol, (i, pr["kind"])
        scale = max(1.0, np.max(np.abs(pr["Jc"])))
        assert np.max(np.abs(Jc[i] - pr["Jc"])) / scale < 1e-6, (i, pr["kind"])
        assert np.max(np.abs(Jp[i] - pr["Jp"])) / scale < 1e-6, (i, pr["kind"])
        if pr["kind"] != "nearz":
            assert np.max(np.abs(Jc[i] - pr["Jc"])) / scale < 1e-11, (i, pr["kind"])
            assert np.max(np.abs(Jp[i] - pr["Jp"])) / scale < 1e-11, (i, pr["kind"])
        assert np.max(np.abs(r[i])) < 1e-9 * max(1.0, np.max(np.abs(pr["uv"])))


def test_golden_total_error(c2b, golden):
    P = golden["problem"]
    ba = c2b.BAProblem.from_bal(P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv_obs"])
    for nrm, want in P["err"].items():
        got = ba.total_reprojection_error(float(nrm))
        assert abs(got - want) / want < 1e-12, nrm


# ---------------------------------------------------------------------------------------------
# oracle parity on seeded problems, incl. ragged / empty / tile-boundary sizes
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_cam,n_pts,opc,seed,empty_every", [
    (1, 3, 2, 0, 0), (7, 60, 5, 1, 3), (64, 900, 9, 2, 0), (300, 6000, 17, 3, 11), (40, 5000, 64, 4, 0),
])
def test_project_bit_exact_without_k2(c2b, n_cam, n_pts, opc, seed, empty_every):
    P = random_problem(n_cam, n_pts, opc, seed=seed, empty_every=empty_every)
    P["cams15"][:, 14] = 0.0                                   # k2 = 0 -> no pow() anywhere
    want = O.project_observations(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"])
    ba = _upload(c2b, P)
    got = ba.project()
    assert got.shape == want.shape
    assert np.array_equal(got, want), "projection must be bit-exact with the CPU oracle when k2 == 0"


@pytest.mark.parametrize("n_cam,n_pts,opc,seed,k_scale", [(200, 4000, 15, 7, 5e-2), (64, 900, 40, 8, 0.5), (1, 50, 50, 9, 1e-3)])
def test_project_with_k2_bit_exact_with_libm_pow(c2b, n_cam, n_pts, opc, seed, k_scale):
    """k2 != 0 brings in |p|^4 = p.magnitude().powf(4.0) (src/baproblem.rs:147-149) = libm's pow.  The device runs a
    restatement of glibc's pow (csrc/pow4_libm.hpp): bit-exact with the oracle, which calls the library itself."""
    P = random_problem(n_cam, n_pts, opc, seed=seed, k_scale=k_scale)
    assert np.all(P["cams15"][:, 14] != 0.0)
    got = _upload(c2b, P).project()
    want_libm = O.project_observations(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"])
    assert np.array_equal(got, want_libm), "projection must be bit-exact with the oracle (libm pow)"


def test_device_pow4_is_libm_pow_bit_for_bit_over_two_million_draws(c2b):
    """the device's x^4 against libm's pow(x, 4.0) on 2 M draws x = sqrt(n), n ~ U(0, 4) -- the
    domain tests/test_pow4.py measures libm's own mis-roundings on -- plus wide and extreme ranges.  The value is read
    out of a projection without rounding: identity rotation, t = 0, point (2^-3, y, -1) => p = (2^-3, y) exactly; k1 = 0,
    k2 = 2^s with k2 * x^4 >= 2^54 => rad = 1 + k2 x^4 = k2 x^4 exactly; f = 1 => u = rad * 2^-3 exactly."""
    rng = np.random.default_rng(61)
    n = rng.uniform(0.0, 4.0, 2000000)
    n = np.concatenate([n, np.exp2(rng.uniform(-40.0, 40.0, 200000))])
    px = 0.125
    y = np.sqrt(np.maximum(n - px * px, 0.0))
    nn = px * px + y * y                                           # what the device sums: px * px + py * py
    x = np.sqrt(nn)
    want, _ = O.pow4_both(x)                                       # libm's pow(x, 4.0)
    s = 70 - np.floor(np.log2(np.maximum(want, 1e-300))).astype(np.int64)      # k2 * x^4 in [2^70, 2^71)
    got = np.empty_like(want)
    pts = np.column_stack([np.full(len(y), px), y, -np.ones(len(y))])
    for sv in np.unique(s):                                        # one camera per power of two
        m = np.nonzero(s == sv)[0]
        cam = O.camera_from_bal([0, 0, 0, 0, 0, 0, 1.0, 0.0, float(np.ldexp(1.0, int(sv)))]).reshape(1, 15)
        ba = c2b.BAProblem.from_visibility(cam, pts[m], np.array([0, len(m)], dtype=np.uint64), np.arange(len(m), dtype=np.uint64), np.zeros((len(m), 2)))
        uv = ba.project()
        ba.close()
        got[m] = np.ldexp(uv[:, 0], 3 - int(sv))                   # u / 2^-3 / k2: exact
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), "%d of %d differ" % (np.sum(got != want), len(want))
    lm, cr = O.pow4_both(x[:2000000])
    assert np.sum(lm != cr) > 500                                  # ... and these draws do include libm's one-ulp cases


def test_device_pow4_takes_glibcs_large_argument_and_overflow_paths(c2b):
    """The rest of the routine that a projection can read out exactly: |p| from 2^100 to beyond 2^256, i.e. the logarithm's large k,
    exp_inline's `specialcase` for k > 0 (512 <= 4 ln x < 1024: the scale's exponent is lowered by 1009 and the result multiplied
    back) and the overflow exit (+inf).  Same read-out as above (k2 = 2^s down to 2^-954, a normal number); where x^4 overflows the
    pixel is +inf on both sides.  (Results in the subnormal range and subnormal arguments cannot be read out through 1 + k2 x^4;
    they are held bit for bit on the host, where the same header compiles: tests/test_pow4.py.)"""
    rng = np.random.default_rng(62)
    py = np.exp2(np.concatenate([rng.uniform(100.0, 257.5, 60000), rng.uniform(184.0, 256.0, 60000), rng.uniform(255.9, 256.1, 20000)]))
    px = 0.125
    nn = px * px + py * py                                         # = py * py here (2^-6 is far below its last place)
    x = np.sqrt(nn)
    want, _ = O.pow4_both(x)
    over = np.isinf(want)
    assert over.sum() > 5000 and (~over).sum() > 100000 and np.sum(want[~over] > 2.0 ** 1000) > 1000
    s = np.where(over, 0, 70 - np.floor(np.log2(np.where(over, 1.0, want))).astype(np.int64))
    got = np.empty_like(want)
    pts = np.column_stack([np.full(len(py), px), py, -np.ones(len(py))])
    for sv in np.unique(s):
        m = np.nonzero(s == sv)[0]
        cam = O.camera_from_bal([0, 0, 0, 0, 0, 0, 1.0, 0.0, float(np.ldexp(1.0, int(sv)))]).reshape(1, 15)
        ba = c2b.BAProblem.from_visibility(cam, pts[m], np.array([0, len(m)], dtype=np.uint64), np.arange(len(m), dtype=np.uint64), np.zeros((len(m), 2)))
        uv = ba.project()
        ba.close()
        got[m] = np.ldexp(uv[:, 0], 3 - int(sv))
        if sv == 0:                                                # the overflow group: both pixel coordinates against the oracle's
            assert np.array_equal(uv[:300].view(np.uint64), np.array([O.project(cam[0], q) for q in pts[m][:300]]).view(np.uint64))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), "%d of %d differ" % (np.sum(got != want), len(want))


@pytest.mark.parametrize("norm", [1.0, 2.0, 1.5, 3.0])
def test_total_reprojection_error(c2b, norm):
    P = random_problem(150, 3000, 13, seed=11, noise=1e-2, empty_every=9)
    want = O.total_reprojection_error(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"], norm)
    got = _upload(c2b, P).total_reprojection_error(norm)
    assert abs(got - want) / want < 1e-12


@pytest.mark.parametrize("norm", [0.5, 1.25, 4.0, 7.3])
def test_total_reprojection_error_unusual_norms(c2b, norm):
    """norms other than 1 and 2 go through the table logarithm / exponential (camera_math.hpp: pow_tab, r05); residuals that
    are exactly zero (their power is 0, not NaN) and tiny ones included"""
    P = random_problem(120, 2500, 11, seed=23, noise=1e-2, empty_every=7)
    ba = _upload(c2b, P)
    uv = np.array(P["uv"], copy=True)
    exact = ba.project()
    uv[::5] = exact[::5]                              # every fifth residual exactly zero
    uv[1::5] = exact[1::5] + 1e-300                   # ... and a few far below the table path's range check
    want = O.total_reprojection_error(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], uv, norm)
    got = c2b.BAProblem.from_visibility(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], uv).total_reprojection_error(norm)
    assert np.isfinite(got) and abs(got - want) / want < 1e-12


def test_zero_error_by_construction_on_gpu(c2b):
    """observations written from the device's own project() => error exactly 0."""
    P = random_problem(100, 2000, 10, seed=13)
    ba = _upload(c2b, P)
    uv = ba.project()
    ba2 = c2b.BAProblem.from_visibility(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], uv)
    assert ba2.total_reprojection_error(2.0) == 0.0
    assert ba2.total_reprojection_error(1.0) == 0.0
    r, _, _ = ba2.residual_jacobian()
    assert np.all(r == 0.0)


@pytest.mark.parametrize("n_cam,n_pts,opc,seed,empty_every", [
    (1, 1, 1, 0, 0), (5, 100, 13, 1, 2), (33, 700, 8, 2, 0), (257, 8000, 31, 3, 5), (9, 4000, 300, 4, 0),
])
def test_residual_jacobian_state_mode(c2b, n_cam, n_pts, opc, seed, empty_every):
    P = random_problem(n_cam, n_pts, opc, seed=seed, noise=1e-3, empty_every=empty_every)
    r0, Jc0, Jp0 = O.residual_jacobian(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    r, Jc, Jp = _upload(c2b, P).residual_jacobian()
    scale = max(1.0, float(np.max(np.abs(Jc0)))) if len(Jc0) else 1.0
    assert np.max(np.abs(r - r0), initial=0.0) < 1e-12
    assert np.max(np.abs(Jc.reshape(-1, 18) - Jc0), initial=0.0) / scale < 1e-10
    assert np.max(np.abs(Jp.reshape(-1, 6) - Jp0), initial=0.0) / scale < 1e-10


def test_residual_jacobian_bal_mode(c2b):
    P = random_problem(120, 2500, 12, seed=5, noise=1e-3)
    r0, Jc0, Jp0 = O.residual_jacobian_bal(P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    r, Jc, Jp = _upload(c2b, P, bal=True).residual_jacobian()
    scale = max(1.0, float(np.max(np.abs(Jc0))))
    assert np.max(np.abs(r - r0)) < 1e-11
    assert np.max(np.abs(Jc.reshape(-1, 18) - Jc0)) / scale < 1e-10
    assert np.max(np.abs(Jp.reshape(-1, 6) - Jp0)) / scale < 1e-10


def test_jacobian_matches_finite_differences_of_gpu_projection(c2b):
    """Size-independent property: central differences of the device's own projection in the 9 BAL
    parameters and in X reproduce the device's analytic blocks."""
    P = random_problem(20, 300, 6, seed=17)
    w = P["bal9"][:, :3]
    keep = np.linalg.norm(w, axis=1) < 2.5                   # stay inside one Rodrigues chart
    P["bal9"][~keep, :3] *= 0.3
    ba = c2b.BAProblem.from_bal(P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], np.zeros_like(P["uv"]))
    _, Jc, Jp = ba.residual_jacobian()
    counts = np.diff(P["row_ptr"].astype(np.int64))
    cam_of = np.repeat(np.arange(len(counts)), counts)
    h = 1e-6
    for j in range(9):
        hi = P["bal9"].copy(); hi[:, j] += h
        lo = P["bal9"].copy(); lo[:, j] -= h
        up = c2b.BAProblem.from_bal(hi, P["pts"], P["row_ptr"], P["pt_idx"], P["uv"]).project()
        dn = c2b.BAProblem.from_bal(lo, P["pts"], P["row_ptr"], P["pt_idx"], P["uv"]).project()
        fd = (up - dn) / (2 * h)
        assert np.allclose(fd, Jc[:, :, j], rtol=2e-5, atol=2e-6), j
    for j in range(3):
        hi = P["pts"].copy(); hi[:, j] += h
        lo = P["pts"].copy(); lo[:, j] -= h
        up = c2b.BAProblem.from_bal(P["bal9"], hi, P["row_ptr"], P["pt_idx"], P["uv"]).project()
        dn = c2b.BAProblem.from_bal(P["bal9"], lo, P["row_ptr"], P["pt_idx"], P["uv"]).project()
        fd = (up - dn) / (2 * h)
        assert np.allclose(fd, Jp[:, :, j], rtol=2e-5, atol=2e-6), j
    del cam_of


def test_empty_and_degenerate_inputs(c2b):
    # no observations at all
    ba = c2b.BAProblem.from_bal([IDENT_CAM], [[0, 0, -1.0]], [0, 0], [], np.zeros((0, 2)))
    assert ba.num_observations() == 0 and ba.project().shape == (0, 2)
    assert ba.total_reprojection_error(2.0) == 0.0
    r, Jc, Jp = ba.residual_jacobian()
    assert r.shape == (0, 2) and Jc.shape == (0, 2, 9) and Jp.shape == (0, 2, 3)
    # z == 0 -> inf/NaN propagate like the reference (not an error)
    ba = c2b.BAProblem.from_bal([IDENT_CAM], [[1.0, 1.0, 0.0]], [0, 1], [0], [[0.0, 0.0]])
    uv = ba.project()
    want = O.project_observations(O.camera_from_bal(IDENT_CAM), [[1.0, 1.0, 0.0]], [0, 1], [0])
    assert np.all(np.isnan(uv) == np.isnan(want)) and np.all(np.isinf(uv) == np.isinf(want))


def test_asserts_become_status_codes(c2b):
    with pytest.raises(c2b.City2baError) as ei:       # assert!(ci < &points.len()), src/baproblem.rs:368
        c2b.BAProblem.from_bal([IDENT_CAM], [[0, 0, -1.0]], [0, 1], [1], [[0.0, 0.0]])
    assert ei.value.status == -2
    with pytest.raises(c2b.City2baError) as ei:       # assert!(cam_i < cams.len()), src/baproblem.rs:345
        c2b.BAProblem.new([IDENT_CAM], [[0, 0, -1.0]], [(1, 0, 0.0, 0.0)])
    assert ei.value.status == -2
    with pytest.raises(c2b.City2baError) as ei:       # row_ptr must be a prefix sum
        c2b.BAProblem.from_bal([IDENT_CAM, IDENT_CAM], [[0, 0, -1.0]], [0, 1, 0], [0], [[0.0, 0.0]])
    assert ei.value.status == -1
    ba = c2b.BAProblem.from_bal([IDENT_CAM], [[0, 0, -1.0]], [0, 1], [0], [[0.0, 0.0]])
    with pytest.raises(c2b.City2baError) as ei:       # Normal::new panics on negative std (rand 0.6)
        c2b.noise.add_noise(ba, -1.0, 0.0, 0.0, 0.0)
    assert ei.value.status == -1


def test_new_keeps_push_order_per_camera(c2b):
    """BAProblem::new pushes observations per camera in arrival order (src/baproblem.rs:344-348)."""
    pts = np.array([[0, 0, -2.0], [0.1, 0, -2.0], [0, 0.1, -3.0]])
    obs = [(1, 2, 0.5, 0.5), (0, 1, 0.1, 0.2), (1, 0, 0.3, 0.4), (0, 0, 0.7, 0.8), (1, 1, 0.9, 1.0)]
    ba = c2b.BAProblem.new([IDENT_CAM, IDENT_CAM], pts, obs)
    assert list(ba.row_ptr) == [0, 2, 5]
    assert list(ba.pt_idx) == [1, 0, 2, 0, 1]
    assert np.array_equal(ba.observations(), [[0.1, 0.2], [0.7, 0.8], [0.5, 0.5], [0.3, 0.4], [0.9, 1.0]])
    assert "2 cameras, 3 points, and 5 observations" in str(ba)


# ---------------------------------------------------------------------------------------------
# camera (de)serialisation, stats
# ---------------------------------------------------------------------------------------------
def test_from_bal_to_bal_against_oracle(c2b):
    P = random_problem(500, 10, 0, seed=19)
    ba = c2b.BAProblem.from_bal(P["bal9"], P["pts"], np.zeros(501, dtype=np.uint64), [], np.zeros((0, 2)))
    cams = ba.cameras()
    assert np.max(np.abs(cams - P["cams15"])) < 5e-16 * 4          # device sincos vs glibc: ulps
    assert np.array_equal(cams[:, 9:], P["cams15"][:, 9:])
    ba2 = c2b.BAProblem.from_visibility(P["cams15"], P["pts"], np.zeros(501, dtype=np.uint64), [], np.zeros((0, 2)))
    got = ba2.cameras_bal()
    want = O.camera_to_bal(P["cams15"])
    assert np.max(np.abs(got - want)) < 1e-12


def test_stats_against_oracle(c2b):
    P = random_problem(700, 5000, 3, seed=23)
    ba = _upload(c2b, P)
    assert _relerr(ba.mean(), O.mean(P["cams15"], P["pts"]), floor=1e-3) < 1e-12
    assert _relerr(ba.std(), O.std(P["cams15"], P["pts"])) < 1e-12
    mn, mx = ba.extent()
    mn0, mx0 = O.extent(P["cams15"], P["pts"])
    assert np.array_equal(mn, mn0) and np.array_equal(mx, mx0)        # min/max: exact
    assert np.array_equal(ba.dimensions(), O.dimensions(P["cams15"], P["pts"]))
    o, idx = ba.drift_origin()
    o0, idx0 = O.drift_origin(P["cams15"], P["pts"])
    assert idx == idx0 and np.array_equal(o, o0)                      # index work: exact


def test_statistics_of_moved_cameras_come_from_the_centre_refresh_and_carry_the_camera_tables_bits(c2b):
    """r06: after something moved the cameras (add_drift, add_noise) the Level-1 statistics derive the centre table ALONE from
    the state (k_cameras_centers) instead of a whole camera table the next entity pass would invalidate unread.  Same cm_center,
    same operands: the 20 statistics are bit-equal to those of a problem freshly uploaded with the moved cameras (whose statistics
    read the table k_cameras_prepare wrote) -- (count, mean, M2) triples, min / max, origin and all."""
    from city2ba_amd import noise as N
    P = random_problem(701, 3000, 4, seed=29)
    ba = _upload(c2b, P)
    N.add_drift_normalized(ba, 1e-3, 1e-4, 1e-2, seed=3)                # leaves the camera table stale
    moved = ba._stats()                                                 # centres refreshed from the state, table left stale
    fresh = c2b.BAProblem.from_visibility(ba.cameras(), ba.points(), P["row_ptr"], P["pt_idx"], P["uv"])
    assert np.array_equal(moved.view(np.uint64), fresh._stats().view(np.uint64))
    assert np.array_equal(ba.project(), fresh.project())                # ... and the next pass re-derives the whole table as before
    assert not np.array_equal(moved, _upload(c2b, P)._stats())          # (the drift did move them)


def test_drift_origin_tie_goes_to_later(c2b):
    pts = np.array([[1.0, 0, 0], [0.0, 0, 0], [0, 0, 2.0], [0.0, 0.0, 0.0], [3.0, 0, 0]])
    ba = c2b.BAProblem.from_bal([IDENT_CAM], pts, [0, 0], [], np.zeros((0, 2)))
    o, idx = ba.drift_origin()
    assert idx == 1 + 3 and np.all(o == 0.0)


# ---------------------------------------------------------------------------------------------
# visibility predicate on the reference's test grid (indices must be bit-exact)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("blocks,cpb,ppb,L", [(3, 10, 20, 5.0), (4, 10, 10, 20.0)])
def test_visibility_pairs_bit_exact(c2b, blocks, cpb, ppb, L):
    cams, pts = grid_cameras_points(blocks, cpb=cpb, ppb=ppb, L=L)
    ci, pi = grid_candidate_pairs(cams, pts, 10.0)
    uv0, keep0 = O.visibility_pairs(cams, pts, ci, pi, 10.0)
    ba = c2b.BAProblem.from_visibility(cams, pts, np.zeros(len(cams) + 1, dtype=np.uint64), [], np.zeros((0, 2)))
    uv, keep = ba.visibility_pairs(ci, pi, 10.0)
    assert keep0.sum() > 1000
    assert np.array_equal(keep, keep0), "kept observation indices must match the CPU path exactly"
    assert np.array_equal(np.isnan(uv), np.isnan(uv0))
    assert np.array_equal(uv[keep == 1], uv0[keep0 == 1])            # k2 == 0 here: bit-exact uv too
    # boundary cases exist in this grid (|u| == 1 exactly) and must be kept by <= / >=
    assert np.any(np.abs(uv0[keep0 == 1]) == 1.0)


# ---------------------------------------------------------------------------------------------
# noise: same draws as the oracle's Philox scheme, reference inequalities (tests/main.rs:134-195)
# ---------------------------------------------------------------------------------------------
def _grid_problem():
    cams, pts = grid_cameras_points(3, cpb=10, ppb=20, L=5.0)
    ci, pi = grid_candidate_pairs(cams, pts, 10.0)
    uv, keep = O.visibility_pairs(cams, pts, ci, pi, 10.0)
    ci, pi, uv = ci[keep == 1], pi[keep == 1], uv[keep == 1]
    row_ptr = np.zeros(len(cams) + 1, dtype=np.int64)
    np.add.at(row_ptr, ci.astype(np.int64) + 1, 1)
    return dict(cams15=cams, pts=pts, row_ptr=np.cumsum(row_ptr).astype(np.uint64), pt_idx=pi.astype(np.uint64), uv=uv)


def test_add_drift_matches_oracle(c2b):
    P = _grid_problem()
    d = np.array([0.3, -0.5, 0.8])
    c0, p0 = O.add_drift(P["cams15"], P["pts"], 1e-3, 2e-3, 0.2, d, seed=42)
    ba = c2b.noise.add_drift(_upload(c2b, P), 1e-3, 2e-3, 0.2, d, seed=42)
    assert np.max(np.abs(ba.cameras() - c0)) < 1e-9
    assert np.max(np.abs(ba.points() - p0)) < 1e-9


def test_add_drift_normalized_matches_oracle_and_inequality(c2b):
    P = _grid_problem()
    ba = _upload(c2b, P)
    e0 = ba.total_reprojection_error(2.0)
    c0, p0 = O.add_drift_normalized(P["cams15"], P["pts"], 0.1, 0.1, 0.1, seed=7)
    ba = c2b.noise.add_drift_normalized(ba, 0.1, 0.1, 0.1, seed=7)
    assert np.max(np.abs(ba.cameras() - c0)) < 1e-7          # strength*d^2 amplifies ulps of |std|
    assert np.max(np.abs(ba.points() - p0)) < 1e-7
    assert ba.total_reprojection_error(2.0) > e0             # tests/main.rs:134-141


def test_add_noise_matches_oracle_and_inequality(c2b):
    P = _grid_problem()
    ba = _upload(c2b, P)
    e0 = ba.total_reprojection_error(2.0)
    c0, p0, uv0 = O.add_noise(P["cams15"], P["pts"], P["uv"], 0.1, 0.1, 0.1, 0.1, seed=99)
    ba = c2b.noise.add_noise(ba, 0.1, 0.1, 0.1, 0.1, seed=99)
    assert np.max(np.abs(ba.cameras() - c0)) < 1e-9
    assert np.max(np.abs(ba.points() - p0)) < 1e-9
    assert np.max(np.abs(ba.observations() - uv0)) < 1e-9
    assert ba.total_reprojection_error(2.0) > e0             # tests/main.rs:143-150
    e_cpu = O.total_reprojection_error(c0, p0, P["row_ptr"], P["pt_idx"], uv0, 2.0)
    assert abs(ba.total_reprojection_error(2.0) - e_cpu) / e_cpu < 1e-7


def test_observation_noise_lean_transcendentals_track_libm_and_the_distribution(c2b):
    """k_add_noise_observations evaluates its own log / sin / cos (camera_math.hpp: log_unit, sincos_turns32 -- the
    fdlibm kernels on exactly the domains the draws need) instead of the library's; pinned here against the CPU
    restatement's glibc to 1e-13 over 400 000 draws (std = 1 on a zero uv: the output IS the draw), far inside the
    1e-9 the draws are compared at.  The draws themselves: direction uniform on the circle, magnitude N(0, 1)
    (src/noise.rs:152-170: unit_random * Normal(0, std))."""
    import torch
    from city2ba_amd import device as D
    n = 400_000
    dev = torch.device("cuda", 0)
    uv = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    D.add_noise_observations(uv, 12345, 1.0, seed=4242)
    got = uv.cpu().numpy()
    one_cam = np.array([[1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0]], dtype=np.float64)
    _, _, want = O.add_noise(one_cam, np.ones((1, 3)), np.zeros((n, 2)), 0.0, 0.0, 0.0, 1.0, seed=4242, obs_offset=12345)
    assert np.max(np.abs(got - want)) < 1e-13 * max(1.0, float(np.max(np.abs(want))))
    r = np.hypot(got[:, 0], got[:, 1])                      # |z|: half-normal -> mean sqrt(2/pi), E r^2 = 1
    assert abs(r.mean() - np.sqrt(2 / np.pi)) < 4e-3 and abs((r * r).mean() - 1.0) < 8e-3
    ang = np.arctan2(got[:, 1], got[:, 0])                  # z's sign folds the direction: still uniform on the circle
    assert abs(np.cos(ang).mean()) < 4e-3 and abs(np.sin(ang).mean()) < 4e-3 and abs(np.cos(2 * ang).mean()) < 4e-3
    hist, _ = np.histogram(ang, bins=16, range=(-np.pi, np.pi))
    assert hist.min() > 0.93 * n / 16 and hist.max() < 1.07 * n / 16


def test_add_sin_noise_matches_oracle_and_inequality(c2b):
    P = _grid_problem()
    ba = _upload(c2b, P)
    e0 = ba.total_reprojection_error(2.0)
    c0, p0 = O.add_sin_noise(P["cams15"], P["pts"], [1.0, 1.0, 0.0], [0.0, 1.0, 0.0], 1.0, 2.0)
    ba = c2b.noise.add_sin_noise(ba, [1.0, 1.0, 0.0], [0.0, 1.0, 0.0], 1.0, 2.0)
    assert np.max(np.abs(ba.cameras() - c0)) < 1e-9
    assert np.max(np.abs(ba.points() - p0)) < 1e-9
    assert ba.total_reprojection_error(2.0) > e0             # tests/main.rs:188-195


def test_zero_strength_noise_only_rounds(c2b):
    """run_noise always calls add_drift/add_noise, even with all-zero parameters
    (src/bin/city2ba.rs:305-340): the result differs from the input by rounding only."""
    P = _grid_problem()
    ba = c2b.noise.add_noise(c2b.noise.add_drift_normalized(_upload(c2b, P), 0.0, 0.0, 0.0, seed=1), 0, 0, 0, 0, seed=2)
    assert np.max(np.abs(ba.cameras() - P["cams15"])) < 1e-12
    assert np.array_equal(ba.points(), P["pts"]) and np.array_equal(ba.observations(), P["uv"])
    c0, p0 = O.add_drift_normalized(P["cams15"], P["pts"], 0.0, 0.0, 0.0, seed=1)
    c0, p0, _ = O.add_noise(c0, p0, P["uv"], 0, 0, 0, 0, seed=2)
    assert np.array_equal(ba.cameras(), c0)                  # pure algebra (no transcendental): bit-exact


def test_noise_is_seed_deterministic_and_seed_sensitive(c2b):
    P = _grid_problem()
    a = c2b.noise.add_noise(_upload(c2b, P), 0.1, 0.1, 0.1, 0.1, seed=5)
    b = c2b.noise.add_noise(_upload(c2b, P), 0.1, 0.1, 0.1, 0.1, seed=5)
    c = c2b.noise.add_noise(_upload(c2b, P), 0.1, 0.1, 0.1, 0.1, seed=6)
    assert np.array_equal(a.observations(), b.observations()) and np.array_equal(a.cameras(), b.cameras())
    assert not np.array_equal(a.observations(), c.observations())


# ---------------------------------------------------------------------------------------------
# full-size properties (BASELINE config sizes; oracle too slow there -> size-independent checks)
# ---------------------------------------------------------------------------------------------
def test_large_problem_properties(c2b):
    rng = np.random.default_rng(31)
    n_cam, n_pts, per = 40000, 120000, 30
    P = random_problem(400, 12000, per, seed=37)             # template, tiled 100x with shifted points
    reps = n_cam // 400
    cams = np.tile(P["cams15"], (reps, 1))
    pts = np.tile(P["pts"], (reps, 1))
    counts = np.tile(np.diff(P["row_ptr"].astype(np.int64)), reps)
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    pt_idx = np.concatenate([P["pt_idx"].astype(np.int64) + k * 12000 for k in range(reps)]).astype(np.uint64)
    n_obs = len(pt_idx)
    assert n_obs > 1_000_000
    ba = c2b.BAProblem.from_visibility(cams, pts, row_ptr, pt_idx, np.zeros((n_obs, 2)))
    uv = ba.project()
    # periodicity: every replica projects identically (bit-exact), and matches the oracle on replica 0
    m = int(P["row_ptr"][-1])
    assert np.array_equal(uv[:m], uv[m:2 * m]) and np.array_equal(uv[:m], uv[-m:])
    want = O.project_observations(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"])
    assert _relerr(uv[:m], want, floor=1e-3) < 1e-13
    # zero error by construction + linear growth of the L1 error under a uniform observation shift
    ba = c2b.BAProblem.from_visibility(cams, pts, row_ptr, pt_idx, uv)
    assert ba.total_reprojection_error(2.0) == 0.0
    shift = rng.normal(size=2) * 1e-3
    ba = c2b.BAProblem.from_visibility(cams, pts, row_ptr, pt_idx, uv + shift)
    l1 = ba.total_reprojection_error(1.0)
    assert abs(l1 - n_obs * np.abs(shift).sum()) / l1 < 1e-9
    r, Jc, Jp = ba.residual_jacobian()
    assert np.allclose(r, -shift, rtol=0, atol=1e-12)
    assert np.array_equal(Jc[:m], Jc[m:2 * m]) and np.array_equal(Jp[:m], Jp[-m:])
    assert np.all(np.isfinite(Jc)) and np.all(np.isfinite(Jp))


def test_level1_jacobian_leaves_in_chunks_pageable_and_pinned(c2b):
    """c2b_problem_residual_jacobian streams its results out in 256k-observation chunks through a ring of device
    buffers (copies overlapped with the next chunk's kernel).  A problem of several chunks, with a ragged last one,
    into fresh pageable arrays, caller-owned arrays, and page-locked arrays: all equal the oracle's."""
    from city2ba_amd.baproblem import pinned_empty
    P = random_problem(700, 30000, 137, seed=31, noise=1e-3)
    # every camera's observation list repeated 19 times (with different observed uv): 570 000 observations
    reps, counts = 19, np.diff(P["row_ptr"].astype(np.int64))
    rng = np.random.default_rng(32)
    P["pt_idx"] = np.concatenate([np.tile(P["pt_idx"][a:a + k], reps) for a, k in zip(P["row_ptr"][:-1].astype(np.int64), counts)])
    P["uv"] = np.concatenate([np.tile(P["uv"][a:a + k], (reps, 1)) for a, k in zip(P["row_ptr"][:-1].astype(np.int64), counts)])
    P["uv"] = P["uv"] + rng.normal(scale=1e-3, size=P["uv"].shape)
    P["row_ptr"] = np.concatenate([[0], np.cumsum(counts * reps)]).astype(np.uint64)
    n = len(P["pt_idx"])
    assert n > 2 * 256 * 1024 and n % (256 * 1024) != 0
    r0, Jc0, Jp0 = O.residual_jacobian(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    ba = _upload(c2b, P)
    scale = max(1.0, float(np.max(np.abs(Jc0))))
    outs = [ba.residual_jacobian(), ba.residual_jacobian(pinned=True)]
    mine = (np.full((n, 2), np.nan), np.full((n, 2, 9), np.nan), np.full((n, 2, 3), np.nan))
    outs.append(ba.residual_jacobian(out=mine))
    assert outs[2][0] is mine[0]
    pin = (pinned_empty((n, 2)), pinned_empty((n, 2, 9)), pinned_empty((n, 2, 3)))
    for _ in range(2):                                           # reuse: ring slots and events are recycled
        outs.append(ba.residual_jacobian(out=pin))
    for r, Jc, Jp in outs:
        assert np.array_equal(r, outs[0][0]) and np.array_equal(Jc, outs[0][1]) and np.array_equal(Jp, outs[0][2])
    r, Jc, Jp = outs[0]
    assert np.max(np.abs(r - r0)) < 1e-12
    # 1e-9: this sample holds a camera with |w| = 8e-4, where to_rodrigues(R) (state mode) is worth ~1e-10 relative
    assert np.max(np.abs(Jc.reshape(n, 18) - Jc0.reshape(n, 18))) / scale < 1e-9
    assert np.max(np.abs(Jp.reshape(n, 6) - Jp0.reshape(n, 6))) / scale < 1e-9
    with pytest.raises(c2b.City2baError):
        ba.residual_jacobian(out=(np.empty((n, 2), dtype=np.float32), mine[1], mine[2]))


def test_shared_reciprocal_division_is_ieee_exact_over_the_whole_exponent_range(c2b):
    """project() divides -q.x and -q.y by the same q.z (src/baproblem.rs:146); the device computes the refined
    reciprocal of the scaled denominator once for both (camera_math.hpp: div2_shared).  With an identity camera,
    f = 1 and no distortion, uv IS the pair of quotients: they must equal the CPU's IEEE divisions bit for bit for
    operands anywhere in the exponent range -- quotients that are subnormal, that overflow, denominators whose
    reciprocal is subnormal, numerators that need opposite scalings (the fallback path)."""
    rng = np.random.default_rng(123)
    n = 200_000
    mant = rng.uniform(1.0, 2.0, (n, 3)) * rng.choice([-1.0, 1.0], (n, 3))
    expo = rng.integers(-1000, 1000, (n, 3))
    expo[: n // 4] = rng.integers(-60, 60, (n // 4, 3))                   # a quarter in the ordinary range
    expo[n // 4: n // 2, 2] = rng.integers(-1022, -900, n // 4)           # tiny denominators
    expo[n // 2: 3 * n // 4, 0] = expo[n // 2: 3 * n // 4, 2] - rng.integers(1000, 1070, n // 4)   # subnormal quotients
    pts = np.ldexp(mant, np.clip(expo, -1073, 1023))
    special = np.array([[0.0, 1.0, 2.0], [1.0, 0.0, -3.0], [5e-324, 1.0, 1.0], [1.0, 1.0, 5e-324], [1e308, -1e308, 1e-308],
                        [1e-308, 1e308, 1e308], [3.0, -7.0, 1.7976931348623157e308], [2.2250738585072014e-308, 1.0, 3.0]])
    pts = np.vstack([pts, special])
    cam = O.camera_from_bal([0, 0, 0, 0, 0, 0, 1.0, 0.0, 0.0]).reshape(1, 15)
    row_ptr = np.array([0, len(pts)], dtype=np.uint64)
    pt_idx = np.arange(len(pts), dtype=np.uint64)
    with np.errstate(all="ignore"):
        want = np.column_stack([-pts[:, 0] / pts[:, 2], -pts[:, 1] / pts[:, 2]])   # numpy = IEEE division
    want_o = O.project_observations(cam, pts, row_ptr, pt_idx)
    fin = np.isfinite(want).all(axis=1) & (np.abs(want) < 1e70).all(axis=1)        # beyond that |p|^4 overflows and 0 * inf = NaN
    assert np.array_equal(want_o[fin].view(np.uint64), want[fin].view(np.uint64))    # the oracle agrees with plain IEEE
    ba = c2b.BAProblem.from_visibility(cam, pts, row_ptr, pt_idx, np.zeros((len(pts), 2)))
    got = ba.project()
    assert np.array_equal(got[fin].view(np.uint64), want[fin].view(np.uint64))
    assert fin.sum() > 0.5 * len(pts)
    sub = fin & (np.abs(want[:, 0]) < 2.3e-308) & (want[:, 0] != 0)
    assert sub.sum() > 1000                                                       # subnormal quotients were exercised
    # everything, finite or not, equals the oracle's projection (same NaN / inf classes)
    both_nan = np.isnan(got) & np.isnan(want_o)
    assert np.array_equal(np.where(both_nan, 0.0, got).view(np.uint64), np.where(both_nan, 0.0, want_o).view(np.uint64))
