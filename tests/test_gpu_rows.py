"""The row-structure forms of the per-observation passes (c2b_rows_pack, c2b_project_rows,
c2b_reprojection_error_sum_rows, c2b_visibility_rows): a camera-major list addressed by row_ptr -- the reference's
one list per camera, src/baproblem.rs:256-260 -- must give the SAME BITS as the cam_idx forms on the expanded list,
whatever the shape of the lists: empty lists (at the start, inside a tile, several in a row, at the end), one
observation per camera (64 cameras in a wave: the kernels' slow path), one huge list, ragged ends.  The tile records
themselves are checked against a numpy restatement of their definition."""
import numpy as np
import pytest

from _problems import random_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as entry
    entry.build()
    import torch
    import city2ba_amd
    from city2ba_amd import device as D
    assert city2ba_amd.device_count() > 0
    return dict(torch=torch, D=D, dev=torch.device("cuda", 0))


def _np_tiles(counts):
    """tile records from their definition: per 64 observations {mask of lanes > 0 whose camera differs from the lane
    before, camera of lane 0 | bit 31 when some step inside the tile skips a camera}"""
    cam_of = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    n = len(cam_of)
    n_t = (n + 63) // 64
    rec = np.zeros((n_t, 4), dtype=np.uint32)
    for t in range(n_t):
        c = cam_of[64 * t:64 * t + 64]
        d = np.diff(c)
        mask = 0
        for l in np.nonzero(d != 0)[0]:
            mask |= 1 << int(l + 1)
        rec[t] = (mask & 0xFFFFFFFF, mask >> 32, int(c[0]) | (0x80000000 if np.any(d > 1) else 0), 0)
    return rec


def _lists(kind, rng):
    if kind == "ragged":
        return rng.integers(1, 60, size=300)
    if kind == "empties":                                   # empty lists everywhere, runs of them too
        c = rng.integers(0, 50, size=400)
        c[rng.random(400) < 0.3] = 0
        c[:3] = 0
        c[-2:] = 0
        c[100:120] = 0
        return c
    if kind == "singles":                                   # one observation per camera: 64 cameras per tile
        return np.ones(1000, dtype=np.int64)
    if kind == "singles_with_gaps":
        c = np.ones(1500, dtype=np.int64)
        c[rng.random(1500) < 0.2] = 0
        return c
    if kind == "one_list":
        return np.array([777])
    if kind == "tiny":
        return np.array([0, 1, 0])
    if kind == "tile_edges":                                # boundaries exactly on multiples of 64 and 192
        return np.array([64, 64, 64, 128, 192, 1, 63, 191, 1, 0, 64])
    raise AssertionError(kind)


@pytest.mark.parametrize("kind", ["ragged", "empties", "singles", "singles_with_gaps", "one_list", "tiny", "tile_edges"])
def test_rows_forms_give_the_bits_of_the_index_forms(env, kind):
    torch, D, dev = env["torch"], env["D"], env["dev"]
    rng = np.random.default_rng(sum(kind.encode()))
    counts = np.asarray(_lists(kind, rng), dtype=np.int64)
    n_cam, n = len(counts), int(counts.sum())
    P = random_problem(n_cam, 2000, 3, seed=11, noise=1e-3)            # cameras and points; the lists are ours
    pt = rng.integers(0, 2000, size=n)
    uv = rng.normal(size=(n, 2))
    row_ptr = np.zeros(n_cam + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(counts)
    cam_of = np.repeat(np.arange(n_cam), counts)

    camblk = D.cameras_prepare_state(torch.from_numpy(P["cams15"]).to(dev))
    pts4 = D.points_pad(torch.from_numpy(P["pts"]).to(dev))
    ci = torch.from_numpy(cam_of.astype(np.int32)).to(dev)
    pi = torch.from_numpy(pt.astype(np.int32)).to(dev)
    uv_d = torch.from_numpy(uv).to(dev)
    rows = D.Rows(torch.from_numpy(row_ptr).to(dev))
    assert rows.n_obs == n and rows.n_cam == n_cam
    torch.cuda.synchronize()
    assert np.array_equal(rows.tiles.cpu().numpy().view(np.uint32), _np_tiles(counts))

    a, b = (torch.full((n, 2), 7.0, dtype=torch.float64, device=dev) for _ in range(2))
    D.project(camblk, pts4, ci, pi, a)
    D.project_rows(camblk, pts4, rows, pi, b)
    torch.cuda.synchronize()
    assert np.array_equal(a.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64))

    ws = D.workspace(n, dev)
    for norm in (2.0, 1.0, 1.5):
        ea, eb = (torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(2))
        D.reprojection_error_sum(camblk, pts4, ci, pi, uv_d, norm, ws, ea)
        D.reprojection_error_sum_rows(camblk, pts4, rows, pi, uv_d, norm, ws, eb)
        torch.cuda.synchronize()
        assert ea.item() == eb.item() and np.isfinite(ea.item())

    # residual + Jacobian: the whole list, then the same list in two launches cut at a multiple of 64
    out = [[torch.full((n, w), 5.0, dtype=torch.float64, device=dev) for w in (2, 18, 6)] for _ in range(3)]
    ea, eb = (torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(2))
    D.residual_jacobian_sum(camblk, pts4, ci, pi, uv_d, *out[0], 2.0, ws, ea)
    D.residual_jacobian_rows(camblk, pts4, rows, pi, uv_d, *out[1], 2.0, ws, eb)
    cut = (n // 2) // 64 * 64
    D.residual_jacobian_rows(camblk, pts4, rows, pi[:cut], uv_d[:cut], *[o[:cut] for o in out[2]], n_obs=cut)
    D.residual_jacobian_rows(camblk, pts4, rows, pi[cut:], uv_d[cut:], *[o[cut:] for o in out[2]], obs_base=cut)
    torch.cuda.synchronize()
    assert ea.item() == eb.item()
    for w in range(3):
        assert torch.equal(out[0][w].view(torch.int64), out[1][w].view(torch.int64))
        assert torch.equal(out[0][w].view(torch.int64), out[2][w].view(torch.int64))

    ka, kb = (torch.full((n,), 9, dtype=torch.uint8, device=dev) for _ in range(2))
    D.visibility_pairs(camblk, pts4, ci, pi, 60.0, a, ka)
    D.visibility_rows(camblk, pts4, rows, pi, 60.0, b, kb)
    torch.cuda.synchronize()
    assert np.array_equal(ka.cpu().numpy(), kb.cpu().numpy())
    assert np.array_equal(a.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64))
    if n > 100:
        assert 0 < int(ka.sum().item())
    # the mask as one ballot word per 64 pairs (c2b_visibility_rows_bits): bit l of word t = pair 64 t + l, nothing past n
    words = torch.full(((n + 63) // 64,), -1, dtype=torch.int64, device=dev)
    c = torch.full_like(a, 3.0)
    D.visibility_rows_bits(camblk, pts4, rows, pi, 60.0, c, words)
    torch.cuda.synchronize()
    bits = np.unpackbits(words.cpu().numpy().view(np.uint8), bitorder="little")
    assert np.array_equal(bits[:n], kb.cpu().numpy()) and not bits[n:].any()
    assert np.array_equal(c.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64))


def test_rows_property_random_list_shapes(env):
    """hypothesis over list shapes (runs of empty lists, lists longer than a tile, boundaries on tile edges): the tile
    records equal their definition and every row-structure launcher gives the index form's bits"""
    from hypothesis import given, settings
    from hypothesis import strategies as st
    torch, D, dev = env["torch"], env["D"], env["dev"]
    P = random_problem(40, 500, 3, seed=5, noise=1e-3)
    camblk_all = D.cameras_prepare_state(torch.from_numpy(P["cams15"]).to(dev))
    pts4 = D.points_pad(torch.from_numpy(P["pts"]).to(dev))
    run_len = st.one_of(st.just(0), st.just(0), st.integers(1, 5), st.integers(60, 70), st.integers(120, 200), st.just(64), st.just(128))

    @settings(max_examples=120, deadline=None)
    @given(st.lists(run_len, min_size=1, max_size=40), st.integers(0, 2 ** 31 - 1))
    def run(counts, seed):
        counts = np.asarray(counts, dtype=np.int64)
        n_cam, n = len(counts), int(counts.sum())
        if n == 0:
            return
        rng = np.random.default_rng(seed)
        row_ptr = np.zeros(n_cam + 1, dtype=np.int64)
        row_ptr[1:] = np.cumsum(counts)
        camblk = camblk_all.prefix(n_cam)
        ci = torch.from_numpy(np.repeat(np.arange(n_cam), counts).astype(np.int32)).to(dev)
        pi = torch.from_numpy(rng.integers(0, 500, size=n).astype(np.int32)).to(dev)
        uv = torch.from_numpy(rng.normal(size=(n, 2))).to(dev)
        rows = D.Rows(torch.from_numpy(row_ptr).to(dev))
        assert np.array_equal(rows.tiles.cpu().numpy().view(np.uint32), _np_tiles(counts))
        a, b = (torch.full((n, 2), 3.0, dtype=torch.float64, device=dev) for _ in range(2))
        D.project(camblk, pts4, ci, pi, a)
        D.project_rows(camblk, pts4, rows, pi, b)
        assert torch.equal(a.view(torch.int64), b.view(torch.int64))
        ws = D.workspace(n, dev)
        ea, eb = (torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(2))
        out = [[torch.full((n, w), 5.0, dtype=torch.float64, device=dev) for w in (2, 18, 6)] for _ in range(2)]
        D.residual_jacobian_sum(camblk, pts4, ci, pi, uv, *out[0], 2.0, ws, ea)
        D.residual_jacobian_rows(camblk, pts4, rows, pi, uv, *out[1], 2.0, ws, eb)
        assert ea.item() == eb.item()
        for w in range(3):
            assert torch.equal(out[0][w].view(torch.int64), out[1][w].view(torch.int64))

    run()


def test_rows_on_the_bench_grid_and_their_argument_checks(env):
    """blocks = 4 of the synthetic grid (cameras without observations exist there: no cull), and the ABI's checks"""
    import argparse
    import bench
    from city2ba_amd import _lib as L
    torch, D, dev = env["torch"], env["D"], env["dev"]
    sh = bench.build_shard(argparse.Namespace(blocks=4), 0, 1, dev)
    n, n_cam = sh["n_obs"], sh["camblk"].shape[0]
    row_ptr = torch.zeros(n_cam + 1, dtype=torch.int64, device=dev)
    row_ptr[1:] = torch.cumsum(torch.bincount(sh["cam_idx"].long(), minlength=n_cam), 0)
    rows = D.Rows(row_ptr)
    a, b = (torch.empty((n, 2), dtype=torch.float64, device=dev) for _ in range(2))
    D.project(sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], a)
    D.project_rows(sh["camblk"], sh["pts4"], rows, sh["pt_idx"], b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)

    # a row_ptr that does not cover the list (a caller's bug: the last lists cut short, the first one starting late)
    # must give wrong cameras, not wild addresses: every camera the kernels derive stays inside [0, n_cam)
    bad = row_ptr.clone()
    bad[-3:] = bad[-4]
    bad[0] = 5
    rows_bad = D.Rows(bad, n)
    D.project_rows(sh["camblk"], sh["pts4"], rows_bad, sh["pt_idx"], b)
    torch.cuda.synchronize()
    cams = rows_bad.tiles[:, 2] & 0x7FFFFFFF
    assert int(cams.max()) < n_cam and int(cams.min()) >= 0

    lib = L.lib()
    assert lib.c2b_rows_tiles_bytes(0) == 0 and lib.c2b_rows_tiles_bytes(1) == 16 and lib.c2b_rows_tiles_bytes(65) == 32
    p = lambda t: t.data_ptr()
    # misaligned tile records, no cameras, missing row_ptr: status codes, never a launch
    assert lib.c2b_project_rows(p(sh["camblk"]), p(sh["pts4"]), p(row_ptr), n_cam, p(rows.tiles) + 4, p(sh["pt_idx"]), n, p(b), None) == L.ERR_INVALID_ARGUMENT
    assert lib.c2b_project_rows(p(sh["camblk"]), p(sh["pts4"]), p(row_ptr), 0, p(rows.tiles), p(sh["pt_idx"]), n, p(b), None) == L.ERR_INVALID_ARGUMENT
    assert lib.c2b_project_rows(p(sh["camblk"]), p(sh["pts4"]), None, n_cam, p(rows.tiles), p(sh["pt_idx"]), n, p(b), None) == L.ERR_INVALID_ARGUMENT
    assert lib.c2b_rows_pack(p(row_ptr), 1 << 31, n, p(rows.tiles), None) == L.ERR_INVALID_ARGUMENT
    assert lib.c2b_rows_pack(p(row_ptr), n_cam, 0, None, None) == L.OK                  # an empty list is fine
    assert lib.c2b_project_rows(None, None, None, 0, None, None, 0, None, None) == L.OK


def test_level1_rebuilds_its_row_structure_when_the_list_changes(env):
    """The Level-1 problem derives row_ptr + tile records lazily and must drop them whenever the observation list
    changes (cull, a second upload): project / total_reprojection_error / residual_jacobian after each change against
    the oracle on what the problem then holds."""
    import oracle as O
    import city2ba_amd as c2b
    from city2ba_amd import synthetic as S
    g = S.synthetic_grid(3, 20, 3, 5.0, 1.0, 1.0, 1.0, 10.0, False, cull=False)   # un-culled grid: cameras without observations
    uv = g.observations() + np.random.default_rng(3).normal(scale=1e-2, size=(g.num_observations(), 2))
    ba = c2b.BAProblem.from_visibility(g.cameras(), g.points(), g.row_ptr.copy(), g.pt_idx.copy(), uv)
    g.close()

    def check(ba):
        cams, pts, rp, pi, uv = ba.cameras(), ba.points(), ba.row_ptr, ba.pt_idx, ba.observations()
        if len(pi) == 0:
            assert ba.total_reprojection_error(2.0) == 0.0
            return
        want = O.project_observations(cams, pts, rp, pi)                        # k2 != 0: libm's pow on both sides
        assert np.array_equal(ba.project(), want)
        assert abs(ba.total_reprojection_error(2.0) - O.total_reprojection_error(cams, pts, rp, pi, uv, 2.0)) <= 1e-12 * max(1.0, ba.total_reprojection_error(2.0))
        r, Jc, Jp = ba.residual_jacobian()
        r0, Jc0, Jp0 = O.residual_jacobian(cams, pts, rp, pi, uv)
        Jc = np.asarray(Jc).reshape(len(r0), 18)
        assert np.max(np.abs(r - r0)) < 1e-12 and np.max(np.abs(Jc - Jc0)) / max(1.0, np.max(np.abs(Jc0))) < 1e-9

    check(ba)
    n0 = ba.num_observations()
    ba.cull(False)         # (the faithful mode's index aliasing, src/baproblem.rs:523, empties a graph with isolated cameras)
    assert 0 < ba.num_observations() <= n0
    check(ba)                                                                   # rows of the culled list, not of the old one
    Q = random_problem(9, 50, 5, seed=22, noise=1e-2)
    ba._upload(Q["cams15"], False, Q["pts"], Q["row_ptr"], Q["pt_idx"], Q["uv"])   # a second upload into the same handle
    check(ba)
    ba.close()
