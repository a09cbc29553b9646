"""Level-0 (device pointers, torch as the memory owner) edge cases that the CSR-based Level-1 API cannot produce:
observations in arbitrary (unsorted) camera order -- the kernels promise correctness there through their
global-memory fallback -- and many cameras per wave (more than the 12-camera LDS tile)."""
import numpy as np
import pytest

import oracle as O
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


def _device_problem(env, P, order):
    torch, D, dev = env["torch"], env["D"], env["dev"]
    counts = np.diff(P["row_ptr"].astype(np.int64))
    cam_of = np.repeat(np.arange(len(counts)), counts)
    cam15 = torch.from_numpy(P["cams15"]).to(dev)
    camblk = D.cameras_prepare_state(cam15)
    pts4 = D.points_pad(torch.from_numpy(P["pts"]).to(dev))
    ci = torch.from_numpy(cam_of[order].astype(np.int32)).to(dev)
    pi = torch.from_numpy(P["pt_idx"].astype(np.int64)[order].astype(np.int32)).to(dev)
    uv = torch.from_numpy(np.ascontiguousarray(P["uv"][order])).to(dev)
    return camblk, pts4, ci, pi, uv


@pytest.mark.parametrize("n_cam,n_pts,opc,seed", [(200, 3000, 9, 1), (900, 4000, 2, 2), (37, 2000, 40, 3)])
def test_unsorted_observation_order(env, n_cam, n_pts, opc, seed):
    torch, D, dev = env["torch"], env["D"], env["dev"]
    P = random_problem(n_cam, n_pts, opc, seed=seed, noise=1e-3)
    n = len(P["pt_idx"])
    rng = np.random.default_rng(seed)
    order = rng.permutation(n)                           # arbitrary COO order: cameras jump on every lane
    camblk, pts4, ci, pi, uv = _device_problem(env, P, order)
    r = torch.empty((n, 2), dtype=torch.float64, device=dev)
    Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
    Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ws = D.workspace(n, dev)
    err = torch.zeros(1, dtype=torch.float64, device=dev)
    D.residual_jacobian(camblk, pts4, ci, pi, uv, r, Jc, Jp, 2.0, ws)
    D.error_sum_finish(ws, n, err)
    proj = torch.empty((n, 2), dtype=torch.float64, device=dev)
    D.project(camblk, pts4, ci, pi, proj)
    e2 = torch.zeros(1, dtype=torch.float64, device=dev)
    D.reprojection_error_sum(camblk, pts4, ci, pi, uv, 2.0, ws, e2)
    torch.cuda.synchronize()
    r0, Jc0, Jp0 = O.residual_jacobian(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    scale = max(1.0, float(np.max(np.abs(Jc0))))
    assert np.max(np.abs(r.cpu().numpy() - r0[order])) < 1e-12
    assert np.max(np.abs(Jc.cpu().numpy() - Jc0[order])) / scale < 1e-10
    assert np.max(np.abs(Jp.cpu().numpy() - Jp0[order])) / scale < 1e-10
    want_uv = O.project_observations(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"])[order]
    assert np.max(np.abs(proj.cpu().numpy() - want_uv) / np.maximum(np.abs(want_uv), 1e-3)) < 1e-13
    want_e = float(np.sum(r0 * r0))
    assert abs(err.item() - want_e) / want_e < 1e-11 and abs(e2.item() - want_e) / want_e < 1e-11


def test_sorted_and_unsorted_launches_agree_bitwise(env):
    """The LDS-tile path (sorted) and the global fallback (unsorted) read the same camera records: same bits."""
    torch, D, dev = env["torch"], env["D"], env["dev"]
    P = random_problem(150, 3000, 12, seed=9, noise=1e-3)
    n = len(P["pt_idx"])
    order = np.random.default_rng(4).permutation(n)
    outs = []
    for o in (np.arange(n), order):
        camblk, pts4, ci, pi, uv = _device_problem(env, P, o)
        r = torch.empty((n, 2), dtype=torch.float64, device=dev)
        Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
        Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
        D.residual_jacobian(camblk, pts4, ci, pi, uv, r, Jc, Jp, 2.0, None)
        torch.cuda.synchronize()
        outs.append((r.cpu().numpy(), Jc.cpu().numpy(), Jp.cpu().numpy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a[order], b)


def test_camera_trait_methods_batched(env):
    """project_world / to_world / transform as stand-alone batched calls: bit-exact with the oracle (pure IEEE
    algebra, no transcendental), and the reference's test_project_isomorphic (src/baproblem.rs:244-249) on the device."""
    torch, D, dev = env["torch"], env["D"], env["dev"]
    P = random_problem(64, 500, 0, seed=12)
    rng = np.random.default_rng(3)
    n = 2000
    ci = rng.integers(0, 64, n).astype(np.int32)
    p = rng.uniform(-20, 20, (n, 3))
    cam15 = torch.from_numpy(P["cams15"]).to(dev)
    ci_d, p_d = torch.from_numpy(ci).to(dev), torch.from_numpy(p).to(dev)
    q = D.project_world(cam15, ci_d, p_d)
    back = D.to_world(cam15, ci_d, q)
    torch.cuda.synchronize()
    want_q = np.array([O.project_world(P["cams15"][c], x) for c, x in zip(ci, p)])
    want_back = np.array([O.to_world(P["cams15"][c], x) for c, x in zip(ci, want_q)])
    assert np.array_equal(q.cpu().numpy(), want_q)
    assert np.array_equal(back.cpu().numpy(), want_back)
    assert np.max(np.abs(back.cpu().numpy() - p)) < 1e-8                     # test_project_isomorphic's tolerance
    # the reference's own KAT
    kat = O.camera_from_bal([3.0, 5.0, -2.0, 0.5, -0.2, 0.1, 1.0, 0.0, 0.0])
    c1 = torch.from_numpy(kat).to(dev)
    z = torch.zeros(1, dtype=torch.int32, device=dev)
    pk = torch.tensor([[1.0, 3.0, -1.0]], dtype=torch.float64, device=dev)
    assert float((D.to_world(c1, z, D.project_world(c1, z, pk)) - pk).abs().max()) <= 1e-8
    # transform: per-camera deltas
    dR = np.array([O.basis_from_axis_angle(a / np.linalg.norm(a), th) for a, th in
                   zip(rng.normal(size=(64, 3)), rng.uniform(-1, 1, 64))])
    dl = rng.normal(size=(64, 3))
    moved = D.cameras_transform(cam15.clone(), torch.from_numpy(dR).to(dev), torch.from_numpy(dl).to(dev))
    torch.cuda.synchronize()
    want = np.array([O.transform(P["cams15"][i], dR[i], dl[i]) for i in range(64)])
    assert np.array_equal(moved.cpu().numpy(), want)


def test_alloc_jacobian_outputs_returns_usable_buffers():
    """device.alloc_jacobian_outputs places r / Jc / Jp by measuring the store pattern (bench.py set-up).  Whatever
    allocation it settles on, the buffers have the right shapes, are distinct, and the kernel's results in them equal
    the results in plainly allocated ones."""
    import argparse
    import torch
    import bench
    from city2ba_amd import device as D
    dev = torch.device("cuda", 0)
    small, log = D.alloc_jacobian_outputs(5000, dev)
    assert [tuple(t.shape) for t in small] == [(5000, 2), (5000, 18), (5000, 6)] and log == []
    sh = bench.build_shard(argparse.Namespace(blocks=32), 0, 1, dev)
    n = sh["n_obs"]
    (r, Jc, Jp), log = D.alloc_jacobian_outputs(n, dev, max_attempts=3)
    assert 1 <= len(log) <= 3 and all(x > 0.0 for x in log)
    assert len({r.data_ptr(), Jc.data_ptr(), Jp.data_ptr()}) == 3
    r0, Jc0, Jp0 = torch.empty_like(r), torch.empty_like(Jc), torch.empty_like(Jp)
    ws = D.workspace(n, dev)
    e0, e1 = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    a = (sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"])
    D.residual_jacobian_sum(*a, r0, Jc0, Jp0, 2.0, ws, e0)
    D.residual_jacobian_sum(*a, r, Jc, Jp, 2.0, ws, e1)
    torch.cuda.synchronize()
    assert torch.equal(r, r0) and torch.equal(Jc, Jc0) and torch.equal(Jp, Jp0) and e0.item() == e1.item()


def _sharded_setup(seed=77):
    import torch
    from city2ba_amd import device as D
    from _problems import random_problem
    P = random_problem(211, 3000, 9, seed=seed, noise=1e-2)
    dev = torch.device("cuda", 0)
    cam15 = torch.from_numpy(P["cams15"]).to(dev)
    pts4 = D.points_pad(torch.from_numpy(P["pts"]).to(dev))
    return P, dev, cam15, pts4


def test_sharded_statistics_equal_unsharded():
    """cameras sharded over ranks, points replicated.  Three shards processed one after the other on
    this GPU through c2b_stats_partial_pass1/2 and combined by the host half of the collective
    (dist.combine_stats_partials / finish_stats -- the same functions the ranks call after their all-gathers) give
    the statistics of the whole problem: min / max / origin exactly, mean / std to rounding."""
    import torch
    import oracle as O
    from city2ba_amd import device as D
    from city2ba_amd import dist as Dist
    P, dev, cam15, pts4 = _sharded_setup()
    n_cam, n_pts = cam15.shape[0], pts4.shape[0]
    n_ent = n_cam + n_pts
    ws = D.workspace(0, dev)
    whole = D.stats(D.cameras_prepare_state(cam15), pts4, ws).cpu().numpy()
    world = 3
    cb = [0, 70, 70, n_cam]                                      # uneven, one EMPTY camera shard
    shards = []
    for r in range(world):
        camblk = D.cameras_prepare_state(cam15[cb[r]:cb[r + 1]].contiguous())
        lo, hi = n_pts * r // world, n_pts * (r + 1) // world
        shards.append((camblk, cb[r], pts4[lo:hi], lo))
    parts = np.stack([D.stats_partial_pass1(c, base, n_cam, p, lo, n_ent, ws).cpu().numpy() for c, base, p, lo in shards])
    mean, mn, mx, origin, oidx = Dist.combine_stats_partials(parts, n_ent)
    mean_d = torch.from_numpy(mean).to(dev)
    sumsq = np.zeros(3)
    for c, base, p, lo in shards:
        sumsq = sumsq + D.stats_partial_pass2(c, p, mean_d, ws).cpu().numpy()
    st = Dist.finish_stats(mean, mn, mx, origin, oidx, sumsq, n_ent)
    assert np.array_equal(st[6:15], whole[6:15])                 # min, max, dimensions: exact
    assert int(st[18]) == int(whole[18]) and np.array_equal(st[15:18], whole[15:18])
    assert np.allclose(st[0:6], whole[0:6], rtol=1e-13, atol=1e-13) and abs(st[19] - whole[19]) <= 1e-13 * whole[19]
    assert np.allclose(st[0:3], O.mean(P["cams15"], P["pts"]), rtol=1e-12, atol=1e-12)
    assert np.allclose(st[3:6], O.std(P["cams15"], P["pts"]), rtol=1e-12)
    # world size 1 through the same entry point is the unsharded result
    one = Dist.stats_sharded(D.cameras_prepare_state(cam15), 0, n_cam, pts4, ws).cpu().numpy()
    assert np.array_equal(one[6:19], whole[6:19]) and np.allclose(one[0:6], whole[0:6], rtol=1e-13, atol=1e-13)


def test_compact_centre_table_is_the_record_field_and_the_statistics_do_not_care_which_they_read():
    """r05: cameras_prepare_* also write cen4 [n_cam][4], the cameras' centres as 32-byte rows (x y z 0), and the
    statistics read it instead of one 128-byte line of the 256-byte record per camera.  The table holds the bits of the
    record's centre field; the statistics -- whole, a shard's shares, the sharded entry -- are bit for bit the same whether
    they read the table or the records (same kernel, same order; only the address differs); both routes against the
    oracle; the bal9 form writes the same table."""
    import torch
    import oracle as O
    from city2ba_amd import device as D
    from city2ba_amd import dist as Dist
    P, dev, cam15, pts4 = _sharded_setup(seed=79)
    n_cam, n_pts = cam15.shape[0], pts4.shape[0]
    ws = D.workspace(0, dev)
    cen = torch.full((n_cam, 4), float("nan"), dtype=torch.float64, device=dev)
    blk = D.cameras_prepare_state(cam15, centers=cen)
    rec = D.camblk_records(blk)                                                   # (the table is blocked by groups of 8 cameras)
    assert torch.equal(cen[:, :3], rec[:, 24:27]) and bool((cen[:, 3] == 0).all())
    assert torch.equal(rec, D.camblk_records(D.cameras_prepare_state(cam15)))     # the records do not change
    assert np.allclose(cen[:, :3].cpu().numpy(), O.centers(P["cams15"]), rtol=0, atol=1e-12)
    bal9 = D.cameras_to_bal(cam15)
    cen_b = torch.full_like(cen, float("nan"))
    blk_b = D.cameras_prepare_bal(bal9, centers=cen_b)
    assert torch.equal(cen_b[:, :3], D.camblk_records(blk_b)[:, 24:27]) and bool((cen_b[:, 3] == 0).all())
    a = D.stats(blk, pts4, ws).cpu().numpy()
    b = D.stats(blk, pts4, ws, centers=cen).cpu().numpy()
    assert np.array_equal(a, b)
    assert np.allclose(b[0:3], O.mean(P["cams15"], P["pts"]), rtol=1e-12, atol=1e-12)
    assert np.allclose(b[3:6], O.std(P["cams15"], P["pts"]), rtol=1e-12)
    o0, idx0 = O.drift_origin(P["cams15"], P["pts"])
    assert int(b[18]) == idx0 and np.array_equal(b[15:18], o0)
    # cameras only / points only / a camera range as a shard
    assert np.array_equal(D.stats(blk, pts4[:0], ws).cpu().numpy(), D.stats(blk, pts4[:0], ws, centers=cen).cpu().numpy())
    lo, hi = 40, 170
    sub, sub_c = D.cameras_prepare_state(cam15[lo:hi].contiguous()), cen[lo:hi].contiguous()      # a shard's table is PREPARED for its cameras, not sliced
    n_ent = n_cam + n_pts
    p1 = D.stats_partial_pass1(sub, lo, n_cam, pts4[100:900], 100, n_ent, ws).cpu().numpy()
    p1c = D.stats_partial_pass1(sub, lo, n_cam, pts4[100:900], 100, n_ent, ws, centers=sub_c).cpu().numpy()
    assert np.array_equal(p1, p1c)
    mean_d = torch.from_numpy(b[0:3].copy()).to(dev)
    assert torch.equal(D.stats_partial_pass2(sub, pts4[100:900], mean_d, ws), D.stats_partial_pass2(sub, pts4[100:900], mean_d, ws, centers=sub_c))
    one = Dist.stats_sharded(blk, 0, n_cam, pts4, ws, centers=cen).cpu().numpy()
    assert np.array_equal(one, Dist.stats_sharded(blk, 0, n_cam, pts4, ws).cpu().numpy())
    # a misshapen table is refused before anything is launched
    with pytest.raises(AssertionError):
        D.stats(blk, pts4, ws, centers=cen[:-1])


def test_origin_search_without_a_square_root_per_entity_keeps_fold1s_rule():
    """r05: the statistics pass takes sqrt only for entities whose SQUARED distance is within 8 ulps of the square of the
    thread's best distance.  fold1 (src/noise.rs:80-86) compares rounded distances and keeps the LATER of equals: points
    whose squared distances differ in the last bits but round to the same distance must still resolve to the later one,
    wherever in the table (thread, wave, workgroup) the pair sits."""
    import torch
    from city2ba_amd import device as D
    dev = torch.device("cuda", 0)
    ws = D.workspace(0, dev)
    rng = np.random.default_rng(5)
    n = 300_000
    pts = rng.uniform(5.0, 50.0, size=(n, 3))
    # two points whose SQUARED distances differ in the last bit and whose distances are the same double
    x0 = 2.4842827116307444
    y0 = float(np.sqrt(np.spacing(x0 * x0)))
    d2a, d2b = x0 * x0, (x0 * x0 + y0 * y0) + 0.0 * 0.0
    assert d2b > d2a and np.sqrt(d2a) == np.sqrt(d2b)
    a, b = np.array([x0, 0.0, 0.0]), np.array([x0, y0, 0.0])
    cam15 = torch.zeros((0, 15), dtype=torch.float64, device=dev)
    blk = torch.zeros((0, 32), dtype=torch.float64, device=dev)
    for ia, ib in ((7, 100_000), (100_000, 7), (131_071, 131_072), (299_999, 0), (5, 6)):
        q = pts.copy()
        q[ia], q[ib] = a, b
        d = np.sqrt((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2])
        want = int(np.flatnonzero(d == d.min())[-1])
        st = D.stats(blk, D.points_pad(torch.from_numpy(q).to(dev)), ws).cpu().numpy()
        assert int(st[18]) == want and np.array_equal(st[15:18], q[want])


def test_sharded_drift_and_noise_equal_unsharded_row_for_row():
    """Every draw is keyed by the GLOBAL camera index (cam_base + i): perturbing camera shards separately -- each with
    the whole, replicated point table -- gives bit for bit the cameras and points of the unsharded call."""
    import torch
    from city2ba_amd import device as D
    P, dev, cam15, pts4 = _sharded_setup(seed=78)
    n_cam = cam15.shape[0]
    ws = D.workspace(0, dev)
    st = D.stats(D.cameras_prepare_state(cam15), pts4, ws)
    bounds = [0, 1, 64, 65, n_cam]
    for what in ("drift_normalized", "drift", "noise"):
        c_ref, p_ref = cam15.clone(), pts4.clone()
        if what == "drift_normalized":
            D.add_drift_normalized(c_ref, p_ref, st, 1e-3, 2e-3, 0.2, 42)
        elif what == "drift":
            D.add_drift_sharded(c_ref, 0, p_ref, st, 1e-3, 2e-3, 0.2, 42, direction=(0.3, -0.5, 0.8))
        else:
            D.add_noise_entities(c_ref, p_ref, st, 0.05, 0.02, 0.1, 99)
        assert not torch.equal(c_ref, cam15) and not torch.equal(p_ref, pts4)
        rows, pts_out = [], []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            c, p = cam15[lo:hi].contiguous(), pts4.clone()
            if what == "drift_normalized":
                D.add_drift_sharded(c, lo, p, st, 1e-3, 2e-3, 0.2, 42)
            elif what == "drift":
                D.add_drift_sharded(c, lo, p, st, 1e-3, 2e-3, 0.2, 42, direction=(0.3, -0.5, 0.8))
            else:
                D.add_noise_entities_sharded(c, lo, p, st, 0.05, 0.02, 0.1, 99)
            rows.append(c)
            pts_out.append(p)
        assert torch.equal(torch.cat(rows), c_ref), what
        assert all(torch.equal(p, p_ref) for p in pts_out), what


def test_one_launch_step_is_graph_capturable_and_replays_cleanly():
    """The fused step (residual + Jacobian + folded error sum) inside a HIP graph: the arrival counters live in the
    workspace the graph's launch was given, the last workgroup of every replay resets them, so each replay folds
    correctly -- with inputs changed in place between replays the replayed sum tracks the eager one bit for bit."""
    import argparse
    import torch
    import bench
    from city2ba_amd import device as D
    dev = torch.device("cuda", 0)
    sh = bench.build_shard(argparse.Namespace(blocks=4), 0, 1, dev)
    n = sh["n_obs"]
    r, Jc, Jp = (torch.empty((n, k), dtype=torch.float64, device=dev) for k in (2, 18, 6))
    ws = D.workspace(n, dev)
    err_g, err_e = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    uv = sh["uv"].clone()
    args = (sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], uv)
    D.residual_jacobian_sum(*args, r, Jc, Jp, 2.0, ws, err_g)         # warm-up
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            D.residual_jacobian_sum(*args, r, Jc, Jp, 2.0, ws, err_g)
    torch.cuda.synchronize()
    seen = set()
    for k in range(6):
        uv.add_(1e-3 * (k + 1))                                      # the graph reads uv in place
        g.replay()
        torch.cuda.synchronize()
        got = err_g.item()
        D.residual_jacobian_sum(*args, r, Jc, Jp, 2.0, ws, err_e)
        torch.cuda.synchronize()
        assert got == err_e.item() and got > 0
        seen.add(got)
    assert len(seen) == 6


def test_level1_jacobian_left_on_the_device_equals_the_host_output_entry():
    """c2b_problem_residual_jacobian_device (VERDICT r03 "Missing" 5): the BAProblem-level caller's route to the placed
    output arrays -- bits equal to c2b_problem_residual_jacobian's host arrays, the folded sum equal to
    total_reprojection_error(2.)^2, the set reusable across calls and refused when the problem changed size"""
    import ctypes as C
    import numpy as np
    import torch
    import city2ba_amd as c2b
    from city2ba_amd import _lib as L
    from _problems import random_problem
    P = random_problem(140, 1800, 13, seed=31, noise=1e-3, empty_every=7)
    ba = c2b.BAProblem.from_bal(P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    r, Jc, Jp = ba.residual_jacobian()
    outs, s = ba.residual_jacobian_device()
    torch.cuda.synchronize()
    assert np.array_equal(outs.r.cpu().numpy().view(np.uint64), np.asarray(r).view(np.uint64))
    assert np.array_equal(outs.Jc.cpu().numpy().view(np.uint64), np.asarray(Jc).reshape(-1, 18).view(np.uint64))
    assert np.array_equal(outs.Jp.cpu().numpy().view(np.uint64), np.asarray(Jp).reshape(-1, 6).view(np.uint64))
    e2 = ba.total_reprojection_error(2.0)
    assert abs(s - e2 * e2) <= 1e-12 * s and abs(s - float((np.asarray(r) ** 2).sum())) <= 1e-12 * s
    # the loop of a solver: perturb, re-evaluate into the SAME set
    c2b.noise.add_noise(ba, 0.01, 0.01, 0.01, 0.001, seed=3)
    ptr = outs.Jc.data_ptr()
    outs2, s2 = ba.residual_jacobian_device(outs)
    assert outs2 is outs and outs.Jc.data_ptr() == ptr and s2 != s
    r2, Jc2, _ = ba.residual_jacobian()
    torch.cuda.synchronize()
    assert np.array_equal(outs.r.cpu().numpy().view(np.uint64), np.asarray(r2).view(np.uint64))
    assert np.array_equal(outs.Jc.cpu().numpy().view(np.uint64), np.asarray(Jc2).reshape(-1, 18).view(np.uint64))
    # a set of the wrong size is refused, not overrun
    ba.cull(False)
    assert ba.num_observations() < len(r)
    h = C.c_void_p(outs.handle.value)
    assert L.lib().c2b_problem_residual_jacobian_device(ba._h, 1, C.byref(h), None) == L.ERR_INVALID_ARGUMENT
    assert L.lib().c2b_problem_residual_jacobian_device(ba._h, 1, None, None) == L.ERR_INVALID_ARGUMENT
    # an empty list: an empty set, a zero sum
    e = c2b.BAProblem.from_visibility(P["cams15"][:3], P["pts"], np.zeros(4, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros((0, 2)))
    o0, s0 = e.residual_jacobian_device()
    assert s0 == 0.0 and o0.r.shape == (0, 2)
    for b in (ba, e):
        b.close()


@pytest.mark.gpu
def test_output_placement_searches_deep_and_returns_what_it_held():
    """c2b_jacobian_outputs_alloc (r05): up to 64 attempts; sets smaller than the 2-GiB stride are followed by a held filler so that the
    search moves through the device memory; every reject and filler is freed before the call returns; the kept set's rate is what
    c2b_jacobian_outputs_store_rate reports also when it is the 9th or a later attempt."""
    import torch
    from city2ba_amd import device as D
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    n = 1_200_000                                                  # >= 10^6: measured; 250 MB per set
    free0 = torch.cuda.mem_get_info()[0]
    o = D.JacobianOutputs(n, dev, max_attempts=12, fast_store_GBs=1e9)      # an unreachable rate: all 12 attempts are made
    assert len(o.log) == 12 and 0 <= o.chosen < 12 and all(r > 0.0 for r in o.log)
    assert o.store_GBs == pytest.approx(o.log[o.chosen], rel=1e-3) and o.log[o.chosen] >= 0.98 * max(o.log)
    o.r.fill_(1.0); o.Jc.fill_(2.0); o.Jp.fill_(3.0)
    torch.cuda.synchronize()
    assert float(o.Jc.sum().item()) == 2.0 * n * 18
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held < 2 * n * 208 + (64 << 20), held                    # the kept set (+ allocator granularity), not twelve strides
    late = D.JacobianOutputs(n, dev, max_attempts=64, fast_store_GBs=1e9)
    assert 1 <= len(late.log) <= 64 and late.store_GBs > 0.0
    del o, late
