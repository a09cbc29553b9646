"""Fusion along the reference's own call sequences (VERDICT r03 "Missing" 2 and 3).

run_noise evaluates total_reprojection_error(1.) and (2.) back to back on the same data, before and after the noise
(src/bin/city2ba.rs:283-287, 350-354), and add_noise's observation pass (src/noise.rs:152-170) is immediately followed
by that pair.  c2b_reprojection_error_sums2_rows folds both norms in one pass; c2b_add_noise_observations_error_sums2_rows
draws, perturbs, stores, projects and folds both norms in one pass.  Bars: every fused result is BIT-IDENTICAL to the
separate launches it replaces, and equal to the CPU oracle at the tolerances the separate launches are held to."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from _problems import random_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as entry
    entry.build()
    import torch
    import city2ba_amd
    from city2ba_amd import device as D
    assert city2ba_amd.device_count() > 0
    return dict(torch=torch, D=D, dev=torch.device("cuda", 0))


def _lists(kind, rng):
    if kind == "ragged":
        return rng.integers(1, 60, size=300)
    if kind == "empties":
        c = rng.integers(0, 50, size=400)
        c[rng.random(400) < 0.3] = 0
        c[:3] = 0
        c[-2:] = 0
        c[100:120] = 0
        return c
    if kind == "singles":                                   # 64 cameras per tile: the kernels' slow path
        return np.ones(1000, dtype=np.int64)
    if kind == "one_list":
        return np.array([777])
    if kind == "tiny":
        return np.array([0, 1, 0])
    if kind == "large":                                     # several workgroups, the last one ragged
        return rng.integers(20, 40, size=4000)
    raise AssertionError(kind)


def _setup(env, kind, seed=11):
    torch, D, dev = env["torch"], env["D"], env["dev"]
    rng = np.random.default_rng(sum(kind.encode()))
    counts = np.asarray(_lists(kind, rng), dtype=np.int64)
    n_cam, n = len(counts), int(counts.sum())
    P = random_problem(n_cam, 2000, 3, seed=seed, noise=1e-3)
    pt = rng.integers(0, 2000, size=n)
    uv = rng.normal(size=(n, 2))
    row_ptr = np.zeros(n_cam + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(counts)
    camblk = D.cameras_prepare_state(torch.from_numpy(P["cams15"]).to(dev))
    pts4 = D.points_pad(torch.from_numpy(P["pts"]).to(dev))
    pi = torch.from_numpy(pt.astype(np.int32)).to(dev)
    rows = D.Rows(torch.from_numpy(row_ptr).to(dev))
    return dict(P=P, n=n, n_cam=n_cam, pt=pt, uv=uv, row_ptr=row_ptr, camblk=camblk, pts4=pts4, pi=pi, rows=rows,
                ws=D.workspace(n, dev))


@pytest.mark.parametrize("kind", ["ragged", "empties", "singles", "one_list", "tiny", "large"])
def test_l1_and_l2_from_one_launch_carry_the_bits_of_the_two_launches(env, kind):
    import oracle as O
    torch, D, dev = env["torch"], env["D"], env["dev"]
    s = _setup(env, kind)
    uv_d = torch.from_numpy(s["uv"]).to(dev)
    e1, e2 = (torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(2))
    both = torch.full((2,), -1.0, dtype=torch.float64, device=dev)
    D.reprojection_error_sum_rows(s["camblk"], s["pts4"], s["rows"], s["pi"], uv_d, 1.0, s["ws"], e1)
    D.reprojection_error_sum_rows(s["camblk"], s["pts4"], s["rows"], s["pi"], uv_d, 2.0, s["ws"], e2)
    D.reprojection_error_sums2_rows(s["camblk"], s["pts4"], s["rows"], s["pi"], uv_d, s["ws"], both)
    torch.cuda.synchronize()
    assert both[0].item() == e1.item() and both[1].item() == e2.item()
    assert np.isfinite(both.cpu().numpy()).all()
    assert D.workspace_selfcheck(s["ws"]) == 0
    # ... and the oracle's sequential sums (src/baproblem.rs:265-279) at the tolerance of the one-norm launches
    rp, pi64 = s["row_ptr"].astype(np.uint64), s["pt"].astype(np.uint64)
    for k, norm in enumerate((1.0, 2.0)):
        want = O.total_reprojection_error(s["P"]["cams15"], s["P"]["pts"], rp, pi64, s["uv"], norm) ** norm
        assert abs(both[k].item() - want) <= 1e-12 * want


def test_both_sums_are_nan_on_a_workspace_that_was_never_initialised(env):
    torch, D, dev = env["torch"], env["D"], env["dev"]
    s = _setup(env, "ragged")
    uv_d = torch.from_numpy(s["uv"]).to(dev)
    raw = torch.zeros_like(s["ws"])                         # no magic word
    both = torch.zeros(2, dtype=torch.float64, device=dev)
    D.reprojection_error_sums2_rows(s["camblk"], s["pts4"], s["rows"], s["pi"], uv_d, raw, both)
    torch.cuda.synchronize()
    assert torch.isnan(both).all()
    # an empty list: both sums are zero without a launch, NULL everything else
    from city2ba_amd import _lib as L
    both.fill_(5.0)
    assert L.lib().c2b_reprojection_error_sums2_rows(None, None, None, 0, None, None, None, 0, None, both.data_ptr(), None) == L.OK
    torch.cuda.synchronize()
    assert both.tolist() == [0.0, 0.0]
    assert L.lib().c2b_reprojection_error_sums2_rows(None, None, None, 0, None, None, None, 0, None, None, None) == L.ERR_INVALID_ARGUMENT
    assert L.lib().c2b_add_noise_observations_error_sums2_rows(s["camblk"].data_ptr(), s["pts4"].data_ptr(), s["rows"].row_ptr.data_ptr(),
                                                               s["n_cam"], s["rows"].tiles.data_ptr(), s["pi"].data_ptr(), uv_d.data_ptr(),
                                                               s["n"], 0, -1.0, 3, s["ws"].data_ptr(), both.data_ptr(), None) == L.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("kind,obs_base", [("ragged", 0), ("empties", 1 << 33), ("singles", 12345), ("tiny", 7), ("large", 99)])
def test_observation_noise_fused_with_the_errors(env, kind, obs_base):
    """uv after the fused pass == uv after c2b_add_noise_observations (bit for bit: same draws, same arithmetic), and the
    two sums == c2b_reprojection_error_sums2_rows on that perturbed uv (bit for bit: same grid, same fold)."""
    import oracle as O
    torch, D, dev = env["torch"], env["D"], env["dev"]
    s = _setup(env, kind, seed=23)
    uv_a = torch.from_numpy(s["uv"]).to(dev)
    uv_b = uv_a.clone()
    want, got = (torch.full((2,), -1.0, dtype=torch.float64, device=dev) for _ in range(2))
    D.add_noise_observations(uv_a, obs_base, 0.02, 77)
    D.reprojection_error_sums2_rows(s["camblk"], s["pts4"], s["rows"], s["pi"], uv_a, s["ws"], want)
    D.add_noise_observations_error_sums2_rows(s["camblk"], s["pts4"], s["rows"], s["pi"], uv_b, obs_base, 0.02, 77, s["ws"], got)
    torch.cuda.synchronize()
    assert torch.equal(uv_a.view(torch.int64), uv_b.view(torch.int64))
    assert got[0].item() == want[0].item() and got[1].item() == want[1].item()
    assert not torch.equal(uv_b, torch.from_numpy(s["uv"]).to(dev))
    # zero strength: uv unchanged bit for bit (0.0 + 0 * z added to every component), sums = the plain sums
    uv_c = torch.from_numpy(s["uv"]).to(dev)
    D.add_noise_observations_error_sums2_rows(s["camblk"], s["pts4"], s["rows"], s["pi"], uv_c, obs_base, 0.0, 77, s["ws"], got)
    D.reprojection_error_sums2_rows(s["camblk"], s["pts4"], s["rows"], s["pi"], torch.from_numpy(s["uv"]).to(dev), s["ws"], want)
    torch.cuda.synchronize()
    assert np.array_equal(uv_c.cpu().numpy(), s["uv"]) and torch.equal(got, want)
    # the oracle's draws for the same counters (add_noise with the entity strengths at zero touches only uv)
    _, _, uv0 = O.add_noise(s["P"]["cams15"], s["P"]["pts"], s["uv"], 0.0, 0.0, 0.0, 0.02, seed=77, obs_offset=obs_base)
    assert np.max(np.abs(uv_b.cpu().numpy() - uv0)) < 1e-9


def test_level1_pairs_and_the_fused_tail_of_run_noise(env):
    """c2b_problem_total_reprojection_errors_l1_l2 == the two Level-1 calls (bits); noise.add_noise_with_errors leaves the
    state add_noise leaves (bits) and returns the errors the two calls return on it (bits); the whole chain against the
    oracle's run_noise chain (src/bin/city2ba.rs:305-354) at the tolerances of tests/test_gpu_dist.py."""
    import oracle as O
    import city2ba_amd as c2b
    from city2ba_amd import noise as N
    P = random_problem(260, 3000, 14, seed=314, noise=1e-3)
    mk = lambda: c2b.BAProblem.from_visibility(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"], device=0)
    a, b = mk(), mk()
    l1, l2 = a.total_reprojection_errors_l1_l2()
    assert l1 == a.total_reprojection_error(1.0) and l2 == a.total_reprojection_error(2.0)
    for ba in (a, b):
        N.add_drift_normalized(ba, 0.05, 0.02, 0.1, seed=7)
    N.add_noise(a, 0.05, 0.02, 0.03, 0.01, seed=8)
    want = (a.total_reprojection_error(1.0), a.total_reprojection_error(2.0))
    _, g1, g2 = N.add_noise_with_errors(b, 0.05, 0.02, 0.03, 0.01, seed=8)
    assert (g1, g2) == want
    assert np.array_equal(a.cameras(), b.cameras()) and np.array_equal(a.points(), b.points())
    assert np.array_equal(a.observations(), b.observations())
    assert b.total_reprojection_errors_l1_l2() == want                    # and the state really is what it reports
    c0, p0 = O.add_drift_normalized(P["cams15"], P["pts"], 0.05, 0.02, 0.1, seed=7)
    c0, p0, uv0 = O.add_noise(c0, p0, P["uv"], 0.05, 0.02, 0.03, 0.01, seed=8)
    for got, norm in ((g1, 1.0), (g2, 2.0)):
        ref = O.total_reprojection_error(c0, p0, P["row_ptr"], P["pt_idx"], uv0, norm)
        assert abs(got - ref) / ref < 1e-7
    # an empty problem: zeros, no launch
    e = c2b.BAProblem.from_visibility(P["cams15"][:3], P["pts"], np.zeros(4, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros((0, 2)))
    assert e.total_reprojection_errors_l1_l2() == (0.0, 0.0)
    for ba in (a, b, e):
        ba.close()


def test_level1_sharded_pairs_at_world_size_one(env):
    """the _sharded forms on a one-rank communicator (RCCL through the C ABI): ONE 2-element all-reduce; equal to the
    unsharded pair (errors: bits -- nothing statistical in between; after add_noise: to rounding, the sharded statistics
    take two passes)."""
    import city2ba_amd as c2b
    from city2ba_amd import _lib as L
    from city2ba_amd import comm as Comm
    P = random_problem(310, 2600, 16, seed=99, noise=1e-4)
    mk = lambda: c2b.BAProblem.from_visibility(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"], device=0)
    plain, shard = mk(), mk()
    c = Comm.Comm(Comm.unique_id(), 0, 1, 0)
    lib = L.lib()
    l1, l2 = C.c_double(), C.c_double()
    L.check(lib.c2b_problem_total_reprojection_errors_l1_l2_sharded(shard._h, c.handle, C.byref(l1), C.byref(l2)))
    assert (l1.value, l2.value) == plain.total_reprojection_errors_l1_l2()
    assert lib.c2b_problem_add_noise_errors_l1_l2_sharded(shard._h, c.handle, 0.02, 0.01, 0.03, 0.004, 5, C.byref(l1), C.byref(l2)) == L.ERR_INVALID_ARGUMENT
    L.check(lib.c2b_problem_set_shard(shard._h, 0, 310, 0))
    L.check(lib.c2b_problem_add_noise_errors_l1_l2_sharded(shard._h, c.handle, 0.02, 0.01, 0.03, 0.004, 5, C.byref(l1), C.byref(l2)))
    from city2ba_amd import noise as N
    _, w1, w2 = N.add_noise_with_errors(plain, 0.02, 0.01, 0.03, 0.004, seed=5)
    assert np.array_equal(shard.observations(), plain.observations())
    scale = max(1.0, float(np.max(np.abs(plain.points()))))
    assert np.max(np.abs(shard.cameras() - plain.cameras())) <= 1e-13 * scale
    assert abs(l1.value - w1) <= 1e-9 * w1 and abs(l2.value - w2) <= 1e-9 * w2
    c.destroy()
    plain.close()
    shard.close()


def test_fused_passes_at_the_headline_size(env):
    """synthetic --blocks 128 (19.3 M observations, BASELINE configs[3]'s grid on one device): the fused sums equal the
    separate launches bit for bit at full size, and the fused noise pass leaves the uv of the separate pass."""
    import argparse
    import bench
    torch, D, dev = env["torch"], env["D"], env["dev"]
    sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
    n, rows = sh["n_obs"], sh["rows"]
    assert n == 19_302_494
    ws = D.workspace(n, dev)
    uv = sh["uv"].clone()                                   # exact projections + the bench's own N(0, 1e-3) observation noise
    e1, e2 = (torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(2))
    both, got = (torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(2))
    D.add_noise_observations(uv, 0, 1e-3, 5)
    D.reprojection_error_sum_rows(sh["camblk"], sh["pts4"], rows, sh["pt_idx"], uv, 1.0, ws, e1)
    D.reprojection_error_sum_rows(sh["camblk"], sh["pts4"], rows, sh["pt_idx"], uv, 2.0, ws, e2)
    D.reprojection_error_sums2_rows(sh["camblk"], sh["pts4"], rows, sh["pt_idx"], uv, ws, both)
    uv2 = sh["uv"].clone()
    D.add_noise_observations_error_sums2_rows(sh["camblk"], sh["pts4"], rows, sh["pt_idx"], uv2, 0, 1e-3, 5, ws, got)
    torch.cuda.synchronize()
    assert both[0].item() == e1.item() and both[1].item() == e2.item() and e2.item() > 0.0
    assert torch.equal(got, both) and torch.equal(uv.view(torch.int64), uv2.view(torch.int64))
    # size-independent property: a uniformly directed offset of magnitude N(0, s) has E |offset|^2 = s^2, and two
    # independent ones (the bench's and this test's, other seeds) add: the L2 sum is ~ 2 n s^2
    assert abs(both[1].item() / (2 * n * 1e-6) - 1.0) < 0.01
