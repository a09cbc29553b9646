"""Full-size properties at every `synthetic --blocks B` configuration BASELINE.json names -- configs[1] B = 4
(21 629 observations), configs[2] B = 32 (1.23 M), configs[3] B = 128 (19.3 M observations, 2.8 GB of camera Jacobian:
byte offsets beyond 2^31) -- where the oracle is too slow to run over everything:
  * tiling independence: the first / last 200k observations computed alone equal the same rows of the full launch;
  * r == project - uv (bit-exact) on every observation, via the separate projection kernel;
  * fused error partials == the stand-alone error kernel == sum of r^2 (to rounding);
  * the oracle on a 20k-observation window in the middle of the problem."""
import argparse

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("blocks,n_expected", [(4, 21_629), (32, 1_225_066), (128, 19_302_494)])
def test_synthetic_blocks_properties(blocks, n_expected):
    import __graft_entry__ as entry
    entry.build()
    import torch
    import bench
    from city2ba_amd import device as D
    dev = torch.device("cuda", 0)
    sh = bench.build_shard(argparse.Namespace(blocks=blocks), 0, 1, dev)
    n = sh["n_obs"]
    assert n == n_expected                                     # the generator is deterministic (no RNG)
    assert blocks < 128 or n * 144 > 2 ** 31
    camblk, pts4, ci, pi, uv = sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"]
    r = torch.empty((n, 2), dtype=torch.float64, device=dev)
    Jc = torch.full((n, 18), float("nan"), dtype=torch.float64, device=dev)
    Jp = torch.full((n, 6), float("nan"), dtype=torch.float64, device=dev)
    ws = D.workspace(n, dev)
    e_fused = torch.zeros(1, dtype=torch.float64, device=dev)
    D.residual_jacobian(camblk, pts4, ci, pi, uv, r, Jc, Jp, 2.0, ws)
    D.error_sum_finish(ws, n, e_fused)
    torch.cuda.synchronize()
    # the one-launch form (what bench.py times) writes the same bits, and the same sum on every repetition
    e_one = torch.zeros(1, dtype=torch.float64, device=dev)
    r_b, Jc_b, Jp_b = torch.empty_like(r), torch.full_like(Jc, float("nan")), torch.full_like(Jp, float("nan"))
    for _ in range(3):
        D.residual_jacobian_sum(camblk, pts4, ci, pi, uv, r_b, Jc_b, Jp_b, 2.0, ws, e_one)
        torch.cuda.synchronize()
        assert e_one.item() == e_fused.item()
    assert torch.equal(r_b, r) and torch.equal(Jc_b, Jc) and torch.equal(Jp_b, Jp)
    # ... and so does the row-structure form (the bench step since r02h): the list addressed by row_ptr + tile
    # records instead of cam_idx, here at full size (byte offsets beyond 2^31, tiles with an empty list inside)
    rows = sh["rows"]
    Jc_b.fill_(float("nan")); Jp_b.fill_(float("nan")); r_b.fill_(float("nan"))
    D.residual_jacobian_rows(camblk, pts4, rows, pi, uv, r_b, Jc_b, Jp_b, 2.0, ws, e_one)
    torch.cuda.synchronize()
    assert e_one.item() == e_fused.item()
    assert torch.equal(r_b, r) and torch.equal(Jc_b, Jc) and torch.equal(Jp_b, Jp)
    # r05: the launch INTO a placed output set picks its workgroup shape by the set's store rate (512 threads x two tiles, or
    # 256 threads x one tile into a slow-store set above 6 M observations): same r / Jc / Jp under either, the folded sum
    # to rounding (another grid), and each shape reproduces its own sum
    outs = D.JacobianOutputs(n, dev, max_attempts=1)
    assert (outs.store_GBs > 0.0) == (n >= 1_000_000)               # measured from a million observations on (its value is the device's business)
    sums = {}
    for rate in (5700.0, 6500.0, 7100.0, 0.0):
        outs.set_store_rate(rate)
        assert D.jacobian_launch_shape(n, rate) == ((16, 1) if n < 6_000_000 else ((4, 1) if 0 < rate < 6300 else ((16, 1) if 0 < rate < 6850 else (8, 2))))
        outs.r.fill_(float("nan")); outs.Jc.fill_(float("nan")); outs.Jp.fill_(float("nan"))
        for _ in range(2):
            D.residual_jacobian_rows_placed(camblk, pts4, rows, pi, uv, outs, 2.0, ws, e_one)
            torch.cuda.synchronize()
            assert sums.setdefault(rate, e_one.item()) == e_one.item()
        assert torch.equal(outs.r, r) and torch.equal(outs.Jc, Jc) and torch.equal(outs.Jp, Jp)
        assert abs(e_one.item() - e_fused.item()) <= 1e-13 * e_fused.item()
    assert sums[7100.0] == sums[0.0] == e_fused.item()
    with pytest.raises(Exception):
        D.residual_jacobian_rows_placed(camblk, pts4, rows, pi, uv, D.JacobianOutputs(n + 64, dev, max_attempts=1), 2.0, ws, e_one)
    del outs
    if blocks == 32:
        # every cache policy of the once-read streams writes the same bits.  n_pts only sizes the working set for
        # that choice (c2b_jacobian_stream_policy), so claiming more points than there are selects the other kernels.
        from city2ba_amd import _lib as L
        lib, p = L.lib(), (lambda t: t.data_ptr())
        for fake_n_pts, want in ((sh["n_pts"], 0), (7_600_000, 2), (8_000_000, 3)):
            assert lib.c2b_jacobian_stream_policy(n, rows.n_cam, fake_n_pts) == want
            Jc_b.fill_(float("nan")); Jp_b.fill_(float("nan")); r_b.fill_(float("nan")); e_one.fill_(-1.0)
            assert lib.c2b_residual_jacobian_rows(p(camblk), p(pts4), fake_n_pts, p(rows.row_ptr), rows.n_cam, p(rows.tiles), 0,
                                                  p(pi), p(uv), n, p(r_b), p(Jc_b), p(Jp_b), 2.0, p(ws), p(e_one), None) == L.OK
            torch.cuda.synchronize()
            assert e_one.item() == e_fused.item()
            assert torch.equal(r_b, r) and torch.equal(Jc_b, Jc) and torch.equal(Jp_b, Jp)
    del r_b, Jc_b, Jp_b
    assert bool(torch.isfinite(Jc).all()) and bool(torch.isfinite(Jp).all())      # every row written, incl. the tail

    # r == project - uv, bit-exact, everywhere
    proj = torch.empty((n, 2), dtype=torch.float64, device=dev)
    D.project(camblk, pts4, ci, pi, proj)
    assert torch.equal(r, proj - uv)
    proj_rows = torch.full_like(proj, float("nan"))
    D.project_rows(camblk, pts4, rows, pi, proj_rows)
    assert torch.equal(proj_rows, proj)
    del proj_rows

    # error: fused partials == stand-alone kernel; both == sum r^2 up to summation order
    e_alone = torch.zeros(1, dtype=torch.float64, device=dev)
    D.reprojection_error_sum(camblk, pts4, ci, pi, uv, 2.0, ws, e_alone)
    torch.cuda.synchronize()
    # (the two kernels group observations differently -- 2 vs 3 tiles per wave -- so the sums agree to rounding only)
    assert abs(e_fused.item() - e_alone.item()) <= 1e-13 * e_alone.item()
    e_again = torch.zeros(1, dtype=torch.float64, device=dev)
    D.reprojection_error_sum(camblk, pts4, ci, pi, uv, 2.0, ws, e_again)
    torch.cuda.synchronize()
    assert e_again.item() == e_alone.item()                    # but each kernel reproduces its own bits
    D.reprojection_error_sum_rows(camblk, pts4, rows, pi, uv, 2.0, ws, e_again)
    torch.cuda.synchronize()
    assert e_again.item() == e_alone.item()                    # the row-structure form: same grid, same bits
    assert abs(e_fused.item() - float((r * r).sum().item())) / e_fused.item() < 1e-11

    # tiling independence at both ends (different tile origin => different wave / XCD assignment)
    w = min(200_000, n // 2)
    for lo, hi in ((0, w), (n - w - 3, n)):
        m = hi - lo
        r2 = torch.empty((m, 2), dtype=torch.float64, device=dev)
        Jc2 = torch.empty((m, 18), dtype=torch.float64, device=dev)
        Jp2 = torch.empty((m, 6), dtype=torch.float64, device=dev)
        D.residual_jacobian(camblk, pts4, ci[lo:hi].contiguous(), pi[lo:hi].contiguous(), uv[lo:hi].contiguous(),
                            r2, Jc2, Jp2, 2.0, None)
        torch.cuda.synchronize()
        assert torch.equal(r2, r[lo:hi]) and torch.equal(Jc2, Jc[lo:hi]) and torch.equal(Jp2, Jp[lo:hi])

    # the oracle on a window in the middle
    lo = n // 2
    hi = min(lo + 20_000, n)
    cams15 = sh["cam15"].cpu().numpy()
    pts = np.ascontiguousarray(sh["pts4"][:, :3].cpu().numpy())
    ci_h = ci[lo:hi].cpu().numpy().astype(np.int64)
    c0, c1 = int(ci_h[0]), int(ci_h[-1]) + 1
    counts = np.bincount(ci_h - c0, minlength=c1 - c0)
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    r0, Jc0, Jp0 = O.residual_jacobian(cams15[c0:c1], pts, row_ptr, pi[lo:hi].cpu().numpy().astype(np.uint64),
                                       uv[lo:hi].cpu().numpy())
    assert np.max(np.abs(r[lo:hi].cpu().numpy() - r0)) < 1e-12
    scale = max(1.0, float(np.max(np.abs(Jc0))))
    assert np.max(np.abs(Jc[lo:hi].cpu().numpy() - Jc0)) / scale < 1e-10
    assert np.max(np.abs(Jp[lo:hi].cpu().numpy() - Jp0)) / scale < 1e-10
