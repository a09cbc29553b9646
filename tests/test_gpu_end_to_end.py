"""GPU end-to-end tests that read like the reference's own tests/main.rs: the library tests over
test_grid() = synthetic_grid(10, 20, 3, 5., 1., 1., 1., 10.) (:130-195), test_line (:197-201) and the CLI
tests (:11-63) against the C++ command line.  The `generate`-based CLI tests (:65-128) are in test_gpu_generate.py."""
import os
import subprocess

import numpy as np
import pytest

import oracle as O
from _problems import grid_cameras_points, grid_candidate_pairs

pytestmark = pytest.mark.gpu


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


def make_test_grid(c2b):
    from city2ba_amd import synthetic as S
    return S.synthetic_grid(10, 20, 3, 5.0, 1.0, 1.0, 1.0, 10.0, False)


# ---- library tests (tests/main.rs:134-195) -------------------------------------------------------------------
def test_synthetic_grid_against_oracle_pipeline(c2b):
    """The whole generator (layout -> candidates -> occlusion -> device predicate -> cull) against the same
    pipeline assembled from the oracle + independent checkers: indices bit-exact."""
    from city2ba_amd import synthetic as S
    from city2ba_amd.baproblem import cull_arrays
    ba = S.synthetic_grid(10, 20, 3, 5.0, 1.0, 1.0, 1.0, 10.0, False)
    cams, pts = grid_cameras_points(3, cpb=10, ppb=20, L=5.0)
    ci, pi = S.candidate_pairs(O.centers(cams), pts, 10.0, occlusion=True, block_length=5.0, block_inset=1.0)
    uv, keep = O.visibility_pairs(cams, pts, ci, pi, 10.0)
    ci, pi, uv = ci[keep == 1], pi[keep == 1], uv[keep == 1]
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(ci, minlength=len(cams)))]).astype(np.uint64)
    want = cull_arrays(cams, pts, row_ptr, pi.astype(np.uint64), uv)
    assert ba.num_cameras() == len(want[0]) and ba.num_points() == len(want[1])
    assert np.array_equal(ba.row_ptr, want[2]) and np.array_equal(ba.pt_idx, want[3])      # indices: exact
    assert np.array_equal(ba.observations(), want[4])                                     # k2 == 0: uv exact
    assert np.array_equal(ba.cameras(), want[0]) and np.array_equal(ba.points(), want[1])
    assert ba.num_cameras() > 100 and ba.num_observations() > 5000
    assert ba.total_reprojection_error(2.0) == 0.0                                        # zero by construction
    # cull post-conditions (src/baproblem.rs:425-453)
    assert np.all(np.diff(ba.row_ptr.astype(np.int64)) > 3)
    assert np.all(np.bincount(ba.pt_idx.astype(np.int64), minlength=ba.num_points()) > 1)


def test_visibility_pairs_compact_equals_mask_then_host_compaction(c2b):
    from city2ba_amd import synthetic as S
    cams, pts = grid_cameras_points(3, cpb=10, ppb=20, L=5.0)
    empty = np.zeros(len(cams) + 1, dtype=np.uint64)
    ba = c2b.BAProblem.from_visibility(cams, pts, empty, [], np.zeros((0, 2)))
    ci, pi = S.candidate_pairs(O.centers(cams), pts, 10.0)
    # leave some cameras without candidates (first, last and a run in the middle)
    sel = (ci > 2) & ((ci < 40) | (ci > 55)) & (ci < len(cams) - 3)
    ci, pi = ci[sel], pi[sel]
    uv, keep = ba.visibility_pairs(ci, pi, 10.0)
    k = keep == 1
    want_row = np.concatenate([[0], np.cumsum(np.bincount(ci[k], minlength=len(cams)))]).astype(np.uint64)
    row, kept, uv_c = ba.visibility_pairs_compact(ci, pi, 10.0)
    assert np.array_equal(row, want_row) and np.array_equal(kept, pi[k].astype(np.uint64)) and np.array_equal(uv_c, uv[k])
    assert 0 < k.sum() < len(k)
    # degenerate inputs: no pairs; unsorted cameras are refused
    row0, kept0, _ = ba.visibility_pairs_compact(ci[:0], pi[:0], 10.0)
    assert np.all(row0 == 0) and len(kept0) == 0
    with pytest.raises(c2b.City2baError, match="non-decreasing"):
        ba.visibility_pairs_compact(ci[::-1].copy(), pi[::-1].copy(), 10.0)


def test_normalized_drift(c2b):                      # tests/main.rs:134-141
    ba = make_test_grid(c2b)
    err_start = ba.total_reprojection_error(2.0)
    ba = c2b.noise.add_drift_normalized(ba, 0.1, 0.1, 0.1, seed=1)
    assert ba.total_reprojection_error(2.0) > err_start


def test_noise(c2b):                                 # tests/main.rs:143-150
    ba = make_test_grid(c2b)
    err_start = ba.total_reprojection_error(2.0)
    ba = c2b.noise.add_noise(ba, 0.1, 0.1, 0.1, 0.1, seed=2)
    assert ba.total_reprojection_error(2.0) > err_start


def test_incorrect_correspondences(c2b):             # tests/main.rs:152-159
    ba = make_test_grid(c2b)
    err_start = ba.total_reprojection_error(2.0)
    out = c2b.noise.add_incorrect_correspondences(ba, 0.01, seed=3)
    assert out.total_reprojection_error(2.0) > err_start
    assert out.num_observations() == ba.num_observations() and np.array_equal(out.observations(), ba.observations())


def test_drop_features(c2b):                         # tests/main.rs:161-168
    ba = make_test_grid(c2b)
    err_start = ba.total_reprojection_error(2.0)
    out = c2b.noise.drop_features(ba, 0.1, seed=3)
    assert out.total_reprojection_error(2.0) >= err_start
    want = np.floor(np.diff(ba.row_ptr.astype(np.int64)) * 0.1).astype(np.int64)
    assert np.array_equal(np.diff(out.row_ptr.astype(np.int64)), want)
    assert out.total_reprojection_error(2.0) == 0.0      # kept observations are still exact projections


def test_split_landmarks(c2b):                       # tests/main.rs:170-177
    ba = make_test_grid(c2b)
    err_start = ba.total_reprojection_error(2.0)
    out = c2b.noise.split_landmarks(ba, 0.1, seed=3)
    assert out.total_reprojection_error(2.0) >= err_start
    assert out.num_points() == ba.num_points() + int(0.1 * ba.num_points())
    assert out.total_reprojection_error(2.0) == 0.0      # copies sit on their sources


def test_join_landmarks(c2b):                        # tests/main.rs:179-186
    ba = make_test_grid(c2b)
    err_start = ba.total_reprojection_error(2.0)
    out = c2b.noise.join_landmarks(ba, 0.01, seed=3)
    assert out.total_reprojection_error(2.0) > err_start
    assert int((out.pt_idx != ba.pt_idx).sum()) == int(0.01 * ba.num_points())


def test_sin_noise(c2b):                             # tests/main.rs:188-195
    ba = make_test_grid(c2b)
    err_start = ba.total_reprojection_error(2.0)
    ba = c2b.noise.add_sin_noise(ba, [1.0, 1.0, 0.0], [0.0, 1.0, 0.0], 1.0, 2.0)
    assert ba.total_reprojection_error(2.0) > err_start


def test_line(c2b):                                  # tests/main.rs:197-201
    from city2ba_amd import synthetic as S
    ba = S.synthetic_line(30, 40, 10.0, 1.0, 1.0, 1.0, 10.0, False)
    assert ba.num_cameras() > 20, "num_cameras: %d" % ba.num_cameras()
    assert ba.total_reprojection_error(2.0) == 0.0


def test_file_roundtrip_through_device(c2b, tmp_path):
    """write -> from_file: to_vec / from_vec run on the device; the round trip reproduces the problem to the
    accuracy of to_rodrigues o from_rodrigues (the reference's own write/read loss)."""
    ba = make_test_grid(c2b)
    for name in ("g.bal", "g.bbal"):
        path = tmp_path / name
        ba.write(path)
        back = c2b.BAProblem.from_file(path)
        assert str(back) == str(ba)
        assert np.array_equal(back.row_ptr, ba.row_ptr) and np.array_equal(back.pt_idx, ba.pt_idx)
        assert np.array_equal(back.points(), ba.points()) and np.array_equal(back.observations(), ba.observations())
        assert np.max(np.abs(back.cameras() - ba.cameras())) < 1e-12
        assert back.total_reprojection_error(2.0) < 1e-10
        want = O.camera_to_bal(ba.cameras())                      # oracle's to_vec of the same state
        assert np.max(np.abs(back.cameras_bal() - want)) < 1e-12


# ---- CLI tests (tests/main.rs:11-63) -------------------------------------------------------------------------------
def _run(cli, *args):
    return subprocess.run([cli] + [str(a) for a in args], capture_output=True, text=True, timeout=300)


def test_cli_synthetic_blocks(cli, tmp_path):        # synthetic_blocks, :11-22
    r = _run(cli, "synthetic", tmp_path / "blocks.bbal")
    assert r.returncode == 0, r.stderr
    assert "Bundle Adjustment Problem" in r.stdout
    assert os.path.getsize(tmp_path / "blocks.bbal") > 1000


def test_cli_bal_output(c2b, cli, tmp_path):         # test_bal_output, :24-35
    r = _run(cli, "synthetic", tmp_path / "blocks.bal")
    assert r.returncode == 0, r.stderr
    assert "Bundle Adjustment Problem" in r.stdout
    ba = c2b.BAProblem.from_file(tmp_path / "blocks.bal")
    assert str(ba) in r.stdout                                   # header of the file == Display line
    assert ba.total_reprojection_error(2.0) < 1e-9               # emitted .bal is self-consistent
    # the same problem from the Python host path: identical indices, cameras/points within 1e-6
    from city2ba_amd import synthetic as S
    ref = S.synthetic_grid(10, 10, 5, 20.0, 1.0, 1.0, 1.0, 10.0, False)
    assert np.array_equal(ba.row_ptr, ref.row_ptr) and np.array_equal(ba.pt_idx, ref.pt_idx)
    assert np.allclose(ba.cameras_bal(), ref.cameras_bal(), rtol=0, atol=1e-9)
    assert np.array_equal(ba.points(), ref.points()) and np.array_equal(ba.observations(), ref.observations())


def test_cli_noise_blocks(cli, tmp_path):            # noise_blocks, :37-63
    bbal = tmp_path / "blocks.bbal"
    r = _run(cli, "synthetic", bbal)
    assert r.returncode == 0 and "Bundle Adjustment Problem" in r.stdout
    r = _run(cli, "noise", bbal, tmp_path / "blocks_noised.bbal", "--drift-strength", "0.00001",
             "--mismatch-chance", "0.00001", "--seed", "3")
    assert r.returncode == 0, r.stderr
    assert "Initial error" in r.stdout and "Final error" in r.stdout
    assert "Initial error: 0.00e0 (L1) 0.00e0 (L2)" in r.stdout
    assert os.path.getsize(tmp_path / "blocks_noised.bbal") == os.path.getsize(bbal)


def test_cli_noise_index_corruption_flags(c2b, cli, tmp_path):
    """run_noise's optional passes (src/bin/city2ba.rs:288-303, :341): drop -> cull, join -> cull, split -> cull
    before the arithmetic noise, mismatches after it"""
    src = tmp_path / "g.bbal"
    assert _run(cli, "synthetic", src, "--blocks", "3", "--points-per-block", "20").returncode == 0
    ba = c2b.BAProblem.from_file(src)
    out = tmp_path / "n.bbal"
    r = _run(cli, "noise", src, out, "--drop-features", "0.8", "--split-landmarks", "0.1", "--join-landmarks", "0.05",
             "--mismatch-chance", "0.05", "--seed", "11")
    assert r.returncode == 0, r.stderr
    got = c2b.BAProblem.from_file(out)
    assert ("BA Problem with %d cameras, %d points, %d correspondences" % (got.num_cameras(), got.num_points(), got.num_observations())) in r.stdout
    assert got.num_observations() < ba.num_observations()                    # 20 % dropped, then culled
    assert got.num_points() != ba.num_points()                               # split adds, cull removes
    assert np.all(np.diff(got.row_ptr.astype(np.int64)) > 3)                 # cull post-conditions hold
    assert np.all(np.bincount(got.pt_idx.astype(np.int64), minlength=got.num_points()) > 1)
    l2 = got.total_reprojection_error(2.0)
    assert l2 > 0 and ("Final error: " in r.stdout)                          # joins / mismatches break projections
    # the same pipeline from the Python host mirror with the CLI's seed assignment gives the same problem
    from city2ba_amd import noise as N
    b = N.drop_features(ba, 0.8, seed=11 + 2).cull()
    b = N.join_landmarks(b, 0.1, seed=11 + 3).cull()                         # the reference passes split_landmarks here (:296)
    b = N.split_landmarks(b, 0.1, seed=11 + 4).cull()
    b = N.add_drift_normalized(b, 0.0, 0.0, 0.0, seed=11)
    b = N.add_noise(b, 0.0, 0.0, 0.0, 0.0, seed=12)
    b = N.add_incorrect_correspondences(b, 0.05, seed=11 + 5)
    assert str(b) == str(got)
    assert np.array_equal(b.row_ptr, got.row_ptr) and np.array_equal(b.pt_idx, got.pt_idx)
    assert np.array_equal(b.points(), got.points()) and np.array_equal(b.observations(), got.observations())


def test_cli_noise_sharded_over_gpus_writes_the_same_file(c2b, cli, tmp_path):
    """`noise --gpus N`: run_noise with the problem sharded over GPUs from one process (c2b_comm_init_all, one
    c2b_problem per GPU marked with c2b_problem_set_shard, one host thread per GPU, the Level-1 *_sharded entries).  On
    this one-GPU box N = 1 still takes that whole path -- RCCL communicator, sharded statistics, all-reduced errors --
    and must write what the classic single-GPU path writes: same indices, points / cameras / observations equal to
    rounding (the sharded statistics make two passes with gathered shares, the classic call one pass)."""
    src = tmp_path / "g.bbal"
    assert _run(cli, "synthetic", src, "--blocks", "4").returncode == 0
    flags = ["--drift-strength", "0.001", "--drift-angle", "0.002", "--drift-std", "0.1", "--rotation-std", "0.01",
             "--translation-std", "0.02", "--point-std", "0.03", "--observation-std", "0.004", "--sin-strength", "0.1",
             "--sin-frequency", "2", "--seed", "5"]
    a, b = tmp_path / "classic.bbal", tmp_path / "sharded.bbal"
    ra = _run(cli, "noise", src, a, *flags)
    rb = _run(cli, "noise", src, b, *flags, "--gpus", "1")
    assert ra.returncode == 0 and rb.returncode == 0, (ra.stderr, rb.stderr)
    assert "noise on 1 GPU through RCCL" in rb.stderr

    def report(r):                                                 # RCCL prints its own banner on stdout when it starts
        return [ln for ln in r.stdout.splitlines() if ln.startswith(("Initial error", "BA Problem", "Final error"))]
    assert len(report(ra)) == 3 and report(ra) == report(rb)       # same Initial / Final error lines (2 decimals)
    pa, pb = c2b.BAProblem.from_file(a), c2b.BAProblem.from_file(b)
    assert np.array_equal(pa.row_ptr, pb.row_ptr) and np.array_equal(pa.pt_idx, pb.pt_idx)
    scale = max(1.0, float(np.max(np.abs(pa.points()))))
    assert np.max(np.abs(pa.points() - pb.points())) <= 1e-13 * scale
    assert np.max(np.abs(pa.cameras_bal() - pb.cameras_bal())) <= 1e-12 * scale
    assert np.array_equal(pa.observations(), pb.observations())
    assert not np.array_equal(pa.points(), c2b.BAProblem.from_file(src).points())      # the noise really happened
    # the index-corruption flags go through the two-pass form (initial error before the host passes)
    rc = _run(cli, "noise", src, tmp_path / "c.bbal", "--drop-features", "0.8", "--mismatch-chance", "0.01", "--seed", "7", "--devices", "0")
    rd = _run(cli, "noise", src, tmp_path / "d.bbal", "--drop-features", "0.8", "--mismatch-chance", "0.01", "--seed", "7")
    assert rc.returncode == 0 and rd.returncode == 0, (rc.stderr, rd.stderr)
    assert len(report(rc)) == 3 and report(rc) == report(rd)
    pc, pd = c2b.BAProblem.from_file(tmp_path / "c.bbal"), c2b.BAProblem.from_file(tmp_path / "d.bbal")
    assert np.array_equal(pc.row_ptr, pd.row_ptr) and np.array_equal(pc.pt_idx, pd.pt_idx)
    assert np.array_equal(pc.observations(), pd.observations())
    # a device that does not exist is an error message, not a crash
    re_ = _run(cli, "noise", src, tmp_path / "e.bbal", "--gpus", "9")
    assert re_.returncode != 0 and "not visible" in re_.stderr


def test_cli_synthetic_line_and_errors(cli, tmp_path):
    r = _run(cli, "synthetic-line", tmp_path / "l.bal", "--cameras", "30", "--points", "40", "--length", "10")
    assert r.returncode == 0 and "Bundle Adjustment Problem with" in r.stdout
    r = _run(cli, "synthetic", tmp_path / "bad.bal", "--block-inset", "10", "--block-length", "20")
    assert r.returncode != 0 and "Block inset" in r.stderr            # assert at src/synthetic.rs:177
    r = _run(cli, "synthetic", tmp_path / "bad.xyz")
    assert r.returncode != 0 and "unknown file extension" in r.stderr
    r = _run(cli, "generate", tmp_path / "a.obj", tmp_path / "b.bal")
    assert r.returncode != 0 and "Could not open file" in r.stderr     # src/bin/city2ba.rs:482-485


def test_cli_noise_equals_python_host_path(c2b, cli, tmp_path):
    """Two independent hosts over the same C ABI (C++ CLI, Python mirror), same seed => identical problems
    (run_noise order, src/bin/city2ba.rs:305-340: drift always, optional sin pair, add_noise always)."""
    src = tmp_path / "g.bbal"
    assert _run(cli, "synthetic", src, "--blocks", "3").returncode == 0
    out = tmp_path / "n.bbal"
    r = _run(cli, "noise", src, out, "--drift-strength", "0.001", "--drift-angle", "0.002", "--drift-std", "0.1",
             "--sin-strength", "0.05", "--sin-frequency", "2", "--rotation-std", "0.01", "--translation-std", "0.02",
             "--point-std", "0.03", "--observation-std", "0.004", "--seed", "5")
    assert r.returncode == 0, r.stderr
    got = c2b.BAProblem.from_file(out)
    ba = c2b.BAProblem.from_file(src)
    ba = c2b.noise.add_drift_normalized(ba, 0.001, 0.002, 0.1, seed=5)
    ba = c2b.noise.add_sin_noise(ba, [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], 0.05, 2.0)
    ba = c2b.noise.add_sin_noise(ba, [0.0, 0.0, 1.0], [0.0, 1.0, 0.0], 0.05, 2.0)
    ba = c2b.noise.add_noise(ba, 0.02, 0.01, 0.03, 0.004, seed=6)
    assert np.array_equal(got.points(), ba.points()) and np.array_equal(got.observations(), ba.observations())
    # cameras went through to_vec -> file -> from_vec once more on the CLI side
    assert np.max(np.abs(got.cameras() - ba.cameras())) < 1e-12
    assert np.array_equal(got.cameras_bal(), ba.cameras_bal())
    l2 = ba.total_reprojection_error(2.0)
    assert ("Final error: " in r.stdout) and l2 > 0
    # fixed-drift variant parses and runs
    r2 = _run(cli, "noise", src, tmp_path / "f.bal", "--fixed-drift", "--drift-strength", "1e-6", "--seed", "1")
    assert r2.returncode == 0 and "Final error" in r2.stdout
