"""A performance floor in the GPU suite: refactors cannot regress the hot path silently, and whatever
MI355X the suite runs on reports what it measured.  `synthetic --blocks 128` (19 302 494 observations):

  * the step kernel (residual + 2x(9+3) Jacobian + folded L2 sum, launched into a placed output set so that the launch
    shape follows the set's store rate) <= 1.15 x the launch's algorithmic bytes at the streaming-store rate THIS device
    sustains into THAT set, measured in the same process (the device-independent figure of merit: 1.06-1.105 measured on
    slow-store, mixed and fast-store devices, profiles/r05*_ab_*);
  * both error norms in one pass, caches swept before every launch <= 1.2 x the recorded 117 us;
  * the statistics pass, back to back <= 50 us (25-33 us measured since the compact centre table).

The bounds are regression tripwires with 10-40 % of air, not targets.  The measured figures go to the test log (printed
past pytest's capture) and, when gpurun_out/ exists, to gpurun_out/perf_floor.json."""
import argparse
import json
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _timed(torch, fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3            # us


def test_hot_path_stays_within_its_floors_at_the_headline_size(capsys):
    import __graft_entry__ as entry
    entry.build()
    import torch
    import bench
    from city2ba_amd import device as D
    dev = torch.device("cuda", 0)
    sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
    n = sh["n_obs"]
    assert n == 19_302_494
    ws = D.workspace(n, dev)
    err = torch.zeros(2, dtype=torch.float64, device=dev)
    a = (sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"])
    out = {"n_observations": n, "setup_s": round(sh["setup_s"], 3)}

    # -- the step kernel against this device's own floor ------------------------------------------------------------
    outs = D.JacobianOutputs(n, dev, max_attempts=1)        # the first set the library hands out, whatever its class
    alg = bench.algorithmic_bytes(n, sh["n_cam_local"], sh["n_pts"])
    store_us = _timed(torch, lambda: D.calib_store_pattern(outs.r, outs.Jc, outs.Jp), 10)
    rate = n * 208 / store_us / 1e3                         # GB/s this set takes the kernel's own stores at
    floor_us = alg / rate / 1e3
    # (the fastest of three measurements: a tripwire must not trip on one noisy reading; floor and kernel interleaved)
    step_us = min(_timed(torch, lambda: D.residual_jacobian_rows_placed(*a, sh["uv"], outs, 2.0, ws, err[:1]), 20) for _ in range(3))
    store_us = min(store_us, _timed(torch, lambda: D.calib_store_pattern(outs.r, outs.Jc, outs.Jp), 10))
    rate = n * 208 / store_us / 1e3
    floor_us = alg / rate / 1e3
    shape = D.jacobian_launch_shape(n, outs.store_GBs)
    out["step"] = {"kernel_us": round(step_us, 1), "store_GBs_of_the_set": round(rate, 1), "store_class": bench.store_class([rate]),
                   "launch_shape_threads_x_tiles": [shape[0] * 64, shape[1]], "algorithmic_floor_us": round(floor_us, 1),
                   "kernel_over_floor": round(step_us / floor_us, 4), "frac_of_8TBs": round(alg / step_us / 1e3 / 8000.0, 4)}

    # -- both norms in one pass, cold ---------------------------------------------------------------------------------
    sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
    cold = []
    for _ in range(7):
        sweep.sum()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        D.reprojection_error_sums2_rows(*a, sh["uv"], ws, err)
        e.record()
        torch.cuda.synchronize()
        cold.append(s.elapsed_time(e) * 1e3)
    cold_us = sorted(cold)[3]
    out["error_sums2_rows_L1_and_L2"] = {"us_cold": round(cold_us, 1), "us_back_to_back": round(_timed(torch, lambda: D.reprojection_error_sums2_rows(*a, sh["uv"], ws, err), 20), 1)}

    # -- statistics ---------------------------------------------------------------------------------------------------
    st = torch.empty(20, dtype=torch.float64, device=dev)
    stats_us = _timed(torch, lambda: D.stats(sh["camblk"], sh["pts4"], ws, st, centers=sh["cen4"]), 50)
    out["stats"] = {"us_back_to_back": round(stats_us, 1)}

    with capsys.disabled():
        print("\n[perf floor] " + json.dumps(out))
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "perf_floor.json"), "w") as fh:
            json.dump(out, fh, indent=1)

    assert step_us <= 1.15 * floor_us, out["step"]
    assert cold_us <= 1.2 * 117.0, out["error_sums2_rows_L1_and_L2"]
    assert stats_us <= 50.0, out["stats"]
