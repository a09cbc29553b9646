"""A performance floor in the GPU suite (tightened per VERDICT r05 item 8 / ADVICE r05): refactors cannot
regress the hot path silently, and whatever MI355X the suite runs on reports what it measured.  `synthetic --blocks 128`
(19 302 494 observations):

  * the step kernel (residual + 2x(9+3) Jacobian + folded L2 sum, launched into a placed output set so that the launch
    shape follows the set's store rate) <= 1.10 x the launch's algorithmic bytes at the streaming-store rate THIS device
    sustains into THAT set, measured in the same process (the device-independent figure of merit: 1.05-1.08 measured on
    slow-store, mixed and fast-store devices over rounds 5-6; 1.062 / 1.064 / 1.074 / 1.082 on r06's four boxes);
  * a rank's EIGHTH of the list (2.41 M observations, what a rank of the 8-GPU run launches) <= 112 us (92-110 measured over
    rounds 5-6 by the class of its 0.5-GB output set; 97.4 us into a 5.5 TB/s set in r06).  Absolute on purpose: at this size
    ramp, tail and fold are a tenth of the launch and neither the store pattern of the same size (0.89 x ... 1.19 x) nor a
    copy is a stable yardstick;
  * both error norms in one pass, caches swept before every launch: <= 120 us (110-117 measured on every device of rounds
    5-6, whatever its store class) AND <= 0.85 x a same-process streaming copy of the pass's algorithmic byte count (the
    device-relative form ADVICE r05 asked for; the copy reads and writes those bytes: 0.72 measured);
  * the statistics pass, back to back <= 29 us (25-26 us measured since the compact centre table).

Tripwires with 4-10 % of air over what five rounds of devices measured -- a 5 % regression of the step kernel trips --,
not targets.  Every figure is the best of three interleaved measurements, so one noisy reading does not trip them.  The
measured figures go to the test log (printed past pytest's capture) and, when gpurun_out/ exists, to
gpurun_out/perf_floor.json."""
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
    # (the fastest of three measurements: a tripwire must not trip on one noisy reading; floor and kernel interleaved.  Slow sets
    # wander by 1-2 % between readings -- 1.062 ... 1.082 over this round's devices --, so a reading above the bound is taken
    # again, twice at most, before it counts: the wire is for an excess that persists)
    step_us = min(_timed(torch, lambda: D.residual_jacobian_rows_placed(*a, sh["uv"], outs, 2.0, ws, err[:1]), 20) for _ in range(3))
    store_us = min(store_us, _timed(torch, lambda: D.calib_store_pattern(outs.r, outs.Jc, outs.Jp), 10))
    for _ in range(2):
        if step_us <= 1.10 * alg / (n * 208 / store_us / 1e3) / 1e3:
            break
        store_us = min(store_us, _timed(torch, lambda: D.calib_store_pattern(outs.r, outs.Jc, outs.Jp), 10))
        step_us = min(step_us, min(_timed(torch, lambda: D.residual_jacobian_rows_placed(*a, sh["uv"], outs, 2.0, ws, err[:1]), 20) for _ in range(3)))
    rate = n * 208 / store_us / 1e3
    floor_us = alg / rate / 1e3
    shape = D.jacobian_launch_shape(n, outs.store_GBs)
    out["step"] = {"kernel_us": round(step_us, 1), "store_GBs_of_the_set": round(rate, 1), "store_class": bench.store_class([rate]),
                   "launch_shape_threads_x_tiles": [shape[0] * 64, shape[1]], "algorithmic_floor_us": round(floor_us, 1),
                   "kernel_over_floor": round(step_us / floor_us, 4), "frac_of_8TBs": round(alg / step_us / 1e3 / 8000.0, 4)}

    # -- a rank's eighth of the list: the launch the 8-GPU run lives on ---------------------------------------------------
    sh8 = bench.build_shard(argparse.Namespace(blocks=128), 0, 8, dev)
    n8 = sh8["n_obs"]
    ws8 = D.workspace(n8, dev)
    outs8 = D.JacobianOutputs(n8, dev, max_attempts=1)
    a8 = (sh8["camblk"], sh8["pts4"], sh8["rows"], sh8["pt_idx"])
    alg8 = bench.algorithmic_bytes(n8, sh8["n_cam_local"], sh8["n_pts"])
    store8 = min(_timed(torch, lambda: D.calib_store_pattern(outs8.r, outs8.Jc, outs8.Jp), 20) for _ in range(3))
    step8 = min(_timed(torch, lambda: D.residual_jacobian_rows_placed(*a8, sh8["uv"], outs8, 2.0, ws8, err[:1]), 50) for _ in range(3))
    floor8 = alg8 / (n8 * 208 / store8 / 1e3) / 1e3
    shape8 = D.jacobian_launch_shape(n8, outs8.store_GBs)
    out["step_rank_eighth"] = {"n_observations": n8, "kernel_us": round(step8, 1), "store_GBs_of_the_set": round(n8 * 208 / store8 / 1e3, 1),
                               "launch_shape_threads_x_tiles": [shape8[0] * 64, shape8[1]], "algorithmic_floor_us": round(floor8, 1),
                               "kernel_over_floor": round(step8 / floor8, 4)}
    del outs8, sh8

    # -- both norms in one pass, cold ---------------------------------------------------------------------------------
    sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)

    def cold_median(fn):
        t = []
        for _ in range(7):
            sweep.sum()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            t.append(s.elapsed_time(e) * 1e3)
        return sorted(t)[3]

    cold_us = min(cold_median(lambda: D.reprojection_error_sums2_rows(*a, sh["uv"], ws, err)) for _ in range(3))
    # the same-process yardstick: a 16-byte-per-lane streaming copy of the pass's algorithmic bytes (20 B per observation + tables)
    light_bytes = (n * 20 + sh["n_cam_local"] * 72 + sh["n_pts"] * 24) // 16 * 16
    src = torch.empty(light_bytes // 8, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    copy_us = min(_timed(torch, lambda: D.calib_copy(src, dst), 20) for _ in range(3))
    out["error_sums2_rows_L1_and_L2"] = {"us_cold": round(cold_us, 1), "us_back_to_back": round(_timed(torch, lambda: D.reprojection_error_sums2_rows(*a, sh["uv"], ws, err), 20), 1),
                                         "copy_of_its_algorithmic_bytes_us": round(copy_us, 1), "cold_over_copy": round(cold_us / copy_us, 3)}

    # -- statistics ---------------------------------------------------------------------------------------------------
    st = torch.empty(20, dtype=torch.float64, device=dev)
    stats_us = min(_timed(torch, lambda: D.stats(sh["camblk"], sh["pts4"], ws, st, centers=sh["cen4"]), 50) for _ in range(3))
    out["stats"] = {"us_back_to_back": round(stats_us, 1)}

    with capsys.disabled():
        print("\n[perf floor] " + json.dumps(out))
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "perf_floor.json"), "w") as fh:
            json.dump(out, fh, indent=1)

    assert step_us <= 1.10 * floor_us, out["step"]
    assert step8 <= 112.0, out["step_rank_eighth"]
    assert cold_us <= 120.0 and cold_us <= 0.85 * copy_us, out["error_sums2_rows_L1_and_L2"]
    assert stats_us <= 29.0, out["stats"]
