#!/usr/bin/env python3
"""bench.py -- the headline measurement: million observations/s of project + 2x(9+3) Jacobian
(+ fused L2 error reduce) on a `city2ba synthetic --blocks B` grid, f64, on N MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole grid: c2b_residual_jacobian_rows (residual, Jc, Jp and
the folded sum of squared residuals, ONE launch; the observation list addressed through its row structure, the
reference's one list per camera) on every rank's shard, then ONE 1-element RCCL all-reduce through this library's C ABI
(N > 1; in line behind the kernel or on its own stream next to the next step's kernel -- both are timed during warm-up
and the faster one runs the timed region, the other is timed right after it; every collective completes inside its loop).
Inputs are resident in HBM before the timed region.  Strong scaling: the same `--blocks 128` problem is sharded
over the ranks by contiguous camera ranges cut on the observation prefix sum (BASELINE.json configs[3]); at
N = 1 one GPU holds all of it.

Only the `cpu_baseline` leg touches oracle/ (rank 0, N = 1, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy
KERNEL_FMT = "k_residual_jacobian_l<2, true, %d, true, %d, %d, true, true, %d>"   # <NORM_2, WITH_ERR, WPB, NT, OPL, MINW, OBUP, CSR, NTL>, as rocprofv3 prints it; WPB, OPL = c2b_jacobian_launch_shape (by the size and by the output set's store rate), NTL = c2b_jacobian_stream_policy of the launch
KERNEL_NAME = KERNEL_FMT % (8, 2, 4, 3)     # the launch the roofline object describes (set in main() from the shard's sizes)


class Watchdog:
    """N > 1: every phase of the run that can block on another rank -- communicator init, the A/B loops, the timed
    region's final synchronize -- calls beat(what) when it starts; a daemon thread checks once a second, and if the last
    beat is older than the limit it prints the rank, what it was waiting for and for how long, then ends the process with
    status 3 (os._exit: a fresh exit, never a re-exec; torch.distributed.run then takes the other ranks down).  The GPU
    calls the main thread blocks in (synchronize, barrier) release the GIL, so the thread runs."""

    def __init__(self, rank, seconds):
        import threading
        self.rank, self.seconds, self.what, self.t = rank, float(seconds), "start", time.monotonic()
        self.limit = self.seconds
        self._stop = threading.Event()
        self._thread = None
        if self.seconds > 0:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def beat(self, what, limit=None):
        """`limit`: this phase's own allowance in seconds (communicator start-up on a cold node can take longer than any
        collective should)"""
        self.what, self.t, self.limit = what, time.monotonic(), (float(limit) if limit else self.seconds)

    def stop(self):
        self._stop.set()

    def _run(self):
        while not self._stop.wait(1.0):
            idle = time.monotonic() - self.t
            if idle > self.limit:
                sys.stderr.write("bench.py watchdog: rank %d made no progress for %.0f s in: %s -- exiting with status 3\n"
                                 % (self.rank, idle, self.what))
                sys.stderr.flush()
                os._exit(3)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--assume-store-GBs", type=float, default=0.0,
                    help="diagnostic: tell the placed launch that its output set streams at this rate instead of the measured one "
                         "(c2b_jacobian_outputs_set_store_rate) -- how the counter passes of live_traffic make their child run the "
                         "parent's kernel instance: under rocprofv3 --pmc the store pattern's own timing is not to be trusted")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two short child passes under rocprofv3 --pmc that measure the step kernel's HBM bytes on this box")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--placement-attempts", type=int, default=48,
                    help="output allocations c2b_jacobian_outputs_alloc may try for streaming-store speed before it keeps "
                         "the best (1 = take the first; the line always ALSO reports the kernel in the first allocation).  "
                         "r05: 48, not 8 -- where in the device memory a set lies decides its store rate, every device mapped "
                         "has 8-12 fast sets among 60 consecutive ones but not always among the first eight "
                         "(profiles/r05ao); the rejects are held only during the search")
    ap.add_argument("--place-inputs", action="store_true",
                    help="also re-place the input arrays by measured kernel time (bench-only experiment, off by default)")
    ap.add_argument("--collective", choices=("auto", "c2b", "torch"), default="auto",
                    help="who issues the per-step all-reduce when N > 1: c2b = RCCL through this library's C ABI "
                         "(c2b_comm_*; what a Rust host would call), torch = torch.distributed; auto = c2b over real "
                         "per-rank GPUs (nccl backend), torch in the shared-GPU gloo rehearsal")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="replay the step (kernel + all-reduce) from a HIP graph (one graph launch instead of a kernel "
                         "launch and a collective enqueue per step).  auto = off: measured at world size 1 on a rank's "
                         "eighth of the problem the graph is 5 us per step SLOWER than eager launches with the C-ABI "
                         "collective (100.2 vs 95.0 us, profiles/r03h_ab_step.txt); kept as an option")
    ap.add_argument("--overlap", choices=("auto", "on", "off"), default="auto",
                    help="N > 1: run the all-reduce of step k on its own stream so that it overlaps the kernel of step "
                         "k + 1 (every step's scalar has its own slot; all collectives complete inside the timed "
                         "region).  auto = time both arrangements during warm-up and run the timed region with the faster "
                         "one (the decision is taken on times all-reduced over the ranks); the line reports both")
    ap.add_argument("--watchdog-seconds", type=float, default=60.0,
                    help="N > 1: a rank that makes no progress for this long (a collective that never completes, a peer that "
                         "died) prints where it was and exits with status 3 instead of hanging the job; 0 disables")
    ap.add_argument("--emulate-allreduce-us", type=float, default=0.0,
                    help="diagnostic (1-GPU boxes): replace the collective by a kernel that spins this many microseconds on "
                         "the collective's stream -- what an N-rank all-reduce of that latency would cost the step in line "
                         "and overlapped (profiles/r03h_ab_step.txt)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path (process group, balanced split, all-reduce) even with one rank: "
                         "how tests exercise the RCCL backend on a 1-GPU box")
    return ap.parse_args()


def build_shard(args, rank, world, dev, balanced=True):
    """This rank's shard of `synthetic --blocks B` (defaults of src/bin/city2ba.rs:113-152), on device -- built ON the
    device (r05): synthetic_grid's layout loops, the candidate search, hits_building and the
    predicate run on the resident problem (c2b_problem_synthetic_grid_layout + c2b_problem_visibility_within_distance:
    src/synthetic.rs:178-299), every rank over the WHOLE grid (tens of milliseconds), so that every rank holds the same
    global row pointer and cuts the same camera ranges from it with no collective (contiguous camera
    ranges split on the observation prefix sum, c2b_partition_cameras); its own range then moves device to device into
    the tensors the Level-0 launchers take (c2b_problem_export_device).  Rounds 1-4 searched the candidates on host
    threads, twice per rank when the split was re-balanced: most of the driver's 20.8 s around a 17-ms timed region."""
    import numpy as np
    import torch
    from city2ba_amd import device as D
    from city2ba_amd import dist as Dist
    from city2ba_amd import synthetic as S

    t_setup = time.perf_counter()
    B, max_dist, L, inset = args.blocks, 10.0, 20.0, 1.0
    ba = S.synthetic_grid(10, 10, B, L, inset, 1.0, 1.0, max_dist, cull=False, device=dev.index, mirror=False)
    n_cam, n_pts, n_obs_total = ba._sizes()
    row_ptr_all = ba._row_ptr                                            # the whole list's row pointer (host, 8 B per camera)
    assert int(row_ptr_all[-1]) == n_obs_total
    bounds = Dist.partition_by_observations(row_ptr_all, world) if (balanced and world > 1) else Dist.camera_count_bounds(n_cam, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    ex = ba.export_device(lo, hi)
    ba.close()
    cam15, pts4, pt_idx, uv, row_ptr = ex["cam15"], ex["pts4"], ex["pt_idx"], ex["uv"], ex["row_ptr"]
    n_obs, obs_base = ex["n_obs"], ex["obs_lo"]
    cen4 = D.centers_table(hi - lo, dev)
    camblk = D.cameras_prepare_state(cam15, centers=cen4)
    # observation noise so that the reduced error is a non-trivial number (seeded, shard-independent: draws keyed by the
    # observation's index in the whole list)
    D.add_noise_observations(uv, obs_base, 1e-3, seed=20243)
    # the list's row structure (what the reference holds: one list per camera) for the *_rows launchers
    rows = D.Rows(row_ptr, n_obs)
    cam_idx = D.expand_rows(row_ptr, n_obs)                               # COO form: the cam_idx launchers, tools/
    torch.cuda.synchronize()
    return dict(camblk=camblk, cen4=cen4, cam15=cam15, pts4=pts4, cam_idx=cam_idx, pt_idx=pt_idx, uv=uv, n_obs=n_obs, rows=rows,
                n_obs_total=n_obs_total, n_cam=n_cam, n_pts=n_pts, n_cam_local=hi - lo, cam_lo=lo, cam_hi=hi, obs_base=obs_base,
                bounds=[int(b) for b in bounds], setup_s=time.perf_counter() - t_setup)


def algorithmic_bytes(n_obs, n_cam, n_pts):
    """SURVEY section 8(d), residual+Jacobian: each camera and point read once, 4-B device indices."""
    return n_obs * (4 + 16 + 16 + 192) + (n_cam + 1) * 8 + n_cam * 72 + n_pts * 24


def store_class(rates_GBs):
    """"slow" = every output set this process tried streams below 6.0 TB/s, "fast" = every one at 6.8 or above, "mixed" =
    both kinds (or in between); None when no set was measured (small problems).  A search that met a fast set first
    stopped there: "fast" then describes the sets it saw."""
    if not rates_GBs:
        return None
    if max(rates_GBs) < 6000.0:
        return "slow"
    return "fast" if min(rates_GBs) >= 6800.0 else "mixed"


def usable_cores():
    """CPUs this process can really use: its affinity mask, capped by the cgroup CPU quota (cpu.max).  The GPU boxes
    show 256 logical CPUs but grant a quota of 16; 256 threads there just get throttled."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    note = "affinity mask: %d CPUs" % n
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period + 0.5))
                if q < n:
                    n, note = q, note + ", cgroup CPU quota: %d" % q
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n), note


def cpu_baseline(sh, seconds):
    """The oracle (C restatement of the reference CPU path + its Jacobian) on this box's host cores, timed inside C
    (oracle.bench_run; no Python in the loop).  Two storage layouts -- "faithful" = the reference's
    Vec<Vec<(usize,(f64,f64))>> with powf per error term (src/baproblem.rs:256-279), "optimised" = flat CSR -- each on one
    thread (a 2 M-observation prefix of the grid) and on all cores (the WHOLE grid, pthreads over equal contiguous camera
    ranges like rayon's par_iter over cameras, src/synthetic.rs:268-269).  Flat keys: `value` is the faithful
    single-thread figure.  Baseline, not target."""
    import numpy as np
    import oracle as O
    n_all, n_cam = sh["n_obs"], sh["n_cam_local"]
    row_ptr = sh["rows"].row_ptr.cpu().numpy().astype(np.uint64)
    pt_idx = sh["pt_idx"].cpu().numpy().astype(np.uint64)
    uv = sh["uv"].cpu().numpy()
    cams15 = sh["cam15"].cpu().numpy()
    pts = np.ascontiguousarray(sh["pts4"][:, :3].cpu().numpy())
    r, Jc, Jp = np.empty((n_all, 2)), np.empty((n_all, 18)), np.empty((n_all, 6))
    cores, cores_note = usable_cores()
    # prefix of whole cameras holding ~2 M observations for the single-thread legs
    c_end = int(np.searchsorted(row_ptr, min(n_all, 2_000_000), side="left"))
    c_end = max(1, min(c_end, n_cam))
    n1 = int(row_ptr[c_end])
    O.lib()
    out = {"unit": "Mobs/s", "cores": 1, "kind": "port"}

    def leg(layout, threads, budget, whole):
        ce, n = (n_cam, n_all) if whole else (c_end, n1)
        args = (cams15[:ce], pts, row_ptr[:ce + 1], pt_idx[:n], uv[:n], r[:n], Jc[:n], Jp[:n])
        O.bench_run(layout, threads, 0.0, *args, max_passes=1)            # touch the output pages once, untimed
        passes, el, _ = O.bench_run(layout, threads, budget, *args)
        return round(n * passes / el / 1e6, 3), "%d cameras / %d observations, %d passes in %.1f s, %d thread%s" % (
            ce, n, passes, el, threads, "" if threads == 1 else "s")

    out["value"], s1 = leg("faithful", 1, seconds * 0.4, False)
    out["value_all_cores"], sa = leg("faithful", cores, seconds * 0.2, True)
    out["cores_all"] = cores
    out["cores_all_note"] = cores_note
    out["value_optimised_1t"], so1 = leg("optimised", 1, seconds * 0.2, False)
    out["value_optimised_all_cores"], soa = leg("optimised", cores, seconds * 0.2, True)
    out["sample"] = ("oracle/ C restatement of project + 2x12 Jacobian + L2 sum on the same grid, timed inside C; faithful "
                     "layout = Vec<Vec<(usize,(f64,f64))>> + powf per term: 1 thread: %s; all cores: %s.  optimised "
                     "(flat CSR) layout: 1 thread: %s; all cores: %s" % (s1, sa, so1, soa))
    try:
        with open("/proc/cpuinfo") as fh:
            out["cpu_model"] = next(line.split(":", 1)[1].strip() for line in fh if line.startswith("model name"))
    except Exception:
        out["cpu_model"] = None
    return out


def generator_and_files(dev_index):
    """The rows of SURVEY section 8 next to the hot path, end to end at the headline size and inside this process (so the HIP
    runtime's start, 60-260 ms of every command-line run, is not in them): `synthetic_grid(--blocks 128)` -- layout,
    candidates, hits_building, predicate, cull, all on the device -- and BAProblem::write / from_file of the result in
    both forms (.bbal words and .bal decimal text assembled / taken apart on the device; the host only moves bytes).
    Wall-clock milliseconds, second of two runs; informational, never `value`."""
    import shutil
    import tempfile
    import time
    import city2ba_amd as c2b
    from city2ba_amd import synthetic as S
    res = {}
    d = tempfile.mkdtemp(prefix="c2b_bench_")
    try:
        g = None
        for _ in range(2):
            if g is not None:
                g.close()
            t = time.perf_counter()
            g = S.synthetic_grid(10, 10, 128, 20.0, 1.0, 1.0, 1.0, 10.0, False, device=dev_index)
            res["synthetic_grid_ms"] = round((time.perf_counter() - t) * 1e3, 1)
        res["cameras_points_observations_after_cull"] = [g.num_cameras(), g.num_points(), g.num_observations()]
        for ext in ("bbal", "bal"):
            path = os.path.join(d, "g128." + ext)
            for _ in range(2):
                if os.path.exists(path):
                    os.remove(path)                                  # (truncating 1 GB of cached pages is not the writer's time)
                t = time.perf_counter()
                g.write(path)
                res[ext + "_write_ms"] = round((time.perf_counter() - t) * 1e3, 1)
            res[ext + "_bytes"] = os.path.getsize(path)
            for _ in range(2):
                t = time.perf_counter()
                back = c2b.BAProblem.from_file(path)
                res[ext + "_read_ms"] = round((time.perf_counter() - t) * 1e3, 1)
                same = back.num_observations() == g.num_observations()
                back.close()
            res[ext + "_read_back_same_size"] = bool(same)
            os.remove(path)
        # run_noise's device flow end to end (src/bin/city2ba.rs:283-354) on that resident problem, ONE number: the two errors before,
        # add_drift_normalized (statistics + drift), add_noise with both errors after (camera table + statistics + entity noise +
        # camera table + the fused observation pass) -- three Level-1 calls, each synchronous; wall clock, second of two rounds
        from city2ba_amd import noise as N
        for k in range(2):
            t = time.perf_counter()
            g.total_reprojection_errors_l1_l2()
            N.add_drift_normalized(g, 1e-4, 1e-5, 1e-3, seed=11 + k)
            N.add_noise_with_errors(g, 1e-3, 1e-4, 1e-3, 1e-3, seed=21 + k)
            res["run_noise_device_flow_us"] = round((time.perf_counter() - t) * 1e6, 1)
        g.close()
    except Exception as exc:                                      # informational: never fails the bench line
        res["failed"] = "%s: %s" % (type(exc).__name__, exc)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return res


def other_configs(dev):
    """BASELINE.json's smaller single-GPU configurations, reported next to the headline (not `value`):
    configs[1] `--blocks 4` project-only (a latency test: 0.6 MB of data) and configs[2] `--blocks 32`
    residual + Jacobian (0.29 GB per launch; fits the 256 MB Infinity Cache)."""
    import argparse as _ap
    import torch
    from city2ba_amd import device as D

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e-3

    res = {}
    s4 = build_shard(_ap.Namespace(blocks=4), 0, 1, dev)
    uv4 = torch.empty_like(s4["uv"])
    t = timed(lambda: D.project_rows(s4["camblk"], s4["pts4"], s4["rows"], s4["pt_idx"], uv4), 500)
    res["blocks4_project_only"] = {"n_observations": s4["n_obs"], "us_per_launch": round(t * 1e6, 2),
                                   "Mobs/s": round(s4["n_obs"] / t / 1e6, 1)}
    # the same launch-bound case replayed from a HIP graph (20 launches per graph): what a caller that loops over
    # small problems gets by capturing its loop -- the kernels are capture-safe (tests/test_gpu_level0.py)
    try:
        side = torch.cuda.Stream(device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(20):
                    D.project_rows(s4["camblk"], s4["pts4"], s4["rows"], s4["pt_idx"], uv4)
        torch.cuda.synchronize()
        tg = timed(g.replay, 50) / 20
        res["blocks4_project_only"]["us_per_launch_in_hip_graph"] = round(tg * 1e6, 2)
    except Exception as exc:                                      # informational
        res["blocks4_project_only"]["us_per_launch_in_hip_graph"] = "failed: %s" % exc
    s32 = build_shard(_ap.Namespace(blocks=32), 0, 1, dev)
    n = s32["n_obs"]
    r = torch.empty((n, 2), dtype=torch.float64, device=dev)
    Jc = torch.empty((n, 18), dtype=torch.float64, device=dev)
    Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ws = D.workspace(n, dev)
    t = timed(lambda: D.residual_jacobian_rows(s32["camblk"], s32["pts4"], s32["rows"], s32["pt_idx"], s32["uv"], r, Jc, Jp,
                                               2.0, ws), 200)
    alg = algorithmic_bytes(n, s32["n_cam"], s32["n_pts"])
    res["blocks32_residual_jacobian"] = {"n_observations": n, "us_per_launch": round(t * 1e6, 2),
                                         "Mobs/s": round(n / t / 1e6, 1), "algorithmic_GB/s": round(alg / t / 1e9, 1)}
    return res


def light_kernels(sh, dev, ws):
    """The other per-observation passes of the hot path on the headline grid, each timed two ways because the two differ
    and both are real: back to back on the same problem (what a solver loop sees: the previous launch left the camera and
    point tables in the 256 MB Infinity Cache) and cold (a 1-GiB read sweeps the caches before every launch: a single
    call on a problem nobody has touched).  Fractions are algorithmic bytes (4-B point index + 16-B
    result or observed uv per observation, every camera and point once) over the 8 TB/s peak."""
    import torch
    from city2ba_amd import device as D
    n, n_cam, n_pts = sh["n_obs"], sh["n_cam_local"], sh["n_pts"]
    uv_out = torch.empty_like(sh["uv"])
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    keep_bits = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.float64, device=dev)
    err2 = torch.zeros(2, dtype=torch.float64, device=dev)
    st = torch.empty(20, dtype=torch.float64, device=dev)
    uv2 = sh["uv"].clone()
    sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
    ent = n_cam * 72 + n_pts * 24
    cases = {
        "project_rows": (lambda: D.project_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], uv_out), n * 20 + ent),
        "error_sum_rows_L2": (lambda: D.reprojection_error_sum_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], 2.0, ws, err), n * 20 + ent),
        # r04: the pair run_noise evaluates back to back (src/bin/city2ba.rs:283-287, 350-354) as ONE pass, and add_noise's
        # observation pass (src/noise.rs:152-170) fused with it
        "error_sums2_rows_L1_and_L2": (lambda: D.reprojection_error_sums2_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], ws, err2), n * 20 + ent),
        "add_noise_observations+error_sums2_rows": (lambda: D.add_noise_observations_error_sums2_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], uv2, 0, 1e-9, 7, ws, err2), n * 36 + ent),
        "visibility_rows": (lambda: D.visibility_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], 10.0, uv_out, keep), n * 21 + ent),
        "visibility_rows_bits": (lambda: D.visibility_rows_bits(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], 10.0, uv_out, keep_bits), n * 20 + n // 8 + ent),
        "add_noise_observations": (lambda: D.add_noise_observations(uv2, 0, 1e-9, 7), n * 32),
        "stats": (lambda: D.stats(sh["camblk"], sh["pts4"], ws, st, centers=sh["cen4"]), n_cam * 24 + n_pts * 24),
    }
    out = {}
    for name, (fn, alg) in cases.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        warm = s.elapsed_time(e) / 20 * 1e-3
        cold = []
        for _ in range(5):
            sweep.sum()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            cold.append(s.elapsed_time(e) * 1e-3)
        cold = sorted(cold)[2]
        out[name] = {"us_back_to_back": round(warm * 1e6, 1), "frac_back_to_back": round(alg / warm / 1e9 / HBM_PEAK_GBS, 3),
                     "us_cold": round(cold * 1e6, 1), "frac_cold": round(alg / cold / 1e9 / HBM_PEAK_GBS, 3)}
        out[name].update(recorded_bound(name))
    return out


# pass of light_kernels -> its kernel in profiles/r06b_light_sq.json (tools/profile_light.sh: rocprofv3 kernel trace, SQ issue /
# wait counters, FETCH_SIZE and WRITE_SIZE, each in its own pass)
_LIGHT_KERNEL = {"project_rows": "k_observations<0, 2, 3, 8, 1, true, true, 0>",
                 "error_sum_rows_L2": "k_observations<1, 2, 3, 8, 1, true, true, 2>",
                 "error_sums2_rows_L1_and_L2": "k_observations<3, 2, 3, 8, 1, true, true, 2>",
                 "add_noise_observations+error_sums2_rows": "k_observations<4, 2, 3, 8, 8, true, true, 2>",
                 "visibility_rows": "k_observations<2, 2, 3, 8, 8, true, true, 1>",
                 "visibility_rows_bits": "k_observations<5, 2, 3, 8, 8, true, true, 1>",
                 "add_noise_observations": "k_add_noise_observations", "stats": "k_stats_pass1<"}


def recorded_bound(name):
    """Which bound binds this pass, from COUNTERS -- a recorded figure of a separate rocprofv3 run of
    these very launches (profiles/r06b_light_sq.json; under the profiler every launch starts from swept caches):
    valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (1 024 SIMDs x 2.4 GHz x duration), valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 /
    the same, hbm_frac = (2 x FETCH_SIZE + WRITE_SIZE) / duration / 8 TB/s, bound = the larger of the last two if it
    reaches one half, else "latency"."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06b_light_sq.json")) as fh:
            d = json.load(fh).get(_LIGHT_KERNEL.get(name, ""), None)
        if not d or "bound" not in d:
            return {}
        return {"recorded_counters": {k: d.get(k) for k in ("bound", "binding_frac", "valu_issue_frac", "valu_busy_frac", "hbm_frac", "avg_us")}}
    except Exception:
        return {}


def adversarial_gather(sh, r, Jc, Jp, ws):
    """grid order is the friendly gather; uniformly random point indices are the adversarial
    one (every lane of a wave hits a different 128-B line of the 63 MB point table).  Timing only."""
    import torch
    from city2ba_amd import device as D
    g = torch.Generator(device=sh["pt_idx"].device)
    g.manual_seed(20244)
    rnd = torch.randint(0, sh["n_pts"], (sh["n_obs"],), dtype=torch.int32, device=sh["pt_idx"].device, generator=g)

    def run():
        D.residual_jacobian_rows(sh["camblk"], sh["pts4"], sh["rows"], rnd, sh["uv"], r, Jc, Jp, 2.0, ws)
    run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        run()
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    return {"us_per_launch": round(t * 1e6, 2), "Mobs/s": round(sh["n_obs"] / t / 1e6, 1)}


def pmc_traffic(kernel_name=None):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 --pmc summary (profiles/pmc_latest*.json,
    produced by tools/profile_bench.sh with the guide's gfx950 FETCH_SIZE correction; one per kernel instance: the 512 x 2
    shape of fast-store sets and the 256 x 1 shape of slow-store ones).  A figure from a SEPARATE run of the same
    command; only reported when a summary names the kernel instance this run launched, and it carries its tag.  With the
    extras on, the line replaces it by this box's own measurement (live_traffic)."""
    import glob
    want = (kernel_name or KERNEL_NAME).replace(" ", "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_latest*.json"))):
        try:
            with open(path) as fh:
                j = json.load(fh)
            profiled = str((j.get("dominant_kernel") or {}).get("name", "")).replace(" ", "")
            if want in profiled and j.get("traffic_bytes_per_launch"):
                return j.get("traffic_bytes_per_launch"), "%s, %s" % (j.get("tag"), os.path.basename(path))
        except Exception:
            continue
    return None, None


def live_traffic(kernel_name, blocks, store_GBs):
    """HBM bytes per launch of the step kernel measured ON THIS BOX, right after the timed region: two child runs of this very
    script (3 steps, no extras) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- each counter in its own pass with
    nothing but --kernel-trace beside it, as the guide prescribes -- and the launches of the instance the timed region ran
    averaged: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the guide's gfx950 correction).  Children, never an exec: this process has
    initialised the GPU.  Anything going wrong (no rocprofv3, a pass failing or timing out, no row of that instance: the child
    may land in an output set of another store class and launch the other shape) returns None and the line falls back to the
    recorded figure.  ~10 s per pass."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3")
    if not tool:
        return None, "no rocprofv3 on PATH"
    want = kernel_name.replace(" ", "")
    got = {}
    base = tempfile.mkdtemp(prefix="c2b_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(base, counter)
            cmd = [tool, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--blocks", str(blocks), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
                   "--placement-attempts", "1", "--assume-store-GBs", "%.1f" % (store_GBs if store_GBs > 0 else 7000.0)]
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
            except subprocess.TimeoutExpired:
                return None, "%s pass timed out" % counter
            if p.returncode != 0:
                return None, "%s pass exited %d" % (counter, p.returncode)
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and want in row.get("Kernel_Name", "").replace(" ", ""):
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, "no %s row of %s (the child launched another instance)" % (counter, kernel_name)
            got[counter] = (sum(vals) / len(vals), len(vals))
    finally:
        shutil.rmtree(base, ignore_errors=True)
    traffic = int((2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024)
    return traffic, "%d + %d launches" % (got["FETCH_SIZE"][1], got["WRITE_SIZE"][1])


def place_inputs(sh, r, Jc, Jp, ws, err):
    """The input arrays come out of build_shard as slices of whatever blocks the caching allocator had at hand; a copy
    in an allocation of its own is sometimes read faster by the very same kernel (r02, docs/log_r01_r03.md: 801 ->
    750 us over uv, camblk, pt_idx).  Greedy and empirical: copy one array, time the kernel, keep the copy
    if the kernel got faster.  Untimed set-up; the log goes into roofline.input_placement."""
    import torch
    from city2ba_amd import device as D

    def kernel_us(reps=6):
        for _ in range(2):
            D.residual_jacobian_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws, err)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            D.residual_jacobian_rows(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws, err)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e3

    if sh["n_obs"] < 1_000_000:
        return {}
    log = {"kernel_us_before": round(kernel_us(), 1)}
    best = log["kernel_us_before"]
    for name in ("uv", "camblk", "pt_idx", "pts4"):
        old = sh[name]
        sh[name] = D.camblk_clone(old) if name == "camblk" else old.clone()
        t = kernel_us()
        if t < best * 0.995:
            log[name] = round(t, 1)
            best = t
        else:
            sh[name] = old
        del old
    log["kernel_us_after"] = round(best, 1)
    return log


def same_run_calibration(n, r, Jc, Jp, dev, alg):
    """What THIS device does for pure streams, measured in this process right after the timed region: the Jacobian
    kernel's store geometry alone (208 B/observation of non-temporal 1-KiB stores, no loads, no arithmetic) and a
    16-B-per-lane copy moving the launch's algorithmic byte count.  Lets a slow box be told from a slow kernel."""
    import torch
    from city2ba_amd import device as D

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e-3
    t_store = timed(lambda: D.calib_store_pattern(r, Jc, Jp))
    numel = int(alg) // 16 // 2 * 2
    src = torch.empty(numel, dtype=torch.float64, device=dev).normal_()
    dst = torch.empty_like(src)
    t_copy = timed(lambda: D.calib_copy(src, dst))
    store_rate = n * 208 / t_store                       # bytes/s this device sustains for the kernel's own stores
    return {"same_run_store_floor_us": round(t_store * 1e6, 2),
            "same_run_store_GBs": round(store_rate / 1e9, 1),
            "same_run_copy_us": round(t_copy * 1e6, 2),
            "same_run_copy_GBs": round(numel * 16 / t_copy / 1e9, 1),
            # the launch's algorithmic bytes at this device's own store rate: what "at the roofline of THIS box" means
            "same_run_algorithmic_floor_us": round(alg / store_rate * 1e6, 2)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    entry.build_hip()
    from city2ba_amd import device as D
    from city2ba_amd import dist as Dist

    rank, local_rank, world = Dist.env_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # Rehearsal knobs (tools/rehearse_multirank.sh): on a 1-GPU box the N-rank code path can be exercised with
    # every rank on GPU 0 and gloo for the two tiny collectives.  Never set by the driver: real runs are one
    # rank per GPU over nccl (= RCCL over xGMI).
    backend = os.environ.get("C2B_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("C2B_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # every rank builds the grid's visibility on its own device and cuts its camera range from the same global row
    # pointer (split on the observation prefix sum): no collective, no second build
    sh = build_shard(args, rank, world, dev)
    bounds = sh["bounds"]
    setup_s = sh["setup_s"]
    if dist_on:
        t = torch.tensor([setup_s], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        setup_s = float(t.item())
    torch.cuda.empty_cache()
    n = sh["n_obs"]
    ws = D.workspace(n, dev)
    err = torch.zeros(1, dtype=torch.float64, device=dev)

    def kernel_us_in(outputs, reps=10):
        """mean duration of the step's kernel writing into the placed set `outputs`, HIP events on the launch stream"""
        for _ in range(3):
            D.residual_jacobian_rows_placed(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], outputs, 2.0, ws, err)
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(reps):
            D.residual_jacobian_rows_placed(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], outputs, 2.0, ws, err)
        e_.record()
        torch.cuda.synchronize()
        return s_.elapsed_time(e_) / reps * 1e3

    # The output arrays.  Where they are allocated changes the kernel's time by up to 20 %, so the
    # line reports BOTH: the kernel in the first allocation the library hands out (what a caller that does not search
    # gets), measured here before anything else, and the timed region in the set c2b_jacobian_outputs_alloc keeps after
    # its bounded search -- the same entry point every caller of the C ABI has.  Untimed set-up.
    torch.cuda.empty_cache()
    first_us, first_GBs = None, None
    if rank == 0 and n >= 1_000_000:
        first = D.JacobianOutputs(n, dev, max_attempts=1)
        first_us, first_GBs = kernel_us_in(first), first.store_GBs
        del first
    # (a rank's shard of a few million observations: its sets' fast class is 6.7-7.0 TB/s, so the search stops at 6.8; ranks that
    # SHARE a device in a rehearsal must not each hold tens of GB of rejects)
    attempts = min(args.placement_attempts, 8) if os.environ.get("C2B_SHARE_GPU") == "1" else args.placement_attempts
    outs = D.JacobianOutputs(n, dev, max_attempts=attempts, fast_store_GBs=7000.0 if n >= 6_000_000 else 6800.0)
    if args.assume_store_GBs > 0.0:
        outs.set_store_rate(args.assume_store_GBs)
    (r, Jc, Jp), placement_log, placement_chosen = (outs.r, outs.Jc, outs.Jp), outs.log, outs.chosen
    input_placement = place_inputs(sh, r, Jc, Jp, ws, err) if args.place_inputs else {}

    # ---- the step -------------------------------------------------------------------------------------------------
    # One launch per step: residual + Jacobian + the folded L2 error sum (in-kernel ticket fold) -> a scalar, then (N > 1)
    # ONE 8-byte all-reduce (src/baproblem.rs:265-279's sum over all cameras).  Two arrangements of the collective exist:
    #   in line      queued behind the kernel on the same stream;
    #   overlapped   on its own stream next to the kernel of the NEXT step -- one event hand-off per step, every step's
    #                scalar in its own slot, every collective complete inside the timed region.
    # Which one is faster depends on what the collective costs next to a CU-saturating kernel on the other ranks, and that
    # cannot be known before it runs on N real GPUs (at world size 1 in line wins by 5 us per step; a spin kernel standing
    # in for a 10-45 us collective says overlapped wins, profiles/r03h, r03z).  So `--overlap auto` MEASURES: both
    # arrangements are timed during warm-up (barrier-bracketed, max over ranks, the decision identical on every rank
    # because it is taken on all-reduced times), the faster one runs the timed region, and the other one is timed again
    # right after it -- the line carries both (config.ms_per_step_in_line / _overlapped) and says which one `value` is.
    dog = Watchdog(rank, args.watchdog_seconds if dist_on else 0.0)
    comm, collective, collective_note, comm_init_ms, rccl_ranks = None, None, None, None, None
    if dist_on:
        want = args.collective
        if want == "auto":
            want = "c2b" if backend == "nccl" else "torch"
        if want == "c2b":
            # RCCL through the C ABI.  The id travels over the process group that already exists; from_process_group
            # raises on EVERY rank together if any rank cannot load RCCL or rank 0 cannot make an id; a rank whose init
            # fails on its own is caught by the gathered flag below (and a rank stuck inside it by the watchdog).
            from city2ba_amd import comm as Comm
            ok, why = 1, ""
            dog.beat("c2b_comm_init_rank (RCCL communicator over %d ranks)" % world, limit=max(300.0, args.watchdog_seconds))
            t_init = time.perf_counter()
            try:
                comm = Comm.Comm.from_process_group(dev_index)
            except Exception as exc:                                  # noqa: BLE001 -- reported in the line
                ok, why = 0, "%s: %s" % (type(exc).__name__, exc)
            comm_init_ms = (time.perf_counter() - t_init) * 1e3
            flag = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if comm is not None:
                    comm.destroy()
                comm, collective_note = None, "c2b communicator unavailable (%s): fell back to torch.distributed" % (why or "another rank failed")
        collective = ("c2b_comm_all_reduce_sum_f64 (%s), 1 x f64 per step" % Comm.backend()) if comm is not None \
            else "%s all_reduce(sum, 1 x f64) per step via torch.distributed" % backend
        # how many ranks the communicator itself says it spans, from every rank (must all equal n_gpus)
        mine = comm.info()[1] if comm is not None else dist.get_world_size()
        rccl_ranks = [None] * world
        dist.all_gather_object(rccl_ranks, int(mine))

    spin_cycles = int(args.emulate_allreduce_us * 2340.0)         # torch.cuda._sleep(100 000) spins 42.8 us on MI355X

    def all_reduce(t):
        if spin_cycles:
            torch.cuda._sleep(spin_cycles)
            return
        if comm is not None:
            comm.all_reduce_sum_(t)
        else:
            Dist.all_reduce_sum_(t)

    on_stream = dist_on and (comm is not None or backend == "nccl")       # gloo stages the scalar through the host
    can_overlap = dist_on and on_stream and args.graph != "on"
    comm_stream = torch.cuda.Stream(device=dev) if can_overlap else None
    n_ab = max(5, min(20, args.steps)) if dist_on else 0
    n_slots = 2 * args.steps + 2 * n_ab + max(args.warmup, 1) + 64
    err_ring = torch.zeros(n_slots if can_overlap else 1, dtype=torch.float64, device=dev)
    handoff = [torch.cuda.Event() for _ in range(n_slots)] if can_overlap else None
    slot = [0]

    def kernel(e):
        # c2b_residual_jacobian_rows_placed: the whole list into the placed output set, the workgroup shape chosen by
        # the store rate c2b_jacobian_outputs_alloc measured for it (c2b_jacobian_launch_shape)
        D.residual_jacobian_rows_placed(sh["camblk"], sh["pts4"], sh["rows"], sh["pt_idx"], sh["uv"], outs, 2.0, ws, e)

    def step_in_line(ev=None):
        if ev is not None:
            ev[0].record()
        kernel(err)
        if ev is not None:
            ev[1].record()
        if dist_on:
            all_reduce(err)
            if ev is not None:
                ev[2].record()                       # after the collective

    def step_overlapped():
        k = slot[0] % n_slots
        slot[0] += 1
        e = err_ring[k:k + 1]
        kernel(e)
        handoff[k].record()
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(handoff[k])
            all_reduce(e)

    def timed_loop(fn, count, what):
        """`count` steps of one arrangement, bracketed like the timed region (barrier + synchronize on both sides);
        returns the MAX over ranks of the wall time in seconds"""
        dog.beat("%s: barrier before %d steps" % (what, count))
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(count):
            if dist_on and (k & 15) == 0:
                dog.beat("%s: enqueueing step %d of %d" % (what, k, count))
            fn()
        dog.beat("%s: waiting for %d enqueued steps (their collectives included) to complete" % (what, count))
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    dog.beat("warm-up")
    for _ in range(max(args.warmup, 1) if dist_on else args.warmup):     # the first collective must not be captured
        step_in_line()
    torch.cuda.synchronize()

    # --graph on: the step replayed from a HIP graph -- ONE graph launch per step instead of a kernel launch plus a
    # collective enqueue.  The kernel's in-launch fold is replay-safe (its counters live in `ws` and every fold leaves them
    # zero); RCCL collectives are capturable.  If capture fails on ANY rank every rank runs eagerly (decision all-reduced).
    # Not the default: at world size 1 a graph launch costs 5 us per step more than the two eager enqueues it replaces.
    graph, graph_note = None, None
    use_graph = args.graph == "on"
    # the collective rides in the graph when it is enqueued on the stream (RCCL, through the C ABI or torch); the gloo
    # rehearsal stages its scalar through the host, so there only the kernel is captured and the all-reduce follows
    # every replay eagerly
    collective_in_graph = on_stream
    if use_graph:
        ok = 1
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    if collective_in_graph or not dist_on:
                        step_in_line()
                    else:
                        kernel(err)
            torch.cuda.synchronize()
            graph = g
        except Exception as exc:                                      # noqa: BLE001
            ok, graph_note = 0, "capture failed (%s: %s): eager steps" % (type(exc).__name__, exc)
        if dist_on:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                graph, graph_note = None, graph_note or "capture failed on another rank: eager steps"
        elif not ok:
            graph = None
        if graph is not None:
            for _ in range(2):
                graph.replay()
                if dist_on and not collective_in_graph:
                    all_reduce(err)
            torch.cuda.synchronize()

    def step_graph():
        graph.replay()
        if dist_on and not collective_in_graph:
            all_reduce(err)

    # which arrangement runs the timed region
    ab = None
    if graph is not None:
        shipped = "hip_graph"
    elif not can_overlap or args.overlap == "off":
        shipped = "in_line"
    elif args.overlap == "on":
        shipped = "overlapped"
    else:                                                         # auto: measure both, keep the faster (same decision on every rank)
        for _ in range(3):                                        # first use of the second stream and its events: untimed
            step_overlapped()
        torch.cuda.synchronize()
        t_a = timed_loop(step_in_line, n_ab, "A/B in line")
        t_b = timed_loop(step_overlapped, n_ab, "A/B overlapped")
        ab = {"steps_each": n_ab, "us_per_step_in_line": round(t_a / n_ab * 1e6, 2), "us_per_step_overlapped": round(t_b / n_ab * 1e6, 2)}
        shipped = "overlapped" if t_b < t_a else "in_line"
    step_of = {"in_line": step_in_line, "overlapped": step_overlapped, "hip_graph": step_graph if graph is not None else None}

    # ---- the timed region: EXACTLY args.steps steps of the shipped arrangement -------------------------------------
    dog.beat("timed region: barrier")
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    events = None
    # N = 1: two HIP event records around the ~700 us kernel of every step (the roofline's kernel time is measured live in
    # the region).  N > 1: none -- three records cost ~13 us of a ~90 us step and stall a two-stream arrangement outright
    # (r03); the per-rank kernel / collective split is measured right after the region instead, on every rank.
    if rank == 0 and not dist_on:
        events = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    if shipped == "in_line":
        for k in range(args.steps):
            if dist_on and (k & 15) == 0:
                dog.beat("timed region (%s): enqueueing step %d of %d" % (shipped, k, args.steps))
            step_in_line(events[k] if events is not None else None)
    else:
        fn = step_of[shipped]
        for k in range(args.steps):
            if dist_on and (k & 15) == 0:
                dog.beat("timed region (%s): enqueueing step %d of %d" % (shipped, k, args.steps))
            fn()
    dog.beat("timed region (%s): waiting for %d enqueued steps (their collectives included) to complete" % (shipped, args.steps))
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- after the region --------------------------------------------------------------------------------------------
    # (1) the OTHER arrangement, the same number of un-instrumented steps, bracketed the same way
    other = {}
    if dist_on and graph is None:
        other["in_line"] = elapsed if shipped == "in_line" else timed_loop(step_in_line, args.steps, "in-line loop after the region")
        if can_overlap:
            other["overlapped"] = elapsed if shipped == "overlapped" else timed_loop(step_overlapped, args.steps, "overlapped loop after the region")
    # (2) the kernel / collective split on EVERY rank: instrumented in-line steps (events would distort the timed loops)
    kernel_us_mine, allreduce_us_mine, graph_step_us = None, None, None
    if dist_on or graph is not None:
        post = min(args.steps, 20)
        dog.beat("instrumented pass: %d in-line steps with HIP events on every rank" % post)
        pev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(post)]
        for k in range(post):
            step_in_line(pev[k])
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        kernel_us_mine = sum(ev[0].elapsed_time(ev[1]) for ev in pev) / post * 1e3
        if dist_on:
            allreduce_us_mine = sum(ev[1].elapsed_time(ev[2]) for ev in pev) / post * 1e3
        if rank == 0:
            events = pev
    else:
        step_in_line()                                            # N = 1: leave the scalar of one more launch in `err`
        torch.cuda.synchronize()
    total_err = Dist.finish_error(err.item(), 2.0)          # every arrangement reduced the same data: `err` holds the in-line sum
    # every overlapped step's slot must hold exactly that sum, bit for bit
    ring_ok = bool((err_ring[:min(slot[0], n_slots)] == err).all().item()) if (can_overlap and slot[0]) else None
    per_rank = [{"n_obs": n, "kernel_us": kernel_us_mine, "allreduce_us": allreduce_us_mine,
                 "store_GBs_kept": (placement_log[placement_chosen] if placement_log and 0 <= placement_chosen < len(placement_log) else None),
                 "ring_ok": ring_ok}]
    if dist_on:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"n_obs": n, "kernel_us": kernel_us_mine, "allreduce_us": allreduce_us_mine,
                                          "store_GBs_kept": (placement_log[placement_chosen] if placement_log and 0 <= placement_chosen < len(placement_log) else None),
                                          "ring_ok": ring_ok})
    per_rank_obs = [q["n_obs"] for q in per_rank]
    dog.stop()

    if rank == 0:
        kern_ms = sorted(ev[0].elapsed_time(ev[1]) for ev in events)
        step_us = elapsed / args.steps * 1e6
        step_breakdown = {"kernel_us_rank0": round(sum(kern_ms) / len(kern_ms) * 1e3, 2)}
        if dist_on:
            ar_us = per_rank[0]["allreduce_us"]
            step_breakdown["allreduce_us"] = round(ar_us, 2)
            # every rank's own figures (instrumented in-line steps right after the region): a slow device among the N --
            # one whose outputs stream at 5.7 instead of 7 TB/s paces every step -- is named here
            step_breakdown["kernel_us_per_rank"] = [round(q["kernel_us"], 2) for q in per_rank]
            step_breakdown["allreduce_us_per_rank"] = [round(q["allreduce_us"], 2) for q in per_rank]
            step_breakdown["store_GBs_kept_per_rank"] = [q["store_GBs_kept"] for q in per_rank]
            step_breakdown["rccl_ranks"] = rccl_ranks
            step_breakdown["comm_init_ms"] = round(comm_init_ms, 1) if comm_init_ms is not None else None
            step_breakdown["arrangement"] = shipped
            step_breakdown["arrangement_chosen_by"] = ("--graph on" if shipped == "hip_graph" else
                                                        "--overlap %s" % args.overlap if (args.overlap != "auto" or not can_overlap) else
                                                        "A/B during warm-up (the faster of the two, decided on times all-reduced over the ranks)")
            step_breakdown["ab_during_warmup"] = ab
            for name, secs in other.items():
                step_breakdown["ms_per_step_" + name] = round(secs / args.steps * 1e3, 5)
            # overlapped: the collective is off the kernel stream's critical path, so what a step costs beyond its kernel
            # is launch gaps, the hand-off and waiting for the slowest rank; in line: kernel + collective + the rest
            slowest = max(q["kernel_us"] for q in per_rank)
            step_breakdown["step_overhead_us"] = round(step_us - slowest - (0.0 if shipped == "overlapped" else ar_us), 2)
            step_breakdown["allreduce_overlaps_next_kernel"] = shipped == "overlapped"
            if can_overlap:
                oks = [q["ring_ok"] for q in per_rank]
                step_breakdown["overlapped_sums_equal_the_in_line_sum"] = (all(o for o in oks if o is not None) if any(o is not None for o in oks) else None)
            step_breakdown["watchdog_seconds"] = args.watchdog_seconds
        else:
            step_breakdown["step_overhead_us"] = round(step_us - step_breakdown["kernel_us_rank0"], 2)
        kern_avg_s = sum(kern_ms) / len(kern_ms) / 1e3
        n_total = sh["n_obs_total"]
        value = n_total * args.steps / elapsed / 1e6
        alg = algorithmic_bytes(n, sh["n_cam_local"], sh["n_pts"])
        achieved = alg / kern_avg_s / 1e9
        policy = D.jacobian_stream_policy(n, sh["n_cam_local"], sh["n_pts"])
        shape = D.jacobian_launch_shape(n, outs.store_GBs)
        kernel_name = KERNEL_FMT % (shape[0], shape[1], 4 if shape[1] == 2 else 1, policy)     # (MINW: 4 with two tiles per wave, capi.hip: launch_jac_l)
        traffic, traffic_tag = pmc_traffic(kernel_name) if (world == 1 and args.blocks == 128) else (None, None)
        out = {
            "metric": "million observations/sec (project+Jacobian)",
            "value": round(value, 3), "unit": "Mobs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "city2ba synthetic --blocks %d (cpb=10 ppb=10 block-length=20 inset=1 max-dist=10): "
                            "residual + 2x(9+3) Jacobian + fused L2 error reduce in ONE launch, f64; 1-scalar all-reduce "
                            "when N>1" % args.blocks,
                "blocks": args.blocks, "n_cameras": sh["n_cam"], "n_points": sh["n_pts"],
                "n_observations": n_total,
                # set-up (untimed): the whole grid's layout + visibility loop on the device and this rank's range exported
                # to the Level-0 tensors, max over ranks
                "setup_s": round(setup_s, 3),
                "occlusion": True, "cull": False,
                "sharding": "contiguous camera ranges balanced on the observation prefix sum (c2b_partition_cameras), "
                            "points replicated, outputs sharded",
                "observations_per_rank": [int(x) for x in per_rank_obs],
                "camera_bounds": [int(x) for x in bounds],
                "total_L2_error": total_err,
                "collective": collective, "collective_note": collective_note,
                "hip_graph": ("one graph launch per step (%s captured)" % ("kernel + all-reduce" if collective_in_graph else "kernel")
                              + ("; step on the GPU %.2f us" % graph_step_us if graph_step_us else ""))
                if graph is not None else (graph_note or False),
                "kernel_time_source": "HIP events around the launch inside the timed region" if not dist_on else
                                      "HIP events around %d instrumented in-line steps right after the timed region, on every "
                                      "rank (event records inside a ~90 us step would cost ~13 us of it)" % min(args.steps, 20),
                # where rank 0's step goes (HIP events around the kernel and around the collective; the rest of the
                # wall-clock step is launch gaps, host dispatch and waiting for the slowest rank)
                **step_breakdown,
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                "launch_shape": "%d threads per workgroup, %d tile%s of 64 observations per wave (c2b_jacobian_launch_shape: by the "
                                "launch's size and the %s GB/s the output set takes streaming stores at; below 6 300 GB/s 256 x 1, below 6 850 GB/s 1 024 x 1: the finer "
                                "grain is 1.6-4.4 %% faster, profiles/r05_ab_slow_store.txt, r05p_ab_step_three_classes.txt)" % (
                                    shape[0] * 64, shape[1], "" if shape[1] == 1 else "s", ("%.0f" % outs.store_GBs) if outs.store_GBs else "unmeasured"),
                "stream_policy": {0: "every load cached", 2: "observed uv non-temporal", 3: "observed uv and point index non-temporal"}[policy]
                + " (c2b_jacobian_stream_policy: tables and streams of this launch against the 256 MB Infinity Cache)",
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                # HBM bytes per launch from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction).
                # Counters cannot be collected inside a timed run: this is the figure of a SEPARATE rocprofv3 --pmc pass of
                # this very command (tools/profile_bench.sh), recorded in profiles/ -- NOT measured in this run
                "traffic": traffic,
                "traffic_recorded_at": ("profiles/ (tag %s): separate rocprofv3 --pmc pass of `python bench.py` on "
                                        "the kernel instance this run launched; a recorded figure, not this run's (replaced below by "
                                        "this box's own measurement when the extras run)" % traffic_tag)
                if traffic is not None else None,
                "algorithmic_bytes_per_launch": alg, "observations_per_launch": n,
                "bytes_per_observation": round(alg / max(n, 1), 2),
                "kernel_avg_us": round(kern_avg_s * 1e6, 2), "kernel_min_us": round(kern_ms[0] * 1e3, 2),
                # MI355X devices differ in how fast they take streaming stores (5.6-5.9 TB/s on some,
                # 7.0-7.1 on others, both inside one process on a third kind) and this launch is 85 % stores: the class of
                # the device this line was measured on, from the store rates of the output sets the placement search tried
                "device_store_class": store_class(placement_log),
                "kept_set_store_GBs": (placement_log[placement_chosen] if placement_log and 0 <= placement_chosen < len(placement_log) else None),
                #  --assume-store-GBs replaces the rate the launch shape is chosen by: said so, next to the measured one
                **({"store_GBs_assumed_for_the_launch_shape": args.assume_store_GBs} if args.assume_store_GBs > 0.0 else {}),
                # the same launch writing into the FIRST allocation the library hands out (no search): what a caller
                # that passes max_attempts = 1 gets on this device
                "kernel_us_first_allocation": round(first_us, 2) if first_us else None,
                "store_GBs_first_allocation": round(first_GBs, 1) if first_GBs else None,
                "frac_first_allocation": round(alg / (first_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if first_us else None,
                "output_placement": {"store_GBs_per_attempt": placement_log, "attempts": max(1, len(placement_log)),
                                     "note": "c2b_jacobian_outputs_alloc (C ABI): r/Jc/Jp allocations tried until the "
                                             "store pattern streams >= 7.0 TB/s, at most --placement-attempts; untimed "
                                             "set-up; `achieved` / `frac` / `value` are measured in the set it kept"},
                "input_placement": input_placement,
            },
        }
        if not dist_on:
            # The honest triple.  `value` is the timed region: steps back to back on one problem (the camera and point
            # records, 232 MB at --blocks 128, are found in the 256 MB Infinity Cache) writing into the output set the
            # placement search kept.  Beside it: the same launch (a) in the FIRST allocation the library hands out -- what
            # a caller that does not search gets on this device -- and (b) with the caches swept by a 1-GiB read before it
            # -- what a single call on a problem nobody has touched costs.  Each as a kernel time, a fraction of the HBM
            # peak, and a whole-step rate (that kernel time + this run's per-step overhead).
            sweep = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
            cold = []
            for _ in range(5):
                sweep.sum()
                ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
                step_in_line(ev)                             # events 0 -> 1 bracket the kernel
                torch.cuda.synchronize()
                cold.append(ev[0].elapsed_time(ev[1]))
            del sweep
            cold_us = sorted(cold)[2] * 1e3
            over_us = max(0.0, step_us - kern_avg_s * 1e6)
            out["roofline"]["kernel_us_caches_swept_before_launch"] = round(cold_us, 2)
            out["roofline"]["frac_caches_swept"] = round(alg / (cold_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            out["value_caches_swept"] = round(n_total / ((cold_us + over_us) * 1e-6) / 1e6, 3)
            out["value_first_allocation"] = round(n_total / ((first_us + over_us) * 1e-6) / 1e6, 3) if first_us else None
            out["value_note"] = ("value: timed region, outputs in the allocation c2b_jacobian_outputs_alloc kept, steps back to back; "
                                 "value_first_allocation: the same launch in the first allocation handed out (no search); "
                                 "value_caches_swept: the same launch after a 1-GiB read swept the caches; the last two = "
                                 "observations / (that kernel's time + this run's per-step overhead of %.1f us)" % over_us)
        if world == 1 and not args.no_extras:
            cal = same_run_calibration(n, r, Jc, Jp, dev, alg)
            out["roofline"].update(cal)
            # THE device-independent figure of merit: this kernel against the launch's algorithmic bytes at the store rate
            # this very device sustains in this very output set (1.0 = every byte at this box's own streaming rate)
            out["roofline"]["kernel_over_same_run_algorithmic_floor"] = round(kern_avg_s * 1e6 / cal["same_run_algorithmic_floor_us"], 4)
            out["roofline"]["kernel_over_same_run_store_floor"] = round(kern_avg_s * 1e6 / cal["same_run_store_floor_us"], 4)
            # the same-run copy moves exactly the launch's algorithmic byte count (read + write)
            out["roofline"]["kernel_over_same_run_copy"] = round(kern_avg_s * 1e6 / cal["same_run_copy_us"], 4)
        #  under a profiler the child rocprofv3 runs would nest: skipped, the recorded figure stays
        under_profiler = any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if under_profiler:
            out["roofline"]["traffic_live_unavailable"] = "this run is itself under a profiler"
        if world == 1 and args.blocks == 128 and not args.no_extras and not args.no_live_traffic and not under_profiler:
            # the counters cannot be collected inside the timed run, but they can be collected on THIS box right after it
            t_live, how = live_traffic(kernel_name, args.blocks, outs.store_GBs)
            out["roofline"]["traffic_recorded"] = traffic
            if t_live is not None:
                out["roofline"]["traffic"] = t_live
                out["roofline"]["traffic_over_algorithmic"] = round(t_live / alg, 4)
                out["roofline"]["traffic_recorded_at"] = ("measured on this box right after the timed region: child runs of this script under "
                                                          "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, %s of this kernel "
                                                          "instance; (2 x FETCH_SIZE + WRITE_SIZE) x 1024)" % how)
            else:
                out["roofline"]["traffic_live_unavailable"] = how
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sh, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        if world == 1 and args.blocks == 128 and not args.no_extras:
            out["other_configs"] = other_configs(dev)
            out["other_configs"]["blocks128_uniform_random_point_gather"] = adversarial_gather(sh, r, Jc, Jp, ws)
            out["other_configs"]["blocks128_other_passes"] = light_kernels(sh, dev, ws)
            out["other_configs"]["blocks128_generator_and_files"] = generator_and_files(dev.index)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        if comm is not None:
            comm.destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
