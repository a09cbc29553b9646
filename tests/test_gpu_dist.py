"""bench.py's multi-rank code path on a ONE-GPU box (the 8-GPU scaling run is the driver's):
  * world size 1 over nccl (= RCCL): the process group, the observation-balanced split and a real
    RCCL all_reduce per step -- the reduced scalar must equal the plain single-process run's bit for bit;
  * two ranks sharing GPU 0 over gloo, launched by torch.distributed.run exactly like the driver launches N ranks:
    observation-balanced camera ranges, every rank exits 0 (round 1's bench crashed on ranks >= 1 after the timed
    loop), the sharded total equals the single-rank total."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--blocks", "32", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, extra_env=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "C2B_DIST_BACKEND", "C2B_SHARE_GPU"):
        env.pop(k, None)
    env.update(extra_env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


@pytest.fixture(scope="module")
def plain():
    rc, out, err = _run([sys.executable, "bench.py", "--gpus", "1"] + COMMON)
    assert rc == 0 and out is not None, err[-2000:]
    assert out["n_gpus"] == 1 and out["config"]["collective"] is None
    return out


@pytest.mark.parametrize("collective,graph,overlap", [("auto", "auto", "auto"), ("c2b", "off", "off"), ("c2b", "on", "off"),
                                                      ("torch", "on", "off"), ("torch", "off", "on")])
def test_rccl_world_size_one_reproduces_the_plain_run(plain, collective, graph, overlap):
    """the multi-rank step as the driver's N-GPU run takes it (auto = RCCL through the C ABI, eager launches, the
    arrangement of the collective chosen by an A/B during warm-up) and four other combinations of who issues the
    collective, HIP-graph replay and overlap: same reduced scalar, bit for bit, as the run without any collective; and
    the line explains itself -- per-rank kernel / collective times, the store rate of every rank's kept output set, how
    many ranks the communicator spans, both arrangements timed in the same run"""
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}
    rc, out, err = _run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--collective", collective,
                         "--graph", graph, "--overlap", overlap] + COMMON, env)
    assert rc == 0 and out is not None, err[-2000:]
    assert "Traceback" not in err, err[-2000:]
    cfg = out["config"]
    assert cfg["collective_note"] is None, cfg["collective_note"]
    if collective in ("auto", "c2b"):
        assert cfg["collective"].startswith("c2b_comm_all_reduce_sum_f64 (RCCL "), cfg["collective"]
    else:
        assert cfg["collective"].startswith("nccl all_reduce"), cfg["collective"]
    if graph == "on":
        assert isinstance(cfg["hip_graph"], str) and cfg["hip_graph"].startswith("one graph launch per step"), cfg["hip_graph"]
    else:
        assert cfg["hip_graph"] is False
    if graph == "on":
        assert cfg["arrangement"] == "hip_graph" and cfg["allreduce_overlaps_next_kernel"] is False
    elif overlap == "auto":                                       # measured, not assumed: both arrangements timed during warm-up
        ab = cfg["ab_during_warmup"]
        assert ab["us_per_step_in_line"] > 0 and ab["us_per_step_overlapped"] > 0 and ab["steps_each"] >= 5
        assert cfg["arrangement"] == ("overlapped" if ab["us_per_step_overlapped"] < ab["us_per_step_in_line"] else "in_line")
        assert cfg["arrangement_chosen_by"].startswith("A/B during warm-up")
    else:
        assert cfg["arrangement"] == ("overlapped" if overlap == "on" else "in_line") and cfg["ab_during_warmup"] is None
    assert cfg["allreduce_overlaps_next_kernel"] is (cfg["arrangement"] == "overlapped")
    if graph != "on":
        # the OTHER arrangement was timed in the same run, and every overlapped step's slot equals the in-line sum bit for bit
        assert cfg["ms_per_step_in_line"] > 0 and cfg["ms_per_step_overlapped"] > 0
        assert out["ms_per_step"] == cfg["ms_per_step_" + cfg["arrangement"]]
        assert cfg["overlapped_sums_equal_the_in_line_sum"] is True
    assert cfg["allreduce_us"] > 0 and cfg["kernel_us_rank0"] > 0
    assert cfg["rccl_ranks"] == [1]
    assert len(cfg["kernel_us_per_rank"]) == 1 and cfg["kernel_us_per_rank"][0] > 0 and cfg["allreduce_us_per_rank"][0] > 0
    assert len(cfg["store_GBs_kept_per_rank"]) == 1 and cfg["store_GBs_kept_per_rank"][0] > 0.0      # 1.2 M observations: measured
    if collective in ("auto", "c2b"):
        assert cfg["comm_init_ms"] > 0
    assert cfg["watchdog_seconds"] == 60.0
    # r05: the set-up is the device route (layout + visibility loop on the resident problem, the range exported device to
    # device): seconds, not the host candidate search's tens of seconds; the line names the device's store class and the
    # launch shape chosen by it
    assert 0 < cfg["setup_s"] < 20.0
    assert out["roofline"]["device_store_class"] in ("slow", "mixed", "fast") and out["roofline"]["kept_set_store_GBs"] > 0.0
    assert "threads per workgroup" in out["roofline"]["launch_shape"]
    assert out["config"]["n_observations"] == plain["config"]["n_observations"]
    assert out["config"]["observations_per_rank"] == [plain["config"]["n_observations"]]
    assert out["config"]["total_L2_error"] == plain["config"]["total_L2_error"]      # all_reduce over one rank: identity
    assert out["roofline"]["kernel_avg_us"] > 0


def test_two_ranks_on_one_gpu_exit_cleanly_and_agree(plain):
    env = {"C2B_DIST_BACKEND": "gloo", "C2B_SHARE_GPU": "1"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + COMMON
    rc, out, err = _run(cmd, env)
    assert rc == 0, err[-3000:]                                  # every rank, not only the one that prints
    assert "Traceback" not in err and "ChildFailedError" not in err, err[-3000:]
    assert out is not None and out["n_gpus"] == 2 and out["scaling"] == "strong"
    per = out["config"]["observations_per_rank"]
    total = plain["config"]["n_observations"]
    assert sum(per) == total == out["config"]["n_observations"]
    assert max(per) - min(per) <= 0.02 * total                    # split on the observation prefix sum
    b = out["config"]["camera_bounds"]
    assert b[0] == 0 and b[-1] == plain["config"]["n_cameras"] and b[1] > 0
    rel = abs(out["config"]["total_L2_error"] - plain["config"]["total_L2_error"]) / plain["config"]["total_L2_error"]
    assert rel < 1e-12
    cfg = out["config"]
    assert 0 < cfg["setup_s"] < 30.0                              # max over the ranks; no second build, no collective inside
    assert cfg["rccl_ranks"] == [2, 2] and cfg["arrangement"] == "in_line"          # gloo stages the scalar through the host
    assert len(cfg["kernel_us_per_rank"]) == 2 and min(cfg["kernel_us_per_rank"]) > 0 and min(cfg["allreduce_us_per_rank"]) > 0
    assert out["ms_per_step"] == cfg["ms_per_step_in_line"] and "ms_per_step_overlapped" not in cfg


def test_comm_through_the_c_abi_at_world_size_one():
    """c2b_comm_* directly (what a Rust host binds): RCCL loads, a one-rank communicator all-reduces / all-gathers the
    identity, the sharded statistics through it equal c2b_stats (mean / min / max / origin bit for bit; std to rounding:
    the unsharded call merges Chan triples in one pass, the sharded one sums squared deviations in a second pass), and
    the Level-1 sharded error equals the unsharded one."""
    import numpy as np
    import torch
    import __graft_entry__ as entry
    entry.build()
    import city2ba_amd as c2b
    from city2ba_amd import _lib as L
    from city2ba_amd import comm as Comm
    from city2ba_amd import device as D
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _problems import random_problem
    import ctypes as C

    assert Comm.backend().startswith("RCCL "), Comm.backend()
    dev = torch.device("cuda", 0)
    c = Comm.Comm(Comm.unique_id(), 0, 1, 0)
    r, w, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    assert L.lib().c2b_comm_info(c.handle, C.byref(r), C.byref(w), C.byref(d)) == L.OK and (r.value, w.value, d.value) == (0, 1, 0)
    x = torch.tensor([1.25, -3.5e300, 7e-310], dtype=torch.float64, device=dev)
    want = x.clone()
    c.all_reduce_sum_(x)
    g = c.all_gather(want)
    torch.cuda.synchronize()
    assert torch.equal(x, want) and g.shape == (1, 3) and torch.equal(g[0], want)

    P = random_problem(300, 4000, 12, seed=77, noise=1e-3)
    camblk = D.cameras_prepare_state(torch.from_numpy(P["cams15"]).to(dev))
    pts4 = D.points_pad(torch.from_numpy(P["pts"]).to(dev))
    ws = D.workspace(len(P["pt_idx"]), dev)
    st_plain = D.stats(camblk, pts4, ws).cpu().numpy()
    st_comm = c.stats_sharded(camblk, 0, 300, pts4, ws).cpu().numpy()
    exact = [0, 1, 2] + list(range(6, 19))
    assert np.array_equal(st_plain[exact], st_comm[exact]), (st_plain, st_comm)
    assert np.allclose(st_plain[[3, 4, 5, 19]], st_comm[[3, 4, 5, 19]], rtol=1e-13, atol=0.0), (st_plain, st_comm)

    ba = c2b.BAProblem.from_visibility(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"], device=0)
    e = C.c_double(0.0)
    L.check(L.lib().c2b_problem_total_reprojection_error_sharded(ba._h, c.handle, 2.0, C.byref(e)))
    assert e.value == ba.total_reprojection_error(2.0)
    # one process driving its GPUs itself (ctx_create(n_dev, dev_ids)): c2b_comm_init_all over the one device
    # of this box, the collective bracketed by a group like NCCL asks of one thread issuing for several ranks
    comms = (C.c_void_p * 1)()
    ids = (C.c_int * 1)(0)
    L.check(L.lib().c2b_comm_init_all(1, ids, comms))
    y = torch.tensor([3.0, -4.5], dtype=torch.float64, device=dev)
    L.check(L.lib().c2b_comm_group_start())
    L.check(L.lib().c2b_comm_all_reduce_sum_f64(comms[0], C.c_void_p(y.data_ptr()), 2, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    L.check(L.lib().c2b_comm_group_end())
    torch.cuda.synchronize()
    assert y.tolist() == [3.0, -4.5]
    L.lib().c2b_comm_destroy(comms[0])
    assert L.lib().c2b_comm_init_all(0, None, comms) == L.ERR_INVALID_ARGUMENT
    # argument checks come back as statuses, never as aborts
    assert L.lib().c2b_comm_all_reduce_sum_f64(None, None, 1, None) == L.ERR_INVALID_ARGUMENT
    h = C.c_void_p()
    assert L.lib().c2b_comm_init_rank(C.create_string_buffer(128), 3, 2, 0, C.byref(h)) == L.ERR_INVALID_ARGUMENT
    c.destroy()


def test_run_noise_flow_through_the_c_abi_matches_the_checker():
    """INTEGRATION.md's multi-GPU run_noise (src/bin/city2ba.rs:280-357: add_drift_normalized -> add_noise -> error) as a
    rank would run it -- statistics, drift, noise and the error all through Level 0 + the C-ABI communicator, world
    size 1 -- against the CPU restatement's chain with the same seeds."""
    import numpy as np
    import torch
    import __graft_entry__ as entry
    entry.build()
    import oracle as O
    from city2ba_amd import comm as Comm
    from city2ba_amd import device as D
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _problems import random_problem

    dev = torch.device("cuda", 0)
    P = random_problem(260, 3000, 14, seed=314, noise=0.0)
    n_cam = len(P["cams15"])
    c = Comm.Comm(Comm.unique_id(), 0, 1, 0)
    cam15 = torch.from_numpy(P["cams15"]).to(dev)
    pts4 = D.points_pad(torch.from_numpy(P["pts"]).to(dev))
    uv = torch.from_numpy(P["uv"]).to(dev)
    rows = D.Rows(torch.from_numpy(P["row_ptr"].astype(np.int64)).to(dev))
    pt_idx = torch.from_numpy(P["pt_idx"].astype(np.int64).astype(np.int32)).to(dev)
    ws = D.workspace(rows.n_obs, dev)

    st = c.stats_sharded(D.cameras_prepare_state(cam15), 0, n_cam, pts4, ws)
    D.add_drift_sharded(cam15, 0, pts4, st, 0.05, 0.02, 0.1, 7)                     # normalized (direction None)
    st = c.stats_sharded(D.cameras_prepare_state(cam15), 0, n_cam, pts4, ws)
    D.add_noise_entities_sharded(cam15, 0, pts4, st, 0.05, 0.02, 0.03, 8)
    D.add_noise_observations(uv, 0, 0.01, 8)
    total = torch.zeros(1, dtype=torch.float64, device=dev)
    D.reprojection_error_sum_rows(D.cameras_prepare_state(cam15), pts4, rows, pt_idx, uv, 2.0, ws, total)
    c.all_reduce_sum_(total)
    got = float(total.item()) ** 0.5

    c0, p0 = O.add_drift_normalized(P["cams15"], P["pts"], 0.05, 0.02, 0.1, seed=7)
    c0, p0, uv0 = O.add_noise(c0, p0, P["uv"], 0.05, 0.02, 0.03, 0.01, seed=8)
    want = O.total_reprojection_error(c0, p0, P["row_ptr"], P["pt_idx"], uv0, 2.0)
    assert np.max(np.abs(cam15.cpu().numpy() - c0)) < 1e-7
    assert np.max(np.abs(pts4[:, :3].cpu().numpy() - p0)) < 1e-7
    assert np.max(np.abs(uv.cpu().numpy() - uv0)) < 1e-9
    assert abs(got - want) / want < 1e-7
    c.destroy()


def test_level1_sharded_entries_equal_the_unsharded_calls_at_world_size_one():
    """c2b_problem_set_shard + c2b_problem_{stats,add_drift,add_sin_noise,add_noise,total_reprojection_error}_sharded on
    a one-rank communicator against the plain Level-1 calls on an identical problem: same draws (global indices = local
    ones here), statistics through the communicator's two-pass form -> results equal to rounding."""
    import ctypes as C
    import numpy as np
    import __graft_entry__ as entry
    entry.build()
    import city2ba_amd as c2b
    from city2ba_amd import _lib as L
    from city2ba_amd import comm as Comm
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _problems import random_problem

    P = random_problem(310, 2600, 16, seed=99, noise=1e-4)
    mk = lambda: c2b.BAProblem.from_visibility(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"], device=0)
    plain, shard = mk(), mk()
    c = Comm.Comm(Comm.unique_id(), 0, 1, 0)
    lib = L.lib()
    h = shard._h
    # before set_shard the sharded entries refuse
    st = np.zeros(20)
    assert lib.c2b_problem_stats_sharded(h, c.handle, st.ctypes.data_as(C.c_void_p)) == L.ERR_INVALID_ARGUMENT
    assert lib.c2b_problem_set_shard(h, 5, 300, 0) == L.ERR_INVALID_ARGUMENT          # 5 + 310 cameras do not fit 300
    L.check(lib.c2b_problem_set_shard(h, 0, 310, 0))
    L.check(lib.c2b_problem_stats_sharded(h, c.handle, st.ctypes.data_as(C.c_void_p)))
    assert np.allclose(st, plain._stats(), rtol=1e-13, atol=0)
    d3 = (C.c_double * 3)(0.3, -0.5, 0.8)
    L.check(lib.c2b_problem_add_drift_sharded(h, c.handle, 0.01, 0.02, 0.1, None, 3))            # normalized
    L.check(lib.c2b_problem_add_drift_sharded(h, c.handle, 0.01, 0.02, 0.1, d3, 4))
    dx, up = (C.c_double * 3)(1, 0, 0), (C.c_double * 3)(0, 1, 0)
    L.check(lib.c2b_problem_add_sin_noise_sharded(h, c.handle, dx, up, 0.1, 2.0))
    L.check(lib.c2b_problem_add_noise_sharded(h, c.handle, 0.02, 0.01, 0.03, 0.004, 5))
    L.check(lib.c2b_problem_add_drift_normalized(plain._h, 0.01, 0.02, 0.1, 3))
    L.check(lib.c2b_problem_add_drift(plain._h, 0.01, 0.02, 0.1, d3, 4))
    L.check(lib.c2b_problem_add_sin_noise(plain._h, dx, up, 0.1, 2.0))
    L.check(lib.c2b_problem_add_noise(plain._h, 0.02, 0.01, 0.03, 0.004, 5))
    scale = max(1.0, float(np.max(np.abs(plain.points()))))        # the drift (strength * d^2) moves things out to ~1e5
    assert np.max(np.abs(shard.cameras() - plain.cameras())) <= 1e-13 * scale
    assert np.max(np.abs(shard.points() - plain.points())) <= 1e-13 * scale
    assert np.array_equal(shard.observations(), plain.observations())       # same draws, same kernel, nothing statistical in between
    e = C.c_double(0.0)
    L.check(lib.c2b_problem_total_reprojection_error_sharded(h, c.handle, 2.0, C.byref(e)))
    assert abs(e.value - plain.total_reprojection_error(2.0)) <= 1e-9 * e.value
    c.destroy()
