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


def test_rccl_world_size_one_reproduces_the_plain_run(plain):
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}
    rc, out, err = _run([sys.executable, "bench.py", "--gpus", "1", "--force-dist"] + COMMON, env)
    assert rc == 0 and out is not None, err[-2000:]
    assert "Traceback" not in err, err[-2000:]
    assert out["config"]["collective"].startswith("nccl")
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
