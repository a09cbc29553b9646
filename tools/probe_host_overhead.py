import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from city2ba_amd import device as D
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=45), 0, 1, dev)     # ~ the per-rank shard of --blocks 128 at N=8
n = sh["n_obs"]
r = torch.empty((n, 2), dtype=torch.float64, device=dev); Jc = torch.empty((n, 18), dtype=torch.float64, device=dev); Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
ws2 = [D.workspace(n, dev), D.workspace(n, dev)]; err = torch.zeros(1, dtype=torch.float64, device=dev)
main = torch.cuda.current_stream(); side = torch.cuda.Stream(device=dev)
pe = [torch.cuda.Event(), torch.cuda.Event()]; fe = [torch.cuda.Event(), torch.cuda.Event()]; fd = [None, None]
def step(i):
    k = i & 1
    if fd[k] is not None: main.wait_event(fd[k])
    D.residual_jacobian(sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws2[k])
    pe[k].record(main)
    with torch.cuda.stream(side):
        side.wait_event(pe[k]); D.error_sum_finish(ws2[k], n, err); fe[k].record(side); fd[k] = fe[k]
for i in range(20): step(i)
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for i in range(K): step(i)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("n_obs %d: host issue %.1f us/step, wall %.1f us/step (GPU-bound if wall > issue)" % (n, t_issue / K * 1e6, t_all / K * 1e6))
