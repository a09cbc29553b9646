import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from city2ba_amd import device as D
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
n = sh["n_obs"]
r = torch.empty((n, 2), dtype=torch.float64, device=dev); Jc = torch.empty((n, 18), dtype=torch.float64, device=dev); Jp = torch.empty((n, 6), dtype=torch.float64, device=dev)
ws = D.workspace(n, dev); err = torch.zeros(1, dtype=torch.float64, device=dev)
jac = lambda: D.residual_jacobian(sh["camblk"], sh["pts4"], sh["cam_idx"], sh["pt_idx"], sh["uv"], r, Jc, Jp, 2.0, ws)
fin = lambda: D.error_sum_finish(ws, n, err)
def total(fn, reps=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
def with_events(reps=30):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); jac(); b.record(); fin()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return sum(ts) / len(ts), ts[0]
for rnd in range(3):
    print("round", rnd, "jac only %.1f | jac+finish %.1f | events(jac) avg %.1f min %.1f" % ((total(jac), total(lambda: (jac(), fin()))) + with_events()))
