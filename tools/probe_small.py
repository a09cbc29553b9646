#!/usr/bin/env python3
"""stats + observation noise on the bench grid, 30 launches each (run under rocprofv3 --kernel-trace --stats / --pmc)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
import bench                                                   # noqa: E402
from city2ba_amd import device as D                            # noqa: E402

dev = torch.device("cuda", 0)
sh = bench.build_shard(argparse.Namespace(blocks=128), 0, 1, dev)
ws = D.workspace(sh["n_obs"], dev)
st = torch.empty(20, dtype=torch.float64, device=dev)
uv = sh["uv"].clone()
for _ in range(30):
    D.stats(sh["camblk"], sh["pts4"], ws, st)
torch.cuda.synchronize()
for _ in range(30):
    D.add_noise_observations(uv, 0, 1e-6, 7)
torch.cuda.synchronize()
