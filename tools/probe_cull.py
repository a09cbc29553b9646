#!/usr/bin/env python3
"""BAProblem::cull on the device at --blocks 128: wall time of c2b_problem_cull over the generator's un-culled graph
(19.3 M observations), three repetitions on fresh uploads (run under rocprofv3 --kernel-trace --stats for the kernel
breakdown)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                             # noqa: E402
import torch                                                   # noqa: E402
import bench                                                   # noqa: E402
import city2ba_amd as c2b                                      # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
sh = bench.build_shard(argparse.Namespace(blocks=blocks), 0, 1, dev)
cams = sh["cam15"].cpu().numpy()
pts = sh["pts4"][:, :3].contiguous().cpu().numpy()
row_ptr = sh["rows"].row_ptr.cpu().numpy().astype(np.uint64)
pt_idx = sh["pt_idx"].cpu().numpy().astype(np.uint64)
uv = sh["uv"].cpu().numpy()
del sh
torch.cuda.empty_cache()
for k in range(3):
    ba = c2b.BAProblem.from_visibility(cams, pts, row_ptr, pt_idx, uv, device=0)
    torch.cuda.synchronize()
    from city2ba_amd import _lib as L
    t0 = time.perf_counter()
    L.check(L.lib().c2b_problem_cull(ba._h, 1))            # the C entry alone: synchronous, the problem stays on the device
    t1 = time.perf_counter()
    ba._refresh_graph()                                    # what the Python mirror adds: the graph back over PCIe (154 MB)
    t2 = time.perf_counter()
    print("cull %d: c2b_problem_cull %.1f ms (+ %.1f ms graph download for the Python mirror) -> %d cameras, %d points, %d observations"
          % (k, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ba.num_cameras(), ba.num_points(), ba.num_observations()))
