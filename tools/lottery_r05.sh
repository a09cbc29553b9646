#!/bin/bash
# a rocprofv3 record of the bench in the SLOW-STORE class.  Every gpurun call lands on another MI355X;
# this looks at the device (store rate of eight output sets, ~4 s) and then
#   * every set below 6.0 TB/s (a slow-store device)  -> tools/profile_bench.sh <tag>_device (the default bench);
#   * otherwise exit 7: nothing else is spent.  (Profiling "the slow set of a mixed device" does not work: every pass is a
#     process of its own and allocates afresh, and under the profiler the store pattern's own timing reads differently.)
#   usage: tools/lottery_r05.sh <tag>
TAG=${1:-r05_slow}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
python3 - <<'PY'
import sys
import torch
import bench
from city2ba_amd import device as D
dev = torch.device("cuda", 0)
n = 19_302_494
rates, keep = [], []
for _ in range(8):
    o = D.JacobianOutputs(n, dev, max_attempts=1)
    keep.append(o)
    rates.append(round(o.store_GBs, 1))
cls = bench.store_class(rates)
print("store GB/s of 8 output sets:", rates, "->", cls, flush=True)
del keep, o
sys.exit(0 if cls == "slow" else 7)
PY
rc=$?
if [ $rc = 0 ]; then
  bash tools/profile_bench.sh "${TAG}_device" --steps 50 --warmup 5 --no-cpu-baseline --no-extras
else
  exit 7
fi
