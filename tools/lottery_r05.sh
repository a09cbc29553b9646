#!/bin/bash
# a rocprofv3 record of the bench in the SLOW-STORE class.  Every gpurun call lands on another MI355X;
# this looks at the device (store rate of eight output sets, ~4 s) and then
#   * every set below 6.0 TB/s (a slow-store device)           -> tools/profile_bench.sh <tag>_device (the default bench);
#   * otherwise, the FIRST set below 6.0 TB/s (a mixed device)  -> tools/profile_bench.sh <tag>_set with --placement-attempts 1,
#     i.e. the kernel in a slow set of a device that also has fast ones (each child process allocates afresh: the bench
#     line of every pass says which rate its own set had);
#   * otherwise exit 7: nothing else is spent.
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
sys.exit(0 if cls == "slow" else (8 if rates[0] < 6000.0 else 7))
PY
rc=$?
if [ $rc = 0 ]; then
  bash tools/profile_bench.sh "${TAG}_device" --steps 50 --warmup 5 --no-cpu-baseline --no-extras
elif [ $rc = 8 ]; then
  bash tools/profile_bench.sh "${TAG}_set" --steps 50 --warmup 5 --no-cpu-baseline --no-extras --placement-attempts 1
else
  exit 7
fi
