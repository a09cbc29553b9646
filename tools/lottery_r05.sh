#!/bin/bash
# a rocprofv3 record of the bench ON A SLOW-STORE DEVICE.  Every gpurun call lands on another MI355X;
# this looks at the device (store rate of eight output sets, ~4 s) and runs tools/profile_bench.sh only when every set
# streams below 6.0 TB/s (exit status 7 otherwise: nothing else is spent).
#   usage: tools/lottery_r05.sh <tag> [class: slow|mixed|fast]
TAG=${1:-r05_slow}; WANT=${2:-slow}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
python3 - "$WANT" <<'PY' || exit 7
import sys
import torch
import bench
from city2ba_amd import device as D
dev = torch.device("cuda", 0)
n = 19_302_494
rates = []
keep = []
for _ in range(8):
    o = D.JacobianOutputs(n, dev, max_attempts=1)
    keep.append(o)
    rates.append(round(o.store_GBs, 1))
cls = bench.store_class(rates)
print("store GB/s of 8 output sets:", rates, "->", cls, flush=True)
del keep, o
sys.exit(0 if cls == sys.argv[1] else 7)
PY
bash tools/profile_bench.sh "$TAG" --steps 50 --warmup 5 --no-cpu-baseline --no-extras
