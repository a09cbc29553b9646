#!/bin/bash
# on the GPU box: look at the device class; on a slow-store device run the full default bench and keep its line
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import sys, torch, bench
from city2ba_amd import device as D
dev = torch.device("cuda", 0)
rates, keep = [], []
for _ in range(8):
    o = D.JacobianOutputs(19_302_494, dev, max_attempts=1); keep.append(o); rates.append(round(o.store_GBs, 1))
cls = bench.store_class(rates)
print("store GB/s of 8 output sets:", rates, "->", cls, flush=True)
sys.exit(0 if cls == "slow" else 7)
PY
rc=$?
if [ $rc = 0 ]; then python3 bench.py > gpurun_out/r05an_bench_slow_device.json 2> gpurun_out/r05an_bench_slow_device.err; echo "bench rc=$?"; fi
exit 0
