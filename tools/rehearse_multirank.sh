#!/bin/bash
# Rehearse bench.py's multi-rank path on a 1-GPU box: N ranks share GPU 0, gloo carries the two tiny
# collectives.  Checks that the sharded run reproduces the single-rank observation count and total error.
#   usage: tools/rehearse_multirank.sh [blocks]
B=${1:-64}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out
python bench.py --blocks $B --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | grep metric > gpurun_out/rehearse_n1.json
for N in 2 4; do
  C2B_DIST_BACKEND=gloo C2B_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --blocks $B --steps 5 --warmup 1 \
      --no-cpu-baseline --no-extras 2>gpurun_out/rehearse_n$N.err | grep metric > gpurun_out/rehearse_n$N.json
done
python - <<PY
import json
ref = json.load(open("gpurun_out/rehearse_n1.json"))
ok = True
for n in (2, 4):
    try:
        d = json.load(open("gpurun_out/rehearse_n%d.json" % n))
    except Exception as e:
        print("N=%d: no JSON line (%s)" % (n, e)); ok = False; continue
    same_obs = d["config"]["n_observations"] == ref["config"]["n_observations"]
    rel = abs(d["config"]["total_L2_error"] - ref["config"]["total_L2_error"]) / ref["config"]["total_L2_error"]
    print("N=%d n_gpus=%d obs=%d (same as N=1: %s) total_L2_error rel diff %.2e scaling=%s" %
          (n, d["n_gpus"], d["config"]["n_observations"], same_obs, rel, d["scaling"]))
    ok = ok and same_obs and rel < 1e-12 and d["n_gpus"] == n
print("REHEARSAL", "OK" if ok else "FAILED")
PY
