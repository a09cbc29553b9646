#!/bin/bash
# N ranks sharing GPU 0 over gloo, launched like the driver launches them: every rank must exit 0.
#   usage: tools/rehearse_n.sh N [blocks]
N=${1:-4}; B=${2:-32}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
set -o pipefail
C2B_DIST_BACKEND=gloo C2B_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
  --master-port $((29700 + N)) bench.py --gpus $N --blocks $B --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/rehearse_$N.err \
  | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); c=j['config']; print('N=%d value %.0f obs/rank %s err %.15g' % (j['n_gpus'], j['value'], c['observations_per_rank'], c['total_L2_error']))"
rc=$?
echo "rc=$rc"; grep -c Traceback /tmp/rehearse_$N.err
