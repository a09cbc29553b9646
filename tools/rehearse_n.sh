#!/bin/bash
# N ranks sharing GPU 0 over gloo, launched like the driver launches them: every rank must exit 0.
#   usage: tools/rehearse_n.sh N [blocks] [graph: auto|on|off] [overlap: auto|on|off]
#          (graph on: every rank replays its step from a HIP graph; overlap on: the collective on its own stream)
N=${1:-4}; B=${2:-32}; G=${3:-on}; O=${4:-auto}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
set -o pipefail
C2B_DIST_BACKEND=gloo C2B_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
  --master-port $((29700 + N)) bench.py --gpus $N --blocks $B --steps 5 --warmup 2 --graph $G --overlap $O --no-cpu-baseline --no-extras 2> /tmp/rehearse_$N.err \
  | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); c=j['config']; print('N=%d value %.0f ms/step %s obs/rank %s err %.15g graph=%s arrangement=%s (%s) A/B=%s in_line=%s overlapped=%s (sums ok: %s) kernel_us/rank=%s allreduce_us/rank=%s store_GBs_kept/rank=%s rccl_ranks=%s comm_init_ms=%s overhead_us=%s' % (j['n_gpus'], j['value'], j['ms_per_step'], c['observations_per_rank'], c['total_L2_error'], c['hip_graph'], c.get('arrangement'), c.get('arrangement_chosen_by'), c.get('ab_during_warmup'), c.get('ms_per_step_in_line'), c.get('ms_per_step_overlapped'), c.get('overlapped_sums_equal_the_in_line_sum'), c.get('kernel_us_per_rank'), c.get('allreduce_us_per_rank'), c.get('store_GBs_kept_per_rank'), c.get('rccl_ranks'), c.get('comm_init_ms'), c.get('step_overhead_us')))"
rc=$?
echo "rc=$rc"; grep -c Traceback /tmp/rehearse_$N.err
