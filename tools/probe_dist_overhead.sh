#!/bin/bash
# How much of a multi-rank step is host overhead?  One rank with the full dist code path (RCCL all_reduce on a side
# stream) at 1/8 of the headline problem (what each of 8 ranks holds): ms_per_step against the kernel's own time.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for B in 45 128; do
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 python bench.py --gpus 1 --force-dist --blocks $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('dist  blocks', j['config']['blocks'], 'n_obs', j['config']['n_observations'], 'ms_per_step', j['ms_per_step'], 'kernel_avg_us', j['roofline']['kernel_avg_us'])"
  python bench.py --gpus 1 --blocks $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('plain blocks', j['config']['blocks'], 'n_obs', j['config']['n_observations'], 'ms_per_step', j['ms_per_step'], 'kernel_avg_us', j['roofline']['kernel_avg_us'])"
done
