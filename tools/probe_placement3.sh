#!/bin/bash
# bench.py five times in fresh processes: with the placement search (default) and without (--placement-attempts 1)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for i in 1 2 3 4 5; do
  for P in 8 1; do
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --placement-attempts $P 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('attempts<=$P', 'value', j['value'], 'frac', r['frac'], 'kernel_us', r['kernel_avg_us'], 'placement', r['output_placement']['store_GBs_per_attempt'])"
  done
done
