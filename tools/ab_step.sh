#!/bin/bash
# The N > 1 step on ONE GPU: a rank's eighth of the headline problem (--blocks 45: 2.4 M observations) through the
# multi-rank code path at world size 1 (RCCL initialised, a real all-reduce per step), six ways: collective through the
# C ABI or through torch.distributed; eager with the collective overlapping the next kernel, eager in line, HIP graph.  Prints the step's breakdown.
#   usage: tools/ab_step.sh [blocks]
B=${1:-45}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for V in "c2b off on" "c2b off off" "c2b on off" "torch off on" "torch off off" "torch on off"; do set -- $V; C=$1; G=$2; O=$3
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29800 + RANDOM % 100)) \
  python bench.py --gpus 1 --force-dist --collective $C --graph $G --overlap $O --blocks $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null \
  | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); c=j['config']
        print('collective=%-5s graph=%-3s overlap=%-3s  step %.2f us (timed loop, no event records)  | instrumented afterwards: kernel %.2f  allreduce %.2f' % ('$C', '$G', '$O', j['ms_per_step']*1e3, c['kernel_us_rank0'], c['allreduce_us']))"
done
# the same shard with the collective EMULATED by a kernel that spins X us on the collective's stream: what an N-rank
# all-reduce of that latency costs the step in line and overlapped
for us in 10 20 30 45; do for O in off on; do
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29800 + RANDOM % 100)) \
  python bench.py --gpus 1 --force-dist --collective c2b --overlap $O --emulate-allreduce-us $us --blocks $B --steps 300 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null \
  | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); c=j['config']
        print('emulated collective %2d us  overlap=%-3s  step %.2f us  | overlapped sums ok: %s' % ($us, '$O', j['ms_per_step']*1e3, c.get('overlapped_sums_equal_the_in_line_sum')))"
done; done
