#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel trace + stats, then the HBM PMC counters in their own
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   usage: tools/profile_bench.sh <tag> [bench args...]
# Output: gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write}/ + summary files written by summarize_prof.py.
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
ARGS="${@:---steps 50 --warmup 5 --no-cpu-baseline --no-extras}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
# The counter passes must launch the instance the trace pass timed: the launch shape follows the output set's measured store
# rate (r05), and under --pmc the store pattern's own timing reads differently -- so tell them the rate the trace pass kept.
RATE=$(python3 -c "
import json,sys
for l in open('$OUT/trace.log'):
    if l.startswith('{') and '\"metric\"' in l:
        print(json.loads(l)['roofline'].get('kept_set_store_GBs') or 7000.0); break
else: print(7000.0)" 2>/dev/null || echo 7000.0)
ARGS="$ARGS --assume-store-GBs $RATE"
echo "counter passes with --assume-store-GBs $RATE"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?"
python3 "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG"
