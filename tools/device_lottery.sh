#!/bin/bash
# Every gpurun call lands on a different MI355X, and the devices differ in how fast they take streaming stores
# (DESIGN.md section 3).  This script looks at the device it got (store rate of eight output-sized allocation sets,
# ~4 s) and then spends time only on what that device can show:
#   * spread between its fastest and slowest set >= 8 %  -> the fast-vs-slow PMC diagnosis (probe_placement_pmc.sh)
#   * a set that streams >= 6.7 TB/s                      -> rocprofv3 kernel stats + PMC of the bench (profile_bench.sh)
#   usage: tools/device_lottery.sh <tag>
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lottery_$TAG
mkdir -p "$OUT"
python3 "$ROOT/tools/probe_placement_pmc.py" > "$OUT/plain.json" 2> "$OUT/plain.err"
read MAXR SPREAD <<< $(python3 -c "
import json,sys
d=json.loads(open('$OUT/plain.json').read().strip().split('\n')[-1]); r=d['store_GBs_per_set']
print(max(r), round((max(r)-min(r))/max(r),4))" 2>/dev/null || echo "0 0")
echo "device: max store rate $MAXR GB/s, spread $SPREAD  ($(head -c 200 $OUT/plain.json))"
if python3 -c "import sys; sys.exit(0 if float('$SPREAD') >= 0.08 else 1)"; then
    echo "mixed device: running the fast-vs-slow PMC diagnosis"
    if [ -z "$C2B_LOTTERY_SKIP_PMC" ]; then bash "$ROOT/tools/probe_placement_pmc.sh" "$TAG"; fi
    echo "per-chunk store rates (is the effect a property of which memory a chunk got?)"
    python3 "$ROOT/tools/probe_chunk_rates.py" 512 192 > "$OUT/chunks_512.json" 2> "$OUT/chunks_512.txt"; tail -2 "$OUT/chunks_512.txt"
    python3 "$ROOT/tools/probe_chunk_rates.py" 64 512 > "$OUT/chunks_64.json" 2> "$OUT/chunks_64.txt"; tail -2 "$OUT/chunks_64.txt" | cut -c1-600
fi
if [ -z "$C2B_LOTTERY_SKIP_PROF" ] && python3 -c "import sys; sys.exit(0 if float('$MAXR') >= 6700 else 1)"; then
    echo "fast-store device: rocprofv3 record of the bench"
    bash "$ROOT/tools/profile_bench.sh" "${TAG}_fast" --steps 50 --warmup 5 --no-cpu-baseline --no-extras
fi
