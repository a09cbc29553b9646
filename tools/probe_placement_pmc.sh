#!/bin/bash
# Run on the GPU box (gpurun): tools/probe_placement_pmc.py plain, then under rocprofv3 --pmc in separate passes (each
# pass is its own process and rolls its own allocations; the script tags the slow / fast dispatches by grid size).
#   usage: tools/probe_placement_pmc.sh <tag>
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/placement_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/probe_placement_pmc.py" > "$OUT/plain.json" 2> "$OUT/plain.err"
echo "plain rc=$? $(cat $OUT/plain.json | head -c 300)"
pass() {
    name=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/probe_placement_pmc.py" > "$OUT/$name.json" 2> "$OUT/$name.err"
    echo "$name rc=$?"
}
pass wr_stall   TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum
pass wr_credit  TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_sum
pass wr_64b     TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_BUSY_sum TCC_TAG_STALL_sum
pass utcl1      TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass utcl1b     TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum
pass grbm       GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
pass per_chan   TCC_EA0_WRREQ
pass per_chan_stall TCC_EA0_WRREQ_STALL
python3 "$ROOT/tools/summarize_placement_pmc.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.err"
echo "summary rc=$?"; head -c 3000 "$OUT/summary.json"
