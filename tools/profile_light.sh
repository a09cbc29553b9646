#!/bin/bash
# rocprofv3 over tools/probe_light.py (run on the GPU box): kernel trace + stats, then -- each in its own pass, as the
# guide prescribes -- the SQ issue / wait counters (two passes), FETCH_SIZE and WRITE_SIZE.  Prints and stores a per-kernel
# table: time, HBM bytes (the guide's gfx950 FETCH correction), vector instructions, and the fraction of the chip's vector
# issue slots and of its HBM peak the kernel used -- which bound binds.
#   usage: tools/profile_light.sh <tag>
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/probe_light.py" > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/pmc_sq1" -- python3 "$ROOT/tools/probe_light.py" > "$OUT/pmc_sq1.log" 2>&1; echo "sq1 rc=$?"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- python3 "$ROOT/tools/probe_light.py" > "$OUT/pmc_sq2.log" 2>&1; echo "sq2 rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/tools/probe_light.py" > "$OUT/pmc_fetch.log" 2>&1; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/tools/probe_light.py" > "$OUT/pmc_write.log" 2>&1; echo "write rc=$?"
python3 "$ROOT/tools/summarize_light.py" "$OUT" "$TAG"
