#!/bin/bash
# SQ counters of the dense sweep kernels (run on the GPU box); under `timeout` like every PMC pass here
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/dense_sq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/tools/bench_dense.py" --reps 1 --cpu-pairs 1000 > "$OUT.log" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
t = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dense" in r["Kernel_Name"]:
            t[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in t.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
