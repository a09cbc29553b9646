#!/bin/bash
# PMC passes over tools/probe_light.py (run on the GPU box): SQ issue/wait split and L2 hit rates per kernel.
# Output: gpurun_out/light_<tag>/pass*/ + a per-kernel table (light_<tag>.json).
# Every pass runs under `timeout`: a pass with TA_* / TCP_* / GRBM_* counters aborted inside rocprofv3 on this pool
# (signal 6) and then sat until the caller's limit, so those counters are not collected here.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/light_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/probe_light.py" > "$OUT/pass$i.log" 2>&1
  echo "pass$i rc=$?"
done
python3 - "$OUT" "$TAG" <<'EOF'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
table = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "c2b::" not in k:
            continue
        short = k.split("c2b::")[1].split("(")[0]
        table[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in table.items()}
json.dump(res, open(out + "/../light_%s.json" % tag, "w"), indent=1)
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-40s %.4g" % (c, v))
EOF
