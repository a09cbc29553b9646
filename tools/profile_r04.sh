#!/bin/bash
# rocprofv3 over tools/probe_r04.py (run on the GPU box): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own
# passes; prints and stores a per-kernel table (time, HBM bytes with the guide's gfx950 FETCH correction).
#   usage: tools/profile_r04.sh <tag>
TAG=${1:-r04f}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/probe_r04.py" > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/tools/probe_r04.py" > "$OUT/pmc_fetch.log" 2>&1; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/tools/probe_r04.py" > "$OUT/pmc_write.log" 2>&1; echo "write rc=$?"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
def find(sub, pat):
    h = glob.glob(out + "/" + sub + "/**/" + pat, recursive=True)
    return h[0] if h else None
stats = {}
f = find("trace", "*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        if "c2b::" in r["Name"]:
            stats[r["Name"].split("c2b::")[1].split("(")[0]] = {"calls": int(float(r["Calls"])), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
def pmc(sub, counter):
    acc = collections.defaultdict(list)
    f = find(sub, "*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and "c2b::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("c2b::")[1].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
for k, d in stats.items():
    if k in fe and k in wr:
        d["hbm_MB_per_launch"] = round((2 * fe[k] + wr[k]) * 1024 / 1e6, 1)      # FETCH_SIZE doubled: the guide's gfx950 correction
        d["fetch_MB"], d["write_MB"] = round(2 * fe[k] * 1024 / 1e6, 1), round(wr[k] * 1024 / 1e6, 1)
json.dump(stats, open(out + "/../%s_kernels.json" % tag, "w"), indent=1)
for k, d in sorted(stats.items(), key=lambda x: -x[1]["avg_us"] * x[1]["calls"]):
    print("%-64s calls %3d  avg %9.2f us  HBM %s MB (read %s, write %s)" % (k[:64], d["calls"], d["avg_us"], d.get("hbm_MB_per_launch"), d.get("fetch_MB"), d.get("write_MB")))
PY
