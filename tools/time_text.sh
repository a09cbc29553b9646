#!/bin/bash
# the text form (.bal) of the headline grid end to end: the device formatter (r04, csrc/text_kernels.hpp) against the host
# formatter over a download (C2B_HOST_TEXT=1), same bytes; then the kernel trace of the device route
#   usage: tools/time_text.sh > profiles/rNN_text_times.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CLI=$ROOT/city2ba_amd/cli/city2ba
python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as e; e.build_cli()" >/dev/null
D=$(mktemp -d)
T() {
  echo "\$ $PREFIX city2ba $*"
  local s=$(date +%s.%N)
  C2B_TIMING=1 timeout 120 "$CLI" "$@" 2>&1 | sed 's/^/  /'
  local e=$(date +%s.%N)
  python -c "print('  wall: %.3f s' % ($e - $s))"
}
for i in 1 2; do rm -f $D/g128.bal; T synthetic $D/g128.bal --blocks 128; done
T noise $D/g128.bal $D/n128.bal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1
export C2B_HOST_TEXT=1; PREFIX="C2B_HOST_TEXT=1"
T synthetic $D/h128.bal --blocks 128
T noise $D/g128.bal $D/nh128.bal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1
unset C2B_HOST_TEXT; PREFIX=
cmp $D/g128.bal $D/h128.bal && echo "synthetic --blocks 128 (.bal): both routes wrote the same file"
cmp $D/n128.bal $D/nh128.bal && echo "noise (.bal): both routes wrote the same file"
ls -la $D | awk 'NR>1 {print "  " $5, $9}'
rm -f $D/n128.bal $D/nh128.bal $D/h128.bal
cd /tmp && export TMPDIR=/tmp
for what in "synthetic $D/p128.bal --blocks 128" "noise $D/g128.bal $D/pn128.bal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1"; do
  rm -rf $D/prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -- "$CLI" $what > $D/prof.log 2>&1
  f=$(find $D/prof -name '*kernel_stats.csv' | head -1)
  echo "# rocprofv3 --kernel-trace --stats -- city2ba ${what//$D\//}"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("  all kernels: %.2f ms" % (tot / 1e6))
for r in rows[:12]:
    name = r["Name"].replace("void ", "").replace("c2b::", "").split("(")[0]
    print("  %-34s calls %3d  avg %9.1f us  total %8.2f ms" % (name, int(float(r["Calls"])), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  else tail -5 $D/prof.log; fi
done
rm -rf "$D"
