#!/bin/bash
# wall-clock of the C++ CLI end to end on the GPU box, with per-phase times (C2B_TIMING=1); each command twice (the
# first start of a process on a fresh box pays for paging the runtime in), the resident route and rounds 1-3's host route
#   usage: tools/time_cli.sh > profiles/rNN_cli_times.txt
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
for B in 32 128 128; do rm -f $D/g$B.bbal; T synthetic $D/g$B.bbal --blocks $B; done
T synthetic $D/g32.bal --blocks 32
# the text form of the headline grid (972 MB of shortest round-trip decimals: formatted / parsed on the device since r04; tools/time_text.sh compares with the host route)
T synthetic $D/g128.bal --blocks 128
T noise $D/g128.bal $D/n128.bal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1
rm -f $D/g128.bal $D/n128.bal
for i in 1 2; do rm -f $D/n128.bbal; T noise $D/g128.bbal $D/n128.bbal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1; done
echo "# rounds 1-3's route: layout, candidate search, hits_building and file (de)serialisation on the host"
export C2B_HOST_CANDIDATES=1 C2B_HOST_IO=1; PREFIX="C2B_HOST_CANDIDATES=1 C2B_HOST_IO=1"
rm -f $D/h128.bbal; T synthetic $D/h128.bbal --blocks 128
rm -f $D/nh128.bbal; T noise $D/g128.bbal $D/nh128.bbal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1
unset C2B_HOST_CANDIDATES C2B_HOST_IO; PREFIX=
cmp $D/g128.bbal $D/h128.bbal && echo "synthetic --blocks 128: both routes wrote the same file"
cmp $D/n128.bbal $D/nh128.bbal && echo "noise: both routes wrote the same file"
ls -la $D | awk 'NR>1 {print "  " $5, $9}'
T ply $D/g32.bbal $D/g32.ply
rm -rf "$D"
