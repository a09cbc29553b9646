#!/bin/bash
# wall-clock of the C++ CLI end to end (generation + cull + to_vec + file) on the GPU box, with per-phase times
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CLI=$ROOT/city2ba_amd/cli/city2ba
python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as e; e.build_cli()" >/dev/null
D=$(mktemp -d)
T() {
  echo "\$ city2ba $*"
  local s=$(date +%s.%N)
  C2B_TIMING=1 "$CLI" "$@" 2>&1 | sed 's/^/  /'
  local e=$(date +%s.%N)
  python -c "print('  wall: %.2f s' % ($e - $s))"
}
for B in 32 128; do T synthetic $D/g$B.bbal --blocks $B; done
T synthetic $D/g32.bal --blocks 32
ls -la $D | awk 'NR>1 {print "  " $5, $9}'
T noise $D/g128.bbal $D/n128.bbal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1
T ply $D/g32.bbal $D/g32.ply
rm -rf "$D"
