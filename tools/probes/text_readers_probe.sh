#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CLI=$ROOT/city2ba_amd/cli/city2ba
D=$(mktemp -d)
"$CLI" synthetic $D/g.bal --blocks 128 > /dev/null 2>&1
for T in 1 2 3 4 6 8 3 6; do
  echo "readers=$T: $(C2B_READ_THREADS=$T C2B_TIMING=1 "$CLI" noise $D/g.bal $D/n.bbal --drift-strength 1e-5 --seed 1 2>&1 | grep 'read (' )"
done
rm -rf $D
