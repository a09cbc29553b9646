#!/bin/bash
# wall-clock of the C++ CLI end to end (generation + cull + to_vec + file) on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CLI=$ROOT/city2ba_amd/cli/city2ba
T() { local s=$(date +%s.%N); "$@" > /tmp/cli.out 2>&1; local rc=$?; local e=$(date +%s.%N); printf "%-60s %6.2f s rc=%d  %s\n" "$*" "$(echo "$e - $s" | bc)" $rc "$(head -1 /tmp/cli.out | cut -c1-90)"; }
for B in 32 128; do T $CLI synthetic /tmp/g$B.bbal --blocks $B; done
T $CLI synthetic /tmp/g32.bal --blocks 32
ls -la /tmp/g32.bal /tmp/g32.bbal /tmp/g128.bbal | awk '{print $5, $9}'
T $CLI noise /tmp/g128.bbal /tmp/n128.bbal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1
tail -2 /tmp/cli.out
