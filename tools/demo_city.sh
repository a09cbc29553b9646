#!/bin/bash
# city mesh -> bundle adjustment problem, end to end through the C++ CLI (run on the GPU box)
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as e; e.build_cli()" >/dev/null
CLI=city2ba_amd/cli/city2ba
T=$(mktemp -d)
run() {
  echo "\$ $*"
  s=$(date +%s.%N)
  "$@" || true
  e=$(date +%s.%N)
  python -c "print('  wall: %.2f s' % ($e - $s))"
}
python tools/make_city_obj.py $T/city32.obj --blocks 32 --detail 4
# faithful cull: the reference's observation filter (src/baproblem.rs:523) empties a graph with many unseen points
run $CLI generate $T/city32.obj $T/city32f.bbal --cameras 5000 --points 200000 --max-dist 40 --path street --seed 1
run $CLI generate $T/city32.obj $T/city32.bbal --cameras 5000 --points 200000 --max-dist 40 --path street --seed 1 --exact-lcc
run $CLI generate $T/city32.obj $T/city32s.bbal --cameras 5000 --points 200000 --max-dist 40 --path street --step-size 4 --seed 1 --exact-lcc
python tools/make_city_obj.py $T/city96.obj --blocks 96 --detail 8
run $CLI generate $T/city96.obj $T/city96.bbal --cameras 40000 --points 1000000 --max-dist 40 --path street --seed 1 --exact-lcc
run $CLI noise $T/city96.bbal $T/city96n.bbal --drift-strength 1e-4 --rotation-std 1e-3 --point-std 1e-2 --observation-std 1e-3 --seed 2
run $CLI ply $T/city96n.bbal $T/city96n.ply
ls -la $T
rm -rf "$T"
