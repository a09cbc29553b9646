#!/bin/bash
# phase timings of `city2ba generate` on a large city mesh (run on the GPU box).  The last run is the size the
# reference's paper quotes ("100,000 cameras, 1,000,000 [points] in less than an hour", paper.md:45).
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as e; e.build_cli()" >/dev/null
T=$(mktemp -d)
python tools/make_city_obj.py $T/city96.obj --blocks 96 --detail 8
run() {
  echo "\$ city2ba generate $*"
  s=$(date +%s.%N)
  C2B_TIMING=1 city2ba_amd/cli/city2ba generate "$@"
  e=$(date +%s.%N)
  python -c "print('  wall: %.2f s' % ($e - $s))"
}
run $T/city96.obj $T/a.bbal --cameras 40000 --points 1000000 --max-dist 40 --path street --seed 1 --exact-lcc
run $T/city96.obj $T/b.bbal --cameras 40000 --points 1000000 --max-dist 40 --path street --step-size 4 --seed 1 --exact-lcc
run $T/city96.obj $T/c.bbal --cameras 100000 --points 1000000 --max-dist 40 --path street --step-size 1.8 --seed 1 --exact-lcc
run $T/city96.obj $T/d.bbal --cameras 100000 --points 1000000 --max-dist 40 --ground 10000 --height 1.7 --seed 1 --exact-lcc
rm -rf "$T"
