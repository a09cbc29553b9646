#!/bin/bash
# prints the `generate` CLI transcripts on the two reference scenes (run on the GPU box)
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as e; e.build_cli()" >/dev/null
CLI=city2ba_amd/cli/city2ba
T=$(mktemp -d)
for args in "tests/golden/box.obj $T/a.bal --cameras 100 --points 100 --path BezierCurve --seed 1" \
            "tests/golden/box.obj $T/b.bal --cameras 100 --points 100 --path BezierCurve --step-size 0.1 --seed 1" \
            "tests/golden/box.obj $T/c.bal --cameras 100 --points 100 --ground -1.0 --seed 1" \
            "tests/golden/test_scene.obj $T/d.bal --cameras 100 --points 200 --seed 1" \
            "tests/golden/test_scene.obj $T/e.bal --cameras 2000 --points 20000 --seed 1" \
            "tests/golden/test_scene.obj $T/f.bbal --cameras 2000 --points 20000 --path path --seed 1"; do
  echo "\$ city2ba generate $args"
  s=$(date +%s.%N)
  $CLI generate $args
  e=$(date +%s.%N)
  python -c "print('  wall: %.2f s' % ($e - $s))"
done
rm -rf "$T"
