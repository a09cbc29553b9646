#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; CLI=$ROOT/city2ba_amd/cli/city2ba; D=$(mktemp -d)
python $ROOT/tools/make_city_obj.py $D/c.obj --blocks 32 --detail 4 > /dev/null
A="--cameras 20000 --points 300000 --max-dist 40 --path street --step-size 1 --seed 3 --exact-lcc"
"$CLI" generate $D/c.obj $D/a.bbal $A > $D/a.txt
C2B_HOST_SAMPLER=1 "$CLI" generate $D/c.obj $D/b.bbal $A > $D/b.txt
C2B_DENSE_SWEEP=1 "$CLI" generate $D/c.obj $D/c.bbal $A > $D/c.txt
C2B_HOST_SAMPLER=1 C2B_DENSE_SWEEP=1 "$CLI" generate $D/c.obj $D/d.bbal $A > $D/d.txt
"$CLI" generate $D/c.obj $D/e.bbal --cameras 20000 --points 300000 --max-dist 40 --ground 10000 --height 1.7 --seed 3 --exact-lcc > $D/e.txt
C2B_HOST_SAMPLER=1 C2B_DENSE_SWEEP=1 "$CLI" generate $D/c.obj $D/f.bbal --cameras 20000 --points 300000 --max-dist 40 --ground 10000 --height 1.7 --seed 3 --exact-lcc > $D/f.txt
tail -2 $D/a.txt
cmp $D/a.bbal $D/b.bbal && cmp $D/a.bbal $D/c.bbal && cmp $D/a.bbal $D/d.bbal && echo "street path: all four routes wrote the same file"
tail -2 $D/e.txt
cmp $D/e.bbal $D/f.bbal && echo "poisson: both routes wrote the same file"
rm -rf $D
