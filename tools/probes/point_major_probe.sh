#!/bin/bash
# a point-major text file (the order of the Bundle Adjustment in the Large datasets) through `city2ba noise`: the device
# reader sorts the observations by camera (stable radix sort, csrc/text_kernels.hpp) -- against the host parser's
# sequential push (C2B_HOST_TEXT=1), files compared.  usage: tools/probes/point_major_probe.sh [blocks]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
CLI=$ROOT/city2ba_amd/cli/city2ba
B=${1:-64}
D=$(mktemp -d)
"$CLI" synthetic $D/g.bal --blocks $B > /dev/null
python3 - $D/g.bal $D/p.bal <<'PY'
import sys, numpy as np
lines = open(sys.argv[1]).read().split("\n")
nc, npts, no = map(int, lines[0].split())
obs = lines[1:1 + no]
pt = np.fromiter((int(l.split(" ", 2)[1]) for l in obs), dtype=np.int64, count=no)
order = np.argsort(pt, kind="stable")
open(sys.argv[2], "w").write("\n".join(lines[:1] + [obs[i] for i in order] + lines[1 + no:]))
print("point-major file: %d cameras, %d points, %d observations" % (nc, npts, no))
PY
for i in 1 2; do
  echo "device:"; C2B_TEXT_DEVICE_STRICT=1 C2B_TIMING=1 "$CLI" noise $D/p.bal $D/d.bbal --drift-strength 1e-5 --seed 1 2>&1 | grep "read (\|Error"
done
echo "host parser:"; C2B_HOST_TEXT=1 C2B_TIMING=1 "$CLI" noise $D/p.bal $D/h.bbal --drift-strength 1e-5 --seed 1 2>&1 | grep "read ("
cmp $D/d.bbal $D/h.bbal && echo "both routes wrote the same file"
rm -rf $D
