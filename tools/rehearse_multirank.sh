#!/bin/bash
# Rehearse bench.py's multi-rank path on a 1-GPU box.  The checks live in tests/test_gpu_dist.py (RCCL at world size 1;
# two ranks sharing GPU 0 over gloo, launched by torch.distributed.run; every rank must exit 0); this wrapper only runs
# them and fails if they fail.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
python -m pytest tests/test_gpu_dist.py -m gpu -x -q
