#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 600 python tools/tune_jac.py --variants 14,104,108,-1 > gpurun_out/r02b_tune_jac.log 2>&1; echo "tune_jac rc=$?"
tail -5 gpurun_out/r02b_tune_jac.log
timeout 600 python tools/tune_jac.py --no-err --variants 14,104,108 > gpurun_out/r02b_tune_jac_noerr.log 2>&1; echo "tune_jac noerr rc=$?"
tail -4 gpurun_out/r02b_tune_jac_noerr.log
timeout 600 python tools/tune_obs.py --variants 1208,1408,1216,2008,2004 > gpurun_out/r02b_tune_obs.log 2>&1; echo "tune_obs rc=$?"
tail -16 gpurun_out/r02b_tune_obs.log
