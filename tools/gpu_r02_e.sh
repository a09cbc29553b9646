#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02e_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02e_pytest.log
timeout 600 python tools/tune_obs.py --variants 1208,1408 > gpurun_out/r02e_tune_obs.log 2>&1; echo "tune_obs rc=$?"; tail -7 gpurun_out/r02e_tune_obs.log
timeout 600 python tools/tune_jac.py --variants 14,-1,-2 > gpurun_out/r02e_tune_jac.log 2>&1; echo "tune_jac rc=$?"; tail -4 gpurun_out/r02e_tune_jac.log
