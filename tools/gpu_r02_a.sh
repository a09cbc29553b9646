#!/bin/bash
# round-2 GPU session A: parity tests on the new pipelined kernels, then A/B of the kernel variants.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a_pytest.log
tail -5 gpurun_out/r02a_pytest.log
timeout 600 python tools/tune_jac.py --variants 14,100,104,105,108,116,-1,-2 > gpurun_out/r02a_tune_jac.log 2>&1; echo "tune_jac rc=$?"
tail -14 gpurun_out/r02a_tune_jac.log
timeout 600 python tools/tune_obs.py --variants 1208,2008,2004,2016 > gpurun_out/r02a_tune_obs.log 2>&1; echo "tune_obs rc=$?"
tail -16 gpurun_out/r02a_tune_obs.log
