#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02d_pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r02d_pytest.log
timeout 600 python tools/tune_jac.py --variants 14,104,-1,-2 > gpurun_out/r02d_tune_jac.log 2>&1; echo "tune_jac rc=$?"
tail -6 gpurun_out/r02d_tune_jac.log
timeout 600 python tools/tune_jac.py --no-err --variants 14,104 > gpurun_out/r02d_tune_jac_noerr.log 2>&1; echo "tune_jac noerr rc=$?"
tail -3 gpurun_out/r02d_tune_jac_noerr.log
timeout 600 python tools/tune_obs.py --variants 1208,2008 > gpurun_out/r02d_tune_obs.log 2>&1; echo "tune_obs rc=$?"
tail -7 gpurun_out/r02d_tune_obs.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r02d_bench.json; tail -3 gpurun_out/r02d_bench.err
