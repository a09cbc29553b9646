#!/bin/bash
# round-2 profile set: rocprofv3 kernel trace + HBM PMC passes + SQ counters of the bench command, per-kernel timings,
# SQ counters of the light kernels.  Summaries are copied to profiles/ by hand afterwards.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=${1:-r02a}
mkdir -p gpurun_out
bash tools/profile_bench.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; tail -12 gpurun_out/prof_$TAG.log
timeout 600 python tools/bench_kernels.py > gpurun_out/${TAG}_per_kernel.json 2> gpurun_out/${TAG}_per_kernel.err; echo "bench_kernels rc=$?"
bash tools/profile_light.sh $TAG > gpurun_out/light_$TAG.log 2>&1; tail -5 gpurun_out/light_$TAG.log
