mkdir -p gpurun_out/r03a
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03a/pytest.log
python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc=$?"; head -c 1500 gpurun_out/r03a/bench.json; tail -3 gpurun_out/r03a/bench.err
python tools/bench_kernels.py > gpurun_out/r03a/per_kernel.json 2> gpurun_out/r03a/per_kernel.err; echo "kernels rc=$?"
bash tools/probe_placement_pmc.sh r03a
