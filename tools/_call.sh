mkdir -p gpurun_out/r03b
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r03b/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r03b/pytest.log
python bench.py > gpurun_out/r03b/bench.json 2> gpurun_out/r03b/bench.err; echo "bench rc=$?"; head -c 2500 gpurun_out/r03b/bench.json; tail -3 gpurun_out/r03b/bench.err
python tools/bench_kernels.py > gpurun_out/r03b/per_kernel.json 2> gpurun_out/r03b/per_kernel.err; echo "kernels rc=$?"
python tools/probe_placement_pmc.py > gpurun_out/r03b/placement_plain.json 2>/dev/null; head -c 400 gpurun_out/r03b/placement_plain.json
