mkdir -p gpurun_out/r03m
(time python -m pytest tests -m gpu -q) > gpurun_out/r03m/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03m/pytest.log
python tools/bench_kernels.py > gpurun_out/r03m/per_kernel.json 2> gpurun_out/r03m/per_kernel.err; echo "kernels rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03m/per_kernel.json'))
for k,v in d['kernels'].items(): print("%-62s %8.1f us  frac %.3f"%(k,v['us'],v['frac_hbm_peak']))
PY
