mkdir -p gpurun_out/r03o
(time python -m pytest tests -m gpu -q) > gpurun_out/r03o/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03o/pytest.log
python tools/bench_kernels.py > gpurun_out/r03o/per_kernel.json 2> gpurun_out/r03o/per_kernel.err; echo "kernels rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03o/per_kernel.json'))
for k in ('rows_pack (once per list)','add_drift_normalized','add_noise_entities','project_rows'): print(k, d['kernels'][k])
PY
