mkdir -p gpurun_out/r03q
(time python -m pytest tests -m gpu -q) > gpurun_out/r03q/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03q/pytest.log
python tools/bench_kernels.py > gpurun_out/r03q/per_kernel.json 2> gpurun_out/r03q/per_kernel.err; echo "kernels rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03q/per_kernel.json'))
for k in ('stats(mean,std,extent,origin)','cameras_prepare_state','cameras_to_bal','cameras_from_bal','add_drift_normalized','add_noise_entities','residual_jacobian_rows (one launch, bench step)'): print(k, d['kernels'][k])
PY
python tools/bench_dense.py 2>/dev/null | tail -3 | cut -c1-600
