mkdir -p gpurun_out/r03v
(time python -m pytest tests -m gpu -q) > gpurun_out/r03v/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03v/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r03v/bench.json 2> gpurun_out/r03v/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r03v/bench.json').read().strip().split('\n')[-1]); r=b['roofline']
print(b['value'], b['ms_per_step'], r['frac'], r['kernel_avg_us'], r['kernel_us_first_allocation'], r['frac_first_allocation'], r['output_placement']['store_GBs_per_attempt'])
for k,v in b['other_configs']['blocks128_other_passes'].items(): print(k, v)
PY
