mkdir -p gpurun_out/r03n
python tools/tune_obs.py --only rows --cold > gpurun_out/r03n/tune_obs_cold.txt 2>&1; grep -v jacobian gpurun_out/r03n/tune_obs_cold.txt | tail -32
