mkdir -p gpurun_out/r03h
(time python -m pytest tests/test_gpu_dist.py -m gpu -q) > gpurun_out/r03h/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03h/pytest.log
bash tools/ab_step.sh 45 > gpurun_out/r03h/ab_step_b45.txt 2>&1; cat gpurun_out/r03h/ab_step_b45.txt
for n in 2 8; do bash tools/rehearse_n.sh $n 32 off; done > gpurun_out/r03h/rehearse.txt 2>&1; cat gpurun_out/r03h/rehearse.txt
bash tools/rehearse_n.sh 4 32 on >> gpurun_out/r03h/rehearse.txt 2>&1; tail -3 gpurun_out/r03h/rehearse.txt
