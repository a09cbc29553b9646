python -m pytest tests/test_gpu_dist.py tests/test_gpu_end_to_end.py -m gpu -q -k "level1_sharded or noise_sharded" 2>&1 | grep -v "^$" | tail -30 | cut -c1-250
