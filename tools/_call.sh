python -m pytest tests/test_gpu_workspace.py -m gpu -q 2>&1 | tail -5
