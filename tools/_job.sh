timeout 600 python -m pytest tests/test_loader_gpu.py -x -q 2>&1 | tail -15
