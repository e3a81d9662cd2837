timeout 600 python -m pytest tests/test_dp_gpu.py -x -q 2>&1 | grep -v "site-packages\|dist-packages\|runpy" | tail -12
