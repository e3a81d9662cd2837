timeout 600 python -m pytest tests/test_pyngp_testbed_gpu.py tests/test_bl_render_gpu.py tests/test_loader_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -30
