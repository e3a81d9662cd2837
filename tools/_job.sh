#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_extrinsics_gpu.py tests/test_error_map_gpu.py tests/test_step_schedule_gpu.py -x -q -m gpu 2>&1 | tail -20
