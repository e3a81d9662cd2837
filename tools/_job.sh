#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_network_gpu.py tests/test_loss_gpu.py -x -q -m gpu -k "cam_gradient" 2>&1 | tail -30
