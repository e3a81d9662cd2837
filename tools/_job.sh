#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu.log 2>&1
tail -5 gpurun_out/full_gpu.log
grep -n "RuntimeError" gpurun_out/full_gpu.log | head
