#!/bin/bash
# round 6: occupancy-grid update with brick-ordered samples: tests, then the timeline of an update step with and without the ordering
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_step_schedule_gpu.py tests/test_sampling_gpu.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_a_tests.log
TIMELINE_UPDATE=1 bash tools/step_timeline.sh > gpurun_out/r06_a_timeline_morton.txt 2>&1
NGP_SCENE_TESTBED_OPTIONS='{"morton_grid_samples": false}' TIMELINE_UPDATE=1 bash tools/step_timeline.sh > gpurun_out/r06_a_timeline_forward_order.txt 2>&1
tail -5 gpurun_out/r06_a_tests.log
TIMELINE_UPDATE=1 bash tools/fox_timeline.sh > gpurun_out/r06_a_fox_timeline_morton.txt 2>&1
NGP_BENCH_MORTON_GRID=0 TIMELINE_UPDATE=1 bash tools/fox_timeline.sh > gpurun_out/r06_a_fox_timeline_forward_order.txt 2>&1
