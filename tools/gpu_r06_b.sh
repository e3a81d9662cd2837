#!/bin/bash
# round 6: the compaction's row index instead of copied encoding rows: tests, then A/B of the main leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_network_gpu.py tests/test_network_pass_gpu.py tests/test_baseline_configs_gpu.py tests/test_pyngp_testbed_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_b_tests.log
tail -4 gpurun_out/r06_b_tests.log
for rep in 1 2; do
  for mode in true false; do
    NGP_SCENE_TESTBED_OPTIONS="{\"x_row_index_mode\": $mode}" timeout 300 python bench.py --steps 1000 --warmup 50 --no_cpu_baseline --no_render --legs none 2>/dev/null | python -c "
import sys, json
l = json.loads([x for x in sys.stdin if x.startswith('{')][-1])
k = l['kernels']
print('x_row_index_mode=$mode rep $rep: %.4f ms/step; loss %.1f us, backward group %.1f us, inference %.1f us' % (l['ms_per_step'], k['compute_loss']['avg_us'], k['nerf_backward']['avg_us'], k['nerf_inference']['avg_us']))
" | tee -a gpurun_out/r06_b_ab.txt
  done
done
