#!/bin/bash
# GPU box: sweep of the run-ahead march's persistent workgroups (NGP_HIP_GEN_WGS) — step time and the backward group beside it
export TMPDIR=/tmp
for w in ${WGS_LIST:-384 512 640 768 1024}; do
  NGP_HIP_GEN_WGS=$w timeout 200 python bench.py --steps 400 --warmup 300 --no_cpu_baseline --no_render --legs none 2>/dev/null | grep '^{' | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels']; print('wgs $w: ms/step', l['ms_per_step'], 'M samples/s', round(l['value']/1e6,1), {a:k[a]['avg_us'] for a in ('generate_training_samples','nerf_inference','compute_loss','nerf_backward','optimizer_step')})"
done
