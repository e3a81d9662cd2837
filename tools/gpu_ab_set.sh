#!/bin/bash
# GPU box: A/B of library variants built with NGP_EXTRA_HIP_FLAGS into blender-ngp_amd/lib_ab/<name>/ (git-ignored): lego + fox step per variant, `base` interleaved.  $@ = variant names
export TMPDIR=/tmp
run() {
  if [ $1 = base ]; then unset NGP_HIP_LIBRARY_DIR; export LD_LIBRARY_PATH=$PWD/blender-ngp_amd/lib; else export NGP_HIP_LIBRARY_DIR=$PWD/blender-ngp_amd/lib_ab/$1 LD_LIBRARY_PATH=$PWD/blender-ngp_amd/lib_ab/$1; fi
  python bench.py --steps 300 --warmup 5 --no_cpu_baseline --no_render --legs fox 2>/dev/null | grep "^{" | python -c "
import json,sys
l=json.loads(sys.stdin.read()); f=l['fox']; k=l['kernels']; fk=f['kernels']
print('%-6s lego %.4f' % ('$1', l['ms_per_step']), {a:k[a]['avg_us'] for a in ('generate_training_samples','nerf_inference','compute_loss','nerf_backward')}, 'fox %.4f' % f['ms_per_step'], {a:fk[a]['avg_us'] for a in ('generate_training_samples','compute_loss','nerf_backward')})"
}
for rep in 1 2; do run base; for v in "$@"; do run $v; done; done
