#!/bin/bash
# GPU box: A/B of env settings on the short bench (no render, no cpu baseline).  Each argument is one "VAR=val VAR2=val" set ("-" = none); prints value + per-group times
for envset in "$@"; do
  [ "$envset" = "-" ] && envset=""
  for rep in 1 2; do
    env $envset timeout 300 python bench.py --steps ${STEPS:-300} --warmup 20 --no_cpu_baseline --no_render 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('[%s] %.1f M samples/s  %.4f ms/step | %s' % ('$envset', d['value']/1e6, d['ms_per_step'], '  '.join('%s %.0f' % (n[:14], v['avg_us']) for n,v in k.items())))
"
  done
done
