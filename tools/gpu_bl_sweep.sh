#!/bin/bash
# GPU box: schedule sweep of the Blender renderer leg (bench_legs.py bl_render): BL_SKIPS x BL_FACTOR
export TMPDIR=/tmp
for cfg in "96 4" "96 3" "48 4" "24 4" "0 4" "200 4" "96 6"; do
  set -- $cfg
  BL_SKIPS=$1 BL_FACTOR=$2 timeout 200 python bench_legs.py bl_render 2>/dev/null | grep '^{' | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('skips $1 factor $2: stock', l['stock_render_ms'], '| 1 nerf', l['bl_render_ms_frames_1nerf'], 'passes', l['bl_passes_1nerf'], 'samples', l['bl_network_samples_1nerf'], '| 2 nerf', l['bl_render_ms_min_2nerf'], l['bl_passes_2nerf'])"
done
