#!/bin/bash
# GPU box, round 5: the optimizer step's Ema stage on stream B — tests, then A/B of the step time (lego, fox, plumbing) in one process each
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=${1:-r05_t}
timeout 900 python -m pytest tests/test_step_schedule_gpu.py tests/test_snapshot_gpu.py tests/test_pyngp_testbed_gpu.py tests/test_dp_gpu.py tests/test_two_testbeds_gpu.py tests/test_plumbing_gpu.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
grep -n "passed\|failed" $out/${tag}_pytest.txt | tail -3
for side in 1 0 1 0; do
NGP_BENCH_SIDE_EMA=$side timeout 300 python bench.py --steps 300 --warmup 5 --no_cpu_baseline --legs fox,plumbing > $out/${tag}_bench_$side.log 2>&1
grep '^{' $out/${tag}_bench_$side.log | tail -1 > $out/${tag}_line_$side.json
python - <<PY
import json
l=json.load(open("$out/${tag}_line_$side.json"))
k=l["kernels"]
print("side_ema=$side lego", l["value"], l["ms_per_step"], "render", l.get("render_MP_per_s"), {a:b.get("avg_us") for a,b in k.items()})
f=l.get("fox",{})
print("   fox", f.get("value"), f.get("ms_per_step"), {a:b.get("avg_us") for a,b in f.get("kernels",{}).items()})
p=l.get("plumbing",{})
print("   image", p.get("image",{}).get("ms_per_step"), p.get("image",{}).get("groups_us"), "sdf", p.get("sdf",{}).get("ms_per_step"), p.get("sdf",{}).get("groups_us"))
PY
done
