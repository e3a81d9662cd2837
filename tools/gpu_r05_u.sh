#!/bin/bash
# GPU box, round 5: the backward pass over the live samples — parity tests, then A/B of the step (lego, fox) in alternating processes
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=${1:-r05_u}
timeout 1700 python -m pytest tests/test_network_gpu.py tests/test_baseline_configs_gpu.py tests/test_step_schedule_gpu.py tests/test_pyngp_testbed_gpu.py tests/test_dp_gpu.py tests/test_network_pass_gpu.py tests/test_snapshot_gpu.py tests/test_extrinsics_gpu.py tests/test_netx_e2e_gpu.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
grep -n "passed\|failed" $out/${tag}_pytest.txt | tail -3
for on in 1 0 1 0; do
NGP_BENCH_COMPACT_BWD=$on timeout 300 python bench.py --steps 300 --warmup 5 --no_cpu_baseline --legs fox > $out/${tag}_bench_$on.log 2>&1
grep '^{' $out/${tag}_bench_$on.log | tail -1 > $out/${tag}_line_$on.json
python - <<PY
import json
l=json.load(open("$out/${tag}_line_$on.json"))
k=l["kernels"]
print("compact=$on lego", l["value"], l["ms_per_step"], "psnr", l.get("psnr"), "render", l.get("render_MP_per_s"), {a:b.get("avg_us") for a,b in k.items()})
f=l.get("fox",{})
print("   fox", f.get("value"), f.get("ms_per_step"), {a:b.get("avg_us") for a,b in f.get("kernels",{}).items()})
PY
done
