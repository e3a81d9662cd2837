#!/bin/bash
# GPU box, round 5: the hash-grid backward's record space packed by level kind (scratch 778 -> 405 MB): parity of every backward path, then the step times
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=${1:-r05_s}
timeout 1700 python -m pytest tests/test_network_gpu.py tests/test_grid_backward_gpu.py tests/test_gridmlp_gpu.py tests/test_netx_gpu.py tests/test_plumbing_gpu.py tests/test_network_pass_gpu.py \
    tests/test_baseline_configs_gpu.py tests/test_step_schedule_gpu.py tests/test_extrinsics_gpu.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
grep -n "passed\|failed" $out/${tag}_pytest.txt | tail -3
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --legs fox,plumbing > $out/${tag}_bench$i.log 2>&1
grep '^{' $out/${tag}_bench$i.log | tail -1 > $out/${tag}_line$i.json
python - <<PY
import json
l=json.load(open("$out/${tag}_line$i.json"))
k=l["kernels"]
print("lego", l["value"], l["ms_per_step"], {a:b.get("avg_us") for a,b in k.items()})
f=l.get("fox",{})
print("fox", f.get("value"), f.get("ms_per_step"), {a:b.get("avg_us") for a,b in f.get("kernels",{}).items()})
p=l.get("plumbing",{})
print("plumbing", json.dumps(p)[:600])
PY
done
