#!/bin/bash
# GPU box, round 5: the two-pass splat, the shared stream pair, fp16 wire / sharded Ema / stale guards, the fox leg with its Blender-renderer comparison, the preflight
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=r05_h
timeout 1500 python -m pytest tests/test_sampling_gpu.py tests/test_dp_gpu.py tests/test_two_testbeds_gpu.py tests/test_step_schedule_gpu.py tests/test_bl_render_gpu.py tests/test_render_gpu.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
grep -n "passed\|failed" $out/${tag}_pytest.txt | tail -3
timeout 120 python bench.py --gpus 1 --preflight > $out/${tag}_preflight.txt 2>&1; grep '^{' $out/${tag}_preflight.txt
timeout 300 python bench_legs.py fox 300 > $out/${tag}_fox.log 2>&1
grep '^{' $out/${tag}_fox.log | tail -1 > $out/${tag}_fox_line.json
python - <<PY
import json
l=json.load(open("$out/${tag}_fox_line.json"))
print("fox", l["value"], l["ms_per_step"], "render", l.get("render_MP_per_s"), l.get("render_ms_frames"), l.get("render_roofline"), {a:b["avg_us"] for a,b in l["kernels"].items()})
print("fox bl", json.dumps(l.get("bl_render"))[:900])
PY
timeout 300 python bench.py --steps 300 --warmup 5 --no_cpu_baseline --legs none > $out/${tag}_lego.log 2>&1
grep '^{' $out/${tag}_lego.log | tail -1 > $out/${tag}_lego_line.json
python - <<PY
import json
l=json.load(open("$out/${tag}_lego_line.json"))
print("lego", l["value"], l["ms_per_step"], "render", l.get("render_MP_per_s"), {a:b.get("avg_us") for a,b in l["kernels"].items()})
PY
rm -rf /tmp/tr_f
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_f -o t -- python bench_legs.py fox 100 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/tr_f/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if any(k in r["Name"] for k in ("splat","ema_kernel","grid_samples","bitfield","reduce_sum","grid_to_bitfield","encode_planes","nerf_forward_kernelILi1"))]
for r in rows: print("   ", r["Name"][:70].ljust(70), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000))
PY
