#!/bin/bash
# GPU box, round 5 second call: the measured network-pass policy (Testbed.network_pass) — tests, fox / lego / plumbing lines under 'auto' and forced — and the
# fox render discrepancy (leg 40 MP/s vs probe 106 MP/s at 168 more training steps)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=r05_b
timeout 900 python -m pytest tests/test_network_pass_gpu.py tests/test_gridmlp_gpu.py tests/test_plumbing_gpu.py tests/test_network_gpu.py tests/test_step_schedule_gpu.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
tail -5 $out/${tag}_pytest.txt
foxline() {  # $1 = label, $2 = organisation or "", rest = env
  label=$1; org=$2; shift; shift
  env "$@" timeout 300 python bench_legs.py fox 300 $org > $out/${tag}_fox_${label}.log 2>&1
  grep '^{' $out/${tag}_fox_${label}.log | tail -1 > $out/${tag}_fox_${label}_line.json
  python - <<PY
import json
try:
    l=json.load(open("$out/${tag}_fox_${label}_line.json"))
    print("fox $label", l["value"], l["ms_per_step"], "render", l.get("render_MP_per_s"), l.get("render_ms_frames"), l.get("render_network_samples_per_frame"), l["network_pass"], {a:b["avg_us"] for a,b in l["kernels"].items()})
except Exception as e: print("fox $label FAILED", e)
PY
}
foxline auto ""
foxline fused fused
foxline two_kernel two_kernel
foxline auto_min1500 "" FOX_MIN_STEP=1500
foxline auto_min3000 "" FOX_MIN_STEP=3000
for st in 1032 1332 1500 3000; do
  FOX_SHORT=1 timeout 300 python tools/fox_render_probe.py train /tmp/fox_$st.msgpack $st > $out/${tag}_probe_$st.txt 2>&1
  echo "probe steps $st"; grep '^{' $out/${tag}_probe_$st.txt | cut -c1-300
done
timeout 600 python bench.py --steps 300 --warmup 5 --no_cpu_baseline --legs plumbing > $out/${tag}_lego.log 2>&1
grep '^{' $out/${tag}_lego.log | tail -1 > $out/${tag}_lego_line.json
python - <<PY
import json
l=json.load(open("$out/${tag}_lego_line.json"))
print("lego", l["value"], l["ms_per_step"], l.get("render_MP_per_s"), l["network_pass"])
print("plumbing", json.dumps(l.get("plumbing"))[:1800])
PY
