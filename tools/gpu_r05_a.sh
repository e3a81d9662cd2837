#!/bin/bash
# GPU box, round 5 first call: where the fox photographs (config #2) and the plumbing configs stand at HEAD —
#   (1) fox training: fused network pass vs the two-kernel pass (NGP_HIP_FWD_WS), line + rocprofv3 kernel stats each
#   (2) launch-constant sweeps on fox and on the lego stand-in (forward workgroup cap, run-ahead march workgroups)
#   (3) fox render: schedule sweep on one trained model, kernel stats of a render-only process, pass structure of one frame
#   (4) plumbing leg (image, SDF) kernel stats
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=r05_a
kstats() {  # $1 = trace dir, $2 = rows
python - <<PY
import csv,glob
f=glob.glob("$1/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("kernel, calls, avg_us, total_ms, pct")
for r in rows[:$2]: print(r["Name"][:120].ljust(120), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000), "%.1f"%(float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
}
foxline() {  # $1 = label, rest = env
  label=$1; shift
  env "$@" timeout 300 python bench_legs.py fox 300 > $out/${tag}_fox_${label}.log 2>&1
  grep '^{' $out/${tag}_fox_${label}.log | tail -1 > $out/${tag}_fox_${label}_line.json
  python - <<PY
import json
try:
    l=json.load(open("$out/${tag}_fox_${label}_line.json"))
    print("fox $label", l["value"], l["ms_per_step"], "render", l.get("render_MP_per_s"), {a:b["avg_us"] for a,b in l["kernels"].items()})
except Exception as e: print("fox $label FAILED", e)
PY
}
legoline() {
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 300 --warmup 5 --no_cpu_baseline --no_render --legs none > $out/${tag}_lego_${label}.log 2>&1
  grep '^{' $out/${tag}_lego_${label}.log | tail -1 > $out/${tag}_lego_${label}_line.json
  python - <<PY
import json
try:
    l=json.load(open("$out/${tag}_lego_${label}_line.json"))
    print("lego $label", l["value"], l["ms_per_step"], {a:b.get("avg_us") for a,b in l.get("kernels",{}).items()})
except Exception as e: print("lego $label FAILED", e)
PY
}
# (1)
foxline fused
foxline ws NGP_HIP_FWD_WS=1
for v in fused ws; do
  rm -rf /tmp/tr_$v
  if [ $v = ws ]; then export NGP_HIP_FWD_WS=1; else unset NGP_HIP_FWD_WS; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -o t -- python bench_legs.py fox 100 > $out/${tag}_fox_${v}_prof.log 2>&1
  kstats /tmp/tr_$v 26 > $out/${tag}_fox_${v}_kernel_stats.txt 2>&1
  head -20 $out/${tag}_fox_${v}_kernel_stats.txt
done
unset NGP_HIP_FWD_WS
# (2)
for c in 512 768 1536 2048; do foxline cap$c NGP_HIP_FWD_CAP=$c; done
for g in 384 512 768 1024 4096; do foxline wgs$g NGP_HIP_GEN_WGS=$g; done
foxline ws_wgs512 NGP_HIP_FWD_WS=1 NGP_HIP_GEN_WGS=512
foxline ws_wgs1024 NGP_HIP_FWD_WS=1 NGP_HIP_GEN_WGS=1024
foxline fused2
legoline base
legoline ws NGP_HIP_FWD_WS=1
for c in 768 1536; do legoline cap$c NGP_HIP_FWD_CAP=$c; done
for g in 512 768 1024; do legoline wgs$g NGP_HIP_GEN_WGS=$g; done
legoline base2
# (3)
timeout 400 python tools/fox_render_probe.py train /tmp/fox.msgpack 1500 > $out/${tag}_fox_render_sweep.txt 2>&1
cat $out/${tag}_fox_render_sweep.txt | cut -c1-400
rm -rf /tmp/tr_r
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_r -o t -- python tools/fox_render_probe.py render /tmp/fox.msgpack 8 > $out/${tag}_fox_render_prof.log 2>&1
tail -1 $out/${tag}_fox_render_prof.log
kstats /tmp/tr_r 20 > $out/${tag}_fox_render_kernel_stats.txt 2>&1
cat $out/${tag}_fox_render_kernel_stats.txt
NGP_HIP_RENDER_TRACE=1 timeout 200 python tools/fox_render_probe.py render /tmp/fox.msgpack 2 2>&1 | grep "render pass" | awk '{print $3, $4, $5}' | tr '\n' ';' > $out/${tag}_fox_render_passes.txt
cut -c1-3000 $out/${tag}_fox_render_passes.txt; echo
# (4)
rm -rf /tmp/tr_p
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_p -o t -- python bench_legs.py plumbing > $out/${tag}_plumbing_prof.log 2>&1
tail -1 $out/${tag}_plumbing_prof.log | cut -c1-1500
kstats /tmp/tr_p 30 > $out/${tag}_plumbing_kernel_stats.txt 2>&1
cat $out/${tag}_plumbing_kernel_stats.txt
