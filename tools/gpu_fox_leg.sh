#!/bin/bash
# GPU box: the fox leg on its own under rocprofv3 --kernel-trace --stats (per-kernel averages) after a plain run (the leg's own line).  $1 = tag, $2.. = env assignments
tag=${1:-fox}; shift
export TMPDIR=/tmp "$@"
out=$PWD/gpurun_out; mkdir -p $out
timeout 300 python bench_legs.py fox 300 > $out/${tag}_fox_line.log 2>&1
grep '^{' $out/${tag}_fox_line.log | tail -1 > $out/${tag}_fox_line.json
rm -rf /tmp/tr_fox
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_fox -o t -- python bench_legs.py fox 100 > $out/${tag}_fox_leg.log 2>&1
python - <<PY > $out/${tag}_fox_kernel_stats.txt 2>&1
import csv,glob
f=glob.glob("/tmp/tr_fox/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("kernel, calls, avg_us, total_ms, pct")
for r in rows[:24]: print(r["Name"][:110].ljust(110), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000), "%.1f"%(float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
python - <<PY
import json
l=json.load(open("$out/${tag}_fox_line.json"))
print("fox", l["value"], l["ms_per_step"], "render", l.get("render_MP_per_s"), {a:b["avg_us"] for a,b in l["kernels"].items()})
PY
head -16 $out/${tag}_fox_kernel_stats.txt
