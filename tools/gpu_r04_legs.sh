#!/bin/bash
# GPU box: the driver's short bench run with its legs (fox, bl_render, plumbing) + rocprofv3 kernel stats of the fox leg and of the Blender-renderer leg on their own.  $1 = tag
tag=${1:-r04_a}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
timeout 900 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.log 2>&1
grep '^{' $out/${tag}_bench_driver.log | tail -1 > $out/${tag}_bench_driver_line.json
for leg in fox bl_render; do
  rm -rf /tmp/tr_$leg
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$leg -o t -- python bench_legs.py $leg > $out/${tag}_${leg}_leg.log 2>&1
  python - <<PY > $out/${tag}_${leg}_kernel_stats.txt 2>&1
import csv,glob
f=glob.glob("/tmp/tr_$leg/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("kernel, calls, avg_us, total_ms, pct")
for r in rows[:28]: print(r["Name"][:110].ljust(110), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000), "%.1f"%(float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
done
python - <<PY
import json
l=json.load(open("$out/${tag}_bench_driver_line.json"))
print("lego", l["value"], l["ms_per_step"], l.get("render_MP_per_s"))
for k in ("fox","bl_render","plumbing"):
    v=l.get(k,{})
    if k=="fox": v={a:b for a,b in v.items() if a!="kernels"}; print("fox kernels", {a:b["avg_us"] for a,b in l.get("fox",{}).get("kernels",{}).items()})
    print(k, json.dumps(v)[:900])
PY
head -14 $out/${tag}_fox_kernel_stats.txt; head -22 $out/${tag}_bl_render_kernel_stats.txt
