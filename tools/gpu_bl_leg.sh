#!/bin/bash
# GPU box: the Blender-renderer leg on its own (plain run for its line, then rocprofv3 --kernel-trace --stats).  $1 = tag, $2.. = env assignments
tag=${1:-bl}; shift
export TMPDIR=/tmp "$@"
out=$PWD/gpurun_out; mkdir -p $out
timeout 300 python bench_legs.py bl_render > $out/${tag}_bl_line.log 2>&1
grep '^{' $out/${tag}_bl_line.log | tail -1 > $out/${tag}_bl_line.json
rm -rf /tmp/tr_bl
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bl -o t -- python bench_legs.py bl_render > $out/${tag}_bl_leg.log 2>&1
python - <<PY > $out/${tag}_bl_kernel_stats.txt 2>&1
import csv,glob
f=glob.glob("/tmp/tr_bl/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("kernel, calls, avg_us, total_ms, pct")
for r in rows[:24]: print(r["Name"][:110].ljust(110), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000), "%.1f"%(float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
cat $out/${tag}_bl_line.json; tail -3 $out/${tag}_bl_line.log | cut -c1-300
grep -v "gb_fx\|adam\|backward\|compute_loss\|generate_training\|grid_combine\|wgrad\|post_and\|splat\|grid_samples\|Cijk" $out/${tag}_bl_kernel_stats.txt | head -16
