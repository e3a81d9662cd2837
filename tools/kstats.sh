#!/bin/bash
# GPU box: per-kernel average durations of a short bench run (rocprofv3 kernel trace), top 16 rows
export TMPDIR=/tmp
rm -rf /tmp/tr
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python bench.py --steps 100 --warmup 300 --no_cpu_baseline --no_render --legs none > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/tr/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:${1:-16}]: print(r["Name"][:86].ljust(86), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000))
PY
