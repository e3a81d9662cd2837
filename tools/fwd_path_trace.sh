#!/bin/bash
# GPU box: per-dispatch kernel times of one forward-pass variant.  $1 = path, env as for fwd_path_probe.py
export TMPDIR=/tmp
N=${N:-519936}
rm -rf /tmp/fp_t
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/fp_t -o t -- python tools/fwd_path_probe.py child $1 $N 10 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/fp_t/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "ngp" in r["Kernel_Name"]]
rows=rows[-${2:-14}:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows: print(r["Kernel_Name"][:50].ljust(50), "start %8.1f us  dur %7.1f us  grid %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size","")))
PY
