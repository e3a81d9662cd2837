#!/bin/bash
# GPU box: timeline of steady-state steps of the fox leg (as tools/step_timeline.sh for the lego bench)
export TMPDIR=/tmp
rm -rf /tmp/tlf
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlf -o t -- ${FOX_CMD:-python bench_legs.py fox 60} > /dev/null 2>&1
python - <<PY
import csv,glob,os
f=glob.glob("/tmp/tlf/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
fw=[i for i,r in enumerate(rows) if "compute_loss_kernel" in r["Kernel_Name"]]   # a step's anchor: its loss kernel
import statistics
per=[(int(rows[fw[k+1]]["Start_Timestamp"])-int(rows[fw[k]]["Start_Timestamp"]))/1000 for k in range(len(fw)-58, len(fw)-1)]
print("periods of the last %d steps: mean %.1f us, median %.1f; above 700 us: %s" % (len(per), statistics.mean(per), statistics.median(per), [round(x) for x in per if x > 700]))
ks=list(range(len(fw)-8, len(fw)-5))
if os.environ.get("TIMELINE_UPDATE"):
    upd=[k for k in range(len(fw)-40, len(fw)-2) if any("splat_max_kernel" in r["Kernel_Name"] for r in rows[fw[k]:fw[k+1]])]
    ks=sorted(set(j for k in upd[-1:] for j in (k-1,k,k+1)))
for k in ks:
    a,b=fw[k],fw[k+1]
    t0=int(rows[a]["Start_Timestamp"])
    print("---- step, period %.1f us" % ((int(rows[b]["Start_Timestamp"])-t0)/1000))
    for r in rows[a:b]:
        n=r["Kernel_Name"].split("(")[0].replace("void ","").replace("ngp::","")[:44]
        print("  q%-3s %-44s %8.1f -> %8.1f  (%6.1f)" % (r.get("Queue_Id","?"), n, (int(r["Start_Timestamp"])-t0)/1000, (int(r["End_Timestamp"])-t0)/1000, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000))
PY
