#!/bin/bash
# GPU box: timeline of a few steady-state training steps (start / end of every kernel relative to the step's network pass, per queue)
export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python bench.py --steps 60 --warmup 300 --no_cpu_baseline --no_render --legs none $TIMELINE_ARGS > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/tl/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# anchor of a step: its loss kernel (the network pass in front of it is one fused kernel or encode_planes + the MLP kernel, whichever the tuner runs)
fw=[i for i,r in enumerate(rows) if "compute_loss_kernel" in r["Kernel_Name"]]
# three consecutive steps near the end that have no occupancy update in between
import os
ks=list(range(len(fw)-8, len(fw)-5))
if os.environ.get("TIMELINE_UPDATE"):   # the steps around an occupancy-grid update instead
    upd=[k for k in range(max(0, len(fw)-40), len(fw)-2) if any("splat_max_kernel" in r["Kernel_Name"] for r in rows[fw[k]:fw[k+1]])]
    ks=sorted(set(j for k in upd[-1:] for j in (k-1,k,k+1,k+2)))
for k in ks:
    a,b=fw[k],fw[k+1]
    t0=int(rows[a]["Start_Timestamp"])
    print("---- step, period %.1f us" % ((int(rows[b]["Start_Timestamp"])-t0)/1000))
    for r in rows[a:b]:
        n=r["Kernel_Name"]
        n=n.split("(")[0].replace("void ","").replace("ngp::","")[:44]
        print("  q%-3s %-44s %8.1f -> %8.1f  (%6.1f)" % (r.get("Queue_Id","?"), n, (int(r["Start_Timestamp"])-t0)/1000, (int(r["End_Timestamp"])-t0)/1000, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000))
PY
