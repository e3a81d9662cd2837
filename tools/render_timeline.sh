#!/bin/bash
# GPU box: kernel-by-kernel timeline of ONE 800x800 frame of the stock renderer (last frame of tools/render_probe.py under rocprofv3 --kernel-trace)
export TMPDIR=/tmp
steps=${1:-1000}
rm -rf /tmp/rtl
NGP_PROBE_PLAIN=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/rtl -o t -- python tools/render_probe.py $steps 3 > /tmp/render_probe.out 2>&1
tail -2 /tmp/render_probe.out
python - <<'PY'
import csv,glob
f=glob.glob("/tmp/rtl/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=max(i for i,r in enumerate(rows) if "init_rays" in r["Kernel_Name"])
fr=rows[last:]
t0=int(fr[0]["Start_Timestamp"]); prev_end=t0
tot={}
for r in fr:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=r["Kernel_Name"].split("(")[0].replace("ngp::","").replace("void ","")[:34]
    print("%8.1f us  gap %6.1f  dur %7.1f  %s  grid %s"%((s-t0)/1e3,(s-prev_end)/1e3,(e-s)/1e3,n,r.get("Grid_Size_X", r.get("Grid_Size","?"))))
    tot[n]=tot.get(n,0)+(e-s)/1e3
    prev_end=max(prev_end,e)
print("frame span %.1f us"%((prev_end-t0)/1e3))
for k,v in sorted(tot.items(),key=lambda kv:-kv[1]): print("  %-36s %8.1f us"%(k,v))
PY
