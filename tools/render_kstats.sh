#!/bin/bash
# GPU box: per-kernel totals of the stock renderer (rocprofv3 kernel trace over tools/render_probe.py: train $1 steps, render $2 frames), per frame, and the pass structure of one frame
export TMPDIR=/tmp
steps=${1:-1000}; frames=${2:-20}
rm -rf /tmp/rtr
NGP_PROBE_ONLY=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/rtr -o t -- python tools/render_probe.py $steps $frames > /tmp/render_probe.out 2>&1
tail -1 /tmp/render_probe.out
python - <<PY
# per-frame kernel totals over the LAST $frames frames only: every dispatch from the first init_rays launch of that window on (round 6: the totals used to be divided over
# the whole run, which charged the frames with the copies of the data set upload and the per-step memsets of the 1000 training steps: "155 copies + 47 fills per frame")
import csv,glob,collections
f=glob.glob("/tmp/rtr/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
starts=[i for i,r in enumerate(rows) if "init_rays_kernel" in r["Kernel_Name"]]
nf=$frames
win=rows[starts[-nf]:]
tot=collections.defaultdict(lambda:[0,0.0])
for r in win:
    n=r["Kernel_Name"]
    if "at::" in n: continue
    t=tot[n]; t[0]+=1; t[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000
span=(int(win[-1]["End_Timestamp"])-int(win[0]["Start_Timestamp"]))/1000/nf
s=0
for n,(c,us) in sorted(tot.items(), key=lambda kv:-kv[1][1]):
    s+=us/nf
    print(n[:70].ljust(70), "calls/frame %6.1f  avg %7.1f us  per frame %8.1f us"%(c/nf, us/c, us/nf))
print("sum of kernel time per frame: %.1f us; first launch to last end of the window, per frame: %.1f us"%(s, span))
PY
NGP_PROBE_ONLY=1 NGP_HIP_RENDER_TRACE=1 timeout 300 python tools/render_probe.py $steps 1 2>&1 | grep "render pass" | tail -60 | awk '{print $4, $5, $6}' | tr '\n' ';' | cut -c1-1500
echo
