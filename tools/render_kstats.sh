#!/bin/bash
# GPU box: per-kernel totals of the stock renderer (rocprofv3 kernel trace over tools/render_probe.py: train $1 steps, render $2 frames), per frame, and the pass structure of one frame
export TMPDIR=/tmp
steps=${1:-1000}; frames=${2:-20}
rm -rf /tmp/rtr
NGP_PROBE_ONLY=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rtr -o t -- python tools/render_probe.py $steps $frames > /tmp/render_probe.out 2>&1
tail -1 /tmp/render_probe.out
python - <<PY
import csv,glob
f=glob.glob("/tmp/rtr/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::" not in r["Name"]]
keep=[r for r in rows if any(k in r["Name"] for k in ("init_rays","advance_pos","compact_rays","generate_next_inputs","composite","shade_kernel","encode_planes","nerf_forward_kernelILi0ELi1","accumulate","tonemap","rocclr"))]
keep.sort(key=lambda r:-float(r["TotalDurationNs"]))
nf=$frames+2   # + the evaluation frame + the warm-up frame
tot=0
for r in keep:
    per=float(r["TotalDurationNs"])/1000/nf; tot+=per
    print(r["Name"][:70].ljust(70), "calls/frame %6.1f  avg %7.1f us  per frame %8.1f us"%(int(r["Calls"])/nf, float(r["AverageNs"])/1000, per))
print("sum of kernel time per frame (the training step's encode_planes calls of the occupancy update are in the encode row): %.1f us"%tot)
PY
NGP_PROBE_ONLY=1 NGP_HIP_RENDER_TRACE=1 timeout 300 python tools/render_probe.py $steps 1 2>&1 | grep "render pass" | tail -60 | awk '{print $4, $5, $6}' | tr '\n' ';' | cut -c1-1500
echo
