#!/bin/bash
# GPU box, round 5: (1) the fox leg's 18 ms in front of every frame's first tracer pass — kernel / copy timeline of the last frames; (2) counters of the scatter pass
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=r05_e
rm -rf /tmp/tr_leg
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr_leg -o t -- python tools/fox_leg_bisect.py leg > $out/${tag}_leg.txt 2>&1
grep '^{' $out/${tag}_leg.txt
python - <<'PY' > gpurun_out/r05_e_leg_timeline.txt 2>&1
import csv, glob
rows = []
for f in glob.glob("/tmp/tr_leg/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70], r.get("Queue_Id", "")))
for f in glob.glob("/tmp/tr_leg/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy")) , ""))
rows.sort()
# the last frame: find the last 3 init_rays kernels
idx = [i for i, r in enumerate(rows) if "init_rays" in r[2]]
print("n rows", len(rows), "init_rays at", idx[-4:])
start = idx[-3] - 12
t0 = rows[start][0]
prev_end = t0
for s, e, name, q in rows[start:]:
    print("%10.3f ms  dur %9.3f ms  gap %8.3f ms  q%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, q, name))
    prev_end = max(prev_end, e)
PY
head -150 gpurun_out/r05_e_leg_timeline.txt | cut -c1-150
# (2) the scatter pass, hashed levels only, round-4 scatter vs the LDS-staged one: SQ counters
base_ld=$LD_LIBRARY_PATH
for v in old dev; do
  if [ $v = dev ]; then d=$PWD/blender-ngp_amd/lib_dev; else d=$PWD/blender-ngp_amd/lib_ab/$v; fi
  export NGP_HIP_LIBRARY_DIR=$d LD_LIBRARY_PATH=$d:$base_ld
  i=0; dirs=""
  while read -r group; do
    i=$((i+1)); rm -rf /tmp/pm_$i
    timeout 200 rocprofv3 --kernel-trace --pmc $group --output-format csv -d /tmp/pm_$i -o p -- python tools/gb_level_probe.py --only 0xffe0 --iters 30 > /dev/null 2>&1
    dirs="$dirs /tmp/pm_$i"
  done <<'G'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INSTS_GDS
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
G
  python tools/pmc_generic.py $out/${tag}_scatter_${v}_counters.json $dirs
  python - <<PY
import json
d=json.load(open("$out/${tag}_scatter_${v}_counters.json"))
for k,v in d.items():
    if "gb_fx_bin_kernelILi3ELb1" in k: print("$v scatter", json.dumps(v))
PY
done
