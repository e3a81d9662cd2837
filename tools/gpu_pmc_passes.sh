#!/bin/bash
# run on the GPU box: one rocprofv3 pass per counter group over a short bench run; only per-kernel averages are kept.  $1 = tag
tag=${1:-pmc}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
small="python bench.py --steps 60 --warmup 200 --no_cpu_baseline --no_render"
i=0
dirs=""
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $out/${tag}_p$i -o p -- $small > $out/${tag}_p$i.log 2>&1
  dirs="$dirs $out/${tag}_p$i"
done <<'GROUPS'
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_WAVES
GROUPS
python tools/pmc_generic.py $out/${tag}_counters.json $dirs
tail -3 $out/${tag}_p1.log
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
find $out -name '*.db' -delete
