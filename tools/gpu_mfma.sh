#!/bin/bash
# GPU box: matrix-core counters of the MFMA kernels over a short bench run (separate --pmc passes, kernel trace only).  $1 = tag
tag=${1:-r02_mfma}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
small="python bench.py --steps 40 --warmup 280 --no_cpu_baseline --no_render --legs none"
i=0; dirs=""
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $out/${tag}_p$i -o p -- $small > $out/${tag}_p$i.log 2>&1
  dirs="$dirs $out/${tag}_p$i"
done <<'GROUPS'
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
MfmaUtil VALUBusy
GROUPS
python tools/mfma_summary.py $out/${tag}.json $dirs
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
find $out -name '*.db' -delete
