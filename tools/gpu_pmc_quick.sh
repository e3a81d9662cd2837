#!/bin/bash
# run on the GPU box: a few counter groups over a short bench run.  $1 = tag; counter groups on stdin (one per line)
tag=${1:-q}
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
done
python tools/pmc_generic.py $out/${tag}_counters.json $dirs
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
find $out -name '*.db' -delete
