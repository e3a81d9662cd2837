#!/bin/bash
# GPU box: the two traffic passes alone (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 --pmc runs with the kernel trace only) -> gpurun_out/<tag>_pmc_traffic.json.  $1 = tag
tag=${1:-pmc}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
small="python bench.py --steps 100 --warmup 300 --no_cpu_baseline --no_render"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o ${tag} -- $small > $out/${tag}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o ${tag} -- $small > $out/${tag}_pmc_write.log 2>&1
python tools/pmc_traffic.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_traffic.json ${tag} > $out/${tag}_pmc_traffic.log 2>&1
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
find $out -name '*.db' -delete
cat $out/${tag}_pmc_traffic.log
