#!/bin/bash
# run on the GPU box (via gpurun): default bench line + kernel trace + the two PMC passes.  $1 = tag (e.g. r01_b)
tag=${1:-r01}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
timeout 600 python bench.py > $out/${tag}_bench.log 2>&1
grep '^{' $out/${tag}_bench.log | tail -1 > $out/${tag}_bench_line.json
small="python bench.py --steps 100 --warmup 300 --no_cpu_baseline --no_render"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o ${tag} -- $small > $out/${tag}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o ${tag} -- $small > $out/${tag}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o ${tag} -- $small > $out/${tag}_pmc_write.log 2>&1
# keep only the summaries (raw traces are large)
python tools/pmc_traffic.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_traffic.json ${tag} > $out/${tag}_pmc_traffic.log 2>&1
head -3 $out/${tag}_pmc_fetch/*counter_collection.csv > $out/${tag}_pmc_fetch_head.csv
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
find $out -name '*.db' -delete
ls -la $out/${tag}_trace $out/${tag}_pmc_fetch $out/${tag}_pmc_write 2>&1 | head -40
tail -3 $out/${tag}_pytest_gpu.log; tail -2 $out/${tag}_smoke.log; tail -2 $out/${tag}_bench_line.json | cut -c1-300
