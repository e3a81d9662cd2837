#!/bin/bash
# GPU box: the round-4 evidence set — whole GPU suite, smoke, the driver's short bench run (with its legs), the default bench line, kernel trace of the training leg,
# the two traffic passes, MFMA counters, render / fox / Blender-renderer / variants kernel stats.  $1 = tag
tag=${1:-r04_k}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -x -q --durations=15 > $out/${tag}_pytest_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.log 2>&1
grep '^{' $out/${tag}_bench_driver.log | tail -1 > $out/${tag}_bench_driver_line.json
timeout 900 python bench.py > $out/${tag}_bench.log 2>&1
grep '^{' $out/${tag}_bench.log | tail -1 > $out/${tag}_bench_line.json
small="python bench.py --steps 100 --warmup 300 --no_cpu_baseline --no_render --legs none"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o ${tag} -- $small > $out/${tag}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o ${tag} -- $small > $out/${tag}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o ${tag} -- $small > $out/${tag}_pmc_write.log 2>&1
python tools/pmc_traffic.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_traffic.json ${tag} > $out/${tag}_pmc_traffic.log 2>&1
cp $out/${tag}_trace/*/*kernel_stats.csv $out/${tag}_bench_kernel_stats.csv 2>/dev/null || find $out/${tag}_trace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
find $out -name '*.db' -delete
bash tools/gpu_mfma.sh ${tag} > $out/${tag}_mfma.log 2>&1
bash tools/render_kstats.sh > $out/${tag}_render_kernel_stats.txt 2>&1
bash tools/gpu_fox_leg.sh ${tag} > /dev/null 2>&1
bash tools/gpu_bl_leg.sh ${tag} > /dev/null 2>&1
timeout 300 python bench_legs.py variants 2>/dev/null | grep '^{' > $out/${tag}_variants_line.json
tail -4 $out/${tag}_pytest_gpu.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; tail -2 $out/${tag}_smoke.log; cut -c1-400 $out/${tag}_bench_driver_line.json; echo; cut -c1-300 $out/${tag}_bench_line.json
