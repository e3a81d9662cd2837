#!/bin/bash
# GPU box: round-3 opening run — whole GPU suite with durations, smoke, the driver's short bench run.  $1 = tag
tag=${1:-r03_a}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $out/${tag}_pytest_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.log 2>&1
grep '^{' $out/${tag}_bench_driver.log | tail -1 > $out/${tag}_bench_driver_line.json
tail -25 $out/${tag}_pytest_gpu.log; tail -2 $out/${tag}_smoke.log; cut -c1-400 $out/${tag}_bench_driver_line.json
