#!/bin/bash
# run on the GPU box: the BASELINE-size tests (verbose), the whole gpu suite, then the driver's bench invocation and the default one.  $1 = tag
tag=${1:-r02}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py -x -q -s > $out/${tag}_baseline_configs.log 2>&1
tail -5 $out/${tag}_baseline_configs.log
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_baseline_configs_gpu.py > $out/${tag}_pytest_gpu.log 2>&1
tail -3 $out/${tag}_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.log 2>&1
grep '^{' $out/${tag}_bench_driver.log | tail -1 > $out/${tag}_bench_driver_line.json
cut -c1-600 $out/${tag}_bench_driver_line.json
timeout 600 python bench.py > $out/${tag}_bench.log 2>&1
grep '^{' $out/${tag}_bench.log | tail -1 > $out/${tag}_bench_line.json
cut -c1-400 $out/${tag}_bench_line.json
