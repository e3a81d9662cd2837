timeout 600 python -m pytest tests/test_network_gpu.py tests/test_sampling_gpu.py tests/test_step_schedule_gpu.py -x -q 2>&1 | tail -3
show() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
for i in 1 2; do echo "== default"; timeout 120 python bench.py --steps 300 --warmup 300 --no_cpu_baseline --no_render 2>&1 | show; done
echo "== lane-per-ray"; NGP_HIP_GEN_MODE=1 timeout 120 python bench.py --steps 300 --warmup 300 --no_cpu_baseline --no_render 2>&1 | show
bash tools/kstats.sh 12
