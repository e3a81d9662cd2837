timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
show() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
echo "== default"; timeout 120 python bench.py --steps 300 --warmup 300 --no_cpu_baseline --no_render 2>&1 | show
echo "== march late"; BENCH_MARCH_LATE=1 timeout 120 python bench.py --steps 300 --warmup 300 --no_cpu_baseline --no_render 2>&1 | show
