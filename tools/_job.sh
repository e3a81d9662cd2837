timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30
timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step'], d.get('psnr_at_bench'), d.get('render_MP_per_s'), {k:v['avg_us'] for k,v in d['kernels'].items()})"
