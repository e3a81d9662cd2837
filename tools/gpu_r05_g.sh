#!/bin/bash
# GPU box, round 5: the staged scatter, second version (count emits per-workgroup histograms, scan kernel, one-barrier scatter) — parity, stand-alone kernel times, the step
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=r05_g
timeout 1500 python -m pytest tests/test_grid_backward_gpu.py tests/test_network_gpu.py tests/test_gridmlp_gpu.py tests/test_netx_gpu.py tests/test_plumbing_gpu.py tests/test_dp_gpu.py tests/test_two_testbeds_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
tail -5 $out/${tag}_pytest.txt
base_ld=$LD_LIBRARY_PATH
for v in old dev hs1024; do
  if [ $v = dev ]; then d=$PWD/blender-ngp_amd/lib_dev; else d=$PWD/blender-ngp_amd/lib_ab/$v; fi
  export NGP_HIP_LIBRARY_DIR=$d LD_LIBRARY_PATH=$d:$base_ld
  for m in 0xffff 0xffe0; do
  rm -rf /tmp/tr_gb
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_gb -o t -- python tools/gb_level_probe.py --only $m --iters 100 > $out/${tag}_gb_${v}_$m.log 2>&1
  echo "== $v $m: $(grep '^mask' $out/${tag}_gb_${v}_$m.log)"
  python - <<PY
import csv,glob
f=glob.glob("/tmp/tr_gb/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if any(k in r["Name"] for k in ("gb_fx","gb_hs","grid_backward","grid_combine","nerf_backward_fused"))]
for r in rows: print("   ", r["Name"][:60].ljust(60), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000))
PY
  done
  timeout 300 python bench.py --steps 300 --warmup 5 --no_cpu_baseline --no_render --legs none > $out/${tag}_lego_$v.log 2>&1
  grep '^{' $out/${tag}_lego_$v.log | tail -1 > $out/${tag}_lego_${v}_line.json
  timeout 300 python bench_legs.py fox 300 > $out/${tag}_fox_$v.log 2>&1
  grep '^{' $out/${tag}_fox_$v.log | tail -1 > $out/${tag}_fox_${v}_line.json
  timeout 300 python bench_legs.py plumbing > $out/${tag}_plumbing_$v.log 2>&1
  python - <<PY
import json
for w in ("lego","fox"):
    try:
        l=json.load(open("$out/${tag}_%s_${v}_line.json"%w))
        print("   %s $v"%w, l["value"], l["ms_per_step"], {a:b.get("avg_us") for a,b in l.get("kernels",{}).items()})
    except Exception as e: print("   %s $v FAILED"%w, e)
try:
    l=json.loads([x for x in open("$out/${tag}_plumbing_$v.log") if x.startswith("{")][-1])
    for k in ("image","sdf"): print("   %s $v"%k, l[k]["ms_per_step"], l[k]["groups_us"], l[k]["network_pass"]["running"])
except Exception as e: print("   plumbing $v FAILED", e)
PY
done
