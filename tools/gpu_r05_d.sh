#!/bin/bash
# GPU box, round 5: the hash-grid backward's scatter pass write-combined through LDS (gb_scatter_staged) — parity, stand-alone kernel times and the step, against the
# round-4 scatter (lib_ab/old) and two other chunk sizes; and the fox render leg's missing 25 ms per frame (stage times of a slow and a fast process)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=r05_d
timeout 900 python -m pytest tests/test_grid_backward_gpu.py tests/test_network_gpu.py tests/test_gridmlp_gpu.py tests/test_netx_gpu.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
tail -4 $out/${tag}_pytest.txt
base_ld=$LD_LIBRARY_PATH
for v in old hs256 dev hs1024; do
  if [ $v = dev ]; then d=$PWD/blender-ngp_amd/lib_dev; else d=$PWD/blender-ngp_amd/lib_ab/$v; fi
  export NGP_HIP_LIBRARY_DIR=$d LD_LIBRARY_PATH=$d:$base_ld
  rm -rf /tmp/tr_gb
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_gb -o t -- python tools/gb_level_probe.py --only 0xffff --iters 100 > $out/${tag}_gb_$v.log 2>&1
  echo "== $v: $(grep '^mask' $out/${tag}_gb_$v.log)"
  python - <<PY
import csv,glob
f=glob.glob("/tmp/tr_gb/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if any(k in r["Name"] for k in ("gb_fx","grid_backward","grid_combine","nerf_backward_fused"))]
for r in rows: print("   ", r["Name"][:60].ljust(60), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000))
PY
  timeout 300 python bench.py --steps 300 --warmup 5 --no_cpu_baseline --no_render --legs none > $out/${tag}_lego_$v.log 2>&1
  grep '^{' $out/${tag}_lego_$v.log | tail -1 > $out/${tag}_lego_${v}_line.json
  timeout 300 python bench_legs.py fox 300 > $out/${tag}_fox_$v.log 2>&1
  grep '^{' $out/${tag}_fox_$v.log | tail -1 > $out/${tag}_fox_${v}_line.json
  python - <<PY
import json
for w in ("lego","fox"):
    try:
        l=json.load(open("$out/${tag}_%s_${v}_line.json"%w))
        print("   %s $v"%w, l["value"], l["ms_per_step"], {a:b.get("avg_us") for a,b in l.get("kernels",{}).items()})
    except Exception as e: print("   %s $v FAILED"%w, e)
PY
done
unset NGP_HIP_LIBRARY_DIR; export LD_LIBRARY_PATH=$base_ld
rm -rf /tmp/tr_k
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_k -o t -- python bench.py --steps 100 --warmup 300 --no_cpu_baseline --no_render --legs none > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/tr_k/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("lego step, product library, beside the march:")
for r in rows[:14]: print("   ", r["Name"][:80].ljust(80), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000))
PY
for v in leg probe; do
  timeout 200 python tools/fox_leg_bisect.py $v > $out/${tag}_bisect_$v.txt 2>&1
  grep "render pass\|render:" $out/${tag}_bisect_$v.txt | cut -c1-120 | tr '\n' ';' | cut -c1-1500; echo
  grep '^{' $out/${tag}_bisect_$v.txt
done
