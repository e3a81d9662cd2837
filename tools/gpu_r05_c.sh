#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
for v in leg noprofile noprops nobench probe; do
  timeout 200 python tools/fox_leg_bisect.py $v > $out/r05_c_bisect_$v.txt 2>&1
  grep "render pass" $out/r05_c_bisect_$v.txt | awk '{print $3, $4, $5}' | tr '\n' ';' | cut -c1-600; echo
  grep '^{' $out/r05_c_bisect_$v.txt
done
# hash-grid backward: where the scatter pass's time is (development build: NGP_HIP_GB_LEVELS masks), per kernel
export NGP_HIP_LIBRARY_DIR=$PWD/blender-ngp_amd/lib_dev LD_LIBRARY_PATH=$PWD/blender-ngp_amd/lib_dev:$LD_LIBRARY_PATH
for m in 0xffff 0x001f 0xffe0 0x0020 0x8000; do
  rm -rf /tmp/tr_gb
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_gb -o t -- python tools/gb_level_probe.py --only $m --iters 100 > $out/r05_c_gb_$m.log 2>&1
  echo "mask $m"; grep "^mask" $out/r05_c_gb_$m.log
  python - <<PY
import csv,glob
f=glob.glob("/tmp/tr_gb/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if any(k in r["Name"] for k in ("gb_fx","grid_backward","grid_combine","nerf_backward_fused"))]
for r in rows: print("   ", r["Name"][:60].ljust(60), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000))
PY
done
