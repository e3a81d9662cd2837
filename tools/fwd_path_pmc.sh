#!/bin/bash
# GPU box: kernel times + L2 / fabric counters of the forward-pass variants (tools/fwd_path_probe.py child runs).  $1 = list of paths, e.g. "0 4"
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
N=${N:-519936}
for path in ${1:-0 4}; do
  export NGP_HIP_FWD_PATH=$path
  rm -rf /tmp/fp_*
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_t -o t -- python tools/fwd_path_probe.py child $path $N 20 > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d /tmp/fp_a -o p -- python tools/fwd_path_probe.py child $path $N 20 > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d /tmp/fp_b -o p -- python tools/fwd_path_probe.py child $path $N 20 > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d /tmp/fp_c -o p -- python tools/fwd_path_probe.py child $path $N 20 > /dev/null 2>&1
  python tools/pmc_generic.py $out/fwd_path_${path}_counters.json /tmp/fp_a /tmp/fp_b /tmp/fp_c
  python - <<PY
import csv,glob,json
f=glob.glob("/tmp/fp_t/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "ngp" in r["Name"]]
for r in rows: print("path $path", r["Name"][:70].ljust(70), r["Calls"], "%.1f us"%(float(r["AverageNs"])/1000))
d=json.load(open("$out/fwd_path_${path}_counters.json"))
for k,v in d.items(): print("path $path", k, v)
PY
done
