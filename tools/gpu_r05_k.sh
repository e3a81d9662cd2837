#!/bin/bash
# GPU box, round 5: the banded grid backward of 2-D ordered batches (image fitting, config #1) — parity and the image leg
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=r05_k
timeout 1500 python -m pytest tests/test_grid_backward_gpu.py tests/test_gridmlp_gpu.py tests/test_plumbing_gpu.py tests/test_network_pass_gpu.py tests/test_baseline_configs_gpu.py tests/test_dp_gpu.py tests/test_two_testbeds_gpu.py tests/test_sampling_gpu.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
grep -n "passed\|failed" $out/${tag}_pytest.txt | tail -3
rm -rf /tmp/tr_p
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_p -o t -- python bench_legs.py plumbing > $out/${tag}_plumbing.log 2>&1
python - <<PY
import json,csv,glob
l=json.loads([x for x in open("$out/${tag}_plumbing.log") if x.startswith("{")][-1])
for k in ("image","sdf"): print(k, l[k]["ms_per_step"], l[k]["groups_us"], l[k]["network_pass"]["running"], "loss", l[k]["loss"])
f=glob.glob("/tmp/tr_p/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:24]: print("   ", r["Name"][:90].ljust(90), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000))
PY
