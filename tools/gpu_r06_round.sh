#!/bin/bash
# GPU box: the round-6 evidence set — whole GPU suite, smoke, the driver's short bench run (with its legs), the default bench line, kernel trace of the training leg,
# the two traffic passes (lego step and fox leg), MFMA counters, render / fox kernel stats, the preflight.  $1 = tag
tag=${1:-r06_p}
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $out/${tag}_pytest_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
timeout 120 python bench.py --gpus 1 --preflight 2>/dev/null | grep '^{' > $out/${tag}_preflight.json
timeout 900 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.log 2>&1
grep '^{' $out/${tag}_bench_driver.log | tail -1 > $out/${tag}_bench_driver_line.json
timeout 900 python bench.py > $out/${tag}_bench.log 2>&1
grep '^{' $out/${tag}_bench.log | tail -1 > $out/${tag}_bench_line.json
small="python bench.py --steps 100 --warmup 300 --no_cpu_baseline --no_render --legs none"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o ${tag} -- $small > $out/${tag}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o ${tag} -- $small > $out/${tag}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o ${tag} -- $small > $out/${tag}_pmc_write.log 2>&1
python tools/pmc_traffic.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_traffic.json ${tag} > $out/${tag}_pmc_traffic.log 2>&1
find $out/${tag}_trace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
foxcmd="python bench_legs.py fox 100"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fox_pmc_fetch -o ${tag} -- $foxcmd > $out/${tag}_fox_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_fox_pmc_write -o ${tag} -- $foxcmd > $out/${tag}_fox_pmc_write.log 2>&1
python tools/pmc_traffic.py $out/${tag}_fox_pmc_fetch $out/${tag}_fox_pmc_write $out/${tag}_pmc_traffic_fox.json ${tag} > $out/${tag}_pmc_traffic_fox.log 2>&1
rm -rf /tmp/tr_fox
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_fox -o t -- $foxcmd > $out/${tag}_fox_leg.log 2>&1
python - <<PY > $out/${tag}_fox_kernel_stats.txt 2>&1
import csv,glob
f=glob.glob("/tmp/tr_fox/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("kernel, calls, avg_us, total_ms, pct")
for r in rows[:28]: print(r["Name"][:110].ljust(110), r["Calls"], "%.1f"%(float(r["AverageNs"])/1000), "%.1f"%(float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
find $out -name '*.db' -delete
bash tools/gpu_mfma.sh ${tag} > $out/${tag}_mfma.log 2>&1
bash tools/render_kstats.sh > $out/${tag}_render_kernel_stats.txt 2>&1
tail -4 $out/${tag}_pytest_gpu.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; tail -2 $out/${tag}_smoke.log; cat $out/${tag}_preflight.json | cut -c1-200; cut -c1-500 $out/${tag}_bench_driver_line.json; echo; cut -c1-300 $out/${tag}_bench_line.json; echo
python - <<PY
import json
l=json.load(open("$out/${tag}_bench_driver_line.json"))
print("lego", l["value"], l["ms_per_step"], l.get("render_MP_per_s"), l.get("network_pass"), l["roofline"]["frac"], l["roofline"].get("traffic"))
for k in ("fox","bl_render","plumbing"):
    v=l.get(k,{})
    if k=="fox": print("fox kernels", {a:b["avg_us"] for a,b in v.get("kernels",{}).items()}); v={a:b for a,b in v.items() if a not in ("kernels",)}
    print(k, json.dumps(v)[:1400])
PY
# ---- round 6: the stand-in written to disk in the stock nerf-synthetic layout, benchmarked through --scene / --test_scene beside the in-memory run
python - <<PY > $out/${tag}_scene_ab.txt 2>&1
import json, os, subprocess, sys, torch
sys.path[:0] = ["blender-ngp_amd"]
import scene
ds = scene.make_dataset(100, 3, 800, torch.device("cuda:0"))
train = scene.write_dataset(ds, "/tmp/standin_lego", with_test=True, stock_keys=True)
del ds
def run(args):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1000", "--warmup", "50", "--no_cpu_baseline", "--legs", "none", "--n_test", "3"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    return json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
mem = run([])
disk = run(["--scene", train, "--test_scene", "/tmp/standin_lego/transforms_test.json"])
for name, l in (("in memory", mem), ("on disk  ", disk)):
    print("%s: %.4f ms/step, %.1f M samples/s, PSNR %.2f dB, render %.1f MP/s, data=%s, workload=%s" % (name, l["ms_per_step"], l["value"] / 1e6, l["psnr_db"], l["render_MP_per_s"], l["data"], l["config"]["workload"][:120]))
print("step time ratio %.4f, PSNR difference %.3f dB" % (disk["ms_per_step"] / mem["ms_per_step"], disk["psnr_db"] - mem["psnr_db"]))
PY
cat $out/${tag}_scene_ab.txt
bash tools/gpu_r06_calib.sh > $out/${tag}_calib.log 2>&1
