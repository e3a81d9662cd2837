#!/bin/bash
# GPU box, round 5: fp16-wire exchange / sharded Ema / stale guards (one-rank), frames through the pinned staging buffer, the preflight, counters of the scatter pass
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
tag=r05_f
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_render_gpu.py tests/test_bl_render_gpu.py tests/test_network_pass_gpu.py tests/test_gridmlp_gpu.py tests/test_render_modes_e2e_gpu.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
tail -6 $out/${tag}_pytest.txt
timeout 120 python bench.py --gpus 1 --preflight 2>&1 | tail -3
timeout 300 python bench_legs.py fox 300 > $out/${tag}_fox.log 2>&1
grep '^{' $out/${tag}_fox.log | tail -1 > $out/${tag}_fox_line.json
python - <<PY
import json
l=json.load(open("$out/${tag}_fox_line.json"))
print("fox", l["value"], l["ms_per_step"], "render", l.get("render_MP_per_s"), l.get("render_ms_frames"), l.get("render_network_samples_per_frame"), l["network_pass"], {a:b["avg_us"] for a,b in l["kernels"].items()})
PY
for v in leg probe; do
  timeout 200 python tools/fox_leg_bisect.py $v > $out/${tag}_bisect_$v.txt 2>&1
  grep "render:" $out/${tag}_bisect_$v.txt | tail -1; grep '^{' $out/${tag}_bisect_$v.txt
done
timeout 300 python bench.py --steps 300 --warmup 5 --no_cpu_baseline --legs bl_render > $out/${tag}_lego.log 2>&1
grep '^{' $out/${tag}_lego.log | tail -1 > $out/${tag}_lego_line.json
python - <<PY
import json
l=json.load(open("$out/${tag}_lego_line.json"))
print("lego", l["value"], l["ms_per_step"], "render", l.get("render_MP_per_s"), l.get("render_ms_per_frame"))
print("bl_render", json.dumps(l.get("bl_render"))[:1200])
PY
base_ld=$LD_LIBRARY_PATH
for v in dev staged; do
  if [ $v = dev ]; then d=$PWD/blender-ngp_amd/lib_dev; else d=$PWD/blender-ngp_amd/lib_ab/$v; fi
  export NGP_HIP_LIBRARY_DIR=$d LD_LIBRARY_PATH=$d:$base_ld
  i=0; dirs=""
  for group in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
               "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC" \
               "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
               "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
               "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/pm_$i
    timeout 200 rocprofv3 --kernel-trace --pmc $group --output-format csv -d /tmp/pm_$i -o p -- python tools/gb_level_probe.py --only 0xffe0 --iters 30 > $out/${tag}_pmc_${v}_$i.log 2>&1 < /dev/null
    dirs="$dirs /tmp/pm_$i"
  done
  python tools/pmc_generic.py $out/${tag}_scatter_${v}_counters.json $dirs
  python - <<PY
import json
d=json.load(open("$out/${tag}_scatter_${v}_counters.json"))
for k,v in d.items():
    if "gb_fx_bin_kernelILi3ELb1" in k: print("$v scatter", json.dumps(v))
PY
done
tail -3 $out/${tag}_pmc_dev_1.log
