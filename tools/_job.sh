#!/bin/bash
cd /root/repo
printf "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY\nSQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS\nSQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR\nSQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_IFETCH\n" | bash tools/gpu_pmc_quick.sh binq
python - <<'PY'
import json
d=json.load(open('gpurun_out/binq_counters.json'))
for k,v in d.items():
    if 'gb_fx' in k or 'grid_backward' in k: print(k, v)
PY
