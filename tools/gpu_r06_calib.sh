#!/bin/bash
# GPU box: FETCH_SIZE calibration on this code base's access patterns (tools/fetch_calibration.hip)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_calibration.hip -o /tmp/fetch_calibration || exit 1
/tmp/fetch_calibration > $out/r06_calib_plain.txt; cat $out/r06_calib_plain.txt
rm -rf /tmp/cal1 /tmp/cal2 /tmp/cal3
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cal1 -o p -- /tmp/fetch_calibration > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d /tmp/cal2 -o p -- /tmp/fetch_calibration > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d /tmp/cal3 -o p -- /tmp/fetch_calibration > /dev/null 2>&1
python tools/fetch_calibration.py $out/r06_calib_plain.txt $out/r06_fetch_calibration.json /tmp/cal1 /tmp/cal2 /tmp/cal3
