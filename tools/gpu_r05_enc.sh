#!/bin/bash
# GPU box: the XCD-affine encoder's persistent-workgroup count and items per claim (development knobs NGP_HIP_ENC_BLOCKS / NGP_HIP_ENC_CLAIM) on the samples of a real fox
# training step and on random points of four pass sizes (tools/encode_sweep_probe.py); "rule" = the library's own size-dependent choice
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 300 python tools/encode_sweep_probe.py capture 2>&1 | tail -1
d=$PWD/blender-ngp_amd/lib_dev
export NGP_HIP_LIBRARY_DIR=$d LD_LIBRARY_PATH=$d:$LD_LIBRARY_PATH
echo "rule $(timeout 300 python tools/encode_sweep_probe.py time 2>/dev/null | grep '^{')" | tee $out/r05_enc_sweep2.txt
for b in 512 768 1024 1536; do for c in 1 2 4 8; do
  echo "blocks $b claim $c $(NGP_HIP_ENC_BLOCKS=$b NGP_HIP_ENC_CLAIM=$c timeout 300 python tools/encode_sweep_probe.py time 2>/dev/null | grep '^{')"
done; done | tee -a $out/r05_enc_sweep2.txt
echo "rule $(timeout 300 python tools/encode_sweep_probe.py time 2>/dev/null | grep '^{')" | tee -a $out/r05_enc_sweep2.txt
