bash tools/gpu_r02_tests.sh r02_b; bash tools/gpu_mfma.sh r02_mfma 2>&1 | tail -20
