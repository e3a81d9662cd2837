for a in 0 1 2 3; do echo "== fwd ablate $a"; NGP_HIP_FWD_ABLATE=$a NGP_BWD_MODES=0 timeout 120 python tools/microbench.py --iters 20 2>&1 | grep "nerf_forward"; done
