#!/usr/bin/env python3
"""GPU box: fixed vs per-iteration cost of the fused MLP kernels.  Runs ngp_hip_nerf_backward / ngp_hip_nerf_forward at batch sizes that give every workgroup
1, 2, 4, 8 iterations; run under `rocprofv3 --kernel-trace --output-format csv` and feed the trace to the same script with --trace to get the per-size averages:
  rocprofv3 --kernel-trace --output-format csv -d /tmp/fx -o t -- python tools/mlp_fixed_cost_probe.py; python tools/mlp_fixed_cost_probe.py --trace /tmp/fx"""
import csv
import glob
import os
import sys

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(_ROOT, "tests"), os.path.join(_ROOT, "blender-ngp_amd")]
SIZES = [1 << 16, 1 << 17, 1 << 18, 1 << 19]
REPS = 12


def run():
    import numpy as np
    import torch
    import capi
    import helpers as H
    from capi import check
    import torch  # noqa: F811 (before the kernel library: one HIP runtime per process)
    lib = os.environ.get("NGP_PROBE_LIB")
    ngp = capi.CLib(lib, os.path.join(_ROOT, "include", "ngp_hip.h"), ("ngp_hip_", "ngp_rccl_")) if lib else capi.load_ngp_hip()
    cuda = torch.device("cuda:0")
    desc = H.make_desc(ngp, log2_hashmap_size=19)
    params = H.random_params(desc, seed=0, grid_amp=0.5)
    d_desc, d_P = H.to_dev(desc, cuda), H.to_dev(params, cuda)
    for n in SIZES:
        coords = H.random_coords(n, seed=1)
        d_c = H.to_dev(coords, cuda)
        out, xs = H.dev_zeros(n * 4 * 2, cuda), H.dev_zeros(n * 32 * 2, cuda)
        dl = H.to_dev((np.random.RandomState(3).randn(n, 4) * 0.01).astype(np.float16), cuda)
        sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
        scratch, grads = H.dev_zeros(sb, cuda), H.dev_zeros(H.n_params(desc) * 2, cuda)
        for _ in range(REPS):
            check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
            check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), dl.data_ptr(), 4,
                                            grads.data_ptr(), scratch.data_ptr(), sb))
        torch.cuda.synchronize()


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for key in ("nerf_backward_fused_kernel", "nerf_forward_kernel"):
        ks = [r for r in rows if key in r["Kernel_Name"]]
        assert len(ks) == REPS * len(SIZES), (key, len(ks))
        print(key)
        prev = None
        for i, n in enumerate(SIZES):
            us = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in ks[i * REPS + 2:(i + 1) * REPS])
            med = us[len(us) // 2]
            print(f"  n = 2^{n.bit_length() - 1}: median {med:7.1f} us" + (f"   (+{med - prev:.1f} for the doubling)" if prev else ""))
            prev = med


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--trace":
        report(sys.argv[2])
    else:
        run()
