#!/usr/bin/env python3
"""GPU box: kernel times of the backward group on ray-ordered synthetic samples (2^18, 48-sample rays), stand-alone (nothing beside it).  Under rocprofv3:
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o t -- python tools/gb_phase_probe.py; python tools/gb_phase_probe.py --trace /tmp/gp
NGP_PROBE_LIB picks a kernel library built with knock-out macros (dev only)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]


def run():
    import numpy as np
    import torch
    import capi
    import helpers as H
    from capi import check
    from microbench import ray_coords
    dev = torch.device("cuda:0")
    lib = os.environ.get("NGP_PROBE_LIB")
    ngp = capi.CLib(lib, os.path.join(ROOT, "include", "ngp_hip.h"), ("ngp_hip_", "ngp_rccl_")) if lib else capi.load_ngp_hip()
    n = 1 << 18
    desc = H.make_desc(ngp, 19)
    P = H.random_params(desc, 0, grid_amp=0.1)
    coords = ray_coords(n)
    dl = (np.random.RandomState(2).randn(n, 4) * 0.01).astype(np.float16)
    d_desc, d_P, d_c, d_dl = H.to_dev(desc, dev), H.to_dev(P, dev), H.to_dev(coords, dev), H.to_dev(dl, dev)
    out, xs = H.dev_zeros(n * 8, dev), H.dev_zeros(n * 64, dev)
    grads = H.dev_zeros(H.n_params(desc) * 2, dev)
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch = H.dev_zeros(sb, dev)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    for _ in range(24):
        check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4,
                                        grads.data_ptr(), scratch.data_ptr(), sb))
    torch.cuda.synchronize()


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        nm = r["Name"]
        if any(k in nm for k in ("gb_fx", "grid_backward", "grid_combine", "nerf_backward_fused", "wgrad")):
            tag = ("scatter" if "ILi3ELb1" in nm else "count") if "gb_fx_bin" in nm else ""
            print("  %-34s %-8s %7.1f us" % (nm.split("(")[0].replace("_ZN3ngp", "")[:34], tag, float(r["AverageNs"]) / 1000))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--trace":
        report(sys.argv[2])
    else:
        run()
