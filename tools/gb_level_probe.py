#!/usr/bin/env python
"""Per-level cost of the hash-grid backward (GPU box only): times ngp_hip_nerf_backward with NGP_HIP_GB_LEVELS masks.  Needs the development build of the kernel
library (python blender-ngp_amd/build.py --dev; NGP_HIP_LIBRARY_DIR=blender-ngp_amd/lib_dev): the product library has no environment knobs.

    python tools/gb_level_probe.py [--n 262144]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 18)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=lambda v: int(v, 0), default=None, help="run this level mask only (for rocprofv3 --kernel-trace --stats runs: per-kernel times of the group under one mask)")
    a = ap.parse_args()
    import torch
    import capi
    import helpers as H
    from capi import check
    from microbench import ray_coords
    dev = torch.device("cuda:0")
    ngp = capi.load_ngp_hip()
    n = a.n
    desc = H.make_desc(ngp, 19)
    P = H.random_params(desc, 0, grid_amp=0.1)
    coords = ray_coords(n)
    dl = (np.random.RandomState(2).randn(n, 4) * 0.01).astype(np.float16)
    d_desc, d_P, d_c, d_dl = H.to_dev(desc, dev), H.to_dev(P, dev), H.to_dev(coords, dev), H.to_dev(dl, dev)
    out, xs = H.dev_zeros(n * 8, dev), H.dev_zeros(n * 64, dev)
    npar = H.n_params(desc)
    grads = H.dev_zeros(npar * 2, dev)
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch = H.dev_zeros(sb, dev)
    st = torch.cuda.current_stream().cuda_stream
    check(ngp.ngp_hip_nerf_forward(st, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))

    def bwd():
        check(ngp.ngp_hip_nerf_backward(st, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4,
                                        grads.data_ptr(), scratch.data_ptr(), sb))

    def timeit(mask):
        os.environ["NGP_HIP_GB_LEVELS"] = hex(mask)
        for _ in range(3):
            bwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            bwd()
        e1.record()
        torch.cuda.synchronize()
        return 1000.0 * e0.elapsed_time(e1) / a.iters

    if a.only is not None:
        print("mask %s: %.1f us per call" % (hex(a.only), timeit(a.only)))
        return
    base = timeit(0)
    print("no level: %.1f us (everything but the owner blocks)" % base)
    for name, mask in [("all", 0xffff), ("dense 0-4", 0x1f), ("hashed 5-15", 0xffe0)] + [("level %d" % l, 1 << l) for l in range(16)]:
        print("%-12s +%.1f us" % (name, timeit(mask) - base), flush=True)


if __name__ == "__main__":
    main()
