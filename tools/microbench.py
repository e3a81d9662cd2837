#!/usr/bin/env python
"""Kernel micro-benchmarks through the C ABI (GPU box only).  Inputs mimic a training step: samples are consecutive
fixed-step points along random rays through the unit cube (coherent like compacted ray samples), lego-sized network.

    python tools/microbench.py [--n 262144] [--iters 20] [--random]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def ray_coords(n, seed=0, run=48):
    rs = np.random.RandomState(seed)
    import capi
    c = np.zeros(n, dtype=capi.COORD)
    n_rays = (n + run - 1) // run
    o = 0.5 + (rs.rand(n_rays, 3) - 0.5) * 0.5
    d = rs.randn(n_rays, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = (np.arange(run) * (np.sqrt(3.0) / 1024.0))[None, :, None]
    pos = (o[:, None, :] + d[:, None, :] * t).reshape(-1, 3)[:n]
    c["pos"] = np.clip(pos, 0.0, 1.0).astype(np.float32)
    c["dir"] = np.repeat((d + 1.0) * 0.5, run, axis=0)[:n].astype(np.float32)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 18)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--random", action="store_true", help="uniform random positions instead of ray-coherent ones")
    a = ap.parse_args()
    import torch
    import capi
    import helpers as H
    from capi import check
    dev = torch.device("cuda:0")
    ngp = capi.load_ngp_hip()
    n = a.n
    desc = H.make_desc(ngp, 19)
    P = H.random_params(desc, 0, grid_amp=0.1)
    coords = H.random_coords(n, 1) if a.random else ray_coords(n)
    dl = (np.random.RandomState(2).randn(n, 4) * 0.01).astype(np.float16)
    d_desc, d_P, d_c, d_dl = H.to_dev(desc, dev), H.to_dev(P, dev), H.to_dev(coords, dev), H.to_dev(dl, dev)
    out, xs = H.dev_zeros(n * 8, dev), H.dev_zeros(n * 64, dev)
    npar = H.n_params(desc)
    grads = H.dev_zeros(npar * 2, dev)
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch = H.dev_zeros(sb, dev)
    master, m1, m2, ema = (H.dev_zeros(npar * 4, dev) for _ in range(4))
    inf = H.dev_zeros(npar * 2, dev)
    st = torch.cuda.current_stream().cuda_stream

    def timeit(name, fn, units, bytes_per_unit=None):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1000.0 * e0.elapsed_time(e1) / a.iters
        extra = "" if bytes_per_unit is None else "  %.0f GB/s algorithmic" % (bytes_per_unit * units / us / 1e3)
        print("%-28s %10.1f us   %8.2f Munits/s%s" % (name, us, units / us, extra), flush=True)
        return us

    timeit("nerf_inference", lambda: check(ngp.ngp_hip_nerf_inference(st, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4)), n, 588)
    timeit("nerf_density", lambda: check(ngp.ngp_hip_nerf_density(st, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr())), n, 588)
    timeit("nerf_forward", lambda: check(ngp.ngp_hip_nerf_forward(st, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr())), n, 588)
    for mode in os.environ.get("NGP_BWD_MODES", "0").split(","):
        os.environ["NGP_HIP_BWD_ABLATE"] = mode
        timeit("nerf_backward[ablate=%s]" % mode, lambda: check(ngp.ngp_hip_nerf_backward(st, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4,
                                                                                       grads.data_ptr(), scratch.data_ptr(), sb)), n, 1100)
    os.environ["NGP_HIP_BWD_ABLATE"] = "0"
    timeit("optimizer_step", lambda: check(ngp.ngp_hip_optimizer_step(st, npar, 10240, 5, H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-15), H.f32(1e-6), H.f32(128.0), H.f32(0.95), grads.data_ptr(),
                                                                      master.data_ptr(), d_P.data_ptr(), m1.data_ptr(), m2.data_ptr(), ema.data_ptr(), inf.data_ptr(), 3)), npar, 36)


if __name__ == "__main__":
    main()
