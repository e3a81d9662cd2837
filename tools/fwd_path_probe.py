#!/usr/bin/env python
"""dev tool (GPU box): the training forward pass on ray-coherent (or RANDOM_POS=1: uniform random) samples of a training step's size, one process per variant:
    path 0  ngp_hip_nerf_forward — the fused kernel (product path of the training step)
    path 5  ngp_hip_nerf_forward_ws — XCD-affine encode into level planes + MLP kernel (product path of the renderers and the occupancy update)
    path 9  the same, queues cut by the measured per-level cost (NGP_HIP_ENC_COST=1)
    path 6 / 7 / 8  ngp_hip_nerf_forward_ws with one encode launch per group of levels (NGP_HIP_ENC_MODE=1): 2.5 MiB groups with / without pair loads, 4.5 MiB groups
prints the time per call and checks that outputs and saved encodings are bit-identical to path 0.
    python tools/fwd_path_probe.py            (runs itself once per path)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def child(path, n, iters):
    import torch
    import capi
    import helpers as H
    from capi import check
    from microbench import ray_coords
    dev = torch.device("cuda:0")
    ngp = capi.load_ngp_hip()
    desc = H.make_desc(ngp, 19)
    P = H.random_params(desc, 0, grid_amp=0.1)
    coords = H.random_coords(n, 1) if os.environ.get("RANDOM_POS") else ray_coords(n, run=int(os.environ.get("RUN", "48")))
    d_desc, d_P, d_c = H.to_dev(desc, dev), H.to_dev(P, dev), H.to_dev(coords, dev)
    out, xs = H.dev_zeros(n * 8, dev), H.dev_zeros(n * 64, dev)
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: check(ngp.ngp_hip_nerf_forward(st, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    if path >= 5:   # the XCD-affine two-kernel path that the renderers use
        wsb = ngp.ngp_hip_nerf_encode_workspace_bytes(n)
        ws = H.dev_zeros(wsb, dev)
        fn = lambda: check(ngp.ngp_hip_nerf_forward_ws(st, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr(), ws.data_ptr(), wsb))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1000.0 * e0.elapsed_time(e1) / iters
    o, x = H.to_host(out, np.uint16), H.to_host(xs, np.uint16)
    np.save("/tmp/fwd_path_%d_out.npy" % path, o)
    np.save("/tmp/fwd_path_%d_x.npy" % path, x)
    same = ""
    if path != 0 and os.path.exists("/tmp/fwd_path_0_out.npy"):
        same = "  out identical: %s  x identical: %s" % ((np.load("/tmp/fwd_path_0_out.npy") == o).all(), (np.load("/tmp/fwd_path_0_x.npy") == x).all())
    print("path %d: %8.1f us for %d samples (%.0f GB/s algorithmic)%s" % (path, us, n, 588.0 * n / us / 1e3, same), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        n = int(os.environ.get("N", str(520000 // 128 * 128)))
        for path in [int(x) for x in os.environ.get("PATHS", "0,5,9,8").split(",")]:
            env = dict(os.environ, NGP_HIP_FWD_PATH=str(path))
            if path == 9:   # the XCD-affine queues cut by the measured per-level cost
                env.update(NGP_HIP_ENC_COST="1")
            elif path >= 6:   # 6: per-level-group launches, pairs; 7: the same with 4-byte gathers; 8: 4.5 MiB groups
                env.update(NGP_HIP_ENC_MODE="1", NGP_HIP_ENC_PAIR="0" if path == 7 else "1", NGP_HIP_ENC_GROUP_KIB="4608" if path == 8 else "2560")
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(path), str(n), "30"], env=env, check=False)
