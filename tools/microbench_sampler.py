#!/usr/bin/env python
"""Times generate_training_samples on a REAL training state (trained occupancy grid, lego-like cameras), with dev variants.
    python tools/microbench_sampler.py [--train_steps 600] [--res 400] [--n_train 50]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train_steps", type=int, default=600)
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--n_train", type=int, default=50)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--variants", default="0,1,2,3")
    a = ap.parse_args()
    import torch
    import capi
    import helpers as H
    import scene
    from capi import check
    dev = torch.device("cuda:0")
    ds = scene.make_dataset(a.n_train, 1, a.res, dev)
    tb = scene.build_testbed(ds)
    scene.train(tb, a.train_steps)
    tb.sync()
    P = tb.debug_pointers()
    ngp = capi.load_ngp_hip()
    R = int(P["rays_per_batch"])
    max_samples = 16 << 18
    aabb = H.unit_aabb()
    bufs = dict(rc=H.dev_zeros(4, dev), nc=H.dev_zeros(4, dev), idx=H.dev_zeros(R * 4, dev), rays=H.dev_zeros(R * 24, dev), ns=H.dev_zeros(R * 8, dev), co=H.dev_zeros(max_samples * 28, dev))
    dist = H.dev_zeros(32 * 32 * 2 * 4, dev)
    dres = np.array([32, 32], np.int32)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        bufs["rc"].zero_(); bufs["nc"].zero_()
        check(ngp.ngp_hip_generate_training_samples(st, R, aabb.ctypes.data, max_samples, int(P["rng_state"]), int(P["rng_inc"]), bufs["rc"].data_ptr(), bufs["nc"].data_ptr(),
                                                    bufs["idx"].data_ptr(), bufs["rays"].data_ptr(), bufs["ns"].data_ptr(), bufs["co"].data_ptr(), int(P["n_images"]), int(P["metadata"]),
                                                    int(P["xforms"]), int(P["bitfield"]), 0, None, 0, 0, H.f32(P["cone_angle_constant"]), dist.data_ptr(), dres.ctypes.data, 0, R, None, None, 0))

    for v in a.variants.split(","):
        os.environ["NGP_HIP_GEN_VARIANT"] = v
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        n_kept = int(H.to_host(bufs["rc"], np.uint32)[0]); n_s = int(H.to_host(bufs["nc"], np.uint32)[0])
        print("variant %s: %8.1f us   rays %d kept %d samples %d (%.1f / kept ray)" % (v, 1000 * e0.elapsed_time(e1) / a.iters, R, n_kept, n_s, n_s / max(n_kept, 1)), flush=True)
    ns = H.to_host(bufs["ns"], np.uint32)[0:2 * n_kept:2]
    print("numsteps per ray: mean %.1f p50 %d p90 %d p99 %d max %d" % (ns.mean(), np.percentile(ns, 50), np.percentile(ns, 90), np.percentile(ns, 99), ns.max()))


if __name__ == "__main__":
    main()
