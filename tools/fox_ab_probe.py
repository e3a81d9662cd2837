import os, sys, gc, json
sys.path[:0] = ["/root/repo", "/root/repo/blender-ngp_amd"]
import torch
import scene, bench_legs
mode = sys.argv[1]
dev = torch.device("cuda:0")
if mode == "torch1g":
    keep = torch.zeros(1 << 28, dtype=torch.float32, device=dev)   # 1 GiB of caching-allocator memory
elif mode == "hip1g":
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ptrs = []
    for _ in range(64):
        q = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(16 << 20)) == 0; ptrs.append(q)
elif mode == "untrained":
    ds = scene.make_dataset(100, 3, 800, dev)
    tb = scene.build_testbed(ds)
elif mode == "dataset_only":
    ds = scene.make_dataset(100, 3, 800, dev)
elif mode != "alone":
    ds = scene.make_dataset(100, 3, 800, dev)
    tb = scene.build_testbed(ds)
    tb.async_training_steps = True
    for _ in range(300): tb.frame()
    tb.sync()
    if mode == "render":
        import numpy as np
        tb.shall_train = False
        tb.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
        for _ in range(3): tb.render(800, 800, 1, True)
    if mode == "deleted":
        del tb; gc.collect()
r = bench_legs.fox_leg(100, {"nerf_backward": 1100, "nerf_inference": 588})
print(mode, r["ms_per_step"], r["value"])
