#!/usr/bin/env python3
"""GPU box, dev: per-ray cycle stamps of compute_loss_kernel (a kernel library built with the stamps, LD_LIBRARY_PATH=blender-ngp_amd/lib_t): where a ray's wave spends its time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd")]
import numpy as np, torch, scene
dev = torch.device("cuda:0")
ds = scene.make_dataset(100, 3, 800, dev)
tb = scene.build_testbed(ds)
tb.async_training_steps = True
tb.prefetch_samples = False   # nothing beside the kernel
while tb.training_step < 1005: tb.frame()
tb.sync()
lib = ctypes.CDLL(os.path.join(ROOT, "blender-ngp_amd", "lib_t", "libngp_hip.so"))
buf = np.zeros(8192 * 8, dtype=np.uint64)
assert lib.ngp_hip_debug_loss_timing(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(-1, 8)
t = t[t[:, 7] != 0]
tag = t[np.argmax(t[:, 0]), 7]          # the last launch's generator state: rows of older launches (more ray slots then) are left out
t = t[t[:, 7] == tag]
t0 = t[:, 0].min()
f = 2.4e9   # s_memtime counts shader-clock cycles (~2.4 GHz under load)
us = lambda x: x.astype(np.float64) / f * 1e6
print("rays with stamps:", len(t), " kernel span (first start -> last end): %.1f us" % us(t[:, 4].max() - t0))
print("start offsets (us): median %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(us(t[:, 0] - t0), [50, 90, 100])))
for name, a, b in (("setup", 0, 1), ("pass 1", 1, 2), ("barrier + atomic", 2, 3), ("loss + pass 2", 3, 4), ("whole wave", 0, 4)):
    d = us(t[:, b] - t[:, a])
    print("%-18s median %6.2f  mean %6.2f  p90 %6.2f  max %6.2f us" % (name, np.median(d), d.mean(), np.percentile(d, 90), d.max()))
ns = t[:, 5].astype(np.int64); d = us(t[:, 2] - t[:, 1])
for lo, hi in ((1, 64), (65, 128), (129, 192), (193, 256), (257, 400)):
    m = (ns >= lo) & (ns <= hi)
    if m.any(): print("pass 1 of rays with %3d-%3d samples: %5d rays, median %.2f us" % (lo, hi, m.sum(), np.median(d[m])))
