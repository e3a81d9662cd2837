"""dev tool: step time of the plumbing configs (image 1024^2-like, sdf sphere) at the reference batch 2^18"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd")]
import numpy as np, torch, pyngp
CFG = os.path.join(ROOT, "blender-ngp_amd", "configs")
B = 1 << 18
rs = np.random.RandomState(0)
y, x = np.mgrid[0:1024, 0:1024].astype(np.float32) / 1024
img = np.stack([0.5 + 0.4 * np.sin(30 * x) * np.cos(23 * y), (x - 0.5) ** 2 + (y - 0.5) ** 2 < 0.1, x * y, np.ones_like(x)], -1).astype(np.float32)
for mode in ("image", "sdf"):
    tb = pyngp.Testbed(pyngp.TestbedMode.Image if mode == "image" else pyngp.TestbedMode.Sdf)
    if mode == "image":
        tb.set_image_data(img)
    else:
        pts = rs.rand(1 << 20, 3).astype(np.float32)
        tb.override_sdf_training_data(pts, (np.linalg.norm(pts - 0.5, axis=1) - 0.3).astype(np.float32))
    tb.reload_network_from_file(os.path.join(CFG, mode, "base.json"))
    tb.shall_train = True
    for _ in range(100):
        tb.train(B)
    tb.set_profiling(True); tb.reset_profile()
    tb.sync(); t0 = time.perf_counter()
    N = 300
    for _ in range(N):
        tb.train(B)
    tb.sync(); dt = (time.perf_counter() - t0) / N
    prof = tb.profile()
    print(mode, "ms/step %.3f  samples/s %.1fM  loss %.5f" % (dt * 1e3, B / dt / 1e6, tb.loss), {k: round(v["ms"] / max(v["launches"], 1) * 1e3) for k, v in prof.items() if v["launches"]})
    if mode == "image":
        print("  image mse", tb.compute_image_mse(False))
