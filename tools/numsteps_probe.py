import sys
sys.path[:0] = ["/root/repo", "/root/repo/blender-ngp_amd"]
import numpy as np, torch, scene
dev = torch.device("cuda:0")
ds = scene.make_dataset(100, 3, 800, dev)
tb = scene.build_testbed(ds)
tb.async_training_steps = True
while tb.training_step < 1003: tb.frame()
tb.debug_capture_next_step()
tb.frame(); tb.sync()
c = tb.debug_captured()
ns = np.asarray(c["numsteps"]).reshape(-1, 2)[: int(c["gen_counters"][0]), 0]
print("rays", ns.size, "samples", ns.sum(), "mean", ns.mean(), "percentiles 50/90/99/99.9/max", np.percentile(ns, [50, 90, 99, 99.9]), ns.max())
print("rays with more than 64 / 128 / 256 / 512 samples:", (ns > 64).sum(), (ns > 128).sum(), (ns > 256).sum(), (ns > 512).sum())
