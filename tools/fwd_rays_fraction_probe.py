"""dev tool: fraction of the marched samples the experimental ray-walking network pass (Testbed.forward_walks_rays) evaluates in one steady-state step of the
bench scene (the rest is zero-filled: behind its ray's termination)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import scene
dev = torch.device("cuda", 0)
ds = scene.make_dataset(100, 2, 800, dev)
tb = scene.build_testbed(ds)
tb.forward_walks_rays = True
scene.train(tb, 1200)
tb.debug_capture_next_step()
tb.frame()
cap = tb.debug_captured()
n_rays = int(cap["gen_counters"][0])
ns = cap["numsteps"].reshape(-1, 2)[:n_rays].astype(np.int64)
out = cap["mlp_out"].view(np.uint16).reshape(-1, 4)
tot = ev = 0
for n, b in ns:
    z = (out[b:b + n] == 0).all(axis=1)
    tot += n; ev += int((~z).sum())
print("marched", tot, "evaluated", ev, "fraction", ev / tot)
