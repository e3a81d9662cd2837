"""dev (GPU box): how many samples of a training step's compacted batch carry NO loss gradient at all (dL/dout = 0 in all four channels after the roll-over / rescale)?
Such a sample contributes exact zeros to every sum of the backward pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import scene
dev = torch.device("cuda", 0)
ds = scene.make_dataset(100, 2, 800, dev)
tb = scene.build_testbed(ds)
tb.async_training_steps = True
for stop in (300, 1100, 3000):
    scene.train(tb, stop)
    tb.sync()
    tb.debug_capture_next_step()
    tb.frame()
    c = tb.debug_captured()
    dl = np.asarray(c["dloss_rolled"]).view(np.float16).reshape(-1, 4).astype(np.float32)
    z_all = (dl == 0).all(axis=1)
    z_rgb = (dl[:, :3] == 0).all(axis=1)
    z_sig = dl[:, 3] == 0
    sub = (np.abs(dl) < 6.2e-5) & (dl != 0)
    print("step %d: %d samples; all four zero %.3f; rgb zero %.3f; sigma zero %.3f; subnormal entries %.3f" % (stop, dl.shape[0], z_all.mean(), z_rgb.mean(), z_sig.mean(), sub.mean()), flush=True)
    # runs of zero-gradient samples (ray order): are they the tails of rays?
    idx = np.flatnonzero(z_all)
    if idx.size:
        runs = np.split(idx, np.flatnonzero(np.diff(idx) != 1) + 1)
        lens = np.array([len(r) for r in runs])
        print("   zero runs: %d, mean length %.1f, >= 8: %.3f of the zero samples" % (len(runs), lens.mean(), lens[lens >= 8].sum() / idx.size))
