"""dev (GPU box): the stock tracer's steps-per-pass cap (nerf.render_max_steps_per_pass) on the lego stand-in at 800 x 800 — ms per frame over the bench's test views."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd")]
import numpy as np
import torch
import scene
dev = torch.device("cuda", 0)
ds = scene.make_dataset(100, 3, 800, dev)
tb = scene.build_testbed(ds)
tb.async_training_steps = True
scene.train(tb, int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
tb.sync()
tb.shall_train = False
tb.background_color = [0.0, 0.0, 0.0, 1.0]
tb.snap_to_pixel_centers = True
tb.nerf.render_min_transmittance = 1e-4
tb.fov_axis = 0
tb.fov = ds["camera_angle_x"] * 180 / np.pi
ref = None
for rep in range(2):
    for cap in (64, 32, 16, 8, 64):
        tb.nerf.render_max_steps_per_pass = cap
        ms, samples = [], []
        for k in range(2 + 9):
            tb.set_nerf_camera_matrix(ds["test_poses"][k % 3][:3, :])
            t0 = time.perf_counter()
            img = np.asarray(tb.render(800, 800, 1, True))
            if k >= 2:
                ms.append((time.perf_counter() - t0) * 1e3); samples.append(int(tb.render_samples_evaluated))
        if ref is None:
            ref = img.copy()
        print("cap %2d: %.2f ms per frame (min %.2f), %.1f MP/s, %.1f network samples per pixel, max |pixel - cap 64| %.4f" % (cap, np.mean(ms), np.min(ms), 0.64 / np.mean(ms) * 1e3, np.mean(samples) / 640000, float(np.abs(img - ref).max())), flush=True)
