"""dev tool (GPU box): the fox photographs through load_training_data; PSNR per view over the pixels the aabb covers"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa
import pyngp, scene, metrics
FOX = os.path.join(ROOT, "tests", "golden", "_generated", "fox", "transforms.json")
tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
tb.load_training_data(FOX)
tb.reload_network_from_file(os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json"))
tr = tb.nerf.training
n_train = int(os.environ.get("N_TRAIN", "45"))
tr.n_images_for_training = n_train
for target in (1000, 2000, 4000):
    tb.shall_train = True
    scene.train(tb, target)
    tb.sync()
    tb.shall_train = False
    tb.background_color = [0.0, 0.0, 0.0, 0.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.render_min_transmittance = 1e-4
    out = []
    for i in (0, 22, 45, 46, 47, 48, 49):
        tb.set_camera_to_training_view(i)
        img = tb.render(1080, 1920, 2, True)
        ref = metrics.read_image_rgba8(np.ascontiguousarray(tr.get_image_rgba8(i)))
        cov = img[..., 3] > 0.99
        a = np.clip(metrics.linear_to_srgb(img[..., :3]), 0, 1)
        b = np.clip(metrics.linear_to_srgb(ref[..., :3]), 0, 1)
        mse = float(((a - b) ** 2)[cov].mean())
        out.append("%d: %.1f dB (cov %.2f)" % (i, -10 * np.log10(mse), cov.mean()))
    print("step %d loss %.5f | %s" % (tb.training_step, tb.loss, " | ".join(out)), flush=True)
