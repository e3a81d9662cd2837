"""dev tool: the Blender add-on path at production size — train the lego stand-in, save a snapshot, then time
Testbed.request_nerf_render_sync for 1 and 2 instances of it at 800x800 (for rocprofv3 --kernel-trace --stats runs too).

    python tools/bl_render_probe.py [train_steps=1000] [frames=10]
"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import scene
import pyngp
import helpers as H

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
res = 800
ds = scene.make_dataset(100, 1, res, dev)
tb = scene.build_testbed(ds)
scene.train(tb, steps)
snap = os.path.join(tempfile.mkdtemp(), "lego.msgpack")
tb.save_snapshot(snap, False)
tb.shall_train = False
tb.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
tb.render(res, res, 1, True)
t0 = time.perf_counter()
for _ in range(frames):
    tb.render(res, res, 1, True)
print("stock renderer: %.2f ms / frame" % ((time.perf_counter() - t0) / frames * 1e3))

cam34 = np.asarray(ds["test_poses"][0][:3, :], np.float32)
focal = float(ds["focal"]) if "focal" in ds else 0.5 * res / np.tan(0.5 * np.deg2rad(tb.fov))


def request(n_nerfs):
    dsi = pyngp.DownsampleInfo.MakeFromMip([res, res], 0)
    out = pyngp.RenderOutputProperties([res, res], dsi, 1, pyngp.ColorSpace.SRGB, pyngp.TonemapCurve.Identity, 0.0, [0.0, 0.0, 0.0, 1.0], False)
    cam = pyngp.RenderCameraProperties(tb.camera_matrix, pyngp.CameraModel.Perspective, focal, 0.0, 0.0, 1.0, pyngp.SphericalQuadrilateralConfig.Zero(), pyngp.QuadrilateralHexahedronConfig.Zero())
    box = pyngp.BoundingBox([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    nerfs = []
    for k in range(n_nerfs):
        xf = np.eye(4, dtype=np.float32)
        xf[0, 3] = 0.45 * k
        nerfs.append(pyngp.NerfDescriptor(snap, box, xf, pyngp.RenderModifiers([]), 1.0))
    big = pyngp.BoundingBox([-1.0, -1.0, -1.0], [2.0, 2.0, 2.0])
    return pyngp.RenderRequest(out, cam, pyngp.RenderModifiers([]), nerfs, big)


bl = pyngp.Testbed(pyngp.TestbedMode.Nerf)
for n in (1, 2):
    req = request(n)
    img = bl.request_nerf_render_sync(req)
    t0 = time.perf_counter()
    for _ in range(frames):
        img = bl.request_nerf_render_sync(req)
    dt = (time.perf_counter() - t0) / frames
    print("blender renderer, %d NeRF(s): %.2f ms / frame, %.1f MP/s, %d network samples, coverage %.2f" % (n, dt * 1e3, res * res / dt / 1e6, bl.bl_render_samples, float((img[..., 3] > 0.5).mean())))
