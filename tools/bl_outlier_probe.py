"""dev tool: per-frame wall time of request_nerf_render_sync over 16 frames (hunting a periodic ~30 ms frame)"""
import os, sys, time, tempfile, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, scene, pyngp
dev = torch.device("cuda", 0)
res = 800
ds = scene.make_dataset(100, 1, res, dev)
tb = scene.build_testbed(ds)
scene.train(tb, 1000)
snap = os.path.join(tempfile.mkdtemp(), "lego.msgpack")
tb.save_snapshot(snap, False)
tb.shall_train = False
tb.fov_axis = 0; tb.fov = ds["camera_angle_x"] * 180 / np.pi; tb.background_color = [0.0, 0.0, 0.0, 1.0]
tb.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
ms = []
for _ in range(12):
    t0 = time.perf_counter(); tb.render(res, res, 1, True); ms.append(round((time.perf_counter() - t0) * 1e3, 2))
print("stock", ms)
dsi = pyngp.DownsampleInfo.MakeFromMip([res, res], 0)
outp = pyngp.RenderOutputProperties([res, res], dsi, 1, pyngp.ColorSpace.SRGB, pyngp.TonemapCurve.Identity, 0.0, [0.0, 0.0, 0.0, 1.0], False)
cam = pyngp.RenderCameraProperties(tb.camera_matrix, pyngp.CameraModel.Perspective, float(ds["focal"]), 0.0, 0.0, 1.0, pyngp.SphericalQuadrilateralConfig.Zero(), pyngp.QuadrilateralHexahedronConfig.Zero())
box = pyngp.BoundingBox([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
req = pyngp.RenderRequest(outp, cam, pyngp.RenderModifiers([]), [pyngp.NerfDescriptor(snap, box, np.eye(4, dtype=np.float32), pyngp.RenderModifiers([]), 1.0)], pyngp.BoundingBox([-1.0, -1.0, -1.0], [2.0, 2.0, 2.0]))
bl = pyngp.Testbed(pyngp.TestbedMode.Nerf)
if os.environ.get("NOGC"): gc.disable()
ms = []
for _ in range(16):
    t0 = time.perf_counter(); img = bl.request_nerf_render_sync(req); ms.append(round((time.perf_counter() - t0) * 1e3, 2))
print("bl   ", ms, "passes", bl.bl_render_passes)
