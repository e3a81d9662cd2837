"""dev (GPU box): train the fox model, write the snapshot + view 0's camera to gpurun_out/ so that the oracle can retrace single rays of both renderers on the CPU
(tools/fox_bl_cpu.py), and render the same small frame with both tracers here for the sample counts."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa
import pyngp
import bench_legs
out_dir = os.path.join(ROOT, "gpurun_out")
tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
tb.load_training_data(bench_legs.FOX)
tb.reload_network_from_file(os.path.join(bench_legs.CFG, "nerf", "base.json"))
tb.async_training_steps = True
tb.shall_train = True
while tb.training_step < 1300:
    tb.frame()
tb.sync()
tb.shall_train = False
snap = os.path.join(out_dir, "fox_snap.msgpack")
tb.save_snapshot(snap, False)
print("snapshot bytes", os.path.getsize(snap))
w, h = int(os.environ.get("FOX_W", 135)), int(os.environ.get("FOX_H", 240))
tb.set_camera_to_training_view(0)
focal_px = 0.5 * float((w, h)[int(tb.fov_axis)]) / float(np.tan(0.5 * float(tb.fov) * np.pi / 180.0))
lo, hi = tb.aabb
tb.background_color = [0.0, 0.0, 0.0, 0.0]
tb.nerf.render_min_transmittance = 0.01
tb.nerf.render_with_lens_distortion = False
tb.snap_to_pixel_centers = True
a = np.asarray(tb.render(w, h, 1, True))
stock_samples = int(tb.render_samples_evaluated)
dsi = pyngp.DownsampleInfo.MakeFromMip([w, h], 0)
outp = pyngp.RenderOutputProperties([w, h], dsi, 1, pyngp.ColorSpace.Linear, pyngp.TonemapCurve.Identity, 0.0, [0.0, 0.0, 0.0, 0.0], False)
cam = pyngp.RenderCameraProperties(tb.camera_matrix, pyngp.CameraModel.Perspective, focal_px, 0.0, 0.0, 1.0, pyngp.SphericalQuadrilateralConfig.Zero(), pyngp.QuadrilateralHexahedronConfig.Zero())
nerf = pyngp.NerfDescriptor(snap, pyngp.BoundingBox(list(lo), list(hi)), np.eye(4, dtype=np.float32), pyngp.RenderModifiers([]), 1.0)
req = pyngp.RenderRequest(outp, cam, pyngp.RenderModifiers([]), [nerf], pyngp.BoundingBox([lo[0] - 1, lo[1] - 1, lo[2] - 1], [hi[0] + 1, hi[1] + 1, hi[2] + 1]))
res = {}
for fused in (True, False):
    bl = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    bl.bl_fused_passes = fused
    b = np.asarray(bl.request_nerf_render_sync(req))
    res["fused" if fused else "reference_sequence"] = {"samples": int(bl.bl_render_samples), "passes": int(bl.bl_render_passes)}
    np.save(os.path.join(out_dir, "fox_bl_%d.npy" % int(fused)), b.astype(np.float16))
np.save(os.path.join(out_dir, "fox_stock.npy"), a.astype(np.float16))
info = {"w": w, "h": h, "focal_px": focal_px, "camera_matrix": np.asarray(tb.camera_matrix, np.float32).tolist(), "aabb": [list(map(float, lo)), list(map(float, hi))], "fov": float(tb.fov), "fov_axis": int(tb.fov_axis),
        "stock_samples": stock_samples, "bl": res, "cone_angle_constant": float(tb.nerf.cone_angle_constant), "stock_alpha_mean": float(a[..., 3].mean())}
json.dump(info, open(os.path.join(out_dir, "fox_view.json"), "w"))
print(json.dumps(info))
