"""dev (GPU box): the Blender renderer against the stock tracer on the fox model, same pinhole view — pass structure of both (render_trace) and sample counts."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa
import pyngp
import bench_legs
FOX = bench_legs.FOX
tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
tb.load_training_data(FOX)
tb.reload_network_from_file(os.path.join(bench_legs.CFG, "nerf", "base.json"))
tb.async_training_steps = True
tb.shall_train = True
while tb.training_step < 1300:
    tb.frame()
tb.sync()
w, h = 1080, 1920
tb.set_camera_to_training_view(0)
focal_px = 0.5 * float((w, h)[int(tb.fov_axis)]) / float(np.tan(0.5 * float(tb.fov) * np.pi / 180.0))
lo, hi = tb.aabb
print("fov", tb.fov, "axis", tb.fov_axis, "focal_px", focal_px, "aabb", lo, hi, "cone", tb.nerf.cone_angle_constant, flush=True)
tb.render_trace = True
def set_view():
    tb.set_camera_to_training_view(0)
out = bench_legs.bl_render_on(tb, w, h, set_view, focal_px, (list(lo), list(hi)), ([lo[0] - 1.0, lo[1] - 1.0, lo[2] - 1.0], [hi[0] + 1.0, hi[1] + 1.0, hi[2] + 1.0]), 0.0, frames=1, n_nerfs_list=(1,))
print(json.dumps(out))

# ---- occupancy of the two renderers' bitfields (the Testbed's own vs what NeuralRadianceField::load_snapshot rebuilds from the snapshot's fp16 grid) and the two images
import msgpack, tempfile
snap = os.path.join(tempfile.mkdtemp(), "fox.msgpack")
tb.render_trace = False
tb.save_snapshot(snap, False)
m = msgpack.unpackb(open(snap, "rb").read(), raw=False)["snapshot"]
g = np.frombuffer(m["density_grid_binary"], np.float16).astype(np.float32)
vol = 128 ** 3
nc = g.size // vol
bits = np.unpackbits(np.frombuffer(tb.debug_scene()["bitfield"].tobytes(), np.uint8), bitorder="little")
for c in range(nc):
    gc = g[c * vol:(c + 1) * vol]
    print("cascade %d: grid > 0: %d, > 0.01: %d, negative: %d, mean(max(v,0)) %.6f; Testbed bitfield popcount %d" % (c, int((gc > 0).sum()), int((gc > 0.01).sum()), int((gc < 0).sum()), float(np.maximum(gc, 0).mean()), int(bits[c * vol:(c + 1) * vol].sum())))
mean0 = float(np.maximum(g[:vol], 0).mean()); mean_all = float(np.maximum(g, 0).mean())
print("threshold bl = min(mean over cascade 0 = %.6f, 0.01); threshold stock = min(mean over all = %.6f, 0.01)" % (mean0, mean_all))
for name, th in (("bl", min(mean0, 0.01)), ("stock", min(mean_all, 0.01))):
    print(name, "occupied cells per cascade", [int((g[c * vol:(c + 1) * vol] > th).sum()) for c in range(nc)])

# ---- the two frames themselves
tb.background_color = [0.0, 0.0, 0.0, 0.0]
tb.nerf.render_min_transmittance = 0.01
tb.set_camera_to_training_view(0)
tb.nerf.render_with_lens_distortion = False
a = np.asarray(tb.render(w, h, 1, True))
print("stock samples", tb.render_samples_evaluated)
dsi = pyngp.DownsampleInfo.MakeFromMip([w, h], 0)
outp = pyngp.RenderOutputProperties([w, h], dsi, 1, pyngp.ColorSpace.Linear, pyngp.TonemapCurve.Identity, 0.0, [0.0, 0.0, 0.0, 0.0], False)
cam = pyngp.RenderCameraProperties(tb.camera_matrix, pyngp.CameraModel.Perspective, focal_px, 0.0, 0.0, 1.0, pyngp.SphericalQuadrilateralConfig.Zero(), pyngp.QuadrilateralHexahedronConfig.Zero())
box = pyngp.BoundingBox(list(lo), list(hi))
nerf = pyngp.NerfDescriptor(snap, box, np.eye(4, dtype=np.float32), pyngp.RenderModifiers([]), 1.0)
req = pyngp.RenderRequest(outp, cam, pyngp.RenderModifiers([]), [nerf], pyngp.BoundingBox([lo[0] - 1, lo[1] - 1, lo[2] - 1], [hi[0] + 1, hi[1] + 1, hi[2] + 1]))
bl = pyngp.Testbed(pyngp.TestbedMode.Nerf)
b = np.asarray(bl.request_nerf_render_sync(req))
print("bl samples", bl.bl_render_samples, "passes", bl.bl_render_passes)
for name, im in (("stock", a), ("bl", b)):
    al = im[..., 3]
    print(name, "alpha mean %.4f  >0.99: %.3f  <0.01: %.3f  rgb mean %s" % (al.mean(), (al > 0.99).mean(), (al < 0.01).mean(), im[..., :3].mean(axis=(0, 1))))
print("mean |alpha diff| %.4f  (flipped vertically: %.4f)  rgb mse %.5f (flipped %.5f)" % (np.abs(a[..., 3] - b[..., 3]).mean(), np.abs(a[..., 3] - b[::-1, :, 3]).mean(), ((a[..., :3] - b[..., :3]) ** 2).mean(), ((a[..., :3] - b[::-1, :, :3]) ** 2).mean()))
