"""dev (GPU box): why bench_legs.fox_leg renders the fox at ~40 MP/s / 38 samples per pixel while tools/fox_render_probe.py renders the same views at ~105 MP/s / 29.
Variants of the leg's sequence, one per process: argv[1] in {leg, noprofile, noprops, nobench, probe}."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa
which = sys.argv[1]
if which != "nobench":
    import bench  # noqa
import pyngp
FOX = os.path.join(ROOT, "tests", "golden", "_generated", "fox", "transforms.json")
CFG = os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json")
tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
tb.load_training_data(FOX)
tb.reload_network_from_file(CFG)
tr = tb.nerf.training
tb.async_training_steps = True
tb.shall_train = True
while tb.training_step < 1000:
    tb.frame()
if which not in ("noprofile", "probe"):
    tb.set_profiling(True); tb.reset_profile()
for _ in range(32):
    tb.frame()
if which not in ("noprofile", "probe"):
    tb.profile(); tb.set_profiling(False)
tb.sync()
for _ in range(300):
    if which not in ("noprops", "probe"):
        _ = tr.rays_per_batch
    tb.frame()
    if which not in ("noprops", "probe"):
        _ = tr.measured_batch_size; _ = tr.measured_batch_size_before_compaction
tb.sync()
if which == "probe":
    tb.save_snapshot("/tmp/fox_bisect.msgpack", False)
tb.shall_train = False
tb.background_color = [0.0, 0.0, 0.0, 1.0]
tb.snap_to_pixel_centers = True
tb.nerf.render_min_transmittance = 1e-4
w, h = 1080, 1920
for i in range(3):
    tb.set_camera_to_training_view(i); tb.render(w, h, 1, True)
ms = []
for i in range(6):
    tb.set_camera_to_training_view((7 * i) % 50)
    if i == 5: tb.render_trace = True
    t1 = time.perf_counter(); tb.render(w, h, 1, True); ms.append((time.perf_counter() - t1) * 1e3)
print(json.dumps({"which": which, "ms": [round(x, 1) for x in ms], "samples": int(tb.render_samples_evaluated), "loss": float(tb.loss), "step": tb.training_step}))
