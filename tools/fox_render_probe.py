"""dev tool (GPU box): the stock tracer on the fox photographs (BASELINE config #2) — train, snapshot, then render training views at 1080 x 1920 under a sweep of the
tracer's schedule knobs; or (argv[1] == "render") render only from the snapshot, for rocprofv3 --kernel-trace runs that must not see training kernels.

    python tools/fox_render_probe.py train /tmp/fox.msgpack [steps]      -> trains, saves the snapshot, prints the sweep table
    python tools/fox_render_probe.py render /tmp/fox.msgpack [frames]    -> loads the snapshot, renders `frames` views (FOX_TRACE=1 prints the pass structure)
Knobs for the render mode: FOX_FACTOR, FOX_SKIPS, FOX_CAP, FOX_FUSED (render_pass_samples_factor, render_max_skips_per_pass, render_max_steps_per_pass, render_fused_network).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd")]
import torch  # noqa: F401,E402  (first: one HIP runtime per process)
import pyngp  # noqa: E402

FOX = os.path.join(ROOT, "tests", "golden", "_generated", "fox", "transforms.json")
CFG = os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json")
W, H = 1080, 1920
VIEWS = [0, 7, 14, 21, 28, 35]


def setup_render(tb):
    tb.shall_train = False
    tb.background_color = [0.0, 0.0, 0.0, 1.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.render_min_transmittance = 1e-4


def timed_frames(tb, views=VIEWS, warm=2):
    for i in range(warm):
        tb.set_camera_to_training_view(views[i % len(views)])
        tb.render(W, H, 1, True)
    ms, samples = [], []
    for v in views:
        tb.set_camera_to_training_view(v)
        t0 = time.perf_counter()
        tb.render(W, H, 1, True)
        ms.append((time.perf_counter() - t0) * 1e3)
        samples.append(int(tb.render_samples_evaluated))
    return ms, samples


def apply_env(tb):
    if os.environ.get("FOX_TRACE"):
        tb.render_trace = True
    for env, name, typ in (("FOX_FACTOR", "render_pass_samples_factor", float), ("FOX_SKIPS", "render_max_skips_per_pass", int), ("FOX_CAP", "render_max_steps_per_pass", int),
                           ("FOX_FUSED", "render_fused_network", lambda s: bool(int(s)))):
        if os.environ.get(env):
            setattr(tb.nerf, name, typ(os.environ[env]))


def main():
    mode, snap = sys.argv[1], sys.argv[2]
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    if mode == "train":
        steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
        tb.load_training_data(FOX)
        tb.reload_network_from_file(CFG)
        tb.async_training_steps = True
        tb.shall_train = True
        while tb.training_step < steps:
            tb.frame()
        tb.sync()
        tb.save_snapshot(snap, False)
        setup_render(tb)
        rows = []
        base = dict(factor=tb.nerf.render_pass_samples_factor, skips=tb.nerf.render_max_skips_per_pass, cap=tb.nerf.render_max_steps_per_pass, fused=False)
        sweep = [base, dict(base, fused=True), dict(base, cap=8), dict(base, cap=16), dict(base, cap=32), dict(base, factor=2.0), dict(base, factor=3.0), dict(base, skips=32), dict(base, skips=200),
                 dict(base, skips=0), dict(base, factor=1.0, cap=8, skips=0), base]
        if os.environ.get("FOX_SHORT"):
            sweep = [base, dict(base, cap=8)]
        for s in sweep:
            tb.nerf.render_pass_samples_factor = s["factor"]; tb.nerf.render_max_skips_per_pass = s["skips"]; tb.nerf.render_max_steps_per_pass = s["cap"]; tb.nerf.render_fused_network = s["fused"]
            ms, samples = timed_frames(tb)
            rows.append(dict(s, ms_mean=round(sum(ms) / len(ms), 2), ms=[round(x, 1) for x in ms], MP_per_s=round(W * H / (sum(ms) / len(ms) * 1e-3) / 1e6, 1), samples_per_pixel=round(sum(samples) / len(samples) / (W * H), 1)))
            print(json.dumps(rows[-1]), flush=True)
    else:
        frames = int(sys.argv[3]) if len(sys.argv) > 3 else 6
        tb.load_snapshot(snap)
        setup_render(tb)
        apply_env(tb)
        views = [(7 * i) % 50 for i in range(frames)]
        ms, samples = timed_frames(tb, views, warm=1)
        print(json.dumps({"frames": frames, "ms": [round(x, 1) for x in ms], "samples": samples, "MP_per_s": round(W * H / (sum(ms) / len(ms) * 1e-3) / 1e6, 1)}))


if __name__ == "__main__":
    main()
