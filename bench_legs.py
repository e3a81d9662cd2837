"""bench_legs.py — the legs of the default `python bench.py` line that are not its headline: every performance figure README / DESIGN quote outside the
lego stand-in is measured HERE, inside the driver's run, or not quoted at all.

    fox        BASELINE config #2 on its own data (the reference's data/nerf/fox photographs, staged by build() under tests/golden/_generated/fox):
               aabb_scale 4 => three cascades, cone stepping (cone_angle_constant 1/256, src/testbed_nerf.cu:2730), OpenCV lens; train samples/s, ms/step, the
               launch-group table, the march's own time, render MP/s at the photographs' 1080 x 1920
    bl_render  the fork's Blender multi-NeRF renderer (src/nerf_renderer.cu:565-791) on the trained headline model: ms per 800 x 800 frame for one and two
               instances next to the stock tracer's ms for the same view
    plumbing   BASELINE configs #1 (image, 1024^2) and #5 (SDF) at the reference batch 2^18: ms/step, samples/s and the MFMA TFLOP/s the step's FLOP model gives

All legs run on rank 0 of a one-GPU run, after the headline's timed region, each bounded to a few seconds.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "blender-ngp_amd")
CFG = os.path.join(PKG, "configs")
FOX = os.path.join(ROOT, "tests", "golden", "_generated", "fox", "transforms.json")
MFMA_PEAK_F16_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak

# FLOP model of one gridmlp step per sample (DESIGN.md 8 "Plumbing configs"): the MLP 32 -> 64 -> 64 -> 16 (padded output) forward = 2 * (32*64 + 64*64 + 64*16) = 14 336;
# the fused backward recomputes the forward (14 336), runs dgrad through the three matrices (14 336) and the three weight-gradient contractions (14 336)
GRIDMLP_FLOP_PER_SAMPLE = 4 * 2 * (32 * 64 + 64 * 64 + 64 * 16)


def _group_table(prof, bytes_per_unit):
    out = {}
    for name, e in prof.items():
        if e["launches"]:
            k = {"launches": int(e["launches"]), "avg_us": round(1000.0 * e["ms"] / e["launches"], 2), "units_per_launch": round(e["units"] / e["launches"], 1)}
            if name in bytes_per_unit:
                k["algorithmic_GBps"] = round(bytes_per_unit[name] * e["units"] / (e["ms"] * 1e-3) / 1e9, 1)
            out[name] = k
    return out


def dev_overrides(tb):
    """dev A / B switches of the bench scripts (never read by the library): NGP_BENCH_SIDE_EMA=0 puts the optimizer step's Ema stage back on the training chain"""
    if os.environ.get("NGP_BENCH_SIDE_EMA") is not None:
        tb.ema_on_side_stream = bool(int(os.environ["NGP_BENCH_SIDE_EMA"]))
    if os.environ.get("NGP_BENCH_COMPACT_BWD") is not None:   # 0 = the backward pass over all B slots
        tb.compact_backward = bool(int(os.environ["NGP_BENCH_COMPACT_BWD"]))
    if os.environ.get("NGP_BENCH_MORTON_GRID") is not None:   # 0 = the occupancy-grid update's samples in the reference's order
        tb.morton_grid_samples = bool(int(os.environ["NGP_BENCH_MORTON_GRID"]))
    return tb


def fox_leg(steps, bytes_per_unit, min_train_step=1000, survey_steps=32, hbm_peak=8000.0, network_pass=None):
    """config #2: the fox photographs through `load_training_data` (host/nerf_loader.cpp + jpeg_reader.cpp), default base.json, B = 2^18."""
    if not os.path.exists(FOX):
        return {"skipped": "tests/golden/_generated/fox is staged by build() where /root/reference exists; not present in this checkout"}
    import pyngp
    B = 1 << 18
    t_load = time.perf_counter()
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    dev_overrides(tb)
    tb.load_training_data(FOX)
    tb.reload_network_from_file(os.path.join(CFG, "nerf", "base.json"))
    t_load = time.perf_counter() - t_load
    tr = tb.nerf.training
    if network_pass:
        tb.network_pass = network_pass          # dev A / B (bench_legs.py fox N <organisation>); the default run leaves the Testbed's own measured choice
    tb.async_training_steps = True
    tb.shall_train = True
    while tb.training_step < min_train_step:
        tb.frame()
    tb.set_profiling(True)
    tb.reset_profile()
    for _ in range(survey_steps):
        tb.frame()
    survey = tb.profile()
    tb.set_profiling(False)
    tb.sync()
    timed_from = tb.training_step
    samples = rays = pre = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        rays += tr.rays_per_batch
        tb.frame()
        samples += min(tr.measured_batch_size, B)
        pre += tr.measured_batch_size_before_compaction
    tb.sync()
    dt = time.perf_counter() - t0
    kernels = _group_table(survey, bytes_per_unit)
    network_pass = dict(tb.network_pass_report)
    out = {"workload": "data/nerf/fox of the reference: %d photographs 1080 x 1920 (.jpg), aabb_scale 4 (3 cascades), cone_angle_constant 1/256, OpenCV lens, configs/nerf/base.json, batch 2^18" % len(list(tr.paths)),
           "value": round(samples / dt, 1), "unit": "samples/s", "ms_per_step": round(1000.0 * dt / steps, 4), "steps": steps, "timed_from_training_step": int(timed_from),
           "rays_per_step": round(rays / steps, 1), "pre_compaction_samples_per_step": round(pre / steps, 1), "load_s": round(t_load, 2), "loss": round(float(tb.loss), 5),
           "n_params": int(tb.n_params()), "network_pass": network_pass, "kernels": kernels, "kernels_note": "HIP events on the launch streams, %d untimed survey steps" % survey_steps}
    if "nerf_backward" in kernels and "algorithmic_GBps" in kernels["nerf_backward"]:
        k = kernels["nerf_backward"]
        traffic, traffic_source = None, None
        try:   # static text, like the headline's: L2 <-> fabric bytes per call of the group from two separate --pmc passes over THIS leg on another box, quoted while the kernel set matches
            import bench
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_fox.json")))
            if tj.get("_meta", {}).get("kernel_set") == bench.KERNEL_SET:
                traffic, traffic_source = tj.get("nerf_backward"), "profiles/pmc_traffic_fox.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `python bench_legs.py fox`" % tj["_meta"].get("tag")
        except Exception:
            pass
        out["roofline"] = {"kernel": "nerf_backward", "bound": "hbm", "achieved": k["algorithmic_GBps"], "peak": hbm_peak, "unit": "GB/s", "frac": round(k["algorithmic_GBps"] / hbm_peak, 4), "traffic": traffic,
                           "traffic_static": traffic is not None, "traffic_source": traffic_source}
    # render at the photographs' size from a training view (pose, intrinsics and lens of the view: testbed.cu:273-281)
    tb.shall_train = False
    tb.background_color = [0.0, 0.0, 0.0, 1.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.render_min_transmittance = 1e-4
    w, h = 1080, 1920
    for i in range(3):
        tb.set_camera_to_training_view(i)
        tb.render(w, h, 1, True)
    ms = []
    for i in range(6):
        tb.set_camera_to_training_view((7 * i) % 50)
        t1 = time.perf_counter()
        tb.render(w, h, 1, True)
        ms.append((time.perf_counter() - t1) * 1e3)
    n_samples = int(tb.render_samples_evaluated)
    out.update({"render_MP_per_s": round(w * h / (sum(ms) / len(ms) * 1e-3) / 1e6, 2), "render_ms_per_frame": round(sum(ms) / len(ms), 2), "render_ms_frames": [round(x, 2) for x in ms],
                "render_res": [w, h], "render_network_samples_per_frame": n_samples, "render_samples_per_pixel": round(n_samples / float(w * h), 1)})
    # the tracer against the network pass's byte model (588 B per network sample) over the WHOLE frame time of the last view: host hops, march, composite and the
    # device -> host copy included, so this is a floor for the encoder's own rate (profiles/r05_a_fox_render_kernel_stats.txt: the encode kernel is 77 % of a frame's kernel time)
    gbps = 588.0 * n_samples / (ms[-1] * 1e-3) / 1e9
    out["render_roofline"] = {"kernel": "tracer frame (encode_planes_kernel + MLP kernel + march + composite)", "bound": "hbm", "achieved": round(gbps, 1), "peak": hbm_peak, "unit": "GB/s",
                              "frac": round(gbps / hbm_peak, 4), "traffic": None, "samples": n_samples, "frame_ms": round(ms[-1], 2)}
    # the Blender add-on's renderer on THIS model (its actual input: a snapshot of a real capture), same pinhole view for both renderers
    try:
        def set_view():
            tb.set_camera_to_training_view(0)
        tb.set_camera_to_training_view(0)
        focal_px = 0.5 * float((w, h)[int(tb.fov_axis)]) / float(np.tan(0.5 * float(tb.fov) * np.pi / 180.0))
        lo, hi = tb.aabb
        bl = bl_render_on(tb, w, h, set_view, focal_px, (list(lo), list(hi)), ([lo[0] - 1.0, lo[1] - 1.0, lo[2] - 1.0], [hi[0] + 1.0, hi[1] + 1.0, hi[2] + 1.0]), 0.0, frames=4, n_nerfs_list=(1,))
        if bl.get("bl_network_samples_1nerf") and bl.get("stock_network_samples"):
            bl["bl_over_stock_samples_1nerf"] = round(bl["bl_network_samples_1nerf"] / bl["stock_network_samples"], 2)
            bl["note"] = ("the fork's renderer takes this many times the stock tracer's samples at the same rate per sample: its rays restart t = 0 (and with it the cone's step size) where they enter the "
                          "render box (src/nerf_renderer.cu:127-145) and this fork's loader (NERF_SCALE 1, offset 0) leaves the fox cameras outside the box; with the camera inside the box the two renderers' "
                          "sample counts agree within 2 % (profiles/r05_experiments.md section 6; tests/test_baseline_configs_gpu.py::test_fox_blender_renderer_frame_matches_oracle_and_its_cone_restarts_at_the_box)")
        out["bl_render"] = bl
    except Exception as e:   # the leg's other numbers stand on their own
        out["bl_render"] = {"failed": "%s: %s" % (type(e).__name__, e)}
    return out


def bl_render_leg(tb, ds, res, frames=6):
    """the Blender add-on's path on the headline model: snapshot -> NerfDescriptor -> request_nerf_render_sync (src/python_api.cu:306-330, src/nerf_renderer.cu:565-791)"""
    def set_view():
        tb.fov_axis = 0
        tb.fov = ds["camera_angle_x"] * 180 / np.pi
        tb.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
    return bl_render_on(tb, res, res, set_view, float(ds["focal"]), ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0]), ([-1.0, -1.0, -1.0], [2.0, 2.0, 2.0]), 0.45, frames)


def bl_render_on(tb, w, h, set_view, focal, nerf_box, scene_box, second_nerf_shift, frames=6, n_nerfs_list=(1, 2)):
    """the stock tracer and the Blender renderer on the SAME trained model, view, pinhole camera, background and termination threshold: ms per w x h frame each"""
    import pyngp
    tmp = tempfile.mkdtemp()
    snap = os.path.join(tmp, "bench_model.msgpack")
    try:
        tb.save_snapshot(snap, False)
        tb.shall_train = False
        # the same view, field of view, background and termination threshold for both renderers: the Blender path reads min_transmittance from the field (0.01, the
        # reference's default for both: testbed.h:725, neural_radiance_field.cuh:63); bench.py's PSNR evaluation had set the stock tracer's to run.py's 1e-4
        tb.background_color = [0.0, 0.0, 0.0, 1.0]
        min_t_before = tb.nerf.render_min_transmittance
        lens_before = tb.nerf.render_with_lens_distortion
        tb.nerf.render_min_transmittance = 0.01
        set_view()
        tb.nerf.render_with_lens_distortion = False   # the Blender camera is a pinhole (CameraModel.Perspective): the stock tracer renders the same rays
        for _ in range(3):
            tb.render(w, h, 1, True)
        ms = []
        for _ in range(frames):
            t0 = time.perf_counter()
            tb.render(w, h, 1, True)
            ms.append((time.perf_counter() - t0) * 1e3)
        tb.nerf.render_min_transmittance = min_t_before
        tb.nerf.render_with_lens_distortion = lens_before
        res = w if w == h else [w, h]
        out = {"res": res, "frames": frames, "min_transmittance": 0.01, "stock_render_ms": round(sum(ms) / frames, 2), "stock_render_ms_frames": [round(x, 2) for x in ms], "stock_network_samples": int(tb.render_samples_evaluated)}

        def request(n_nerfs):
            dsi = pyngp.DownsampleInfo.MakeFromMip([w, h], 0)
            outp = pyngp.RenderOutputProperties([w, h], dsi, 1, pyngp.ColorSpace.SRGB, pyngp.TonemapCurve.Identity, 0.0, [0.0, 0.0, 0.0, 1.0], False)
            cam = pyngp.RenderCameraProperties(tb.camera_matrix, pyngp.CameraModel.Perspective, focal, 0.0, 0.0, 1.0, pyngp.SphericalQuadrilateralConfig.Zero(), pyngp.QuadrilateralHexahedronConfig.Zero())
            box = pyngp.BoundingBox(list(nerf_box[0]), list(nerf_box[1]))
            nerfs = []
            for k in range(n_nerfs):
                xf = np.eye(4, dtype=np.float32)
                xf[0, 3] = second_nerf_shift * k
                nerfs.append(pyngp.NerfDescriptor(snap, box, xf, pyngp.RenderModifiers([]), 1.0))
            big = pyngp.BoundingBox(list(scene_box[0]), list(scene_box[1]))
            return pyngp.RenderRequest(outp, cam, pyngp.RenderModifiers([]), nerfs, big)

        bl = pyngp.Testbed(pyngp.TestbedMode.Nerf)
        bl.render_trace = bool(tb.render_trace)
        for knob, env in (("bl_max_skips_per_pass", "BL_SKIPS"), ("bl_max_steps_per_pass", "BL_STEPS"), ("bl_pass_samples_factor", "BL_FACTOR"), ("bl_fused_passes", "BL_FUSED")):   # dev: schedule sweeps
            if os.environ.get(env):
                setattr(bl, knob, type(getattr(bl, knob))(float(os.environ[env])))
        out["bl_schedule"] = {"fused_passes": bool(bl.bl_fused_passes), "max_skips_per_pass": int(bl.bl_max_skips_per_pass), "pass_samples_factor": float(bl.bl_pass_samples_factor), "max_steps_per_pass": int(bl.bl_max_steps_per_pass)}
        for n in n_nerfs_list:
            req = request(n)
            for _ in range(5):                        # loads the snapshot(s), warms up (the 4th request of a fresh renderer takes 20-30 ms once: tools/bl_outlier_probe.py)
                img = bl.request_nerf_render_sync(req)
            ms = []
            for _ in range(frames):
                t0 = time.perf_counter()
                img = bl.request_nerf_render_sync(req)
                ms.append((time.perf_counter() - t0) * 1e3)
            out["bl_render_ms_%dnerf" % n] = round(sum(ms) / frames, 2)
            out["bl_render_ms_min_%dnerf" % n] = round(min(ms), 2)
            out["bl_render_ms_frames_%dnerf" % n] = [round(x, 2) for x in ms]
            out["bl_network_samples_%dnerf" % n] = int(bl.bl_render_samples)
            out["bl_passes_%dnerf" % n] = int(bl.bl_render_passes)
            out["bl_coverage_%dnerf" % n] = round(float((img[..., 3] > 0.5).mean()), 3)
        out["bl_over_stock_1nerf"] = round(out["bl_render_ms_1nerf"] / out["stock_render_ms"], 3)
        return out
    finally:
        try:
            os.unlink(snap)
            os.rmdir(tmp)
        except OSError:
            pass


def _albert():
    gen = os.path.join(ROOT, "tests", "golden", "_generated", "albert.bin")
    if os.path.exists(gen):
        import struct
        raw = open(gen, "rb").read()
        h, w = struct.unpack("ii", raw[:8])
        return np.frombuffer(raw[8:], np.float16).reshape(h, w, 4).astype(np.float32), "data/image/albert.exr of the reference (1024 x 1024)"
    crop = np.load(os.path.join(ROOT, "tests", "golden", "albert_crop_128.npy")).astype(np.float32)
    return np.tile(crop, (8, 8, 1)), "albert.exr 128 x 128 crop tiled to 1024 x 1024"


def plumbing_leg(steps=200, warmup=100):
    """configs #1 (image) and #5 (SDF): hash-grid encode + one 64-wide MLP, no ray marching, B = 2^18 (src/testbed_image.cu:220-291, src/testbed_sdf.cu training step)"""
    import pyngp
    B = 1 << 18
    out = {}
    rs = np.random.RandomState(0)
    for mode in ("image", "sdf"):
        tb = pyngp.Testbed(pyngp.TestbedMode.Image if mode == "image" else pyngp.TestbedMode.Sdf)
        dev_overrides(tb)
        if mode == "image":
            img, what = _albert()
            tb.set_image_data(np.ascontiguousarray(img))
        else:
            pts = rs.rand(1 << 20, 3).astype(np.float32)
            tb.override_sdf_training_data(pts, (np.linalg.norm(pts - 0.5, axis=1) - 0.3).astype(np.float32))
            what = "analytic sphere SDF, 2^20 surface-free samples (the armadillo mesh + BVH sampler are out of scope: SURVEY row P2)"
        tb.reload_network_from_file(os.path.join(CFG, mode, "base.json"))
        tb.shall_train = True
        for _ in range(warmup):
            tb.train(B)
        tb.set_profiling(True)
        tb.reset_profile()
        for _ in range(16):
            tb.train(B)
        prof = tb.profile()
        tb.set_profiling(False)
        tb.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            tb.train(B)
        tb.sync()
        dt = (time.perf_counter() - t0) / steps
        out[mode] = {"workload": what, "ms_per_step": round(dt * 1e3, 4), "samples_per_s": round(B / dt, 1), "steps": steps, "batch": B, "loss": round(float(tb.loss), 6),
                     "mfma_TFLOPs": round(GRIDMLP_FLOP_PER_SAMPLE * B / dt / 1e12, 2), "mfma_frac_of_peak": round(GRIDMLP_FLOP_PER_SAMPLE * B / dt / 1e12 / MFMA_PEAK_F16_TFLOPS, 4),
                     "flop_per_sample": GRIDMLP_FLOP_PER_SAMPLE, "network_pass": dict(tb.network_pass_report), "groups_us": {k: round(v["ms"] / v["launches"] * 1e3, 1) for k, v in prof.items() if v["launches"]}}
        del tb
    return out


def variants_leg(ds, steps=120, warmup=300):
    """network variants on the headline scene at B = 2^18: configs/nerf/base_3layer.json and 4 per-image latent dims (`n_extra_learnable_dims`,
    optimize_extra_dims), each on the MFMA kernels (csrc/network_netx_mfma.cuh) and — 10 steps — on the scalar checker kernels they replaced"""
    import pyngp
    out = {}
    for name, cfg, n_extra in (("base_3layer", "base_3layer.json", 0), ("latent4", "base.json", 4), ("base_0layer", "base_0layer.json", 0)):
        row = {}
        for scalar in (False, True):
            t = pyngp.Testbed(pyngp.TestbedMode.Nerf)
            n = len(ds["train_images"])
            t.create_empty_nerf_dataset(n, ds["aabb_scale"], False)
            t.nerf.training.set_dataset_transform(ds["scale"], ds["offset"])
            w, h = ds.get("w", ds["res"]), ds.get("h", ds["res"])
            for i in range(n):
                t.nerf.training.set_image_rgba8(i, ds["train_images"][i])
                t.nerf.training.set_camera_intrinsics(i, ds["focal"], ds["focal"], 0.5 * w, 0.5 * h)
                t.nerf.training.set_camera_extrinsics(i, ds["train_poses"][i][:3, :], True)
            t.nerf.training.n_images_for_training = n
            if n_extra:
                t.nerf.training.dataset.n_extra_learnable_dims = n_extra
            t.reload_network_from_file(os.path.join(CFG, "nerf", cfg))
            if n_extra:
                t.nerf.training.optimize_extra_dims = True
            t.netx_scalar_kernels = scalar
            t.async_training_steps = True
            t.shall_train = True
            k_warm, k_steps = (warmup, steps) if not scalar else (20, 10)
            for _ in range(k_warm):
                t.frame()
            t.set_profiling(True)
            t.reset_profile()
            for _ in range(8):
                t.frame()
            prof = t.profile()
            t.set_profiling(False)
            t.sync()
            t0 = time.perf_counter()
            samples = 0
            for _ in range(k_steps):
                t.frame()
                samples += min(t.nerf.training.measured_batch_size, 1 << 18)
            t.sync()
            dt = time.perf_counter() - t0
            row["scalar_checker" if scalar else "mfma"] = {"ms_per_step": round(1000.0 * dt / k_steps, 4), "samples_per_s": round(samples / dt, 1), "steps": k_steps, "loss": round(float(t.loss), 5),
                                                             "groups_us": {k: round(v["ms"] / v["launches"] * 1e3, 1) for k, v in prof.items() if v["launches"]}}
            row["n_mlp_params"] = int(t.n_mlp_params)
            del t
        out[name] = row
    return out


if __name__ == "__main__":   # dev: one leg on its own, e.g. under rocprofv3:  python bench_legs.py fox 200
    for p in (PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import json
    import torch  # noqa: F401  (first: one HIP runtime per process)
    which = sys.argv[1] if len(sys.argv) > 1 else "fox"
    if which == "fox":
        import bench
        print(json.dumps(fox_leg(int(sys.argv[2]) if len(sys.argv) > 2 else 200, bench.BYTES_PER_UNIT, min_train_step=int(os.environ.get("FOX_MIN_STEP", "1000")),
                                 network_pass=sys.argv[3] if len(sys.argv) > 3 else None)))
    elif which == "plumbing":
        print(json.dumps(plumbing_leg()))
    elif which == "variants":
        import scene
        ds = scene.make_dataset(100, 1, 800, torch.device("cuda", 0))
        print(json.dumps(variants_leg(ds)))
    elif which == "bl_render":
        import scene
        dev = torch.device("cuda", 0)
        ds = scene.make_dataset(100, 1, 800, dev)
        tb = scene.build_testbed(ds)
        dev_overrides(tb)
        scene.train(tb, int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
        print(json.dumps(bl_render_leg(tb, ds, 800)))
