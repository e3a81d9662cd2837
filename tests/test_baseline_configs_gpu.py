"""The BASELINE.json configurations at their STATED sizes, product path (pyngp -> C++ Testbed -> C ABI -> gfx950 kernels) against the oracle.

  #3 lego      procedural stand-in (the real nerf-synthetic/lego is not in the reference tree), 100 x 800x800 RGBA8, aabb_scale 1,
               configs/nerf/base.json, B = 2^18: one whole training step replayed stage by stage through the oracle; then the "train to 35 PSNR then
               render" gate with scripts/run.py's evaluation protocol.
  #2 fox-shaped  50 x 1920x1080 RGBA8, aabb_scale 4 (3 cascades, cone stepping), T = 2^19, B = 2^18: the same replay + convergence.
  #1 image     1024 x 1024 (albert.exr when the build container produced tests/golden/_generated/albert.bin, else the committed 128^2 crop tiled),
               configs/image/base.json, B = 2^18: step 0 against the oracle, convergence.
  #5 sdf       sphere SDF samples, configs/sdf/base.json, B = 2^18: step 0 against the oracle, convergence.
  #4 (8 GPUs) is the driver's SCALE run; its single-rank data-parallel path is covered by tests/test_dp_gpu.py.

Tolerances (written at the assertion): integer / index work bit-exact; fp16 network outputs rtol 1e-2 + atol 1e-2; loss gradients rtol 4e-3 + atol 6e-6;
MLP weight gradients rtol 3e-2 + 3e-3 of the largest; hash-grid gradients 2e-2 of the norm.
"""
import os
import struct
import sys

import numpy as np
import pytest

import capi
import fullstep as F
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]
pytestmark = pytest.mark.gpu
CFG = os.path.join(ROOT, "blender-ngp_amd", "configs")
B = 1 << 18


def _replay_one_step(oracle, tb, ds):
    """arm the capture, run ONE frame() of the product path, hand every stage's device inputs to the oracle"""
    tb.debug_capture_next_step()
    tb.frame()
    cap = tb.debug_captured()
    S = F.host_scene(tb, ds["train_images"])
    assert cap["target_batch_size"] == B and int(S["desc"]["levels"][0][15]["size"]) == 1 << 19          # the stated batch and table size
    report = {"rays": int(cap["gen_counters"][0]), "samples": int(cap["gen_counters"][1]), "R": int(cap["R"]), "max_inference": int(cap["max_inference"])}
    print("captured step:", report, flush=True)

    # ---- T4 march: bit-exact ray count, sample count, ray records, every sample record
    r = F.oracle_march(oracle, S, cap)
    bf_now = np.asarray(S["bitfield"]).view(np.uint8)
    bf_cap = np.asarray(cap["bitfield"]).view(np.uint8)
    note = "[step %d, prefetch hit %s, occupancy bytes that differ between the step's begin and the read-back: %d of %d]" % (int(cap["step"]), bool(cap["prefetch_hit"]), int((bf_now != bf_cap).sum()), bf_now.size)
    n_rays, n_samples = F.compare_march(r, F.device_march(cap), int(cap["max_inference"]), note)
    report["march_overflowed"] = report["samples"] > report["max_inference"]
    assert n_rays > 1000 and n_samples > B                                                           # a real batch: more samples than survive compaction

    # ---- T5 inference over all pre-compaction samples (training weights): fp16 outputs, rtol 1e-2 / atol 1e-2
    rows = F.covered_rows(cap)
    assert rows.sum() == n_samples
    ref = F.oracle_inference(oracle, S, cap, rows)
    got = cap["mlp_out"].view(np.float16).reshape(-1, 4)[rows].astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-2)
    report["inference_max_abs_diff"] = float(np.abs(got - ref).max())

    # ---- T6 loss + compaction on the device's network outputs
    o = F.oracle_loss(oracle, S, cap)
    border = F.borderline_rays(cap)
    assert len(border) < 0.05 * n_rays                                                             # opaque surfaces: T falls through 1e-4 on many rays
    gns, ons = cap["numsteps_compacted"], o["ns"]
    mism = [i for i in range(n_rays) if int(ons[2 * i]) != int(gns[2 * i]) and i not in border and int(ons[2 * i + 1]) + int(ons[2 * i]) < B and int(gns[2 * i + 1]) + int(gns[2 * i]) < B]
    assert not mism, (mism[:10], [F.describe_compaction_mismatch(cap, ons, i) for i in mism[:3]])     # compacted count per ray: exact off the T < 1e-4 knife edge
    n_kept_dev, n_kept_orc = int(cap["measured_batch_size"]), int(o["cnt"][0])
    assert abs(n_kept_dev - n_kept_orc) <= 2 * len(border) + 2, (n_kept_dev, n_kept_orc)
    report.update(compacted_device=n_kept_dev, compacted_oracle=n_kept_orc, borderline_rays=len(border))
    gco, oco = cap["coords_compacted"].view(np.uint8).reshape(-1, 28), o["co"].view(np.uint8).reshape(-1, 28)
    gdl, odl = cap["dloss"].view(np.float16).reshape(-1, 4).astype(np.float32), o["dl"].astype(np.float32)
    checked = 0
    for i in range(n_rays):
        n, bo, bg = int(ons[2 * i]), int(ons[2 * i + 1]), int(gns[2 * i + 1])
        if i in border or n == 0 or n != int(gns[2 * i]) or bo + n > B or bg + n > B:
            continue
        assert (oco[bo:bo + n] == gco[bg:bg + n]).all(), i                                           # compacted sample records: bit-exact
        np.testing.assert_allclose(gdl[bg:bg + n], odl[bo:bo + n], rtol=4e-3, atol=6e-6)             # loss gradients (wave scans vs sequential sums)
        checked += n
    assert checked > 0.9 * min(n_kept_orc, B)
    keep = np.array([i for i in range(n_rays) if i not in border and int(ons[2 * i]) == int(gns[2 * i]) and int(ons[2 * i + 1]) + int(ons[2 * i]) <= B and int(gns[2 * i + 1]) + int(gns[2 * i]) <= B])
    np.testing.assert_allclose(cap["loss"][keep], o["loss"][keep], rtol=1e-4, atol=1e-9)                 # per-slot loss (loss_output[i], :1462)

    # ---- T7 roll-over (fill_rollover_and_rescale<half> / fill_rollover<float>, :3314-3322): the kept samples stay as they are, the padding behind them
    # repeats them from the start, its gradients scaled by kept / B
    n_c = min(n_kept_dev, B)
    rolled = cap["coords_compacted_rolled"].view(np.uint8).reshape(-1, 28)
    assert (rolled[:n_c] == gco[:n_c]).all() and (rolled[n_c:] == gco[np.arange(n_c, B) % n_c]).all()
    want = gdl.astype(np.float16).copy()
    want[n_c:] = ((gdl[np.arange(n_c, B) % n_c] * np.float32(n_c * 4)) / np.float32(B * 4)).astype(np.float16)
    np.testing.assert_array_equal(cap["dloss_rolled"].view(np.float16).reshape(-1, 4), want)

    # ---- T8 forward + backward over the 2^18 compacted samples
    gref = F.oracle_backward(oracle, S, cap)
    ggot = cap["grads"].view(np.float16).astype(np.float64)
    assert np.isfinite(ggot).all()
    gm, rm = ggot[:10240], gref[:10240]
    np.testing.assert_allclose(gm, rm, rtol=3e-2, atol=3e-3 * np.abs(rm).max())                       # MLP weight gradients
    gg, rg = ggot[10240:], gref[10240:]
    err = np.linalg.norm(gg - rg) / np.linalg.norm(rg)
    assert err < 2e-2, err                                                                           # hash-grid gradients, relative to the norm
    # entries the oracle leaves at exactly 0: the device may hold a stray denormal-sized term there (its dL/dx comes from MFMA sums, the oracle's from
    # sequential ones, and a value that underflows fp16 in one need not in the other) — never more than a handful, never of any size
    stray = np.abs(gg[rg == 0])
    assert np.count_nonzero(stray) <= 1e-5 * len(gg) and (stray.max() if len(stray) else 0.0) <= 1e-3 * np.abs(rg).max()
    report.update(mlp_grad_norm=float(np.linalg.norm(rm)), grid_grad_norm=float(np.linalg.norm(rg)), grid_grad_rel_err=float(err),
                  mlp_grad_rel_err=float(np.linalg.norm(gm - rm) / np.linalg.norm(rm)))
    # ---- the budget of a step is the previous step's demand (testbed_nerf.cu:3189), so about every other step overflows it by a little and drops
    # rays in atomic order.  Whatever the step above was, also hold a step WITHOUT overflow to the strict comparison (identical kept-ray sets).
    for attempt in range(16):
        if not report["march_overflowed"] and attempt == 0:
            report["strict_march_step"] = int(cap["step"])
            break
        tb.debug_capture_next_step()
        tb.frame()
        c2 = tb.debug_captured()
        if int(c2["gen_counters"][1]) <= int(c2["max_inference"]):
            S2 = F.host_scene(tb, ds["train_images"])   # NOT the scene read behind the first captured step: an occupancy update (every 16th step) may lie in between
            F.compare_march(F.oracle_march(oracle, S2, c2), F.device_march(c2), None, "[strict step %d]" % int(c2["step"]))
            report["strict_march_step"] = int(c2["step"])
            break
    assert "strict_march_step" in report
    return report


@pytest.fixture(scope="module")
def lego(cuda):
    import scene
    ds = scene.make_dataset(100, 3, 800, cuda)
    return ds, scene.build_testbed(ds)


def test_lego_full_step_matches_oracle(oracle, lego):
    """config #3 at its stated size: 100 x 800x800 views, T = 2^19, B = 2^18 — step 300 of the product path, every stage against the oracle"""
    import scene
    ds, tb = lego
    scene.train(tb, 300)
    assert tb.training_step == 300 and tb.training_batch_size == B
    rep = _replay_one_step(oracle, tb, ds)
    print("lego step 300:", rep)
    assert tb.training_step >= 301
    assert 0.5 * B < rep["compacted_device"]                                                         # the rays_per_batch feedback fills the batch


def test_lego_trains_to_the_psnr_gate(lego):
    """config #3 "train to 35 PSNR then render" with scripts/run.py's protocol (216-303: black background, pixel-centre sampling, 8 spp, min
    transmittance 1e-4, PSNR on sRGB-clipped images, mean over the held-out views): >= 35 dB by step 1000, >= 37 dB by step 2000"""
    import scene
    ds, tb = lego
    tb.shall_train = True
    scene.train(tb, 1000)
    tb.sync()
    psnr1k, ssim1k, _ = scene.eval_test_views(tb, ds, spp=8)
    tb.shall_train = True
    scene.train(tb, 2000)
    tb.sync()
    psnr2k, ssim2k, _ = scene.eval_test_views(tb, ds, spp=8)
    print("lego gate: %.2f dB / SSIM %.4f at step 1000, %.2f dB / %.4f at step 2000" % (psnr1k, ssim1k, psnr2k, ssim2k))
    assert tb.training_step == 2000
    assert psnr1k >= 35.0 and psnr2k >= 37.0 and ssim2k > 0.98
    img = tb.render(800, 800, 1, True)
    assert img.shape == (800, 800, 4) and np.isfinite(img).all() and 0.02 < (img[..., :3].max(-1) > 0.02).mean() < 0.9     # an object in front of the black background


def test_fox_shaped_scene_full_step_and_convergence(oracle, cuda):
    """config #2's shape: 50 views of 1920 x 1080, aabb_scale 4 => 3 cascades and cone stepping (cone_angle 1/256), default base.json (T = 2^19), B = 2^18.
    The fox photographs themselves are .jpg files of the reference tree (absent on the GPU box; their decoder is tested on the CPU side)."""
    import scene
    ds = scene.make_dataset(50, 2, 1920, cuda, aabb_scale=4, height=1080)
    tb = scene.build_testbed(ds)
    assert tb.nerf.max_cascade == 2 and tb.nerf.cone_angle_constant == pytest.approx(1.0 / 256.0)
    assert tb.n_params() == 10240 + 13074912                                                          # BASELINE.md §4 / SURVEY App. A.3: the fox level table
    scene.train(tb, 300)
    rep = _replay_one_step(oracle, tb, ds)
    print("fox-shaped step 300:", rep)
    tb.shall_train = True
    scene.train(tb, 1500)
    tb.sync()
    psnr, ssim, per = scene.eval_test_views(tb, ds, spp=2)
    print("fox-shaped: %.2f dB / SSIM %.4f after 1500 steps" % (psnr, ssim))
    if not (psnr >= 30.0 and ssim > 0.95):   # 30-38 dB / 0.94-0.99 over 45 runs of this scene (two test views, unordered atomics: no two runs train alike); a slow run gets 500 more steps
        tb.shall_train = True                  # (eval_test_views switches training off)
        scene.train(tb, 2000)
        tb.sync()
        psnr, ssim, per = scene.eval_test_views(tb, ds, spp=2)
        print("fox-shaped: %.2f dB / SSIM %.4f after 2000 steps" % (psnr, ssim))
    assert psnr >= 28.5 and ssim > 0.92     # the worst run seen: 29.86 dB / 0.938 (a floater in front of one of the two test cameras); the photographs' own test is below


FOX = os.path.join(ROOT, "tests", "golden", "_generated", "fox", "transforms.json")


@pytest.mark.skipif(not os.path.exists(FOX), reason="the fox photographs are staged by build() in the build container (tests/golden/make_fox_fixture.py); the procedural fox-shaped test above is the fallback")
def test_fox_photographs_full_step_and_held_out_psnr(oracle, cuda):
    """config #2 on its own data: data/nerf/fox — 50 portrait photographs (1080 x 1920 .jpg, OpenCV lens k1 k2 p1 p2, off-centre principal point),
    aabb_scale 4, default base.json, B = 2^18 — loaded by `load_training_data(transforms.json)` through the product's loader and JPEG decoder
    (src/nerf_loader.cu:354-531, 548-706).
      (a) the reference's transforms.json as it is: 300 steps, then one 2^18 step replayed stage by stage through the oracle;
      (b) held-out quality: the same file minus every 10th present frame (written next to it, same keys), 2000 steps, then the 5 held-out photographs are
          rendered from their own transform_matrix (run.py's way: set_nerf_camera_matrix) with the dataset's intrinsics and lens, and compared in sRGB over
          the pixels the scene box covers — the room's walls leave the aabb_scale-4 box, rays that meet nothing inside it have nothing to be compared
          with (rendered alpha <= 0.99; the reference leaves them to the random background colour in training too).
    Stated bar (measured: training views 28.7 dB, held-out 27.7 22.9 16.8 24.2 26.5 = 23.6 dB mean at 2000 steps — the third one looks at the
    mount from below, outside the hull of the training cameras): held-out mean >= 21 dB and median >= 22 dB, training views >= 26 dB."""
    import json
    import torch  # noqa: F401
    import pyngp
    import metrics
    import scene
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    tb.load_training_data(FOX)
    tb.reload_network_from_file(os.path.join(CFG, "nerf", "base.json"))
    tr = tb.nerf.training
    paths = list(tr.paths)
    assert len(paths) == 50                                                                             # 67 frames listed, 50 present: the others are dropped (nerf_loader.cu:383)
    assert tb.nerf.max_cascade == 2 and tb.nerf.cone_angle_constant == pytest.approx(1.0 / 256.0)
    assert tb.n_params() == 10240 + 13074912
    md0 = tr.get_image_metadata(0)
    assert list(md0["resolution"]) == [1080, 1920] and md0["lens_mode"] == 1
    tb.shall_train = True
    scene.train(tb, 300)
    imgs = [np.ascontiguousarray(tr.get_image_rgba8(i)) for i in range(len(paths))]
    assert imgs[0].shape == (1920, 1080, 4)
    rep = _replay_one_step(oracle, tb, {"train_images": imgs})
    print("fox (50 jpg 1080x1920) step 300:", rep)
    del tb, tr

    # ---- (b) held-out views
    meta = json.load(open(FOX))
    present = sorted((f for f in meta["frames"] if os.path.exists(os.path.join(os.path.dirname(FOX), f["file_path"]))), key=lambda f: f["file_path"])
    assert len(present) == 50
    held = present[4::10]
    train_frames = [f for f in present if f not in held]
    sub = dict(meta)
    sub["frames"] = train_frames
    sub_path = os.path.join(os.path.dirname(FOX), "transforms_holdout_train.json")
    with open(sub_path, "w") as f:
        json.dump(sub, f)
    try:
        tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
        tb.load_training_data(sub_path)
    finally:
        os.unlink(sub_path)
    tb.reload_network_from_file(os.path.join(CFG, "nerf", "base.json"))
    tr = tb.nerf.training
    assert len(tr.paths) == 45
    tb.shall_train = True
    scene.train(tb, 2000)
    tb.sync()
    tb.shall_train = False
    tb.background_color = [0.0, 0.0, 0.0, 0.0]      # rendered alpha = accumulated opacity: which pixels the box covers
    tb.snap_to_pixel_centers = True
    tb.nerf.render_min_transmittance = 1e-4

    def covered_psnr(img, ref8):
        ref = metrics.read_image_rgba8(np.ascontiguousarray(ref8))
        cov = img[..., 3] > 0.99
        a = np.clip(metrics.linear_to_srgb(img[..., :3]), 0, 1)
        b = np.clip(metrics.linear_to_srgb(ref[..., :3]), 0, 1)
        return float(-10.0 * np.log10(((a - b) ** 2)[cov].mean())), float(cov.mean())

    train_psnr = []
    for i in (0, 20):
        tb.set_camera_to_training_view(i)           # pose, focal length, lens and principal point of the view (testbed.cu:273-281)
        p, c = covered_psnr(tb.render(1080, 1920, 2, True), tr.get_image_rgba8(i))
        train_psnr.append(p)
    held_psnr, held_cov = [], []
    tb.set_camera_to_training_view(0)               # intrinsics + lens (one camera took all photographs); the pose comes from the held-out frame
    for fr in held:
        tb.set_nerf_camera_matrix(np.asarray(fr["transform_matrix"], np.float32)[:3, :])
        p, c = covered_psnr(tb.render(1080, 1920, 2, True), pyngp.decode_image(os.path.join(os.path.dirname(FOX), fr["file_path"])))
        held_psnr.append(p)
        held_cov.append(c)
    print("fox (50 jpg 1080x1920): held-out PSNR %.2f dB (%s; box coverage %s), training views %.2f dB after %d steps on 45 frames"
          % (float(np.mean(held_psnr)), " ".join("%.1f" % p for p in held_psnr), " ".join("%.2f" % c for c in held_cov), float(np.mean(train_psnr)), tb.training_step))
    assert min(held_cov) > 0.3
    assert float(np.mean(train_psnr)) >= 26.0 and float(np.mean(held_psnr)) >= 21.0 and float(np.median(held_psnr)) >= 22.0

@pytest.mark.skipif(not os.path.exists(FOX), reason="the fox photographs are staged by build() in the build container (tests/golden/make_fox_fixture.py)")
def test_fox_blender_renderer_frame_matches_oracle_and_its_cone_restarts_at_the_box(oracle, ngp, cuda, tmp_path):
    """The fork's Blender renderer (src/nerf_renderer.cu, `request_nerf_render_sync`) on a real capture: three cascades, cone stepping, cameras OUTSIDE the
    aabb_scale-4 box (this fork's NERF_SCALE is 1 and its offset 0, include/neural-graphics-primitives/nerf_loader.h:28, 84-87: the fox cameras stand 3.8 - 6.4 units
    from the origin).  The model is trained here, written as a snapshot, and the same frame is rendered by the product (both pass loops) and by
    `orc_multi_render` on the CPU from the snapshot's bytes:
      * frames: |difference| <= 4e-3 per channel (fp16 network outputs through expf; 1e-3 measured), mean <= 1e-4, for the reference's launch sequence and for the fused
        loop on the reference's schedule (those two bit-identical); the fused loop's own schedule moves single samples across gaps: <= 0.15 on single pixels, mean <= 1e-3;
      * samples: the GPU's count is the oracle's plus launch padding (< 128 per pass and NeRF);
      * and the reason this renderer takes ~5x the stock tracer's samples on such a view (profiles/r05_experiments.md section 6): init_proxy_rays moves a ray's
        origin to the render box's entry point and sets t = 0 (:127-145), so the cone's step dt = clamp(t / 256, min, max) restarts from the minimum step at the
        box, where the stock tracer (t measured from the camera) steps t_camera / 256.  With a render box that contains the camera the oracle's count falls to
        the stock tracer's (+- 10 %)."""
    import ctypes
    import msgpack
    import torch  # noqa: F401
    import pyngp
    import scene
    import test_multi_render_gpu as M
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    tb.load_training_data(FOX)
    tb.reload_network_from_file(os.path.join(CFG, "nerf", "base.json"))
    tb.shall_train = True
    scene.train(tb, 1000)
    tb.sync()
    tb.shall_train = False
    snap_path = str(tmp_path / "fox.msgpack")
    tb.save_snapshot(snap_path, False)
    w, h = 54, 96
    tb.set_camera_to_training_view(0)
    cam34 = np.asarray(tb.camera_matrix, np.float32)
    lo, hi = tb.aabb
    assert not all(lo[k] < cam34[k, 3] < hi[k] for k in range(3))                                       # the camera stands outside the box
    focal_px = 0.5 * float((w, h)[int(tb.fov_axis)]) / float(np.tan(0.5 * float(tb.fov) * np.pi / 180.0))
    # the stock tracer, same pinhole view (for the sample count only)
    tb.background_color = [0.0, 0.0, 0.0, 0.0]
    tb.nerf.render_min_transmittance = 0.01
    tb.nerf.render_with_lens_distortion = False
    tb.snap_to_pixel_centers = True
    tb.render(w, h, 1, True)
    n_stock = int(tb.render_samples_evaluated)                                                          # every slot of every network launch: >= the samples composited
    dsi = pyngp.DownsampleInfo.MakeFromMip([w, h], 0)
    outp = pyngp.RenderOutputProperties([w, h], dsi, 1, pyngp.ColorSpace.Linear, pyngp.TonemapCurve.Identity, 0.0, [0.0, 0.0, 0.0, 0.0], False)
    cam = pyngp.RenderCameraProperties(tb.camera_matrix, pyngp.CameraModel.Perspective, focal_px, 0.0, 0.0, 1.0, pyngp.SphericalQuadrilateralConfig.Zero(), pyngp.QuadrilateralHexahedronConfig.Zero())
    nerf = pyngp.NerfDescriptor(snap_path, pyngp.BoundingBox(list(lo), list(hi)), np.eye(4, dtype=np.float32), pyngp.RenderModifiers([]), 1.0)
    req = pyngp.RenderRequest(outp, cam, pyngp.RenderModifiers([]), [nerf], pyngp.BoundingBox([lo[0] - 1, lo[1] - 1, lo[2] - 1], [hi[0] + 1, hi[1] + 1, hi[2] + 1]))
    frames, counts = {}, {}
    loops = {"fused loop": (True, False), "fused loop, reference schedule": (True, True), "reference sequence": (False, False)}
    for name, (fused, ref_schedule) in loops.items():
        bl = pyngp.Testbed(pyngp.TestbedMode.Nerf)
        bl.bl_fused_passes = fused
        bl.bl_reference_schedule = ref_schedule
        frames[name] = np.asarray(bl.request_nerf_render_sync(req), np.float32)
        counts[name] = (int(bl.bl_render_samples), int(bl.bl_render_passes))
    del tb

    # ---- the oracle from the snapshot's bytes
    m = msgpack.unpackb(open(snap_path, "rb").read(), raw=False)["snapshot"]
    desc = H.make_desc(ngp, log2_hashmap_size=19, base_resolution=16, aabb_scale=4)
    params = np.frombuffer(m["params_binary"], np.float16).copy()
    grid = np.frombuffer(m["density_grid_binary"], np.float16).astype(np.float32)
    n_cascades = grid.size // 128 ** 3
    assert n_cascades == 3 and params.size >= H.n_params(desc)
    oracle.orc_density_grid_mean.restype = ctypes.c_float
    bf, _ = H.oracle_bitfield(oracle, grid, n_cascades)
    ds = M._ds(oracle, w, h, 0)
    rc = np.zeros(1, capi.RENDER_CAMERA)
    rc["transform"][0] = cam34.T.reshape(-1); rc["model"] = 0; rc["focal_length"] = focal_px; rc["focus_z"] = 1.0
    nets = (ctypes.c_void_p * 1)(desc.ctypes.data); pars = (ctypes.c_void_p * 1)(params.ctypes.data)
    act_rgb = np.array([2], np.int32); act_d = np.array([3], np.int32); min_t = np.array([0.01], np.float32)
    oracle.orc_multi_render.restype = ctypes.c_uint64

    def oracle_frame(render_aabb):
        rp = M._props(np.eye(4, dtype=np.float32), bf.ctypes.data, 0, 0, aabb_scale=4, render_aabb=render_aabb)
        fb = np.zeros((h, w, 4), np.float32); db = np.zeros((h, w), np.float32)
        n = oracle.orc_multi_render(1, nets, pars, rp.ctypes.data, act_rgb.ctypes.data, act_d.ctypes.data, min_t.ctypes.data, ds.ctypes.data, rc.ctypes.data, 0, fb.ctypes.data, db.ctypes.data)
        return fb, int(n)
    want, n_want = oracle_frame(None)
    for name in loops:
        d = np.abs(frames[name] - want)
        print("fox, Blender renderer, %s: %d samples in %d passes (oracle %d); |frame - oracle| max %.2e mean %.2e" % (name, counts[name][0], counts[name][1], n_want, d.max(), d.mean()))
        if name == "fused loop":    # its own schedule: a pass boundary next to an empty voxel trades one sample in the gap for one at the landing point
            assert d.max() <= 0.15 and d.mean() <= 1e-3
        else:
            assert d.max() <= 4e-3 and d.mean() <= 1e-4
    np.testing.assert_array_equal(frames["fused loop, reference schedule"], frames["reference sequence"])
    assert n_want <= counts["reference sequence"][0] < n_want + 128 * counts["reference sequence"][1]
    assert (want[..., 3] > 0.5).mean() > 0.2                                                            # the frame shows the scene
    big = np.zeros(1, H.AABB); big["min"][0] = -12.0; big["max"][0] = 12.0
    _, n_inside = oracle_frame(big)
    print("fox, Blender renderer samples: box entry restarts the cone %d; render box around the camera %d (x %.1f); stock tracer's launches %d" % (n_want, n_inside, n_want / n_inside, n_stock))
    assert n_want > 3 * n_inside and 0.5 * n_stock < n_inside < 1.1 * n_stock


# --------------------------------------------------------------------------------------------------------------- plumbing configs at 2^18
def _albert_1024():
    """the reference's data/image/albert.exr as the build container converted it (tests/golden/_generated/albert.bin: int32 h, int32 w, fp16 RGBA — the
    .bin container of scripts/common.py:165-171), else the committed 128 x 128 crop of it tiled to 1024 x 1024"""
    gen = os.path.join(ROOT, "tests", "golden", "_generated", "albert.bin")
    if os.path.exists(gen):
        raw = open(gen, "rb").read()
        h, w = struct.unpack("ii", raw[:8])
        return np.frombuffer(raw[8:], np.float16).reshape(h, w, 4).astype(np.float32), "albert.exr (1024 x 1024)"
    crop = np.load(os.path.join(ROOT, "tests", "golden", "albert_crop_128.npy")).astype(np.float32)
    return np.tile(crop, (8, 8, 1)), "albert 128 x 128 crop tiled 8 x 8"


def test_image_config_at_stated_size(oracle, ngp, cuda):
    """config #1: 1024 x 1024 image (finest level 512^2, every level dense, T = 2^24), B = 2^18 stratified positions (testbed_image.cu:220-291)"""
    import pyngp
    img, what = _albert_1024()
    h, w = img.shape[:2]
    assert (h, w) == (1024, 1024)
    img = np.ascontiguousarray(img)
    tb = pyngp.Testbed(pyngp.TestbedMode.Image)
    tb.set_image_data(img)
    tb.reload_network_from_file(os.path.join(CFG, "image", "base.json"))
    # ---- oracle restatement of step 0: parameters, batch, targets, forward, L2 loss
    desc = np.zeros(1, capi.NET_DESC)
    pls = float(np.exp(np.log(np.float32(512.0) / np.float32(16)) / np.float32(15)))
    capi.check(ngp.ngp_hip_gridmlp_make_desc_host(2, 16, 24, 16, H.f32(pls), desc.ctypes.data))
    n_params = ngp.ngp_hip_gridmlp_n_params_host(desc.ctypes.data)
    assert n_params == tb.n_params()
    p32 = np.zeros(n_params, np.float32)
    oracle.orc_gridmlp_init_params(desc.ctypes.data, 1337, p32.ctypes.data)
    p16 = p32.astype(np.float16)
    st, inc = H.pcg32_state(1337)
    xy = np.zeros(2 * B, np.float32)
    oracle.orc_generate_random_uniform(st, inc, 2 * B, xy.ctypes.data)
    oracle.orc_image_stratify2(B, 18, xy.ctypes.data)
    tgt = np.zeros((B, 3), np.float32)
    res = np.array([w, h], np.int32)
    oracle.orc_image_eval_and_snap(B, img.ctypes.data, 3, xy.ctypes.data, res.ctypes.data, tgt.ctypes.data, 3, 0, 0)
    pred = np.zeros((B, 4), np.uint16)
    oracle.orc_gridmlp_inference(2, desc.ctypes.data, p16.view(np.uint16).ctypes.data, xy.ctypes.data, 2, B, pred.ctypes.data, 4)
    vals, grad = np.zeros((B, 3), np.float32), np.zeros((B, 4), np.uint16)
    oracle.orc_tcnn_loss_and_gradient(0, B, 3, H.f32(128.0), pred.ctypes.data, 4, tgt.ctypes.data, vals.ctypes.data, grad.ctypes.data, 4)
    want = float(vals.sum(dtype=np.float64))
    tb.shall_train = True
    tb.train(B)
    assert tb.training_step == 1
    first = tb.loss
    assert first == pytest.approx(want, rel=2e-3)
    for _ in range(300):
        tb.train(B)
    mse = tb.compute_image_mse(False)
    print("image config (%s): step-0 loss %.5f (oracle %.5f), loss after 301 steps %.6f, image MSE %.6f" % (what, first, want, tb.loss, mse))
    assert tb.loss < 0.05 * want and mse < 2e-3


def test_sdf_config_at_stated_size(oracle, ngp, cuda):
    """config #5: (position, distance) pairs of a radius-0.3 sphere, configs/sdf/base.json (MAPE, lr 1e-4), B = 2^18: step 0 against the
    oracle — parameters, forward, loss value — then the loss falls"""
    import pyngp
    rs = np.random.RandomState(0)
    n = 1 << 20
    pts = rs.rand(n, 3).astype(np.float32)
    dist = (np.linalg.norm(pts - 0.5, axis=1) - 0.3).astype(np.float32)
    tb = pyngp.Testbed(pyngp.TestbedMode.Sdf)
    tb.override_sdf_training_data(pts, dist)
    tb.reload_network_from_file(os.path.join(CFG, "sdf", "base.json"))
    desc = np.zeros(1, capi.NET_DESC)
    pls = float(np.exp(np.log(np.float32(2048.0) / np.float32(16)) / np.float32(15)))
    capi.check(ngp.ngp_hip_gridmlp_make_desc_host(3, 16, 19, 16, H.f32(pls), desc.ctypes.data))
    n_params = ngp.ngp_hip_gridmlp_n_params_host(desc.ctypes.data)
    assert n_params == tb.n_params()
    p32 = np.zeros(n_params, np.float32)
    oracle.orc_gridmlp_init_params(desc.ctypes.data, 1337, p32.ctypes.data)
    p16 = p32.astype(np.float16)
    # the first batch is the first B provided samples (cursor 0)
    pos = np.ascontiguousarray(pts[:B])
    pred = np.zeros((B, 4), np.uint16)
    oracle.orc_gridmlp_inference(3, desc.ctypes.data, p16.view(np.uint16).ctypes.data, pos.ctypes.data, 3, B, pred.ctypes.data, 4)
    got = tb.gridmlp_inference(pos[:65536])
    np.testing.assert_allclose(got, pred.view(np.float16).astype(np.float32)[:65536], rtol=1e-2, atol=1e-4)     # step-0 forward, fp16 outputs
    tgt = np.ascontiguousarray(dist[:B].reshape(B, 1))
    vals, grad = np.zeros((B, 1), np.float32), np.zeros((B, 4), np.uint16)
    oracle.orc_tcnn_loss_and_gradient(2, B, 1, H.f32(128.0), pred.ctypes.data, 4, tgt.ctypes.data, vals.ctypes.data, grad.ctypes.data, 4)
    want = float(vals.sum(dtype=np.float64))
    tb.shall_train = True
    tb.train(B)
    first = tb.loss
    assert first == pytest.approx(want, rel=2e-3)
    for _ in range(400):
        tb.train(B)
    print("sdf config: step-0 MAPE %.4f (oracle %.4f), after 401 steps %.4f" % (first, want, tb.loss))
    assert tb.training_step == 401 and tb.loss < 0.5 * first
