"""GPU tests of the Blender add-on path end to end through `pyngp`: snapshots written with python-msgpack -> NerfDescriptor / RenderRequest ->
Testbed.request_nerf_render_sync / _async (python_api.cu:192-260, 577-583) against the CPU oracle's frame (oracle/orc_multi.c +
orc_accumulate / orc_tonemap)."""
import ctypes
import os
import sys
import threading

import numpy as np
import pytest

import capi
import helpers as H
from test_multi_render_gpu import _props, _trs, _with_all_mask, _mask, _ds, _camera

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]
pytestmark = pytest.mark.gpu
msgpack = pytest.importorskip("msgpack")

W, Hh = 40, 30


def _write_snapshot(path, desc, params16, grid_fp32, log2_hashmap_size):
    cfg = {
        "loss": {"otype": "Huber"},
        "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2_hashmap_size, "base_resolution": 16},
        "network": {"otype": "FullyFusedMLP", "n_neurons": 64, "n_hidden_layers": 1},
        "rgb_network": {"otype": "FullyFusedMLP", "n_neurons": 64, "n_hidden_layers": 2},
        "snapshot": {
            "version": 1, "density_grid_size": 128, "density_grid_binary": grid_fp32.astype(np.float16).tobytes(),
            "nerf": {"aabb_scale": 1, "rgb": {"rays_per_batch": 4096, "measured_batch_size": 0, "measured_batch_size_before_compaction": 0}},
            "training_step": 100, "loss": 0.01, "aabb": {"min": [0.0, 0.0, 0.0], "max": [1.0, 1.0, 1.0]}, "bounding_radius": 1.0,
            "n_params": int(params16.size), "params_type": "__half", "params_binary": params16.tobytes(),
        },
    }
    open(path, "wb").write(msgpack.packb(cfg, use_bin_type=True))


@pytest.fixture(scope="module")
def two_snapshots(ngp, oracle, cuda, tmp_path_factory):
    d = tmp_path_factory.mktemp("bl")
    desc = H.make_desc(ngp, log2_hashmap_size=12)
    out = []
    for k in range(2):
        P = H.random_params(desc, seed=20 + k, grid_amp=2.0)
        P[2048:2048 + 64] *= np.float16(4.0)
        grid = H.blob_density_grid(1, seed=3 + 8 * k)
        # the snapshot stores the grid as fp16: the oracle must threshold the same rounded values
        grid16 = grid.astype(np.float16).astype(np.float32)
        bf, mean = H.oracle_bitfield(oracle, grid16, 1)
        path = str(d / ("nerf%d.msgpack" % k))
        _write_snapshot(path, desc, P.view(np.uint16), grid, 12)
        out.append(dict(path=path, params=P.view(np.uint16).copy(), bitfield=bf))
    return desc, out


def _request(pyngp, paths, xfs, opacities, masks_per_nerf, global_masks, cam_pos, focal, mip=0, flip_y=True, exposure=0.0, bg=(0.1, 0.2, 0.3, 1.0)):
    ds = pyngp.DownsampleInfo.MakeFromMip([W, Hh], mip)
    output = pyngp.RenderOutputProperties([W, Hh], ds, 1, pyngp.ColorSpace.SRGB, pyngp.TonemapCurve.Identity, exposure, list(bg), flip_y)
    cam34 = H.look_at_xform(cam_pos).reshape(4, 3).T      # column-major flat -> 3x4
    camera = pyngp.RenderCameraProperties(cam34, pyngp.CameraModel.Perspective, focal, 0.0, 0.0, 1.0, pyngp.SphericalQuadrilateralConfig.Zero(), pyngp.QuadrilateralHexahedronConfig.Zero())
    box = pyngp.BoundingBox([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    nerfs = [pyngp.NerfDescriptor(p, box, xf.astype(np.float32), pyngp.RenderModifiers(m), op) for p, xf, op, m in zip(paths, xfs, opacities, masks_per_nerf)]
    return pyngp.RenderRequest(output, camera, pyngp.RenderModifiers(global_masks), nerfs, box)


def _oracle_frame(oracle, desc, snaps, xfs, opacities, mask_arrays, cam_pos, focal, mip, flip_y, exposure, bg):
    ds = _ds(oracle, W, Hh, mip)
    cam = _camera(pos=cam_pos, focal=focal)
    props = np.concatenate([_props(xf, s["bitfield"].ctypes.data, m.ctypes.data if len(m) else 0, len(m), opacity=op) for xf, s, op, m in zip(xfs, snaps, opacities, mask_arrays)])
    n = len(snaps)
    net_ptrs = (ctypes.c_void_p * n)(*[desc.ctypes.data] * n)
    par_ptrs = (ctypes.c_void_p * n)(*[s["params"].ctypes.data for s in snaps])
    rgb_act, dens_act, min_t = np.full(n, 2, np.int32), np.full(n, 3, np.int32), np.full(n, 0.01, np.float32)
    fb = np.zeros((Hh, W, 4), np.float32); db = np.zeros(W * Hh, np.float32)
    oracle.orc_multi_render(n, net_ptrs, par_ptrs, props.ctypes.data, rgb_act.ctypes.data, dens_act.ctypes.data, min_t.ctypes.data, ds.ctypes.data, cam.ctypes.data, flip_y, fb.ctypes.data, db.ctypes.data)
    # bl_render_frame (testbed.cu:2688-2692): accumulate (first sample) + tonemap with the request's colour space on both sides
    res = np.array([W, Hh], np.int32)
    acc = np.zeros_like(fb); surf = np.zeros_like(fb)
    oracle.orc_accumulate(res.ctypes.data, fb.ctypes.data, acc.ctypes.data, ctypes.c_float(0.0), 1)
    bgc = np.array(bg, np.float32)
    oracle.orc_tonemap(res.ctypes.data, ctypes.c_float(exposure), bgc.ctypes.data, acc.ctypes.data, 1, 1, 0, 0, surf.ctypes.data)
    return surf, fb


def test_request_nerf_render_sync_matches_oracle(oracle, cuda, two_snapshots):
    import pyngp
    desc, snaps = two_snapshots
    xfs = [_trs((0.0, 0.0, 0.0), 0.0, 1.0), _trs((0.55, 0.2, 0.1), 0.7, 0.8)]
    ops = [1.0, 0.7]
    cam_pos, focal = (1.7, -1.3, 1.0), 42.0
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    req = _request(pyngp, [s["path"] for s in snaps], xfs, ops, [[], []], [], cam_pos, focal)
    print("request built", flush=True)
    img = tb.request_nerf_render_sync(req)
    print("rendered", tb.bl_render_samples, flush=True)
    assert img.shape == (Hh, W, 4) and tb.bl_render_samples > 1000
    ref, raw = _oracle_frame(oracle, desc, snaps, xfs, ops, [np.zeros(0, capi.MASK3D)] * 2, cam_pos, focal, 0, 1, 0.0, (0.1, 0.2, 0.3, 1.0))
    assert (raw[..., 3] > 0.05).mean() > 0.05
    diff = np.abs(img - ref)
    assert np.mean(diff) < 2e-3 and np.mean(diff.max(axis=-1) > 3e-2) < 0.01
    # background shows through where nothing was hit
    empty = raw[..., 3] == 0
    assert empty.any() and np.allclose(img[empty][:, 3], 1.0, atol=1e-6)
    # second request with the same descriptors re-uses the loaded fields
    img2 = tb.request_nerf_render_sync(req)
    np.testing.assert_array_equal(img2, img)


@pytest.mark.parametrize("n_nerfs", [1, 2])
def test_fused_pass_loop_renders_the_unfused_frame(oracle, cuda, two_snapshots, n_nerfs):
    """NerfRenderer's fused pass loop (ngp_hip_multi_advance: march + cull + compact + per-NeRF lists in one launch, counts through a host mailbox, resting rays,
    the stock tracer's sample budget per pass) against the reference's launch sequence (nerf_renderer.cu:664-791: compact, read back, march, cull, read back, ...).
    On the SAME schedule (bl_reference_schedule: n_steps from the rays that entered the pass, no rest) the frames are identical bit for bit.  The fork's sampler is not
    independent of the schedule — inside a pass it emits the sample at t + dt before testing the position, at a pass boundary the ray is first moved to the next occupied
    voxel, and between two passes the cull decides which NeRF samples next — so the default schedule (fewer, larger passes) differs from the unfused frame in under 3 %
    of the pixels by a few 1e-2, like the reference's own frames do between two resolutions; it stays inside the oracle tolerance of the tests above."""
    import pyngp
    desc, snaps = two_snapshots
    xfs = [_trs((0.0, 0.0, 0.0), 0.0, 1.0), _trs((0.55, 0.2, 0.1), 0.7, 0.8)][:n_nerfs]
    ops = [1.0, 0.7][:n_nerfs]
    cam_pos, focal = (1.7, -1.3, 1.0), 42.0
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    req = _request(pyngp, [s["path"] for s in snaps[:n_nerfs]], xfs, ops, [[]] * n_nerfs, [], cam_pos, focal)
    tb.bl_fused_passes = False
    ref = tb.request_nerf_render_sync(req)
    passes_ref = tb.bl_render_passes
    tb.bl_fused_passes, tb.bl_reference_schedule = True, True
    same = tb.request_nerf_render_sync(req)
    np.testing.assert_array_equal(same, ref)
    tb.bl_reference_schedule = False
    frames = {}
    for skips, factor, cap in ((24, 3.0, 64), (3, 3.0, 64), (0, 2.0, 8), (1, 1.0, 2)):
        tb.bl_max_skips_per_pass, tb.bl_pass_samples_factor, tb.bl_max_steps_per_pass = skips, factor, cap
        frames[(skips, factor, cap)] = (tb.request_nerf_render_sync(req), tb.bl_render_passes)
    for key, (img, passes) in frames.items():
        d = np.abs(img - ref)
        assert np.mean(d) < 4e-3 and np.mean(d.max(axis=-1) > 5e-2) < 0.03, key
    assert passes_ref > 0 and all(p > 0 for _, p in frames.values())   # (pass counts at production size: bench.py's bl_render leg)
    # ... and the default still matches the oracle's dense loop
    oref, raw = _oracle_frame(oracle, desc, snaps[:n_nerfs], xfs, ops, [np.zeros(0, capi.MASK3D)] * n_nerfs, cam_pos, focal, 0, 1, 0.0, (0.1, 0.2, 0.3, 1.0))
    diff = np.abs(frames[(24, 3.0, 64)][0] - oref)
    assert np.mean(diff) < 2e-3 and np.mean(diff.max(axis=-1) > 3e-2) < 0.01


def test_masks_mip_and_global_modifiers(oracle, cuda, two_snapshots):
    import pyngp
    desc, snaps = two_snapshots
    xfs = [_trs((0.0, 0.0, 0.0), 0.0, 1.0), _trs((0.45, 0.25, 0.05), -0.4, 0.9)]
    ops = [0.9, 1.0]
    cam_pos, focal = (1.9, -1.0, 1.1), 40.0
    box_t, sph_t = _trs((0.5, 0.5, 0.5), 0.4), _trs((0.55, 0.45, 0.55), 0.0, 1.1)
    m_box = pyngp.Mask3D.Box([0.7, 0.5, 0.9], box_t.astype(np.float32), pyngp.MaskMode.Add, 0.1, 0.8)
    g_sph = pyngp.Mask3D.Sphere(0.45, sph_t.astype(np.float32), pyngp.MaskMode.Add, 0.05, 1.0)
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    req = _request(pyngp, [s["path"] for s in snaps], xfs, ops, [[m_box], []], [g_sph], cam_pos, focal, mip=1, flip_y=False, exposure=0.5)
    img = tb.request_nerf_render_sync(req)
    # oracle side: local masks, then the global ones brought into each NeRF's frame (RenderModifiers ctor, render_modifiers.cuh:30-46)
    ref_masks = []
    for k, xf in enumerate(xfs):
        local = [_mask(0, 0, box_t, [0.7, 0.5, 0.9], 0.1, 0.8)] if k == 0 else []
        glob = _mask(2, 0, np.linalg.inv(xf) @ sph_t, [0.45], 0.05, 1.0)
        ref_masks.append(_with_all_mask(local + [glob]))
    ref, raw = _oracle_frame(oracle, desc, snaps, xfs, ops, ref_masks, cam_pos, focal, 1, 0, 0.5, (0.1, 0.2, 0.3, 1.0))
    assert (raw[..., 3] > 0.05).mean() > 0.03
    diff = np.abs(img - ref)
    assert np.mean(diff) < 3e-3 and np.mean(diff.max(axis=-1) > 3e-2) < 0.02
    unmasked, _ = _oracle_frame(oracle, desc, snaps, xfs, ops, [np.zeros(0, capi.MASK3D)] * 2, cam_pos, focal, 1, 0, 0.5, (0.1, 0.2, 0.3, 1.0))
    assert np.abs(unmasked - ref).max() > 0.05


def test_request_nerf_render_async_and_errors(cuda, two_snapshots, tmp_path):
    import pyngp
    desc, snaps = two_snapshots
    xfs = [_trs(), _trs((0.5, 0.2, 0.1), 0.7, 0.8)]
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    req = _request(pyngp, [s["path"] for s in snaps], xfs, [1.0, 1.0], [[], []], [], (1.7, -1.3, 1.0), 42.0)
    sync_img = tb.request_nerf_render_sync(req)
    got, done = [], threading.Event()

    def cb(arr):
        got.append(np.array(arr))
        done.set()

    tb.request_nerf_render_async(req, cb)
    assert done.wait(60.0)
    tb.wait_for_render()
    np.testing.assert_array_equal(got[0], sync_img)
    # the add-on's progressive preview: the callback queues the NEXT request itself (it runs on the worker thread), by keyword like
    # python_api.cu:577-580; three frames chain without a join of the worker on itself, and the busy flag is free afterwards
    chain, chain_done = [], threading.Event()

    def chained(arr):
        chain.append(np.array(arr))
        if len(chain) < 3:
            tb.request_nerf_render_async(render_request=req, render_callback=chained)
        else:
            chain_done.set()

    tb.request_nerf_render_async(render_request=req, render_callback=chained)
    assert chain_done.wait(120.0) and len(chain) == 3
    tb.wait_for_render()
    for c in chain:
        np.testing.assert_array_equal(c, sync_img)
    # a failing async request (missing snapshot) reports on stderr, never calls back, and leaves the renderer usable
    bad_async = _request(pyngp, [str(tmp_path / "nope_async.msgpack")], [_trs()], [1.0], [[]], [], (1.7, -1.3, 1.0), 42.0)
    never = []
    tb.request_nerf_render_async(bad_async, lambda a: never.append(1))
    tb.wait_for_render()
    assert not never
    np.testing.assert_array_equal(tb.request_nerf_render_sync(req), sync_img)
    # a Testbed collected while its worker is still rendering waits for it (the GIL is released around the wait; the worker needs it for the callback)
    tb2 = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    late, late_done = [], threading.Event()
    tb2.request_nerf_render_async(req, lambda a: (late.append(1), late_done.set()))
    del tb2
    import gc
    gc.collect()
    assert late_done.wait(60.0) and late == [1]
    # an empty request renders the background only
    empty = _request(pyngp, [], [], [], [], [], (1.7, -1.3, 1.0), 42.0)
    img = tb.request_nerf_render_sync(empty)
    assert np.allclose(img[..., 3], 1.0)
    # missing snapshot file -> RuntimeError, and the renderer stays usable afterwards
    bad = _request(pyngp, [str(tmp_path / "nope.msgpack")], [_trs()], [1.0], [[]], [], (1.7, -1.3, 1.0), 42.0)
    with pytest.raises(RuntimeError):
        tb.request_nerf_render_sync(bad)
    np.testing.assert_array_equal(tb.request_nerf_render_sync(req), sync_img)
    # request-schema helpers the add-on uses
    cam = pyngp.RenderCameraProperties(np.eye(4, dtype=np.float32)[:3], pyngp.CameraModel.Perspective, 50.0, 0.0, 0.0, 1.0, pyngp.SphericalQuadrilateralConfig.Zero(), pyngp.QuadrilateralHexahedronConfig.Zero())
    cam2 = pyngp.RenderCameraProperties(np.eye(4, dtype=np.float32)[:3], pyngp.CameraModel.Perspective, 51.0, 0.0, 0.0, 1.0, pyngp.SphericalQuadrilateralConfig.Zero(), pyngp.QuadrilateralHexahedronConfig.Zero())
    assert cam == cam and cam != cam2
    bb = pyngp.BoundingBox([0, 0, 0], [1, 2, 3])
    assert bb.contains([0.5, 1.0, 2.9]) and not bb.contains([1.5, 0, 0]) and np.allclose(bb.center(), [0.5, 1.0, 1.5]) and np.allclose(bb.diag(), [1, 2, 3])
