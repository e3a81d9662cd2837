"""GPU tests of save_snapshot / load_snapshot (src/testbed.cu:3006-3106): the file is read back by an independent MessagePack
implementation, a fresh Testbed renders the same image from it, and training resumes from the stored optimizer state."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trained(cuda, tmp_path_factory):
    import scene
    ds = scene.make_dataset(n_train=8, n_test=1, res=64, device=cuda)
    tb = scene.build_testbed(ds)
    scene.train(tb, 60)
    d = tmp_path_factory.mktemp("snap")
    return ds, tb, str(d)


def _render(tb, ds):
    tb.background_color = [0.0, 0.0, 0.0, 1.0]
    tb.snap_to_pixel_centers = True
    tb.fov_axis = 0
    tb.fov = ds["camera_angle_x"] * 180 / np.pi
    tb.shall_train = False
    tb.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
    return tb.render(64, 64, 1, True)


def test_snapshot_file_schema(trained):
    import msgpack
    ds, tb, d = trained
    path = os.path.join(d, "a.msgpack")
    tb.save_snapshot(path, False)
    cfg = msgpack.unpackb(open(path, "rb").read(), raw=False)
    assert cfg["encoding"]["otype"] == "HashGrid" and "optimizer" in cfg and "rgb_network" in cfg   # the whole network config travels
    s = cfg["snapshot"]
    assert s["version"] == 1 and s["density_grid_size"] == 128
    assert s["params_type"] == "__half" and s["n_params"] == tb.n_params() and len(s["params_binary"]) == 2 * tb.n_params()
    assert len(s["density_grid_binary"]) == 2 * 128 ** 3          # aabb_scale 1 -> one cascade, fp16
    assert s["training_step"] == tb.training_step == 60
    assert s["aabb"] == {"min": [0.0, 0.0, 0.0], "max": [1.0, 1.0, 1.0]}
    assert s["nerf"]["aabb_scale"] == 1 and s["nerf"]["rgb"]["rays_per_batch"] == tb.nerf.training.rays_per_batch
    dj = s["nerf"]["dataset"]
    assert dj["n_images"] == 8 and len(dj["xforms"]) == 8 and len(dj["metadata"]) == 8
    assert dj["metadata"][0]["resolution"] == [64, 64] and len(dj["xforms"][0]["start"]) == 3 and len(dj["xforms"][0]["start"][0]) == 4
    assert "optimizer" not in s
    grid = np.frombuffer(s["density_grid_binary"], np.float16)
    assert np.isfinite(grid.astype(np.float32)).all() and (grid > 0).any()


def test_fresh_testbed_renders_the_same_from_snapshot(trained):
    import pyngp
    ds, tb, d = trained
    path = os.path.join(d, "b.msgpack")
    tb.save_snapshot(path, False)
    ref = _render(tb, ds)
    t2 = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    t2.load_snapshot(path)
    assert t2.training_step == 60 and t2.n_params() == tb.n_params()
    got = _render(t2, ds)
    assert ref[..., 3].max() > 0.5                              # something was rendered
    # same weights; the occupancy grid went through fp16, so allow the few pixels a flipped cell can touch
    diff = np.abs(got - ref)
    assert np.mean(diff) < 2e-3 and np.mean(diff > 1e-2) < 0.02
    # and a snapshot of the loaded state carries identical weights
    import msgpack
    path2 = os.path.join(d, "b2.msgpack")
    t2.save_snapshot(path2, False)
    a = msgpack.unpackb(open(path, "rb").read(), raw=False)["snapshot"]
    b = msgpack.unpackb(open(path2, "rb").read(), raw=False)["snapshot"]
    assert a["params_binary"] == b["params_binary"] and a["density_grid_binary"] == b["density_grid_binary"]


def test_resume_training_with_optimizer_state(trained):
    import scene
    ds, tb, d = trained
    path = os.path.join(d, "c.msgpack")
    tb.save_snapshot(path, True)
    t3 = scene.build_testbed(ds)
    t3.load_snapshot(path)
    assert t3.training_step == 60
    loss_before = tb.loss
    t3.shall_train = True
    scene.train(t3, 80)
    assert t3.training_step == 80 and np.isfinite(t3.loss)
    assert t3.loss < loss_before * 1.5                          # continues from the trained state, not from scratch
    img = _render(t3, ds)
    assert np.isfinite(img).all()


def test_load_snapshot_errors(trained, tmp_path):
    import msgpack
    import pyngp
    ds, tb, d = trained
    t = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    p = str(tmp_path / "no_snapshot.msgpack")
    open(p, "wb").write(msgpack.packb({"encoding": {}}))
    with pytest.raises(RuntimeError, match="does not contain a snapshot"):
        t.load_snapshot(p)
    path = os.path.join(d, "e.msgpack")
    tb.save_snapshot(path, False)
    cfg = msgpack.unpackb(open(path, "rb").read(), raw=False)
    cfg["snapshot"]["density_grid_size"] = 64
    p2 = str(tmp_path / "bad_grid.msgpack")
    open(p2, "wb").write(msgpack.packb(cfg, use_bin_type=True))
    with pytest.raises(RuntimeError, match="Incompatible grid size"):
        t.load_snapshot(p2)
    cfg["snapshot"]["density_grid_size"] = 128
    cfg["snapshot"]["version"] = 0
    open(p2, "wb").write(msgpack.packb(cfg, use_bin_type=True))
    with pytest.raises(RuntimeError, match="old format"):
        t.load_snapshot(p2)


def test_render_schedule_does_not_change_the_image(trained):
    """the tracer's pass structure (steps per compaction, number of independent pixel ranges / streams) is a schedule, not arithmetic"""
    ds, tb, d = trained
    tb.nerf.render_n_streams, tb.nerf.render_max_steps_per_pass = 1, 8          # the reference's schedule (testbed_nerf.cu:2231)
    ref = _render(tb, ds)
    for streams, cap in [(1, 64), (2, 8), (3, 64), (8, 64)]:
        tb.nerf.render_n_streams, tb.nerf.render_max_steps_per_pass = streams, cap
        np.testing.assert_array_equal(_render(tb, ds), ref)
    assert ref[..., 3].max() > 0.5


def test_script_facing_properties(trained):
    """names scripts/run.py and the add-on read / write on the NeRF path (python_api.cu:650-761)"""
    import pyngp
    ds, tb, d = trained
    tb.nerf.rendering_min_transmittance = 1e-4
    assert tb.nerf.render_min_transmittance == pytest.approx(1e-4)
    tb.nerf.render_with_camera_distortion = True
    assert tb.nerf.render_with_lens_distortion
    tb.camera_smoothing = False; tb.loop_animation = False; tb.dynamic_res = False
    assert tb.nerf.training.n_images == 8 and tb.nerf.training.aabb_scale == 1 and not tb.nerf.training.is_hdr
    assert tb.nerf.training.loss_type == tb.nerf.training.loss
    bb = tb.render_aabb
    assert np.allclose(bb.min, [0, 0, 0]) and np.allclose(bb.max, [1, 1, 1]) and np.allclose(tb.raw_aabb.max, [1, 1, 1])
    tb.render_mode = pyngp.RenderMode.Shade
    tb.render_mode = pyngp.RenderMode.Normals                 # every ERenderMode is built since round 2 (tests/test_render_modes_e2e_gpu.py)
    assert tb.render_mode == pyngp.RenderMode.Normals
    tb.render_mode = pyngp.RenderMode.Shade
    tb.set_camera_to_training_view(3)
    tb.shall_train = False
    tb.background_color = [0.0, 0.0, 0.0, 0.0]
    img = tb.render(64, 64, 1, True)
    ref = ds["train_images"][3]
    ref = np.asarray(ref.cpu().numpy() if hasattr(ref, "cpu") else ref).astype(np.float32) / 255.0
    # rendered from the training pose with the training intrinsics, the picture correlates with the training image's alpha
    a, b = img[..., 3].reshape(-1), ref[..., 3].reshape(-1)
    assert a.std() > 0 and b.std() > 0, (float(a.min()), float(a.max()), float(b.min()), float(b.max()), ref.shape)
    assert np.corrcoef(a, b)[0, 1] > 0.5
