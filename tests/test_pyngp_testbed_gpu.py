"""Testbed-level pieces of the `pyngp` boundary that need a device (python_api.cu:540-761): the data + config constructor, render_with_rolling_shutter,
the camera / crop-box helpers of testbed.cu:223-445, shall_train_encoding / shall_train_network, the dataset accessors, loud refusals."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]
pytestmark = pytest.mark.gpu
CONFIG = os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json")


@pytest.fixture(scope="module")
def trained(cuda, tmp_path_factory):
    pytest.importorskip("PIL.Image")
    import pyngp
    import scene
    ds = scene.make_dataset(n_train=8, n_test=1, res=64, device=cuda)
    ds["train_images"] = [np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x) for x in ds["train_images"]]
    path = scene.write_dataset(ds, str(tmp_path_factory.mktemp("scene")))
    t = pyngp.Testbed(pyngp.TestbedMode.Nerf, path, CONFIG)              # python_api.cu:541-543
    assert t.nerf.training.dataset.n_images == 8 and t.n_params() > 10240
    t.shall_train = True
    scene.train(t, 150)
    t.shall_train = False
    return t, ds, path


def test_constructor_with_json_config_and_dataset_accessors(trained):
    import json
    import pyngp
    t, ds, path = trained
    t2 = pyngp.Testbed(pyngp.TestbedMode.Nerf, path, json.load(open(CONFIG)))
    assert t2.n_params() == t.n_params()
    d = t.nerf.training.dataset
    assert d.aabb_scale == 1 and abs(d.scale - 0.33) < 1e-6 and np.allclose(d.offset, [0.5, 0.5, 0.5]) and not d.is_hdr and not d.from_mitsuba
    assert len(d.metadata) == 8 and len(d.transforms) == 8 and len(d.paths) == 8 and d.envmap_resolution == [0, 0]
    md = d.metadata[3]
    assert md.resolution == [64, 64] and md.lens.mode == pyngp.LensMode.Perspective and md.principal_point.tolist() == [0.5, 0.5]
    assert abs(md.focal_length[0] - ds["focal"]) < 1e-3 and md.rolling_shutter.tolist() == [0, 0, 0, 0] and md.camera_distortion.mode == md.lens.mode
    np.testing.assert_array_equal(d.transforms[3][0], t.nerf.training.transforms[3][0])
    t.nerf.render_lens = md.lens
    assert t.nerf.render_lens.mode == pyngp.LensMode.Perspective and t.nerf.render_distortion.params.shape == (7,)


def test_render_with_rolling_shutter(trained):
    t, ds, _ = trained
    p0 = np.asarray(ds["test_poses"][0], np.float32)[:3, :]
    t.set_nerf_camera_matrix(p0)
    t.fov_axis = 0
    t.fov = ds["camera_angle_x"] * 180 / np.pi
    ref = t.render(64, 64, 1, True)
    same = t.render_with_rolling_shutter(p0, p0, [0, 0, 0, 0], 64, 64, 1, True)
    np.testing.assert_array_equal(same, ref)                                 # both poses equal, no per-ray time: the plain frame
    p1 = p0.copy(); p1[:, 3] += np.float32([0.15, 0.0, 0.1])
    end = t.render_with_rolling_shutter(p1, p1, [0, 0, 0, 0], 64, 64, 1, True)
    rs_rows = t.render_with_rolling_shutter(transform_matrix_start=p0, transform_matrix_end=p1, rolling_shutter=[0.0, 0.0, 1.0, 0.0], width=64, height=64, spp=1, linear=True)
    # per-ray time = v (python_api.cu:584: A + B u + C v + D t), and the renderer blends camera_matrix0 * time + camera_matrix1 * (1 - time)
    # (testbed_nerf.cu:1864): time 0 — the TOP rows — is the END matrix, time 1 the start matrix.  Kept as the reference has it.
    to_ref, to_end = np.abs(rs_rows - ref).mean(axis=(1, 2)), np.abs(rs_rows - end).mean(axis=(1, 2))
    print("rolling shutter rows: |rs - start|", to_ref[::8], "|rs - end|", to_end[::8])
    assert to_end[:24].sum() < to_ref[:24].sum() and to_ref[40:].sum() < to_end[40:].sum()
    assert np.abs(rs_rows - ref).mean() > 1e-4 and np.abs(end - ref).mean() > 1e-3
    assert np.abs(t.render_with_rolling_shutter(p0, p1, [1.0, 0, 0, 0], 64, 64, 1, True) - ref).mean() < 1e-6   # A = 1: every ray at camera_matrix0
    assert np.abs(t.render_with_rolling_shutter(p0, p1, [0.0, 0, 0, 0], 64, 64, 1, True) - end).mean() < 1e-6   # time 0: camera_matrix1


def test_camera_and_crop_box_helpers(trained):
    t, ds, _ = trained
    t.reset_camera()
    assert abs(t.fov - 50.625) < 1e-4 and np.allclose(t.fov_xy, [50.625, 50.625], atol=1e-4)
    t.fov_xy = [40.0, 30.0]
    assert np.allclose(t.fov_xy, [40.0, 30.0], atol=1e-4)
    t.fov_axis = 0
    assert abs(t.fov - 40.0) < 1e-4
    t.reset_camera()
    # look_at = position + view_dir * scale (testbed.cu:223-235); scale moves the camera along the view ray, look_at stays
    la = t.look_at.copy()
    assert np.allclose(la, [0.5, 0.5, 0.5], atol=1e-6) and abs(t.scale - 1.5) < 1e-6
    t.scale = 3.0
    assert np.allclose(t.look_at, la, atol=1e-6) and np.allclose(t.camera_matrix[:, 3], la - 3.0 * t.view_dir, atol=1e-6)
    t.look_at = [0.1, 0.2, 0.3]
    assert np.allclose(t.look_at, [0.1, 0.2, 0.3], atol=1e-6)
    t.view_dir = [1.0, 0.0, 0.0]
    m = t.camera_matrix
    assert np.allclose(m[:, 2], [1, 0, 0], atol=1e-6) and np.allclose(m[:, :3].T @ m[:, :3], np.eye(3), atol=1e-5) and np.allclose(t.look_at, [0.1, 0.2, 0.3], atol=1e-5)
    # training-view navigation (testbed.cu:245-281)
    t.first_training_view()
    np.testing.assert_array_equal(t.camera_matrix, t.nerf.training.transforms[0][0])
    t.previous_training_view()
    np.testing.assert_array_equal(t.camera_matrix, t.nerf.training.transforms[0][0])
    t.next_training_view()
    np.testing.assert_array_equal(t.camera_matrix, t.nerf.training.transforms[1][0])
    t.last_training_view(); t.next_training_view()
    np.testing.assert_array_equal(t.camera_matrix, t.nerf.training.transforms[7][0])
    assert t.nerf.render_with_lens_distortion
    # crop box: columns = half axes, last column = centre; NGP space round trip and NeRF space round trip (testbed.cu:395-445)
    box = t.crop_box(False)
    assert np.allclose(box, [[0.5, 0, 0, 0.5], [0, 0.5, 0, 0.5], [0, 0, 0.5, 0.5]], atol=1e-6)
    th = 0.3
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
    want = np.concatenate([rot * np.float32([0.3, 0.2, 0.1])[None, :], np.float32([[0.45], [0.55], [0.5]])], 1)
    t.set_crop_box(want, False)
    assert np.allclose(t.crop_box(False), want, atol=1e-6)
    assert np.allclose(t.render_aabb_to_local, rot.T, atol=1e-6) and np.allclose(t.render_aabb.diag(), [0.6, 0.4, 0.2], atol=1e-6)
    nerf = t.crop_box(True)
    t.set_crop_box(nerf, True)
    assert np.allclose(t.crop_box(False), want, atol=1e-5)
    corners = np.array(t.crop_box_corners(False))
    assert corners.shape == (8, 3) and np.allclose(corners.mean(0), want[:, 3], atol=1e-6)
    assert np.allclose(corners[7], want[:, :3].sum(1) + want[:, 3], atol=1e-6)
    # ... and the renderer honours it: outside the rotated box nothing is accumulated
    t.set_nerf_camera_matrix(np.asarray(ds["test_poses"][0], np.float32)[:3, :])
    small = t.render(64, 64, 1, True)
    t.set_crop_box(np.float32([[0.5, 0, 0, 0.5], [0, 0.5, 0, 0.5], [0, 0, 0.5, 0.5]]), False)
    t.render_aabb_to_local = np.eye(3, dtype=np.float32)
    full = t.render(64, 64, 1, True)
    assert small[..., :3].sum() < 0.8 * full[..., :3].sum()


def test_shall_train_encoding_and_network(trained):
    """Adam's optimize_matrix_params / optimize_non_matrix_params (testbed.cu:2556-2563): the switched-off class keeps its weights bit for bit"""
    import scene
    t, ds, _ = trained
    n_mlp = t.n_params() - t.n_encoding_params()
    assert n_mlp == 10240
    base = t.debug_params("training").copy()
    t.shall_train = True
    t.shall_train_network = False
    scene.train(t, t.training_step + 5)
    p = t.debug_params("training").copy()
    np.testing.assert_array_equal(p[:n_mlp], base[:n_mlp])
    assert (p[n_mlp:] != base[n_mlp:]).sum() > 1000
    t.shall_train_network = True
    t.shall_train_encoding = False
    scene.train(t, t.training_step + 5)
    q = t.debug_params("training").copy()
    np.testing.assert_array_equal(q[n_mlp:], p[n_mlp:])
    assert (q[:n_mlp] != p[:n_mlp]).sum() > 1000
    t.shall_train_encoding = True
    # (every training switch of python_api.cu:804-818 is built: extrinsics / focal length in tests/test_extrinsics_gpu.py, distortion / envmap in
    # tests/test_render_modes_e2e_gpu.py, latent codes / sharpness in tests/test_netx_e2e_gpu.py)
    t.frame()
    t.shall_train = False
    with pytest.raises(RuntimeError, match="not part of this build"):
        t.calculate_iou()
    with pytest.raises(RuntimeError, match="DLSS"):
        t.dlss = True
    t.nerf.glow_mode = 1                                                        # built since round 2 (tests/test_render_modes_e2e_gpu.py)
    assert np.isfinite(t.render(16, 16, 1, True)).all()
    t.nerf.glow_mode = 0
