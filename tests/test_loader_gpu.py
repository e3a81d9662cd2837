"""GPU test of Testbed.load_training_data (python_api.cu:546 -> Testbed::load_nerf, src/testbed_nerf.cu:2735-2759): a dataset written to disk
in the nerf-synthetic layout and loaded through the transforms.json + PNG path trains exactly like the same data fed through
create_empty_nerf_dataset + set_image / set_camera_* (the in-memory route the other tests use)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]
pytestmark = pytest.mark.gpu


def test_file_loaded_dataset_trains_identically(cuda, tmp_path):
    pytest.importorskip("PIL.Image")
    import pyngp
    import scene
    ds = scene.make_dataset(n_train=6, n_test=1, res=48, device=cuda)
    ds["train_images"] = [np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x) for x in ds["train_images"]]
    path = scene.write_dataset(ds, str(tmp_path))
    a = scene.build_testbed(ds)
    b = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    b.load_training_data(path)
    b.reload_network_from_file(os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json"))
    b.nerf.render_with_lens_distortion = True
    b.exposure = 0.0
    b.shall_train = True
    assert b.nerf.training.n_images_for_training == 6
    # identical training inputs on the device: poses, intrinsics / lens records, pixels
    for i in range(6):
        np.testing.assert_array_equal(a.nerf.training.get_camera_extrinsics(i), b.nerf.training.get_camera_extrinsics(i))
        assert a.nerf.training.get_image_metadata(i) == b.nerf.training.get_image_metadata(i)
        np.testing.assert_array_equal(a.nerf.training.get_image_rgba8(i), b.nerf.training.get_image_rgba8(i))
        np.testing.assert_array_equal(b.nerf.training.get_image_rgba8(i), ds["train_images"][i])
    # training itself is not bit-reproducible run to run (the compaction order of the samples and the MLP weight-gradient partial sums
    # depend on scheduling, as in the reference), so the two runs are compared statistically
    scene.train(a, 60)
    scene.train(b, 60)
    assert a.training_step == b.training_step == 60
    # two short independent runs on a 6-image 48x48 scene scatter quite a bit: the comparison only has to catch a wrong dataset
    # (a swapped axis or a wrong focal length leaves the loss several times higher and the image unrelated)
    assert 0.5 < a.loss / b.loss < 2.0
    ra, rb = a.nerf.training.rays_per_batch, b.nerf.training.rays_per_batch
    assert 0.6 < ra / rb < 1.67
    pose = ds["test_poses"][0][:3, :]
    for t in (a, b):
        t.shall_train = False
        t.set_nerf_camera_matrix(pose)
    assert np.mean(np.abs(a.render(48, 48, 1, True) - b.render(48, 48, 1, True))) < 0.05
    # a snapshot path is accepted too and switches training off (testbed_nerf.cu:2744-2747)
    snap = str(tmp_path / "s.msgpack")
    a.save_snapshot(snap, False)
    c = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    c.load_training_data(snap)
    assert c.training_step == 60 and not c.shall_train


def test_load_time_sharpening(ngp, oracle, cuda, tmp_path):
    """`sharpen` in transforms.json / Testbed.nerf.sharpen (scripts/run.py --sharpen; nerf_loader.cu:803-825): Byte images become premultiplied
    linear half4 and go through the 5-tap unsharp filter on the device; against the oracle's restatement of both kernels"""
    pytest.importorskip("PIL.Image")
    import json
    import pyngp
    import scene
    import helpers as H
    from capi import check
    ds = scene.make_dataset(n_train=3, n_test=1, res=40, device=cuda)
    ds["train_images"] = [np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x) for x in ds["train_images"]]
    path = scene.write_dataset(ds, str(tmp_path))
    meta = json.load(open(path)); meta["sharpen"] = 0.7; json.dump(meta, open(path, "w"))
    t = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    t.load_training_data(path)
    assert t.nerf.training.get_image_metadata(0)["image_data_type"] == 2           # Half
    for i in range(3):
        img = np.ascontiguousarray(ds["train_images"][i])
        n = img.shape[0] * img.shape[1]
        h4 = np.zeros((n, 4), np.uint16); ref = np.zeros((n, 4), np.uint16)
        oracle.orc_image_from_rgba32_f16(n, img.ctypes.data, h4.ctypes.data, 0x00FF00FF)
        oracle.orc_image_sharpen(n, img.shape[1], h4.ctypes.data, ref.ctypes.data, 1, H.f32(0.7))
        got = t.nerf.training.get_image_pixels(i).astype(np.float32).reshape(n, 4)
        want = ref.view(np.float16).astype(np.float32)
        # srgb_to_linear goes through powf (device intrinsic vs libm): a half ulp on the converted pixel, amplified by the centre weight 5.4 / 1.4
        np.testing.assert_allclose(got, want, rtol=8e-3, atol=2e-3)
        assert np.abs(want - h4.view(np.float16).astype(np.float32)).max() > 0.05     # it did sharpen
    # the kernels alone, same inputs on both sides: exact for float4, one rounding for half4
    rs = np.random.RandomState(0)
    f4 = rs.rand(50 * 30, 4).astype(np.float32)
    want = np.zeros_like(f4); oracle.orc_image_sharpen(1500, 50, f4.ctypes.data, want.ctypes.data, 0, H.f32(0.25))
    d_in, d_out = H.to_dev(f4, cuda), H.dev_zeros(f4.nbytes, cuda)
    check(ngp.ngp_hip_image_sharpen(None, 1500, 50, d_in.data_ptr(), d_out.data_ptr(), 3, H.f32(0.25)))
    np.testing.assert_array_equal(H.to_host(d_out, np.float32).reshape(1500, 4), want)
    assert ngp.ngp_hip_image_sharpen(None, 1500, 50, d_in.data_ptr(), d_in.data_ptr(), 3, H.f32(0.25)) != 0      # in place is refused
    # training on the sharpened half4 images runs
    t.reload_network_from_file(os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json"))
    t.shall_train = True
    scene.train(t, 20)
    assert np.isfinite(t.loss) and t.nerf.training.measured_batch_size > 0
