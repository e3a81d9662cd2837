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


def test_fox_like_dataset_jpeg_depth_rays_and_exr_train(cuda, tmp_path):
    """A dataset in the layout of the reference's own data/nerf/fox (`.jpg` frames, BASELINE config #2) plus the optional companions of
    nerf_loader.cu:574-668 — 16-bit depth images, rays_<name>.dat per-pixel rays — and one with EXR frames: loaded through
    Testbed.load_training_data, the device holds what the host stage decoded, and training runs on each."""
    Image = pytest.importorskip("PIL.Image")
    import json
    import pyngp
    import scene
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_image_io_cpu import _write_exr
    ds = scene.make_dataset(n_train=5, n_test=1, res=64, device=cuda)
    imgs = [np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x) for x in ds["train_images"]]
    ds["train_images"] = imgs
    path = scene.write_dataset(ds, str(tmp_path))
    meta = json.load(open(path))
    base = os.path.dirname(path)
    rs = np.random.RandomState(0)
    for k, f in enumerate(meta["frames"]):
        png = os.path.join(base, f["file_path"] + ("" if f["file_path"].endswith(".png") else ".png"))
        rgba = np.asarray(Image.open(png).convert("RGBA"))
        jpg = os.path.splitext(png)[0] + ".jpg"
        bg = (rgba[..., :3].astype(np.float32) * (rgba[..., 3:4] / 255.0)).astype(np.uint8)   # JPEG has no alpha: composite on black
        Image.fromarray(bg, "RGB").save(jpg, quality=95)
        os.remove(png)
        f["file_path"] = os.path.relpath(jpg, base)
        depth = (rs.rand(64, 64) * 4000 + 500).astype(np.uint16)
        Image.fromarray(depth).save(os.path.splitext(png)[0] + "_depth.png")
        f["depth_path"] = os.path.relpath(os.path.splitext(png)[0] + "_depth.png", base)
    meta["integer_depth_scale"] = 1.0 / 1000.0
    json.dump(meta, open(path, "w"))
    host = pyngp.load_nerf_host(path)
    t = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    t.load_training_data(path)
    assert t.nerf.training.n_images_for_training == 5
    for i in range(5):
        md = t.nerf.training.get_image_metadata(i)
        assert md["image_data_type"] == 1 and md.get("has_depth") and not md.get("has_rays")
        np.testing.assert_array_equal(t.nerf.training.get_image_rgba8(i), host["pixels"][i])
    t.reload_network_from_file(os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json"))
    t.nerf.training.depth_supervision_lambda = 0.1
    t.shall_train = True
    scene.train(t, 30)
    assert np.isfinite(t.loss) and t.nerf.training.measured_batch_size > 0

    # per-pixel rays: the pinhole rays of every view written out as rays_<name>.dat (NeRF frame) must train like the pinhole model itself
    for k, f in enumerate(meta["frames"]):
        m = np.array(f["transform_matrix"], np.float32)
        jpg = os.path.join(base, f["file_path"])
        fl = np.float32(meta["fl_x"])
        ys, xs = np.meshgrid(np.arange(64, dtype=np.float32) + 0.5, np.arange(64, dtype=np.float32) + 0.5, indexing="ij")
        d_cam = np.stack([(xs - 32) / fl, -(ys - 32) / fl, -np.ones_like(xs)], -1)      # NeRF camera: x right, y up, looking down -z
        d_world = d_cam @ m[:3, :3].T
        rays = np.concatenate([np.broadcast_to(m[:3, 3], d_world.shape), d_world], -1).astype(np.float32)
        rays.tofile(os.path.join(os.path.dirname(jpg), "rays_" + os.path.splitext(os.path.basename(jpg))[0] + ".dat"))
    r = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    r.load_training_data(path)
    assert all(r.nerf.training.get_image_metadata(i).get("has_rays") for i in range(5))
    assert r.nerf.training.near_distance == 0.0                                           # testbed_nerf.cu:2670-2671
    r.reload_network_from_file(os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json"))
    r.shall_train = True
    scene.train(r, 30)
    assert np.isfinite(r.loss) and r.nerf.training.measured_batch_size > 0
    assert 0.3 < r.loss / t.loss < 3.0

    # EXR frames: RGBA fp16 on the device, HDR activation
    for k, f in enumerate(meta["frames"]):
        lin = imgs[k].astype(np.float32) / 255.0
        _write_exr(os.path.join(base, "hdr_%d.exr" % k), np.ascontiguousarray(lin), 3, 1)
        f["file_path"] = "hdr_%d" % k       # extension-less: .png missing -> .exr (nerf_loader.cu:562-570)
        f.pop("depth_path")
    json.dump(meta, open(path, "w"))
    e = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    e.load_training_data(path)
    assert e.nerf.training.is_hdr and e.nerf.rgb_activation == pyngp.NerfActivation.Exponential
    assert e.nerf.training.get_image_metadata(0)["image_data_type"] == 2
    np.testing.assert_array_equal(e.nerf.training.get_image_pixels(0), (imgs[0].astype(np.float32) / 255.0).astype(np.float16))
    e.reload_network_from_file(os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json"))
    e.shall_train = True
    scene.train(e, 20)
    assert np.isfinite(e.loss) and e.nerf.training.measured_batch_size > 0
