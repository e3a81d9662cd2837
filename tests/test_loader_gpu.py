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
