"""T11: error map -> CDFs inside Testbed::train (testbed_nerf.cu:2933-2939, 2971-3023) and training with the importance-sampling switches."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_testbed_builds_cdfs_and_trains_with_importance_sampling(cuda, oracle):
    import scene
    ds = scene.make_dataset(n_train=6, n_test=1, res=48, device=cuda)
    t = scene.build_testbed(ds)
    tr = t.nerf.training
    tr.n_steps_between_error_map_updates = 6
    assert not tr.is_cdf_valid
    scene.train(t, 6)
    # the step that reached the update interval built the CDFs from the error map accumulated over those steps
    assert tr.is_cdf_valid and tr.n_steps_since_error_map_update == 0 and tr.n_steps_between_error_map_updates == 9   # * 1.5 (3022)
    em = tr.get_error_map()
    n_img, h, w = em.shape
    assert n_img == 6 and em.sum() > 0 and (em >= 0).all()
    x, y, im, pmf = tr.get_error_map_cdfs()
    rx, ry, rsum = np.zeros_like(em), np.zeros((n_img, h), np.float32), np.zeros(n_img, np.float32)
    oracle.orc_construct_cdf_2d(n_img, h, w, em.ctypes.data, rx.ctypes.data, ry.ctypes.data)
    oracle.orc_construct_cdf_1d(n_img, h, ry.ctypes.data, rsum.ctypes.data)
    rp, rc = np.zeros(n_img, np.float32), np.zeros(n_img, np.float32)
    oracle.orc_image_cdf_host(n_img, rsum.ctypes.data, rp.ctypes.data, rc.ctypes.data)
    np.testing.assert_array_equal(x, rx)
    np.testing.assert_array_equal(y, ry)
    np.testing.assert_array_equal(im, rc)
    np.testing.assert_array_equal(pmf, rp)
    assert abs(float(pmf.sum()) - 1.0) < 1e-5
    # switch both samplers on: training continues on importance-sampled rays (a prefetched march made with the old setting is discarded)
    loss_before = t.loss
    tr.sample_focal_plane_proportional_to_error = True
    tr.sample_image_proportional_to_error = True
    scene.train(t, 46)                    # trains UP TO step 46
    assert t.training_step == 46 and np.isfinite(t.loss) and t.loss < loss_before
    assert tr.measured_batch_size > 0


def test_testbed_depth_supervision(cuda):
    """nerf.training.set_image(frame, img, depth_img, depth_scale) + depth_supervision_lambda (python_api.cu:53-72, 809, 828): the depth
    term pulls the expected termination depth of the training rays towards the supplied depth maps"""
    import scene
    import pyngp
    ds = scene.make_dataset(n_train=6, n_test=1, res=48, device=cuda)
    imgs = [np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x) for x in ds["train_images"]]

    def run(lam):
        t = scene.build_testbed(ds)
        tr = t.nerf.training
        for i, im in enumerate(imgs):
            rgba = im.astype(np.float32) / 255.0 if im.dtype == np.uint8 else im.astype(np.float32)
            rgba = rgba.copy(); rgba[..., :3] = np.where(rgba[..., :3] <= 0.04045, rgba[..., :3] / 12.92, ((rgba[..., :3] + 0.055) / 1.055) ** 2.4) * rgba[..., 3:4]
            tr.set_image(i, rgba, np.full(rgba.shape[:2], 0.25, np.float32), 1.0)     # a (wrong) constant depth of 0.25 for every pixel
        assert tr.depth_loss_type == pyngp.LossType.L1 and tr.depth_supervision_lambda == 0.0
        tr.depth_supervision_lambda = lam
        scene.train(t, 60)
        assert np.isfinite(t.loss)
        t.shall_train = False
        t.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
        return t.render(32, 32, 1, True)

    free, pulled = run(0.0), run(5.0)
    # with a strong pull towards a surface 0.25 in front of every camera the reconstruction differs visibly
    assert np.abs(free - pulled).mean() > 5e-3


def test_testbed_exposure_optimisation(cuda):
    """nerf.training.optimize_exposure (python_api.cu:813): images darkened / brightened by known factors -> the per-image log2 exposures move
    the right way (they are defined up to their mean, which the update removes)"""
    import scene
    ds = scene.make_dataset(n_train=32, n_test=1, res=48, device=cuda)
    gains = np.array([0.5, 2.0] * 16, np.float32)        # exposure of image i = log2(1 / gain): the target is 2^e * pixel
    imgs = []
    for x, gain in zip(ds["train_images"], gains):
        im = np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x)
        rgba = im.astype(np.float32) / 255.0 if im.dtype == np.uint8 else im.astype(np.float32)
        lin = np.where(rgba[..., :3] <= 0.04045, rgba[..., :3] / 12.92, ((rgba[..., :3] + 0.055) / 1.055) ** 2.4)
        out = rgba.copy(); out[..., :3] = np.clip(lin * gain, 0, 1) * rgba[..., 3:4]
        imgs.append(out)
    t = scene.build_testbed(ds)
    tr = t.nerf.training
    for i, im in enumerate(imgs):
        tr.set_image(i, im)
    tr.optimize_exposure = True
    assert np.abs(tr.get_camera_exposures()).max() == 0
    hist = []
    for stop in (64, 128, 256, 512, 800):
        scene.train(t, stop)
        e = tr.get_camera_exposures().mean(axis=1)
        hist.append((stop, float(e[gains < 1].mean() - e[gains > 1].mean())))
    print("dark - bright exposure over training:", hist)
    e = tr.get_camera_exposures().mean(axis=1)
    assert np.isfinite(e).all() and abs(float(e.mean())) < 1e-4                 # renormalised to zero mean
    # every 16 steps Adam moves each exposure by at most the learning rate and the mean is removed; with 32 views the colour head cannot explain
    # a per-image brightness, so the darkened images drift above the brightened ones (towards log2(1/gain) = +1 / -1)
    assert hist[-1][1] > 0.1 and hist[-1][1] > hist[0][1], hist
