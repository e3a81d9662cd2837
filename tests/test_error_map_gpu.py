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
