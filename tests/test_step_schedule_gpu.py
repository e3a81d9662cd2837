"""GPU tests of the host-side step schedule options that are not in the reference's pyngp (INTEGRATION.md "Host-side options"):
`async_training_steps` (frame() without the per-step stream drain of testbed.cu:2570) must train the same model, and the live
profiling brackets can be limited to named launch groups."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]

pytestmark = pytest.mark.gpu


def _train(cuda, async_steps, n_steps=48):
    import scene
    ds = scene.make_dataset(n_train=12, n_test=1, res=96, device=cuda)    # (enough rays per step for the per-step counts to average out: ADVICE r05)
    tb = scene.build_testbed(ds)
    tb.network_pass = "fused"            # the auto tuner picks the organisation of the pass by timing: pinned, so that both runs execute the same kernels
    tb.async_training_steps = async_steps
    rays, sizes = [], []
    for _ in range(n_steps):
        tb.frame()
        rays.append(tb.nerf.training.rays_per_batch)
        sizes.append(tb.nerf.training.measured_batch_size)
    tb.sync()
    return ds, tb, np.array(rays), np.array(sizes)


def _render(tb, ds):
    tb.background_color = [0.0, 0.0, 0.0, 1.0]
    tb.snap_to_pixel_centers = True
    tb.fov_axis = 0
    tb.fov = ds["camera_angle_x"] * 180 / np.pi
    tb.shall_train = False
    tb.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
    return np.asarray(tb.render(64, 64, 1, True))


def test_async_training_steps_train_the_same_model(cuda):
    ds, a, rays_a, sizes_a = _train(cuda, False)
    _, b, rays_b, sizes_b = _train(cuda, True)
    assert a.training_step == b.training_step == 48
    # the counter feedback (rays_per_batch <- measured sizes) follows the same trajectory.  Not bit for bit even between two runs of the same
    # mode: the compaction assigns batch slots by atomics, so the weight-gradient sums see the samples in a different order, the weights
    # differ in the last bits after one step and with them the number of samples that survive the transmittance cut (~0.1 %)
    assert rays_a[0] == rays_b[0] and sizes_a[0] == sizes_b[0]
    # ... and over the whole run the counter feedback holds both runs at the same batch: the mean compacted batch size agrees to 3 % (six pairs of runs measured:
    # <= 1.1 %).  rays_per_batch is the variable the feedback MOVES to get there — it absorbs how the occupancy grid happened to evolve behind step 16 — and
    # wanders by up to 19 % between two runs of the SAME mode on this scene (2116 vs 2609 seen), so its bar is 30 %; up to the first grid update it is held to 5 % above
    np.testing.assert_allclose(sizes_a[:12], sizes_b[:12], rtol=0.05)
    np.testing.assert_allclose(rays_a[:12], rays_b[:12], rtol=0.05)
    print("async vs sync: mean batch size %.0f / %.0f, mean rays %.0f / %.0f" % (sizes_b[1:].mean(), sizes_a[1:].mean(), rays_b[1:].mean(), rays_a[1:].mean()))
    assert abs(sizes_a[16:].mean() - sizes_b[16:].mean()) < 0.03 * sizes_a[16:].mean()
    assert abs(rays_a[1:].mean() - rays_b[1:].mean()) < 0.30 * rays_a[1:].mean()
    assert np.isfinite(a.loss) and np.isfinite(b.loss) and abs(a.loss - b.loss) < 0.5 * max(a.loss, b.loss)
    ia, ib = _render(a, ds), _render(b, ds)
    mse = float(np.mean((ia[..., :3] - ib[..., :3]) ** 2))
    assert mse < 5e-3, mse   # two runs of the same short training: the same picture up to training noise


def test_profiling_brackets_can_be_limited_to_named_groups(cuda):
    import scene
    ds = scene.make_dataset(n_train=4, n_test=1, res=32, device=cuda)
    tb = scene.build_testbed(ds)
    tb.set_profiling(True)
    tb.reset_profile()
    for _ in range(3):
        tb.frame()
    p = tb.profile()
    assert p["nerf_backward"]["launches"] == 3 and p["optimizer_step"]["launches"] == 3 and p["compute_loss"]["launches"] == 3
    assert p["nerf_backward"]["ms"] > 0 and p["nerf_backward"]["units"] == 3 * tb.training_batch_size
    tb.set_profiling(True, ["nerf_backward"])
    tb.reset_profile()
    tb.async_training_steps = True   # brackets are folded in when their events have finished, or by profile()
    for _ in range(5):
        tb.frame()
    p = tb.profile()
    assert p["nerf_backward"]["launches"] == 5
    assert all(p[k]["launches"] == 0 for k in p if k != "nerf_backward")
    with pytest.raises(Exception):
        tb.set_profiling(True, ["no_such_group"])
    tb.set_profiling(False)


def test_grid_update_samples_generated_ahead_are_the_in_order_ones(cuda):
    """From step 256 on the step before an occupancy-grid update has no march to run ahead; stream B generates the update's sample positions instead
    (Testbed::maybe_prefetch_grid_samples).  They depend on the density grid and its generator state only: what stream B wrote must be bit for bit what the update
    would generate in stream order, every such update must take them, and a Testbed with the run-ahead switched off generates them itself."""
    import scene
    ds = scene.make_dataset(n_train=8, n_test=1, res=64, device=cuda)
    tb = scene.build_testbed(ds)
    tb.async_training_steps = True
    while tb.training_step < 255:
        tb.frame()
    assert tb.grid_prefetch_hits == 0 and not tb.debug_grid_update_samples()["pending"]   # before step 256 the updates sample the whole grid, every step at first
    tb.frame()                                                                            # step 255: the samples of the update before step 256 go ahead
    ahead = tb.debug_grid_update_samples()
    assert ahead["pending"] and ahead["step"] == 256
    again = tb.debug_grid_update_samples(regenerate=True)
    assert again["pending"]
    assert ahead["positions"].size == again["positions"].size > 0
    np.testing.assert_array_equal(ahead["indices"], again["indices"])                    # (round 6: Morton order of the cells — still a pure function of the generator state)
    np.testing.assert_array_equal(ahead["positions"].view(np.uint32), again["positions"].view(np.uint32))
    tb.frame()                                                                            # step 256 with its update
    assert tb.grid_prefetch_hits == 1 and not tb.debug_grid_update_samples()["pending"]
    while tb.training_step < 305:
        tb.frame()
    tb.sync()
    assert tb.grid_prefetch_hits == 4 and np.isfinite(tb.loss)                           # 256, 272, 288, 304
    # anything that touches the training inputs voids a pending set: the update then generates its own
    while tb.training_step < 320:
        tb.frame()
    assert tb.debug_grid_update_samples()["pending"]
    tb.nerf.training.n_images_for_training = 7
    tb.frame()
    tb.sync()
    assert tb.grid_prefetch_hits == 4 and tb.training_step == 321 and np.isfinite(tb.loss)

    off = scene.build_testbed(ds)
    off.async_training_steps = True
    off.prefetch_samples = False
    while off.training_step < 290:
        off.frame()
    off.sync()
    assert off.grid_prefetch_hits == 0 and np.isfinite(off.loss)


@pytest.mark.parametrize("side", [True, False])
def test_ema_stage_on_the_second_stream_is_the_ema_of_the_weights(cuda, tmp_path, side):
    """`ema_on_side_stream` (opt-in): the optimizer step's Ema stage — what render() and snapshots read — runs on stream B beside the next step's network pass instead of on
    the training chain.  After a run of asynchronous steps the snapshot's Ema state must be exactly what tcnn's EmaOptimizer makes of the previous average and the NEW weights
    ((e * decay * (1 - decay^(t-1)) + w * (1 - decay)) / (1 - decay^t): rtol 2e-6 for numpy's float32 pow against powf), the fp16 inference weights its rounding (bit for bit),
    and a render right behind frame() reads them without the caller synchronising anything.  Run in both modes: the check does not depend on where the stage ran."""
    import msgpack
    import scene
    ds = scene.make_dataset(n_train=8, n_test=1, res=64, device=cuda)
    tb = scene.build_testbed(ds)
    tb.async_training_steps = True
    tb.ema_on_side_stream = side
    for _ in range(20):
        tb.frame()

    def state(name):
        path = str(tmp_path / name)
        tb.save_snapshot(path, True)
        m = msgpack.unpackb(open(path, "rb").read(), raw=False)["snapshot"]
        ema = m["optimizer"]
        assert ema["otype"] == "Ema"
        return np.frombuffer(ema["ema_binary"], np.float32).copy(), np.frombuffer(m["params_binary"], np.uint16).copy(), int(ema["ema_step"])
    e1, i1, t1 = state("a.msgpack")
    assert t1 == 20
    np.testing.assert_array_equal(i1, e1.astype(np.float16).view(np.uint16))
    tb.frame()
    img = _render(tb, ds)                                   # reads the inference weights of step 21: ordered behind the Ema stage by render() itself
    assert np.isfinite(img).all() and img[..., :3].max() > 0.05
    e2, i2, t2 = state("b.msgpack")
    w2 = np.asarray(tb.debug_params("training")).view(np.float16).astype(np.float32)
    assert t2 == 21 and np.any(e2 != e1)
    d = np.float32(0.95)
    want = (e1 * d * (np.float32(1) - np.float32(d) ** np.float32(t2 - 1)) + w2 * (np.float32(1) - d)) * (np.float32(1) / (np.float32(1) - np.float32(d) ** np.float32(t2)))
    np.testing.assert_allclose(e2, want, rtol=2e-6, atol=1e-9)
    np.testing.assert_array_equal(i2, e2.astype(np.float16).view(np.uint16))
    tb.shall_train = True
    for _ in range(3):                                      # and training goes on (the next Adam stage waits for the Ema stage that reads the weights it overwrites)
        tb.frame()
    tb.sync()
    assert tb.training_step == 24 and np.isfinite(tb.loss)
