"""GPU, end to end through `pyngp`: the stock renderer's remaining modes and the two trainable 2-D buffers.

Normals / EncodingVis / Slice / Distortion render modes, render_masks, glow, quilting (src/testbed_nerf.cu:2354-2500) on a briefly trained scene;
the environment map loaded from transforms.json (`envmap` key), rendered behind the scene and trained (train_envmap); the lens-distortion map
trained back towards zero (optimize_distortion).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trained(cuda):
    import scene
    ds = scene.make_dataset(n_train=12, n_test=1, res=64, device=cuda)
    t = scene.build_testbed(ds)
    scene.train(t, 250)
    t.shall_train = False
    t.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
    t.background_color = [0.0, 0.0, 0.0, 0.0]
    return t, ds


def test_normals_mode(trained):
    import pyngp
    t, ds = trained
    t.render_mode = pyngp.RenderMode.Shade
    shade = t.render(48, 48, 1, True)
    t.render_mode = pyngp.RenderMode.Normals
    nrm = t.render(48, 48, 1, True)
    t.render_mode = pyngp.RenderMode.Shade
    assert np.isfinite(nrm).all()
    np.testing.assert_allclose(nrm[..., 3], shade[..., 3], atol=2e-3)          # same geometry: the input-gradient pass leaves dt alone at aabb_scale 1 (warped dt = 0)
    hit = nrm[..., 3] > 0.97
    assert hit.sum() > 100
    n = nrm[hit][:, :3] / nrm[hit][:, 3:4] * 2.0 - 1.0                          # (0.5 n + 0.5) * alpha (shade_kernel_nerf, :1764-1767)
    ln = np.linalg.norm(n, axis=1)
    assert abs(np.median(ln) - 1.0) < 2e-2 and (np.abs(ln - 1.0) < 5e-2).mean() > 0.9
    # normals of visible surfaces face the camera
    cam = np.asarray(t.camera_matrix)
    view_dir = cam[:, 2]
    assert (n @ view_dir < 0.2).mean() > 0.85


def test_encoding_vis_and_slice_modes(trained):
    import pyngp
    t, ds = trained
    shade = t.render(48, 48, 1, True)
    t.visualized_layer, t.visualized_dimension = 1, 5                           # a density-network neuron
    vis = t.render(48, 48, 1, True)
    assert np.isfinite(vis).all() and np.abs(vis[..., :3] - shade[..., :3]).max() > 5e-2
    assert vis[..., 0].max() == 0.0 and vis[..., 2].max() == 0.0 and vis[..., 1].max() > 0   # post-ReLU: no negative part (red), blue is always 0
    t.visualized_layer, t.visualized_dimension = 0, 31                          # an encoding feature: signed
    vis0 = t.render(48, 48, 1, True)
    assert vis0[..., 0].max() > 0 and vis0[..., 1].max() > 0 and vis0[..., 2].max() == 0.0
    with pytest.raises(RuntimeError):
        t.visualized_layer, t.visualized_dimension = 0, 32
        t.render(48, 48, 1, True)
    t.visualized_layer, t.visualized_dimension = 0, -1
    # Slice: the network on the plane slice_plane_z + scale in front of the camera; every pixel of the frame is shaded
    t.render_mode = pyngp.RenderMode.Slice
    sl = t.render(48, 48, 1, True)
    assert np.isfinite(sl).all() and sl[..., 3].min() >= 0 and sl[..., 3].max() <= 1 and sl[..., 3].max() > 0.05
    t.slice_plane_z = t.slice_plane_z + 0.3
    assert np.abs(t.render(48, 48, 1, True) - sl).max() > 1e-2
    t.slice_plane_z = t.slice_plane_z - 0.3
    t.visualized_layer, t.visualized_dimension = 1, 5
    slv = t.render(48, 48, 1, True)                                             # Slice + neuron visualisation: rgba = (neg, pos, 0, 1)
    assert (slv[..., 3] == 1.0).all() and slv[..., 2].max() == 0.0 and slv[..., 1].max() > 0
    t.visualized_dimension = -1
    t.render_mode = pyngp.RenderMode.Shade
    np.testing.assert_allclose(t.render(48, 48, 1, True), shade, atol=1e-6)


def test_render_masks_glow_quilting_and_distortion_mode(trained):
    import pyngp
    t, ds = trained
    shade = t.render(48, 48, 1, True)
    ident = np.eye(4, dtype=np.float32)
    tr = ident.copy(); tr[:3, 3] = [0.5, 0.5, 0.5]
    # keep only what lies inside a thin slab through the scene centre
    t.render_masks = [pyngp.Mask3D.Box([1.0, 1.0, 0.12], tr, pyngp.MaskMode.Add, 0.0, 1.0)]
    masked = t.render(48, 48, 1, True)
    assert len(t.render_masks) == 2                                             # prepare_nerf_masks put the implicit `All` mask in front (testbed_nerf.cu:2345-2350)
    assert masked[..., 3].sum() < 0.7 * shade[..., 3].sum() and masked[..., 3].sum() > 0.02 * shade[..., 3].sum()
    t.render_masks = []
    np.testing.assert_allclose(t.render(48, 48, 1, True), shade, atol=1e-6)
    t.nerf.glow_mode, t.nerf.glow_y_cutoff = 1, 0.7
    glow = t.render(48, 48, 1, True)
    assert np.isfinite(glow).all() and np.abs(glow - shade).max() > 1e-2
    t.nerf.glow_mode = 0
    # quilting: a 2 x 1 quilt shows the scene twice, from two eyes
    t.quilting_dims = [2, 1]
    t.parallax_shift = [0.05, 0.0, 0.0]
    quilt = t.render(96, 48, 1, True)
    left, right = quilt[:, :48], quilt[:, 48:]
    assert left[..., 3].sum() > 50 and abs(left[..., 3].sum() - right[..., 3].sum()) < 0.2 * left[..., 3].sum() and np.abs(left - right).max() > 1e-2
    t.quilting_dims = [1, 1]
    t.parallax_shift = [0.0, 0.0, 0.0]
    # Distortion mode: the (untrained, zero) map paints black with alpha 1 wherever a ray meets the box
    t.render_mode = pyngp.RenderMode.Distortion
    dist = t.render(48, 48, 1, True)
    assert (dist[..., 3] > 0.99).sum() > 500 and np.abs(dist[..., :3]).max() < 1e-6
    m = np.zeros((32, 32, 2), np.float32); m[..., 0] = 0.01
    t.set_distortion_map(m)
    dist2 = t.render(48, 48, 1, True)
    on = dist2[..., 3] > 0.99
    np.testing.assert_allclose(dist2[on][:, :3], np.tile([0.0, 0.5, 0.5], (on.sum(), 1)), atol=1e-4)   # hsv(0.5, 1, |0.01 * 50|)
    t.set_distortion_map(np.zeros((32, 32, 2), np.float32))
    t.render_mode = pyngp.RenderMode.Shade


def test_envmap_from_transforms_json_renders_and_trains(cuda, tmp_path):
    """`envmap` key -> dataset.envmap_data -> the envmap TrainableBuffer: shown behind the scene by the renderer, composited behind every training ray, trained
    from the rays that see the background (train_envmap)."""
    import pyngp  # noqa: F401
    import scene
    from PIL import Image
    ds = scene.make_dataset(n_train=12, n_test=1, res=64, device=cuda)
    d = str(tmp_path)
    path = scene.write_dataset(ds, d)
    Image.fromarray(np.zeros((16, 32, 4), np.uint8), "RGBA").save(os.path.join(d, "env.png"))
    meta = json.load(open(path)); meta["envmap"] = "env.png"
    json.dump(meta, open(path, "w"))
    t = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    t.load_training_data(path)
    t.reload_network_from_file(os.path.join(os.path.dirname(scene.__file__), "configs", "nerf", "base.json"))
    assert list(t.nerf.training.dataset.envmap_resolution) == [32, 16]
    env0 = t.get_envmap()
    assert env0.shape == (16, 32, 4) and np.abs(env0).max() == 0
    # stage 1: the scene itself, from the transparent-background images (random background colours keep the empty space empty)
    t.shall_train = True
    for _ in range(300):
        t.frame()
    assert np.abs(t.get_envmap()).max() == 0                                     # the map is composited (all zeros: no effect) but does not train by itself
    # stage 2: the same views, now opaque on a constant backdrop, with the network frozen: the only thing that can explain that colour is the environment map
    backdrop = np.array([60, 140, 220], np.uint8)
    for i, im in enumerate(ds["train_images"]):
        im = np.asarray(im).astype(np.float32)
        a = im[..., 3:4] / 255.0
        out = np.concatenate([im[..., :3] * a + backdrop.astype(np.float32) * (1 - a), np.full_like(a, 255.0)], -1)
        t.nerf.training.set_image_rgba8(i, out.round().astype(np.uint8))
    t.shall_train_network = False
    t.shall_train_encoding = False
    t.nerf.training.random_bg_color = False
    t.background_color = [0.0, 0.0, 0.0, 1.0]
    t.nerf.training.train_envmap = True
    for _ in range(400):
        t.frame()
    env = t.get_envmap()
    lin = lambda c: np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    want = lin(backdrop / 255.0)
    seen = env[..., :3].sum(-1) > 0.5 * want.sum()                               # texels in the middle of what the training rays looked at
    assert seen.sum() > 30 and np.abs(env[..., 3]).max() == 0                    # alpha gets no gradient
    assert np.abs(np.median(env[seen][:, :3], axis=0) - want).max() < 0.08
    # the renderer shows the map behind the scene: from a TRAINING pose (directions the map was trained on) the pixels the scene leaves transparent carry the backdrop colour
    t.shall_train = False
    t.background_color = [0.0, 0.0, 0.0, 0.0]
    t.set_nerf_camera_matrix(ds["train_poses"][0][:3, :])
    img = t.render(64, 64, 1, True)
    bg = img[..., 3] < 0.05                                                      # (the map's alpha stays 0: these pixels are colour without coverage)
    assert bg.sum() > 500
    got = np.median(img[bg][:, :3], axis=0)                                      # (the renderer reads the Ema copy of the map, decay 0.99: it trails the trained values)
    assert np.abs(got - want).max() < 0.2 and got[2] > 2 * got[1] > 4 * got[0]
    # without train_envmap the map stays what it is
    t.shall_train = True
    t.nerf.training.train_envmap = False
    for _ in range(20):
        t.frame()
    np.testing.assert_array_equal(t.get_envmap(), env)


def test_optimize_distortion_pulls_a_wrong_map_back(cuda):
    """optimize_distortion (testbed_nerf.cu:1671-1685, 3085-3092): pinhole images, a distortion map seeded with a constant offset -> the camera-gradient kernel's
    image-plane gradients move the map back towards zero, one Adam step (lr 1e-4) every n_steps_between_cam_updates."""
    import scene
    ds = scene.make_dataset(n_train=16, n_test=1, res=96, device=cuda)
    t = scene.build_testbed(ds)
    scene.train(t, 400)
    assert np.abs(t.get_distortion_map()).max() == 0
    seed = np.zeros((32, 32, 2), np.float32); seed[..., 0] = 0.004
    t.set_distortion_map(seed)
    t.nerf.training.optimize_distortion = True
    scene.train(t, 400 + 16 * 24)                                                # (scene.train runs up to an absolute step count)
    m = t.get_distortion_map()
    inner = m[6:26, 6:26]                                                        # texels the training pixels weigh on from all sides
    assert np.isfinite(m).all() and np.abs(m - seed).max() > 5e-4                # it trains ...
    assert inner[..., 0].mean() < 0.004 - 6e-4                                   # ... in the right direction: ~1e-4 per update for 24 updates
    assert abs(inner[..., 1].mean()) < 1.5e-3
    t.nerf.training.optimize_distortion = False
    for _ in range(20):
        t.frame()
    np.testing.assert_array_equal(t.get_distortion_map(), m)
