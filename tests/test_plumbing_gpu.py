"""GPU tests of the plumbing configurations through `pyngp`: P1 image fitting (src/testbed_image.cu) and P2 SDF fitting on
provided samples (training step of src/testbed_sdf.cu).  The first step is checked against the oracle end to end (init -> batch -> forward -> loss),
the rest by convergence."""
import ctypes
import os
import struct
import sys

import numpy as np
import pytest

import capi
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]
pytestmark = pytest.mark.gpu
CFG = os.path.join(ROOT, "blender-ngp_amd", "configs")


def _test_image(w=256, h=192):
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    u, v = x / w, y / h
    img = np.zeros((h, w, 4), np.float32)
    img[..., 0] = 0.5 + 0.4 * np.sin(9 * u) * np.cos(7 * v)
    img[..., 1] = ((u - 0.5) ** 2 + (v - 0.5) ** 2 < 0.09) * 0.8 + 0.1
    img[..., 2] = u * v
    img[..., 3] = 1.0
    return img


def test_image_first_step_matches_oracle_and_converges(oracle, ngp, cuda, tmp_path):
    import pyngp
    img = _test_image()
    h, w = img.shape[:2]
    tb = pyngp.Testbed(pyngp.TestbedMode.Image)
    tb.set_image_data(img)
    tb.reload_network_from_file(os.path.join(CFG, "image", "base.json"))
    assert tb.image_resolution == [w, h] and tb.n_params() == 7168 + tb.n_encoding_params()
    B = 1 << 14
    # ---- oracle restatement of step 0: parameters, batch, forward, L2 loss
    desc = np.zeros(1, capi.NET_DESC)
    pls = float(np.exp(np.log(np.float32(max(w, h) / 2.0) / np.float32(16)) / np.float32(15)))
    capi.check(ngp.ngp_hip_gridmlp_make_desc_host(2, 16, 24, 16, H.f32(pls), desc.ctypes.data))
    n_params = ngp.ngp_hip_gridmlp_n_params_host(desc.ctypes.data)
    assert n_params == tb.n_params()
    p32 = np.zeros(n_params, np.float32)
    oracle.orc_gridmlp_init_params(desc.ctypes.data, 1337, p32.ctypes.data)
    p16 = p32.astype(np.float16)
    st, inc = H.pcg32_state(1337)
    xy = np.zeros(2 * B, np.float32)
    oracle.orc_generate_random_uniform(st, inc, 2 * B, xy.ctypes.data)
    oracle.orc_image_stratify2(B, 14, xy.ctypes.data)
    tgt = np.zeros((B, 3), np.float32)
    res = np.array([w, h], np.int32)
    oracle.orc_image_eval_and_snap(B, img.ctypes.data, 3, xy.ctypes.data, res.ctypes.data, tgt.ctypes.data, 3, 0, 0)
    pred = np.zeros((B, 4), np.uint16)
    oracle.orc_gridmlp_inference(2, desc.ctypes.data, p16.view(np.uint16).ctypes.data, xy.ctypes.data, 2, B, pred.ctypes.data, 4)
    vals, grad = np.zeros((B, 3), np.float32), np.zeros((B, 4), np.uint16)
    oracle.orc_tcnn_loss_and_gradient(0, B, 3, H.f32(128.0), pred.ctypes.data, 4, tgt.ctypes.data, vals.ctypes.data, grad.ctypes.data, 4)
    want = float(vals.sum(dtype=np.float64))
    tb.shall_train = True
    tb.train(B)
    assert tb.training_step == 1
    assert tb.loss == pytest.approx(want, rel=2e-3)            # the fresh network outputs ~0: the loss is the mean square of the sRGB targets
    assert want > 0.05
    # ---- convergence
    for _ in range(400):
        tb.train(B)
    assert tb.loss < 0.02 * want
    mse = tb.compute_image_mse(False)
    assert mse < 1.5e-3 and tb.compute_image_mse(True) < 2e-3
    # predictions at pixel centres reproduce the sRGB image
    ys, xs = np.mgrid[0:h:7, 0:w:5]
    pos = np.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], 1).astype(np.float32)
    out = tb.gridmlp_inference(pos)[:, :3]
    lin = img[ys.reshape(-1), xs.reshape(-1), :3]
    srgb = np.where(lin < 0.0031308, 12.92 * lin, 1.055 * np.power(np.maximum(lin, 1e-8), 0.41666) - 0.055)
    assert np.mean((out - srgb) ** 2) < 3e-3
    # ---- render: with scale 1 the frame shows exactly the image area (shade converts sRGB predictions back to linear)
    tb.scale = 1.0
    tb.snap_to_pixel_centers = True
    tb.background_color = [0.0, 0.0, 0.0, 1.0]
    frame = tb.render(w, h, 1, True)
    assert frame.shape == (h, w, 4) and np.allclose(frame[..., 3], 1.0)
    assert np.mean((frame[..., :3] - img[..., :3]) ** 2) < 4e-3
    # ---- the .bin container (scripts/common.py:165-171): int32 h, int32 w, fp16 RGBA
    p = str(tmp_path / "img.bin")
    with open(p, "wb") as f:
        f.write(struct.pack("ii", h, w))
        f.write(img.astype(np.float16).tobytes())
    t2 = pyngp.Testbed(pyngp.TestbedMode.Image)
    t2.load_training_data(p)
    assert t2.image_resolution == [w, h]
    t2.reload_network_from_file(os.path.join(CFG, "image", "base.json"))
    t2.train(B)
    assert t2.loss == pytest.approx(want, rel=5e-3)            # same data (fp16-rounded), same seed
    with pytest.raises(RuntimeError):
        t2.load_training_data(str(tmp_path / "img.exr"))


def test_sdf_training_on_provided_samples(cuda):
    import pyngp
    rs = np.random.RandomState(0)
    n = 1 << 17
    pts = rs.rand(n, 3).astype(np.float32)
    dist = (np.linalg.norm(pts - 0.5, axis=1) - 0.3).astype(np.float32)
    tb = pyngp.Testbed(pyngp.TestbedMode.Sdf)
    tb.override_sdf_training_data(pts, dist)
    tb.reload_network_from_file(os.path.join(CFG, "sdf", "base.json"))
    assert tb.n_params() == 7168 + tb.n_encoding_params() and tb.n_encoding_params() > 2 * 11 * (1 << 19)   # 11 hashed levels of 2^19 entries
    B = 1 << 14
    tb.shall_train = True
    tb.train(B)
    first = tb.loss                                             # MAPE of a ~zero prediction: mean |t| / (|t| + 0.01), close to 1
    assert 0.7 < first < 1.05
    for _ in range(700):
        tb.train(B)
    assert tb.training_step == 701 and tb.loss < 0.35 * first
    q = rs.rand(4096, 3).astype(np.float32)
    pred = tb.gridmlp_inference(q)[:, 0]
    truth = np.linalg.norm(q - 0.5, axis=1) - 0.3
    assert np.corrcoef(pred, truth)[0, 1] > 0.97
    with pytest.raises(RuntimeError):
        tb.render(32, 32, 1, True)
    with pytest.raises(RuntimeError):
        pyngp.Testbed(pyngp.TestbedMode.Volume)
