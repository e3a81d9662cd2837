"""Image metrics of the evaluation harness (row H1): counterpart of the reference's scripts/common.py helpers.

  mse2psnr          scripts/common.py:51
  srgb_to_linear    scripts/common.py:141-143
  linear_to_srgb    scripts/common.py:145-147
  luminance / SSIM  scripts/common.py:188-208
  compute_error     scripts/common.py:232-271 (MAE, MAPE, SMAPE, MSE, MScE, MRSE, MRScE, SSIM)

Pinned by tests/golden/common_py_metrics.npz, captured from the reference module itself (tests/golden/make_common_py_fixtures.py).
"""
import numpy as np
from scipy.ndimage import convolve1d


def mse2psnr(x):
    return -10.0 * np.log(x) / np.log(10.0)


def srgb_to_linear(img):
    limit = 0.04045
    return np.where(img > limit, np.power((img + 0.055) / 1.055, 2.4), img / 12.92)


def linear_to_srgb(img):
    limit = 0.0031308
    return np.where(img > limit, 1.055 * (img ** (1.0 / 2.4)) - 0.055, 12.92 * img)


def luminance(a):
    return 0.2126 * a[:, :, 0] + 0.7152 * a[:, :, 1] + 0.0722 * a[:, :, 2]


def SSIM(a, b):
    k = np.array([0.120078, 0.233881, 0.292082, 0.233881, 0.120078])

    def blur(x):
        return convolve1d(convolve1d(x, k, axis=0), k, axis=1)

    a = luminance(a)
    b = luminance(b)
    mA, mB = blur(a), blur(b)
    sA = blur(a * a) - mA ** 2
    sB = blur(b * b) - mB ** 2
    sAB = blur(a * b) - mA * mB
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    p1 = (2.0 * mA * mB + c1) / (mA * mA + mB * mB + c1)
    p2 = (2.0 * sAB + c2) / (sA + sB + c2)
    return p1 * p2


def compute_error_img(metric, img, ref):
    img = np.array(img, copy=True)
    img[np.logical_not(np.isfinite(img))] = 0
    img = np.maximum(img, 0.0)
    if metric == "MAE":
        return np.abs(img - ref)
    if metric == "MAPE":
        return np.abs(img - ref) / (1e-2 + ref)
    if metric == "SMAPE":
        return np.abs(img - ref) / (1e-2 + (ref + img) / 2.0)
    if metric == "MSE":
        return (img - ref) ** 2
    if metric == "MScE":
        return (np.clip(img, 0.0, 1.0) - np.clip(ref, 0.0, 1.0)) ** 2
    if metric == "MRSE":
        return (img - ref) ** 2 / (1e-2 + ref ** 2)
    if metric == "MRScE":
        i, r = np.clip(img, 0, 100), np.clip(ref, 0, 100)
        return (i - r) ** 2 / (1e-2 + r ** 2)
    if metric == "SSIM":
        return SSIM(np.clip(img, 0.0, 1.0), np.clip(ref, 0.0, 1.0))
    raise ValueError("Unknown metric: %s." % metric)


def compute_error(metric, img, ref):
    metric_map = compute_error_img(metric, img, ref)
    metric_map[np.logical_not(np.isfinite(metric_map))] = 0
    if metric_map.ndim == 3:
        metric_map = np.mean(metric_map, axis=2)
    return np.mean(metric_map)


def read_image_rgba8(rgba8):
    """scripts/common.py:149-163 read_image for an 8-bit RGBA PNG already in memory: sRGB -> linear, premultiplied alpha."""
    img = np.asarray(rgba8).astype(np.float32) / 255.0
    if img.shape[2] == 4:
        img[..., 0:3] = srgb_to_linear(img[..., 0:3])
        img[..., 0:3] *= img[..., 3:4]
    else:
        img = srgb_to_linear(img)
    return img


def eval_psnr_ssim(image_rgba_linear, ref_image, color_space_srgb=False, background_color=(0.0, 0.0, 0.0, 1.0)):
    """PSNR/SSIM of one test view exactly as scripts/run.py:258-293 does it.

    image_rgba_linear: testbed.render(w, h, spp, linear=True) output, H x W x 4.
    ref_image: read_image() of the dataset frame (premultiplied linear RGBA).
    Black background, sRGB-clipped comparison (run.py:228-229, 283-290).
    """
    ref = np.array(ref_image, dtype=np.float32, copy=True)
    img = np.array(image_rgba_linear, dtype=np.float32, copy=True)
    if color_space_srgb and ref.shape[2] == 4:
        ref[..., :3] = np.divide(ref[..., :3], ref[..., 3:4], out=np.zeros_like(ref[..., :3]), where=ref[..., 3:4] != 0)
        ref[..., :3] = linear_to_srgb(ref[..., :3])
        ref[..., :3] *= ref[..., 3:4]
        ref += (1.0 - ref[..., 3:4]) * np.asarray(background_color, dtype=np.float32)
        ref[..., :3] = srgb_to_linear(ref[..., :3])
    a = np.clip(linear_to_srgb(img[..., :3]), 0.0, 1.0)
    r = np.clip(linear_to_srgb(ref[..., :3]), 0.0, 1.0)
    mse = float(compute_error("MSE", a, r))
    ssim = float(compute_error("SSIM", a, r))
    return mse2psnr(mse), ssim
