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

# IEC 61966-2-1 transfer function, knees on the encoded / linear side
_KNEE_ENCODED, _KNEE_LINEAR = 0.04045, 0.0031308
_REC709_LUMA = (0.2126, 0.7152, 0.0722)
_GAUSS5 = np.array([0.120078, 0.233881, 0.292082, 0.233881, 0.120078])   # the 5-tap blur of the SSIM windows
_SSIM_C1, _SSIM_C2 = 0.01 ** 2, 0.03 ** 2


def mse2psnr(x):
    return -10.0 * np.log(x) / np.log(10.0)


def srgb_to_linear(img):
    above = img > _KNEE_ENCODED
    return np.where(above, np.power((img + 0.055) / 1.055, 2.4), img / 12.92)


def linear_to_srgb(img):
    above = img > _KNEE_LINEAR
    return np.where(above, 1.055 * (img ** (1.0 / 2.4)) - 0.055, 12.92 * img)


def luminance(rgb):
    r, g, b = (w * rgb[:, :, c] for c, w in enumerate(_REC709_LUMA))
    return r + g + b


def _window(x):
    """separable 5-tap Gaussian, rows then columns"""
    return convolve1d(convolve1d(x, _GAUSS5, axis=0), _GAUSS5, axis=1)


def SSIM(a, b):
    """per-pixel structural similarity of the luminance of two RGB images: (2 mu_a mu_b + c1)(2 cov + c2) / ((mu_a^2 + mu_b^2 + c1)(var_a + var_b + c2))"""
    ya, yb = luminance(a), luminance(b)
    mu_a, mu_b = _window(ya), _window(yb)
    var_a = _window(ya * ya) - mu_a ** 2
    var_b = _window(yb * yb) - mu_b ** 2
    cov = _window(ya * yb) - mu_a * mu_b
    mean_term = (2.0 * mu_a * mu_b + _SSIM_C1) / (mu_a * mu_a + mu_b * mu_b + _SSIM_C1)
    structure_term = (2.0 * cov + _SSIM_C2) / (var_a + var_b + _SSIM_C2)
    return mean_term * structure_term


def _clip01(x):
    return np.clip(x, 0.0, 1.0)


# per-pixel error maps by name (image first, reference second)
_ERROR_MAPS = {
    "MAE": lambda i, r: np.abs(i - r),
    "MAPE": lambda i, r: np.abs(i - r) / (1e-2 + r),
    "SMAPE": lambda i, r: np.abs(i - r) / (1e-2 + (r + i) / 2.0),
    "MSE": lambda i, r: (i - r) ** 2,
    "MScE": lambda i, r: (_clip01(i) - _clip01(r)) ** 2,
    "MRSE": lambda i, r: (i - r) ** 2 / (1e-2 + r ** 2),
    "MRScE": lambda i, r: (np.clip(i, 0, 100) - np.clip(r, 0, 100)) ** 2 / (1e-2 + np.clip(r, 0, 100) ** 2),
    "SSIM": lambda i, r: SSIM(_clip01(i), _clip01(r)),
}


def compute_error_img(metric, img, ref):
    if metric not in _ERROR_MAPS:
        raise ValueError("Unknown metric: %s." % metric)
    sane = np.array(img, copy=True)
    sane[~np.isfinite(sane)] = 0          # non-finite pixels count as black ...
    sane = np.maximum(sane, 0.0)          # ... and negative ones too
    return _ERROR_MAPS[metric](sane, ref)


def compute_error(metric, img, ref):
    per_pixel = compute_error_img(metric, img, ref)
    per_pixel[~np.isfinite(per_pixel)] = 0
    if per_pixel.ndim == 3:
        per_pixel = np.mean(per_pixel, axis=2)
    return np.mean(per_pixel)


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
