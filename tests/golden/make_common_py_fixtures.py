"""Generates tests/golden/common_py_metrics.npz by importing the REFERENCE's scripts/common.py in the build container.

Run once, here: `python tests/golden/make_common_py_fixtures.py`.  The reference never travels to the GPU box; only the
resulting .npz (inputs + expected outputs) is committed.  scripts/common.py needs `imageio` only for file IO, which the
captured helpers do not touch, so an empty stub module stands in for it.
"""
import os
import sys
import types

import numpy as np

sys.modules.setdefault("imageio", types.ModuleType("imageio"))
sys.path.insert(0, "/root/reference/scripts")
import common  # noqa: E402  (the reference's module)

rs = np.random.RandomState(0)
img = rs.rand(32, 32, 3).astype(np.float32)
ref = np.clip(img + 0.01, 0, 1).astype(np.float32)
hdr = (rs.rand(32, 32, 3).astype(np.float32) * 4.0)
ramp = np.array([0, 0.001, 0.0031308, 0.01, 0.04045, 0.25, 0.5, 0.75, 1.0], dtype=np.float32)

out = {
    "img": img, "ref": ref, "hdr": hdr, "ramp": ramp,
    "linear_to_srgb_ramp": common.linear_to_srgb(ramp),
    "srgb_to_linear_ramp": common.srgb_to_linear(ramp),
    "linear_to_srgb_img": common.linear_to_srgb(img),
    "srgb_to_linear_img": common.srgb_to_linear(img),
    "mse2psnr_1e-3": np.float64(common.mse2psnr(1e-3)),
    "mse2psnr_vals": np.array([common.mse2psnr(x) for x in (1.0, 0.1, 3.3e-4, 1e-5)], dtype=np.float64),
    "mse": np.float64(common.compute_error("MSE", img, ref)),
    "mae": np.float64(common.compute_error("MAE", img, ref)),
    "rmse_map_mean": np.float64(common.compute_error("RMSE", img, ref)) if "RMSE" in getattr(common, "ERROR_NAMES", []) else np.float64("nan"),
    "ssim": np.float64(common.compute_error("SSIM", img, ref)),
    "ssim_hdr": np.float64(common.SSIM(hdr, np.clip(hdr * 0.9, 0, None))),
    "luminance": common.luminance(img) if hasattr(common, "luminance") else np.zeros(1),
}
for name in ("MAPE", "SMAPE", "MRSE", "MSE", "MAE"):
    try:
        out["err_" + name] = np.float64(common.compute_error(name, img, ref))
    except Exception as e:  # metric not present in this fork
        print("skip", name, e)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "common_py_metrics.npz"), **out)
for k, v in out.items():
    if np.ndim(v) == 0:
        print(k, float(v))
