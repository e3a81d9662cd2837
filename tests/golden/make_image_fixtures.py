"""Generates tests/golden/images/*: PIZ-compressed OpenEXR files written by the reference's own vendored tinyexr (oracle/_ref/libref_imageio.so, `make -C oracle ref`,
build container only).  The pixel values are closed formulas of (x, y) (exr_fixture_image below), so the tests rebuild the expected arrays instead of storing them.
Run from the repository root:  python tests/golden/make_image_fixtures.py"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

# name -> (width, height, pixel type 1 HALF / 2 FLOAT, kind)
EXR_FIXTURES = {
    "piz_half_37x45.exr": (37, 45, 1, "smooth"),        # two blocks of 32 lines, odd sizes: the wavelet's odd row / column steps
    "piz_float_64x33.exr": (64, 33, 2, "smooth"),       # FLOAT: two 16-bit planes per channel
    "piz_half_600x35_wide_range.exr": (600, 35, 1, "ramp"),   # > 2^14 distinct values in a block: the modulo-2^16 wavelet
}


def exr_fixture_image(w, h, pixel_type, kind):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.float32)
    if kind == "ramp":
        bits = ((xx + 600 * (yy % 32)) % 31000).astype(np.uint16)   # 19200 different half bit patterns per 32-line block
        img[..., 0] = bits.view(np.float16).astype(np.float32)
        img[..., 1] = 0.5
        img[..., 3] = 1.0
    else:
        img[..., 0] = np.sin(xx * 0.1) * np.cos(yy * 0.07) + 1.5
        img[..., 1] = yy * 0.01 + xx * 0.003 + 0.1
        img[..., 2] = (xx * yy) * 0.0005
        img[..., 3] = np.where((xx + yy) % 7 == 0, 0.5, 1.0)
    if pixel_type == 1:
        img = img.astype(np.float16).astype(np.float32)   # exactly representable: the writer's float -> half step cannot round
    return np.ascontiguousarray(img)


if __name__ == "__main__":
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_imageio.so"))
    out = os.path.join(HERE, "images")
    os.makedirs(out, exist_ok=True)
    for name, (w, h, pt, kind) in EXR_FIXTURES.items():
        img = exr_fixture_image(w, h, pt, kind)
        r = ref.ref_save_exr_rgba(os.path.join(out, name).encode(), img.ctypes.data_as(ctypes.c_void_p), w, h, 4, pt)   # 4 = TINYEXR_COMPRESSIONTYPE_PIZ
        assert r == 0, name
        print(name, os.path.getsize(os.path.join(out, name)), "bytes")
