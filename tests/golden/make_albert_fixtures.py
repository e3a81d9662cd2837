"""Fixtures of BASELINE config #1 (configs/image on data/image/albert.exr), made in the build container where /root/reference exists:

  tests/golden/albert_crop_128.npy            committed: the centre 128 x 128 texels of albert.exr as fp16 RGBA (SURVEY.md §8c "Data availability")
  tests/golden/_generated/albert.bin          NOT committed (8 MB; git-ignored, travels to the GPU box with the other built artefacts): the whole image
                                              in the .bin container of scripts/common.py:165-171 / testbed_image.cu:416-434 (int32 h, int32 w, fp16 RGBA)

The image is decoded by this build's own EXR reader (blender-ngp_amd/host/exr_reader.cpp via pyngp.decode_exr); its RGBA mean (0.18612, 0.18612, 0.18612, 1)
equals what the reference's vendored tinyexr returned for the same file (SURVEY.md §8c probe).  usage: python tests/golden/make_albert_fixtures.py [--crop]"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "blender-ngp_amd"))
SRC = "/root/reference/data/image/albert.exr"


def main():
    import pyngp
    img = pyngp.decode_exr(SRC)
    assert img.shape == (1024, 1024, 4)
    assert np.allclose(img.reshape(-1, 4).mean(0), [0.18612, 0.18612, 0.18612, 1.0], atol=2e-5)   # tinyexr's answer for this file
    os.makedirs(os.path.join(HERE, "_generated"), exist_ok=True)
    with open(os.path.join(HERE, "_generated", "albert.bin"), "wb") as f:
        f.write(struct.pack("ii", 1024, 1024))
        f.write(img.astype(np.float16).tobytes())
    if "--crop" in sys.argv:
        np.save(os.path.join(HERE, "albert_crop_128.npy"), img[448:576, 448:576].astype(np.float16))


if __name__ == "__main__":
    if os.path.exists(SRC):
        main()
