"""BASELINE config #2's own data: the fox photographs of the reference tree (data/nerf/fox: transforms.json + 50 portrait .jpg frames, 18 MB), staged in
the build container where /root/reference exists:

    tests/golden/_generated/fox/transforms.json, images/*.jpg     NOT committed (git-ignored); travels to the GPU box with the other built artefacts

Data only — photographs and their camera file, byte for byte; no reference source.  tests/test_baseline_configs_gpu.py loads the staged transforms.json
through the product's own loader (host/nerf_loader.cpp, host/jpeg_reader.cpp).  usage: python tests/golden/make_fox_fixture.py"""
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/data/nerf/fox"
DST = os.path.join(HERE, "_generated", "fox")


def main():
    os.makedirs(os.path.join(DST, "images"), exist_ok=True)
    n = 0
    for rel in ["transforms.json"] + [os.path.join("images", f) for f in sorted(os.listdir(os.path.join(SRC, "images")))]:
        s, d = os.path.join(SRC, rel), os.path.join(DST, rel)
        if not os.path.exists(d) or os.path.getsize(d) != os.path.getsize(s):
            shutil.copyfile(s, d)
            n += 1
    return n


if __name__ == "__main__":
    if os.path.isdir(SRC):
        print("fox fixture: %d files copied to %s" % (main(), DST))
