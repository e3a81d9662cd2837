"""CPU tests of the transforms.json loader's host stage (src/nerf_loader.cu:197-747): PNG decode against PIL, JSON keys, frame ordering /
culling, path resolution, intrinsics precedence, the NeRF->NGP matrix convention, RGBA8 fix-ups."""
import json
import os
import sys
import zlib
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]
Image = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def pyngp():
    import torch  # noqa: F401
    import pyngp as m
    return m


def _png_bytes(w, h, color_type, depth, rows, palette=None, trns=None):
    """hand-rolled PNG writer so that every filter type / colour type / bit depth can be produced"""
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    raw = b"".join(rows)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", palette)
    if trns is not None:
        out += chunk(b"tRNS", trns)
    half = len(raw) // 2
    comp = zlib.compress(raw, 6)
    out += chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:])   # split IDAT
    return out + chunk(b"IEND", b"")


def _filter_rows(img_bytes_rows, bpp):
    """apply filter types 0..4 cyclically (PNG spec 9.2) to unfiltered rows"""
    out, prev = [], bytes(len(img_bytes_rows[0]))
    for y, row in enumerate(img_bytes_rows):
        f = y % 5
        cur = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0: pred = 0
            elif f == 1: pred = a
            elif f == 2: pred = b
            elif f == 3: pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[i] = (v - pred) & 0xff
        out.append(bytes([f]) + bytes(cur))
        prev = row
    return out


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA", "P"])
def test_decode_png_matches_pil(pyngp, tmp_path, mode):
    rs = np.random.RandomState(1)
    w, h = 37, 23
    if mode == "P":
        im = Image.fromarray(rs.randint(0, 255, (h, w, 3), dtype=np.uint8)).convert("P", palette=Image.ADAPTIVE, colors=17)
    else:
        ch = {"RGBA": 4, "RGB": 3, "L": 1, "LA": 2}[mode]
        arr = rs.randint(0, 255, (h, w, ch), dtype=np.uint8)
        im = Image.fromarray(arr[..., 0] if ch == 1 else arr, mode)
    path = str(tmp_path / ("t_%s.png" % mode))
    im.save(path)
    got = pyngp.decode_png(path)
    want = np.asarray(im.convert("RGBA"))
    np.testing.assert_array_equal(got, want)


def test_decode_png_all_filters_16bit_and_trns(pyngp, tmp_path):
    rs = np.random.RandomState(2)
    w, h = 19, 15
    rgba = rs.randint(0, 255, (h, w, 4), dtype=np.uint8)
    rows = _filter_rows([bytes(rgba[y].reshape(-1)) for y in range(h)], 4)
    p = tmp_path / "filters.png"; p.write_bytes(_png_bytes(w, h, 6, 8, rows))
    np.testing.assert_array_equal(pyngp.decode_png(str(p)), rgba)
    np.testing.assert_array_equal(np.asarray(Image.open(str(p)).convert("RGBA")), rgba)   # the writer itself is sound
    # 16-bit RGB: the high byte is kept (stb_image's 16 -> 8 conversion)
    rgb16 = rs.randint(0, 65535, (h, w, 3)).astype(">u2")
    rows = _filter_rows([rgb16[y].tobytes() for y in range(h)], 6)
    p = tmp_path / "rgb16.png"; p.write_bytes(_png_bytes(w, h, 2, 16, rows))
    got = pyngp.decode_png(str(p))
    np.testing.assert_array_equal(got[..., :3], (rgb16.astype(np.uint16) >> 8).astype(np.uint8))
    assert (got[..., 3] == 255).all()
    # grey with a tRNS colour key, 2-bit palette
    g = rs.randint(0, 4, (h, w), dtype=np.uint8) * 85
    rows = [b"\x00" + bytes(g[y]) for y in range(h)]
    p = tmp_path / "grey_trns.png"; p.write_bytes(_png_bytes(w, h, 0, 8, rows, trns=struct.pack(">H", 85)))
    got = pyngp.decode_png(str(p))
    np.testing.assert_array_equal(got[..., 0], g)
    np.testing.assert_array_equal(got[..., 3], np.where(g == 85, 0, 255))
    idx = rs.randint(0, 4, (h, w))
    packed = []
    for y in range(h):
        bits = "".join(format(int(v), "02b") for v in idx[y]); bits += "0" * (-len(bits) % 8)
        packed.append(b"\x00" + int(bits, 2).to_bytes(len(bits) // 8, "big"))
    pal = bytes([10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120])
    p = tmp_path / "pal2.png"; p.write_bytes(_png_bytes(w, h, 3, 2, packed, palette=pal, trns=bytes([255, 0, 128])))
    got = pyngp.decode_png(str(p))
    np.testing.assert_array_equal(got[..., :3], np.frombuffer(pal, np.uint8).reshape(4, 3)[idx])
    np.testing.assert_array_equal(got[..., 3], np.array([255, 0, 128, 255], np.uint8)[idx])
    # errors
    bad = tmp_path / "bad.png"; bad.write_bytes(b"\xff\xd8\xff\xe0 jpeg")
    with pytest.raises(RuntimeError, match="JPEG"):
        pyngp.decode_png(str(bad))
    with pytest.raises(RuntimeError, match="Could not open"):
        pyngp.decode_png(str(tmp_path / "missing.png"))


def _write_scene(d, n=5, w=16, h=12, extra=None, frame_extra=None, names=None, with_ext=True):
    rs = np.random.RandomState(7)
    os.makedirs(os.path.join(d, "train"), exist_ok=True)
    frames, images, mats = [], {}, {}
    names = names or ["r_%d" % i for i in range(n)]
    for i, name in enumerate(names):
        img = rs.randint(0, 255, (h, w, 4), dtype=np.uint8)
        Image.fromarray(img, "RGBA").save(os.path.join(d, "train", name + ".png"))
        m = np.eye(4); m[:3, :4] = rs.uniform(-1, 1, (3, 4))
        f = {"file_path": "./train/" + name + (".png" if with_ext else ""), "transform_matrix": m.tolist()}
        if frame_extra:
            f.update(frame_extra(i))
        frames.append(f); images[f["file_path"]] = img; mats[f["file_path"]] = m
    j = {"camera_angle_x": 0.6911112070083618, "frames": frames}
    j.update(extra or {})
    path = os.path.join(d, "transforms_train.json")
    open(path, "w").write("// a comment, like the reference's parser allows\n" + json.dumps(j))
    return path, images, mats


def _expected_ngp(m, scale, offset):
    r = np.array(m[:3, :4], np.float32)
    r[:, 1] *= -1; r[:, 2] *= -1
    r[:, 3] = r[:, 3] * np.float32(scale) + np.asarray(offset, np.float32)
    return r[[1, 2, 0], :]


def test_loader_synthetic_conventions(pyngp, tmp_path):
    d = str(tmp_path)
    names = ["r_10", "r_2", "r_0", "r_1", "r_3"]          # frames are sorted by file_path STRING (nerf_loader.cu:359-361)
    path, images, mats = _write_scene(d, names=names, with_ext=False, extra={"aabb_scale": 4, "scale": 0.5, "offset": [0.1, 0.2, 0.3]})
    out = pyngp.load_nerf_host(path)
    assert out["n_images"] == 5 and out["aabb_scale"] == 4 and out["scale"] == 0.5 and np.allclose(out["offset"], [0.1, 0.2, 0.3])
    assert out["paths"] == sorted("./train/" + n for n in names)            # extension-less paths get .png appended on disk only
    focal = np.float32(0.5) * np.float32(16) / np.tan(np.float32(0.5) * (np.float32(0.6911112070083618) * 180 / np.float32(np.pi)) * np.float32(np.pi) / 180)
    for i, p in enumerate(out["paths"]):
        np.testing.assert_array_equal(out["pixels"][i], images[p])
        np.testing.assert_allclose(out["xforms"][i][0], _expected_ngp(mats[p], 0.5, [0.1, 0.2, 0.3]), rtol=0, atol=1e-7)
        np.testing.assert_array_equal(out["xforms"][i][0], out["xforms"][i][1])
        md = out["metadata"][i]
        assert md["resolution"] == [16, 12] and md["principal_point"] == [0.5, 0.5] and md["lens_mode"] == 0
        assert abs(md["focal_length"][0] - focal) < 1e-3 and md["focal_length"][0] == md["focal_length"][1]
    # a directory loads every json in it (testbed_nerf.cu:2738-2743)
    out2 = pyngp.load_nerf_host(d)
    assert out2["n_images"] == 5


def test_loader_keys_and_overrides(pyngp, tmp_path):
    d = str(tmp_path)
    extra = {"fl_x": 20.0, "fl_y": 21.0, "cx": 9.0, "cy": 5.0, "w": 16, "h": 12, "k1": 0.01, "k2": 0.0, "p1": 0.0, "p2": 0.002, "n_frames": 4,
             "white_transparent": True, "aabb": [[-1, -2, -3], [1, 2, 5]], "render_aabb": [[0.1, 0.2, 0.3], [0.7, 0.8, 0.9]], "up": [0, 0, 1],
             "rolling_shutter": [0.0, 0.0, 1.0], "sharpness_discard_threshold": 0.9}
    sharp = [10.0, 10.0, 1.0, 10.0, 10.0]
    path, images, mats = _write_scene(d, extra=extra, frame_extra=lambda i: {"sharpness": sharp[i], **({"fl_x": 33.0} if i == 1 else {})})
    # make one pixel pure white in the first kept image
    first = "./train/r_0.png"
    img = images[first].copy(); img[0, 0] = [255, 255, 255, 200]
    Image.fromarray(img, "RGBA").save(os.path.join(d, "train", "r_0.png"))
    out = pyngp.load_nerf_host(path)
    # n_frames keeps r_0..r_3; the blurry r_2 is discarded against its neighbourhood mean
    assert out["paths"] == ["./train/r_0.png", "./train/r_1.png", "./train/r_3.png"]
    assert out["pixels"][0][0, 0].tolist() == [255, 255, 255, 0]          # white -> transparent
    assert out["pixels"][0][1, 1].tolist() == images[first][1, 1].tolist()
    assert out["metadata"][0]["focal_length"] == [20.0, 21.0] and out["metadata"][1]["focal_length"] == [33.0, 33.0]   # a per-frame fl_x alone resets both axes (read_focal_length, 291-295)
    assert out["metadata"][0]["lens_mode"] == 1 and np.allclose(out["metadata"][0]["lens_params"][:4], [0.01, 0.0, 0.0, 0.002])
    assert np.allclose(out["metadata"][0]["principal_point"], [9.0 / 16, 5.0 / 12])
    assert out["metadata"][0]["rolling_shutter"] == [0.0, 0.0, 1.0, 0.0]
    # "aabb": longest side 8 -> scale 1/8, centre mapped to 0.5
    assert abs(out["scale"] - 0.125) < 1e-7 and np.allclose(out["offset"], [0.5, 0.5, 0.5 - 1.0 * 0.125])
    assert np.allclose(out["render_aabb"], [0.1, 0.2, 0.3, 0.7, 0.8, 0.9]) and out["up"] == [0.0, 1.0, 0.0]     # up axes permuted yzx


def test_loader_errors(pyngp, tmp_path):
    d = str(tmp_path)
    p = os.path.join(d, "empty.json"); open(p, "w").write(json.dumps({"frames": []}))
    with pytest.raises(ValueError, match="No training images"):
        pyngp.load_nerf_host(p)
    p2 = os.path.join(d, "nofov.json")
    Image.fromarray(np.zeros((4, 4, 4), np.uint8), "RGBA").save(os.path.join(d, "a.png"))
    open(p2, "w").write(json.dumps({"frames": [{"file_path": "a.png", "transform_matrix": np.eye(4).tolist()}]}))
    with pytest.raises(RuntimeError, match="fov"):
        pyngp.load_nerf_host(p2)
    p3 = os.path.join(d, "missing.json")
    open(p3, "w").write(json.dumps({"camera_angle_x": 0.5, "frames": [{"file_path": "nope", "transform_matrix": np.eye(4).tolist()}]}))
    with pytest.raises(RuntimeError, match="Could not find image file"):
        pyngp.load_nerf_host(p3)
    with pytest.raises(RuntimeError, match="json file or a directory"):
        pyngp.load_nerf_host(os.path.join(d, "a.png"))


def test_loader_jpeg_exr_alpha_depth_and_ray_files(pyngp, tmp_path):
    """The image kinds of nerf_loader.cu:560-668 besides plain PNG: JPEG frames (stb_image decides by content), EXR frames (RGBA fp16, `fix_premult`,
    is_hdr), `<path>.alpha.<ext>` companions (red channel, sRGB -> linear), 16-bit `depth_path` images with `integer_depth_scale`,
    rays_<name>.dat per-pixel rays converted to the NGP frame, and the switches `enable_depth_loading` / `enable_ray_loading`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_image_io_cpu import _write_exr
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "im"))
    rs = np.random.RandomState(4)
    w, h = 24, 16
    rgb = (rs.rand(h, w, 3) * 255).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(os.path.join(d, "im", "a.jpg"), quality=92)
    alpha = (rs.rand(h, w) * 255).astype(np.uint8)
    Image.fromarray(np.stack([alpha] * 3, -1), "RGB").save(os.path.join(d, "im", "a.jpg.alpha.jpg"), quality=100, subsampling=0)
    depth = (rs.rand(h, w) * 65535).astype(np.uint16)
    Image.fromarray(depth).save(os.path.join(d, "im", "a_depth.png"))
    rays = rs.randn(h, w, 6).astype(np.float32)
    rays.tofile(os.path.join(d, "im", "rays_a.dat"))
    exr = rs.rand(h, w, 4).astype(np.float32)
    _write_exr(os.path.join(d, "im", "b.exr"), exr, 3, 1)
    m = np.eye(4).tolist()
    base = {"camera_angle_x": 0.7, "scale": 0.5, "offset": [0.1, 0.2, 0.3], "integer_depth_scale": 0.001, "fix_premult": True,
            "frames": [{"file_path": "im/a.jpg", "depth_path": "im/a_depth.png", "transform_matrix": m}, {"file_path": "im/b", "transform_matrix": m}]}
    p = os.path.join(d, "t.json"); open(p, "w").write(json.dumps(base))
    out = pyngp.load_nerf_host(p)
    assert out["n_images"] == 2 and out["is_hdr"] and out["has_rays"]
    a, b = out["pixels"]
    assert a.dtype == np.uint8 and out["metadata"][0]["image_data_type"] == 1
    np.testing.assert_array_equal(a[..., :3], pyngp.decode_image(os.path.join(d, "im", "a.jpg"))[..., :3])
    al = pyngp.decode_image(os.path.join(d, "im", "a.jpg.alpha.jpg"))[..., 0].astype(np.float32) / np.float32(255)
    lin = np.where(al <= 0.04045, al / np.float32(12.92), ((al + np.float32(0.055)) / np.float32(1.055)) ** np.float32(2.4))
    assert np.abs(a[..., 3].astype(int) - (np.float32(255) * lin).astype(np.uint8).astype(int)).max() <= 1   # (uint8)(255 * srgb_to_linear(red / 255)), powf vs numpy
    np.testing.assert_array_equal(out["depth16"][0], depth)
    assert abs(out["depth_scale"][0] - 0.001) < 1e-9 and out["depth16"][1] is None
    # rays: origin * scale + offset, then (x, y, z) <- (y, z, x) for origin and direction (nerf_loader.h:165-180)
    o = rays[..., :3] * np.float32(0.5) + np.array([0.1, 0.2, 0.3], np.float32)
    np.testing.assert_array_equal(out["rays"][0][..., :3], o[..., [1, 2, 0]])
    np.testing.assert_array_equal(out["rays"][0][..., 3:], rays[..., 3:][..., [1, 2, 0]])
    # EXR: extension-less path falls back to .exr; colour * alpha (fix_premult), fp16
    assert b.dtype == np.float16 and out["metadata"][1]["image_data_type"] == 2
    src = pyngp.decode_exr(os.path.join(d, "im", "b.exr"))
    want = np.concatenate([src[..., :3] * src[..., 3:4], src[..., 3:4]], -1).astype(np.float16)
    np.testing.assert_array_equal(b, want)
    # the two switches
    base.update(enable_depth_loading=False, enable_ray_loading=False)
    open(p, "w").write(json.dumps(base))
    out = pyngp.load_nerf_host(p)
    assert out["depth16"][0] is None and out["rays"][0] is None and not out["has_rays"]
    # `envmap` key (nerf_loader.cu:533-546): an 8-bit file goes through from_rgba32<float> (sRGB decode, premultiplied by alpha), an .exr is taken as it is
    assert out["envmap"] is None and out["envmap_resolution"] == [0, 0]
    base["envmap"] = "env.png"
    open(p, "w").write(json.dumps(base))
    with pytest.raises(RuntimeError, match="Environment map .* does not exist"):
        pyngp.load_nerf_host(p)
    rs = np.random.RandomState(4)
    env8 = rs.randint(0, 256, (6, 12, 4)).astype(np.uint8)
    Image.fromarray(env8, "RGBA").save(os.path.join(d, "env.png"))
    out = pyngp.load_nerf_host(p)
    assert out["envmap_resolution"] == [12, 6] and out["envmap"].shape == (6, 12, 4)
    c = env8[..., :3].astype(np.float32) * np.float32(1 / 255.0)
    a = env8[..., 3:4].astype(np.float32) * np.float32(1 / 255.0)
    lin = np.where(c <= 0.04045, c / np.float32(12.92), ((c + np.float32(0.055)) / np.float32(1.055)) ** np.float32(2.4)).astype(np.float32)
    np.testing.assert_allclose(out["envmap"], np.concatenate([lin * a, a], -1), rtol=2e-6, atol=1e-7)
    base["envmap"] = "im/b.exr"
    open(p, "w").write(json.dumps(base))
    out = pyngp.load_nerf_host(p)
    np.testing.assert_array_equal(out["envmap"], pyngp.decode_exr(os.path.join(d, "im", "b.exr")))


REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_imageio.so")


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libref_imageio.so (the reference's vendored stb_image + tinyexr, `make -C oracle ref`) is built in the build container")
def test_decoders_against_the_references_own_stb_image_and_tinyexr(pyngp, tmp_path):
    """oracle/_ref: the reference's decoders compiled from /root/reference/dependencies.  PNG (8 / 16 bit, grey, grey + alpha, palette) and the 16-bit depth
    read are bit-identical to stb_image; EXR is bit-identical to tinyexr's LoadEXR; JPEG stays within 3 of 255 (this build's inverse DCT is an exact
    float transform rounded once, stb_image's is its fixed-point one), 4:2:2 included (stb_image's last-column weighting is mirrored)."""
    import ctypes
    ref = ctypes.CDLL(REF_SO)
    for f in ("ref_stbi_load_rgba8", "ref_stbi_load_16_gray", "ref_load_exr_rgba"):
        getattr(ref, f).restype = ctypes.c_void_p

    def grab(fn, path, ctype, ch):
        w, h = ctypes.c_int(), ctypes.c_int()
        p = getattr(ref, fn)(path.encode(), ctypes.byref(w), ctypes.byref(h))
        assert p, path
        shape = (h.value, w.value, ch) if ch > 1 else (h.value, w.value)
        a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctype)), shape).copy()
        ref.ref_free(ctypes.c_void_p(p))
        return a
    rs = np.random.RandomState(1)
    d = str(tmp_path)
    img = (rs.rand(37, 53, 3) * 255).astype(np.uint8)
    cases = [("RGB", img), ("RGBA", (rs.rand(20, 30, 4) * 255).astype(np.uint8)), ("L", img[..., 0]), ("LA", (rs.rand(20, 30, 2) * 255).astype(np.uint8))]
    for mode, arr in cases:
        p = os.path.join(d, mode + ".png"); Image.fromarray(arr, mode).save(p)
        np.testing.assert_array_equal(pyngp.decode_image(p), grab("ref_stbi_load_rgba8", p, ctypes.c_ubyte, 4))
        np.testing.assert_array_equal(pyngp.decode_png_gray16(p), grab("ref_stbi_load_16_gray", p, ctypes.c_ushort, 1))
    p16 = os.path.join(d, "g16.png"); Image.fromarray((rs.rand(20, 30) * 65535).astype(np.uint16)).save(p16)
    np.testing.assert_array_equal(pyngp.decode_image(p16), grab("ref_stbi_load_rgba8", p16, ctypes.c_ubyte, 4))
    np.testing.assert_array_equal(pyngp.decode_png_gray16(p16), grab("ref_stbi_load_16_gray", p16, ctypes.c_ushort, 1))
    pp = os.path.join(d, "pal.png"); Image.fromarray(img).convert("P").save(pp)
    np.testing.assert_array_equal(pyngp.decode_image(pp), grab("ref_stbi_load_rgba8", pp, ctypes.c_ubyte, 4))
    for q in (60, 92):
        for sub in (0, 1, 2):
            for prog in (False, True):
                pj = os.path.join(d, "t.jpg"); Image.fromarray(img).save(pj, quality=q, subsampling=sub, progressive=prog)
                diff = np.abs(pyngp.decode_image(pj).astype(int) - grab("ref_stbi_load_rgba8", pj, ctypes.c_ubyte, 4).astype(int))
                assert diff.max() <= 3, (q, sub, prog, diff.max())
    fox = "/root/reference/data/nerf/fox/images/0001.jpg"
    if os.path.exists(fox):
        diff = np.abs(pyngp.decode_image(fox).astype(int) - grab("ref_stbi_load_rgba8", fox, ctypes.c_ubyte, 4).astype(int))
        assert diff.max() <= 3 and (diff > 0).mean() < 0.03
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_image_io_cpu import _write_exr
    for comp, pt in ((0, 2), (3, 1)):
        pe = os.path.join(d, "t.exr"); _write_exr(pe, rs.rand(19, 23, 4).astype(np.float32), comp, pt)
        np.testing.assert_array_equal(pyngp.decode_exr(pe), grab("ref_load_exr_rgba", pe, ctypes.c_float, 4))
