"""Host-side image decoders of the loader (row f2) against independent decoders: JPEG (baseline + progressive Huffman, grey / 4:4:4 / 4:2:2 / 4:2:0,
restart intervals, odd sizes) against PIL (libjpeg), OpenEXR scanline files (NONE / RLE / ZIPS / ZIP, HALF / FLOAT) against a writer in this file that
follows the published layout.  The reference reads these through the vendored stb_image / tinyexr (src/nerf_loader.cu:575-581).  No GPU needed."""
import io
import os
import struct
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]


def _photo(w, h, seed=0):
    """smooth content with edges and noise (what JPEG is made for, plus what breaks it)"""
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([128 + 100 * np.sin(x / 17.0) * np.cos(y / 23.0), 128 + 90 * np.cos((x + y) / 31.0), 60 + 0.5 * x * (h - y) / max(h, 1) * 255.0 / max(w, 1)], -1)
    img[h // 3: h // 2, w // 4: w // 2] = [250, 20, 40]
    img += rs.randn(h, w, 3) * 6.0
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(64, 48), (67, 53), (1, 1), (17, 8)])
@pytest.mark.parametrize("kw", [dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, progressive=True), dict(subsampling=0, progressive=True, optimize=True),
                                dict(subsampling=1, progressive=True), dict(subsampling=2, restart_marker_blocks=3), dict(subsampling=0, restart_marker_rows=1, progressive=True), dict(grey=True),
                                dict(grey=True, progressive=True)])
def test_jpeg_matches_libjpeg(tmp_path, w, h, kw):
    import pyngp
    from PIL import Image
    kw = dict(kw)
    grey = kw.pop("grey", False)
    src = _photo(w, h)
    im = Image.fromarray(src[..., 1] if grey else src, "L" if grey else "RGB")
    p = str(tmp_path / "t.jpg")
    im.save(p, quality=88, **kw)
    ref = np.asarray(Image.open(p).convert("RGB")).astype(np.int32)
    got = pyngp.decode_image(p)
    assert got.shape == (h, w, 4) and got.dtype == np.uint8 and (got[..., 3] == 255).all()
    d = np.abs(got[..., :3].astype(np.int32) - ref)
    if kw.get("subsampling") == 1 and w > 2:
        d[:, 2 * ((w + 1) // 2 - 1)] = 0   # 4:2:2: the column where stb_image (the reference's decoder, mirrored here; tests/test_loader_cpu.py) weights the last chroma pair unlike libjpeg
    # the inverse DCT here is an exact float transform rounded once; libjpeg's integer IDCT, its chroma filter and its fixed-point colour matrix
    # each round in between: a few code values at isolated pixels, a small fraction of one on average
    assert d.max() <= 5 and d.mean() < 0.5, (d.max(), d.mean())


def test_jpeg_low_quality_and_large(tmp_path):
    import pyngp
    from PIL import Image
    src = _photo(640, 360, seed=3)
    for q, prog in ((35, False), (35, True), (97, True)):
        buf = io.BytesIO()
        Image.fromarray(src, "RGB").save(buf, "JPEG", quality=q, progressive=prog)
        p = str(tmp_path / ("q%d%d.jpg" % (q, prog)))
        open(p, "wb").write(buf.getvalue())
        ref = np.asarray(Image.open(p).convert("RGB")).astype(np.int32)
        d = np.abs(pyngp.decode_image(p)[..., :3].astype(np.int32) - ref)
        assert d.max() <= 6 and d.mean() < 0.5, (q, prog, d.max(), d.mean())


def test_image_signature_dispatch_and_errors(tmp_path):
    import pyngp
    from PIL import Image
    src = _photo(40, 30)
    p = str(tmp_path / "really_a_png.jpg")        # stb_image looks at the content, not at the extension
    Image.fromarray(src, "RGB").save(p, "PNG")
    np.testing.assert_array_equal(pyngp.decode_image(p)[..., :3], src)
    bad = str(tmp_path / "x.jpg")
    open(bad, "wb").write(b"GIF89a" + bytes(64))
    with pytest.raises(RuntimeError):
        pyngp.decode_image(bad)
    trunc = str(tmp_path / "t.jpg")
    buf = io.BytesIO()
    Image.fromarray(src, "RGB").save(buf, "JPEG")
    open(trunc, "wb").write(buf.getvalue()[:200])
    img = None
    try:
        img = pyngp.decode_image(trunc)            # a truncated scan decodes to something (like stb_image) or raises — it must not crash
    except RuntimeError:
        pass
    assert img is None or img.shape == (30, 40, 4)
    with pytest.raises(RuntimeError):
        pyngp.decode_image(str(tmp_path / "missing.jpg"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/data/nerf/fox/images"), reason="the reference tree (fox photographs) only exists in the build container")
def test_fox_frames_decode_like_libjpeg():
    """BASELINE config #2's own data: data/nerf/fox/images/*.jpg (1080 x 1920 portraits)"""
    import glob
    import pyngp
    from PIL import Image
    files = sorted(glob.glob("/root/reference/data/nerf/fox/images/*.jpg"))
    assert len(files) == 50
    for f in files[::17]:
        ref = np.asarray(Image.open(f).convert("RGB")).astype(np.int32)
        got = pyngp.decode_image(f)
        assert got.shape == (1920, 1080, 4)
        d = np.abs(got[..., :3].astype(np.int32) - ref)
        assert d.max() <= 4 and d.mean() < 0.1


# ---------------------------------------------------------------------------------------------------------------- OpenEXR
def _write_exr(path, img, compression, pixel_type, channel_names="ABGR", tile=None, extra_levels=0):
    """scanline / tiled (tile=(tw, th), ONE_LEVEL or, with extra_levels, MIPMAP_LEVELS whose further levels hold zeros) OpenEXR writer (test-side): img (H, W, C)
    float32, channels stored in alphabetical order of their names"""
    h, w, _ = img.shape
    names = sorted(channel_names)
    src_index = {"R": 0, "G": 1, "B": 2, "A": 3, "Y": 0}
    head = struct.pack("<II", 20000630, 2 | (0x200 if tile else 0))

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(val)) + val
    ch = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", pixel_type, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    head += attr("channels", "chlist", ch) + attr("compression", "compression", bytes([compression]))
    head += attr("dataWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1)) + attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    head += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    if tile:
        head += attr("tiles", "tiledesc", struct.pack("<IIB", tile[0], tile[1], 1 if extra_levels else 0))
    head += b"\0"
    lines = 16 if compression == 3 else 1
    dt = np.float16 if pixel_type == 1 else np.float32
    blocks = []
    if tile:
        regions = [(struct.pack("<iiii", tx, ty, 0, 0), tx * tile[0], ty * tile[1], min(tile[0], w - tx * tile[0]), min(tile[1], h - ty * tile[1]))
                   for ty in range((h + tile[1] - 1) // tile[1]) for tx in range((w + tile[0] - 1) // tile[0])]
    else:
        regions = [(struct.pack("<i", y0), 0, y0, w, min(lines, h - y0)) for y0 in range(0, h, lines)]
    for prefix, x0, y0, nx, ny in regions:
        raw = b"".join(img[y, x0:x0 + nx, src_index[n]].astype(dt).tobytes() for y in range(y0, y0 + ny) for n in names)
        data = raw
        if compression:
            a = np.frombuffer(raw, np.uint8)
            split = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
            pred = split.copy()
            pred[1:] = (split[1:] - split[:-1] + 128 + 256) % 256
            pb = pred.astype(np.uint8).tobytes()
            if compression == 1:   # RLE: literal runs only, except one repeat run when the block starts with a repeated byte
                out = bytearray()
                i = 0
                while i < len(pb):
                    j = i
                    while j + 1 < len(pb) and pb[j + 1] == pb[i] and j - i < 126:
                        j += 1
                    if j - i >= 2:
                        out += struct.pack("b", j - i) + pb[i:i + 1]
                        i = j + 1
                    else:
                        k = min(len(pb), i + 100)
                        out += struct.pack("b", -(k - i)) + pb[i:k]
                        i = k
                comp = bytes(out)
            else:
                comp = zlib.compress(pb)
            data = comp if len(comp) < len(raw) else raw
        blocks.append(prefix + struct.pack("<i", len(data)) + data)
    for lvl in range(1, extra_levels + 1):   # one zero tile per further mip level (the reader takes level (0, 0) only; its tiles come first in the offset table)
        lw, lh = max(1, w >> lvl), max(1, h >> lvl)
        for ty in range((lh + tile[1] - 1) // tile[1]):
            for tx in range((lw + tile[0] - 1) // tile[0]):
                nx, ny = min(tile[0], lw - tx * tile[0]), min(tile[1], lh - ty * tile[1])
                data = bytes(nx * ny * len(names) * (2 if pixel_type == 1 else 4))
                blocks.append(struct.pack("<iiii", tx, ty, lvl, lvl) + struct.pack("<i", len(data)) + data)
    off = len(head) + 8 * len(blocks)
    table = b""
    for b in blocks:
        table += struct.pack("<Q", off)
        off += len(b)
    open(path, "wb").write(head + table + b"".join(blocks))


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
@pytest.mark.parametrize("pixel_type", [1, 2])
def test_exr_scanline_round_trip(tmp_path, compression, pixel_type):
    import pyngp
    rs = np.random.RandomState(compression * 2 + pixel_type)
    h, w = 37, 29
    img = (rs.rand(h, w, 4) ** 3 * 4.0).astype(np.float32)
    img[5:20, 3:9] = 0.25          # flat areas: runs for RLE, long matches for zlib
    p = str(tmp_path / "t.exr")
    _write_exr(p, img, compression, pixel_type)
    got = pyngp.decode_exr(p)
    want = img.astype(np.float16).astype(np.float32) if pixel_type == 1 else img
    np.testing.assert_array_equal(got, want)


def test_exr_rgb_without_alpha_and_luminance(tmp_path):
    import pyngp
    img = np.random.RandomState(1).rand(8, 5, 4).astype(np.float32)
    p = str(tmp_path / "rgb.exr")
    _write_exr(p, img, 3, 2, "BGR")
    got = pyngp.decode_exr(p)
    np.testing.assert_array_equal(got[..., :3], img[..., :3])
    assert (got[..., 3] == 1.0).all()
    _write_exr(p, img, 2, 1, "Y")
    got = pyngp.decode_exr(p)
    want = img[..., 0].astype(np.float16).astype(np.float32)
    for c in range(3):
        np.testing.assert_array_equal(got[..., c], want)


def test_exr_rejects_what_it_cannot_read(tmp_path):
    import pyngp
    img = np.zeros((4, 4, 4), np.float32)
    p = str(tmp_path / "pxr24.exr")
    _write_exr(p, img, 0, 2)
    raw = bytearray(open(p, "rb").read())
    i = raw.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    raw[i] = 5    # PXR24
    open(p, "wb").write(bytes(raw))
    with pytest.raises(RuntimeError, match="PXR24"):
        pyngp.decode_exr(p)
    open(p, "wb").write(b"not an exr file at all")
    with pytest.raises(RuntimeError):
        pyngp.decode_exr(p)


@pytest.mark.skipif(not os.path.exists("/root/reference/data/image/albert.exr"), reason="the reference tree only exists in the build container")
def test_albert_exr_matches_the_reference_decoder_and_the_committed_crop():
    """data/image/albert.exr (ZIP, four FLOAT channels): RGBA mean as the reference's vendored tinyexr reports it, and the committed crop"""
    import pyngp
    img = pyngp.decode_exr("/root/reference/data/image/albert.exr")
    assert img.shape == (1024, 1024, 4)
    np.testing.assert_allclose(img.reshape(-1, 4).mean(0), [0.18612, 0.18612, 0.18612, 1.0], atol=2e-5)
    crop = np.load(os.path.join(ROOT, "tests", "golden", "albert_crop_128.npy"))
    np.testing.assert_array_equal(crop, img[448:576, 448:576].astype(np.float16))


# ---------------------------------------------------------------------------------------------------------------- hostile files
def _replace_dht(jpeg_bytes, new_segments):
    """drop every DHT segment of a baseline JPEG and put `new_segments` (payloads without marker / length) in front of the first SOS"""
    out, i = bytearray(jpeg_bytes[:2]), 2
    while i < len(jpeg_bytes):
        assert jpeg_bytes[i] == 0xFF
        m = jpeg_bytes[i + 1]
        ln = struct.unpack(">H", jpeg_bytes[i + 2:i + 4])[0]
        if m == 0xDA:
            for seg in new_segments:
                out += b"\xFF\xC4" + struct.pack(">H", len(seg) + 2) + seg
            out += jpeg_bytes[i:]
            return bytes(out)
        if m != 0xC4:
            out += jpeg_bytes[i:i + 2 + ln]
        i += 2 + ln
    raise AssertionError("no SOS")


def _baseline_jpeg(tmp_path):
    from PIL import Image
    p = str(tmp_path / "ok.jpg")
    Image.fromarray(_photo(32, 24)).save(p, quality=90, subsampling=0)
    return open(p, "rb").read()


@pytest.mark.parametrize("counts", [[200] + [0] * 15, [3] + [0] * 15, [1, 3] + [0] * 14, [0] * 8 + [255, 1] + [0] * 6, [2, 0, 0, 0, 0, 0, 0, 0, 1] + [0] * 7])
def test_jpeg_oversubscribed_huffman_tables_are_rejected_before_any_write(tmp_path, counts):
    """a DHT whose code lengths over-subscribe the code space (e.g. 200 codes of length 1) used to index past the 512-entry prefix tables of the
    decoder object on the stack; it must raise, for DC and AC classes alike"""
    import pyngp
    base = _baseline_jpeg(tmp_path)
    total = sum(counts)
    for tc in (0, 1):
        seg = bytes([tc << 4]) + bytes(counts) + bytes(range(total % 256)) * 0 + bytes([i % 256 for i in range(total)])
        p = str(tmp_path / ("dht_%d.jpg" % tc))
        open(p, "wb").write(_replace_dht(base, [seg]))
        with pytest.raises(RuntimeError, match="Huffman"):
            pyngp.decode_image(p)


def test_jpeg_out_of_range_magnitude_categories_raise(tmp_path):
    """DC categories above 11 / AC sizes above 10 would shift by 32 or more in receive_extend (undefined behaviour): one-symbol tables that
    decode every code to such a category must raise, not shift"""
    import pyngp
    base = _baseline_jpeg(tmp_path)
    one = [2] + [0] * 15                                               # the two 1-bit codes: every bit pattern decodes
    ok_ac = bytes([0x10]) + bytes(one) + bytes([0x00, 0x00])           # AC: every code = end of block
    bad_dc = bytes([0x00]) + bytes(one) + bytes([31, 31])              # DC: category 31
    ok_dc = bytes([0x00]) + bytes(one) + bytes([0x00, 0x00])
    bad_ac = bytes([0x10]) + bytes(one) + bytes([0x0F, 0x0F])          # AC: run 0, size 15
    for name, segs in (("dc", [bad_dc, ok_ac]), ("ac", [ok_dc, bad_ac])):
        both = []
        for th in (0, 1):                                              # the file uses tables 0 and 1 of each class
            both += [bytes([s[0] | th]) + s[1:] for s in segs]
        p = str(tmp_path / ("cat_%s.jpg" % name))
        open(p, "wb").write(_replace_dht(base, both))
        with pytest.raises(RuntimeError, match="category|size"):
            pyngp.decode_image(p)


def test_jpeg_random_corruption_never_crashes(tmp_path):
    """cheap fuzz: flip bytes of valid baseline / progressive files; every outcome is an image or a RuntimeError"""
    import pyngp
    from PIL import Image
    rs = np.random.RandomState(5)
    for kw in (dict(subsampling=2), dict(subsampling=1, progressive=True)):
        p = str(tmp_path / "src.jpg")
        Image.fromarray(_photo(40, 40)).save(p, quality=85, **kw)
        src = open(p, "rb").read()
        for it in range(150):
            b = bytearray(src)
            for _ in range(rs.randint(1, 6)):
                b[rs.randint(2, len(b))] = rs.randint(0, 256)
            q = str(tmp_path / "fuzz.jpg")
            open(q, "wb").write(bytes(b))
            try:
                img = pyngp.decode_image(q)
                assert img.ndim == 3 and img.shape[2] == 4
            except RuntimeError:
                pass


def test_exr_hostile_headers_and_blocks_raise(tmp_path):
    import pyngp
    img = np.random.RandomState(0).rand(8, 8, 4).astype(np.float32)
    p = str(tmp_path / "src.exr")
    _write_exr(p, img, 1, 2)
    raw = open(p, "rb").read()
    key = b"dataWindow\0box2i\0" + struct.pack("<I", 16)
    i = raw.index(key) + len(key)
    # a data window whose extent overflows int32
    bad = raw[:i] + struct.pack("<iiii", -2147483647, 0, 2147483647, 7) + raw[i + 16:]
    q = str(tmp_path / "bad.exr")
    open(q, "wb").write(bad)
    with pytest.raises(RuntimeError, match="data window"):
        pyngp.decode_exr(q)
    # a block offset near 2^64 (offset + 8 wraps around)
    j = raw.index(b"screenWindowWidth\0float\0") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1
    bad = raw[:j] + struct.pack("<Q", 0xFFFFFFFFFFFFFFFC) + raw[j + 8:]
    open(q, "wb").write(bad)
    with pytest.raises(RuntimeError, match="offset"):
        pyngp.decode_exr(q)
    # an RLE block that expands far past the block's size (a flat image, so that the writer keeps the RLE form)
    _write_exr(p, np.full((8, 8, 4), 0.25, np.float32), 1, 2)
    raw = open(p, "rb").read()
    off0 = struct.unpack("<Q", raw[j:j + 8])[0]
    n0 = struct.unpack("<i", raw[off0 + 4:off0 + 8])[0]
    bomb = (struct.pack("b", 127) + b"\x80") * (n0 // 2)
    bad = raw[:off0 + 8] + bomb + raw[off0 + 8 + len(bomb):]
    open(q, "wb").write(bad)
    with pytest.raises(RuntimeError, match="RLE"):
        pyngp.decode_exr(q)


# ---------------------------------------------------------------------------------------------------------------- PIZ, tiles, Adam7, Radiance HDR
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_imageio.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libref_imageio.so (the reference's vendored stb_image + tinyexr, `make -C oracle ref`) is built in the build container")


def _ref():
    import ctypes
    ref = ctypes.CDLL(REF_SO)
    for f in ("ref_stbi_load_rgba8", "ref_stbi_load_16_gray", "ref_load_exr_rgba"):
        getattr(ref, f).restype = ctypes.c_void_p

    def grab(fn, path, ctype, ch):
        w, h = ctypes.c_int(), ctypes.c_int()
        p = getattr(ref, fn)(str(path).encode(), ctypes.byref(w), ctypes.byref(h))
        assert p, path
        shape = (h.value, w.value, ch) if ch > 1 else (h.value, w.value)
        a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctype)), shape).copy()
        ref.ref_free(ctypes.c_void_p(p))
        return a
    return ref, grab


def test_exr_piz_fixtures_written_by_the_references_tinyexr():
    """tests/golden/images/piz_*.exr: PIZ files the reference's vendored tinyexr wrote (tests/golden/make_image_fixtures.py) from closed-form images — HALF and FLOAT
    channels, odd sizes (the wavelet's odd row / column steps), and a block with more than 2^14 different values (the modulo-2^16 wavelet).  Decoded bit for bit."""
    import pyngp
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_image_fixtures import EXR_FIXTURES, exr_fixture_image
    for name, (w, h, pt, kind) in EXR_FIXTURES.items():
        got = pyngp.decode_exr(os.path.join(ROOT, "tests", "golden", "images", name))
        want = exr_fixture_image(w, h, pt, kind)
        assert got.shape == (h, w, 4)
        assert np.array_equal(np.asarray(got).view(np.uint32), want.view(np.uint32)), name


@needs_ref
@pytest.mark.parametrize("w,h", [(1, 1), (3, 2), (33, 65), (130, 33), (64, 64), (7, 100)])
def test_exr_piz_matches_the_references_tinyexr(tmp_path, w, h):
    """files written AND read back by the reference's tinyexr vs this build's PIZ decoder: smooth, noisy (stored raw when PIZ does not shrink the block) and
    heavy-tailed images, HALF and FLOAT"""
    import ctypes
    import pyngp
    ref, grab = _ref()
    rs = np.random.RandomState(w * 131 + h)
    yy, xx = np.mgrid[0:h, 0:w]
    images = [np.stack([np.sin(xx * 0.1) + 1.5, yy * 0.01 + 0.1, (xx + yy) * 0.005, np.ones_like(xx) * 1.0], -1), rs.rand(h, w, 4), rs.randn(h, w, 4) ** 3 * 100,
              np.where(rs.rand(h, w, 4) < 0.9, 0.25, rs.rand(h, w, 4))]
    for k, img in enumerate(images):
        for pt in (1, 2):
            p = str(tmp_path / "t.exr")
            img32 = np.ascontiguousarray(img.astype(np.float32))
            assert ref.ref_save_exr_rgba(p.encode(), img32.ctypes.data_as(ctypes.c_void_p), w, h, 4, pt) == 0
            want = grab("ref_load_exr_rgba", p, ctypes.c_float, 4)
            got = np.asarray(pyngp.decode_exr(p))
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, pt)


@pytest.mark.parametrize("compression,pixel_type,tile,levels", [(0, 2, (16, 16), 0), (3, 1, (32, 8), 0), (2, 2, (64, 64), 0), (1, 1, (5, 7), 0), (3, 2, (16, 16), 2)])
def test_exr_tiled_files(tmp_path, compression, pixel_type, tile, levels):
    """tiled files (single part, level (0, 0)): tiles cropped at the right / bottom edge, every compression of the scanline reader (a tile is ONE compressed block),
    and a mip-mapped file whose further levels are ignored; same pixels as the scanline file of the same image — and as the reference's tinyexr where it is built"""
    import pyngp
    rs = np.random.RandomState(7)
    h, w = 45, 70
    img = (rs.rand(h, w, 4) ** 3 * 4.0).astype(np.float32)
    img[5:20, 3:30] = 0.25
    p = str(tmp_path / "tiled.exr")
    _write_exr(p, img, compression, pixel_type, tile=tile, extra_levels=levels)
    got = np.asarray(pyngp.decode_exr(p))
    want = img.astype(np.float16).astype(np.float32) if pixel_type == 1 else img
    np.testing.assert_array_equal(got, want)
    if os.path.exists(REF_SO) and not levels:
        import ctypes
        _, grab = _ref()
        np.testing.assert_array_equal(got, grab("ref_load_exr_rgba", p, ctypes.c_float, 4))


def test_exr_piz_corruption_never_crashes():
    """a PIZ block with flipped bytes raises or decodes (to whatever the damaged code says) — it never reads or writes out of bounds (run under the same process: a crash fails the suite)"""
    import pyngp
    import tempfile
    data = open(os.path.join(ROOT, "tests", "golden", "images", "piz_half_37x45.exr"), "rb").read()
    rs = np.random.RandomState(3)
    start = data.index(b"\0", data.index(b"screenWindowWidth")) + 1   # somewhere behind the header attributes
    raised = 0
    with tempfile.TemporaryDirectory() as d:
        for trial in range(150):
            b = bytearray(data)
            for _ in range(1 + trial % 4):
                i = rs.randint(start, len(b))
                b[i] = rs.randint(0, 256)
            if trial % 10 == 0:
                b = b[:rs.randint(start, len(b))]
            p = os.path.join(d, "c.exr")
            open(p, "wb").write(bytes(b))
            try:
                out = pyngp.decode_exr(p)
                assert out.shape == (45, 37, 4)
            except RuntimeError:
                raised += 1
    assert raised > 20


def _adam7_png(w, h, color_type, depth, pixel_rows, bits_per_pixel, palette=None, trns=None):
    """interlaced PNG from UNFILTERED full-image rows (bytes, packed at `depth`): the seven passes, each filtered with the cyclic filter choice of _filter_rows"""
    from test_loader_cpu import _filter_rows
    X0, Y0, DX, DY = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]

    def pixel(y, x):
        if bits_per_pixel >= 8:
            b = bits_per_pixel // 8
            return pixel_rows[y][x * b:(x + 1) * b]
        bit = x * depth
        return (pixel_rows[y][bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)
    stream = b""
    for k in range(7):
        xs, ys = range(X0[k], w, DX[k]), range(Y0[k], h, DY[k])
        if not len(xs) or not len(ys):
            continue
        rows = []
        for y in ys:
            if bits_per_pixel >= 8:
                rows.append(b"".join(pixel(y, x) for x in xs))
            else:
                acc = bytearray((len(xs) * depth + 7) // 8)
                for i, x in enumerate(xs):
                    bit = i * depth
                    acc[bit >> 3] |= pixel(y, x) << (8 - depth - (bit & 7))
                rows.append(bytes(acc))
        stream += b"".join(_filter_rows(rows, max(1, bits_per_pixel // 8)))

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1))
    if palette is not None:
        out += chunk(b"PLTE", palette)
    if trns is not None:
        out += chunk(b"tRNS", trns)
    return out + chunk(b"IDAT", zlib.compress(stream, 6)) + chunk(b"IEND", b"")


@pytest.mark.parametrize("w,h", [(1, 1), (2, 3), (5, 5), (8, 8), (9, 17), (33, 20)])
@pytest.mark.parametrize("color_type,depth", [(6, 8), (2, 8), (2, 16), (0, 8), (0, 1), (0, 4), (3, 2), (3, 8), (4, 16)])
def test_png_adam7_matches_the_progressive_scan_free_file(tmp_path, w, h, color_type, depth):
    """Adam7-interlaced PNGs (every colour type, sub-byte depths, sizes where some of the seven passes are empty) decode to the pixels of the non-interlaced file of the
    same image — RGBA8 and the 16-bit grey depth read — and to what the reference's stb_image returns where it is built"""
    import pyngp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_loader_cpu import _filter_rows, _png_bytes
    rs = np.random.RandomState(w * 7 + h + color_type * 100 + depth)
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    bpp = channels * depth
    row_bytes = (w * bpp + 7) // 8
    rows = []
    for _ in range(h):
        r = bytearray(rs.randint(0, 256, row_bytes).astype(np.uint8).tobytes())
        if bpp < 8 and (w * bpp) % 8:
            r[-1] &= (0xff << (8 - (w * bpp) % 8)) & 0xff   # padding bits zero
        rows.append(bytes(r))
    palette = rs.randint(0, 256, 3 * (1 << depth)).astype(np.uint8).tobytes() if color_type == 3 else None
    trns = rs.randint(0, 256, 1 << depth).astype(np.uint8).tobytes() if color_type == 3 else None
    flat = tmp_path / "flat.png"
    flat.write_bytes(_png_bytes(w, h, color_type, depth, _filter_rows(rows, max(1, bpp // 8)), palette, trns))
    inter = tmp_path / "adam7.png"
    inter.write_bytes(_adam7_png(w, h, color_type, depth, rows, bpp, palette, trns))
    np.testing.assert_array_equal(pyngp.decode_image(str(inter)), pyngp.decode_image(str(flat)))
    np.testing.assert_array_equal(pyngp.decode_png_gray16(str(inter)), pyngp.decode_png_gray16(str(flat)))
    if os.path.exists(REF_SO):
        import ctypes
        _, grab = _ref()
        np.testing.assert_array_equal(pyngp.decode_image(str(inter)), grab("ref_stbi_load_rgba8", inter, ctypes.c_ubyte, 4))
        np.testing.assert_array_equal(pyngp.decode_png_gray16(str(inter)), grab("ref_stbi_load_16_gray", inter, ctypes.c_ushort, 1))


def _hdr_bytes(rgbe, rle, magic=b"#?RADIANCE"):
    """Radiance file from an (H, W, 4) uint8 RGBE array: new-style run-length scanlines (runs where a byte repeats >= 3 times, dumps otherwise) or flat quadruples"""
    h, w, _ = rgbe.shape
    out = magic + b"\n# test file\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + ("-Y %d +X %d\n" % (h, w)).encode()
    if not rle:
        return out + rgbe.tobytes()
    for y in range(h):
        out += bytes([2, 2, w >> 8, w & 255])
        for k in range(4):
            plane = rgbe[y, :, k].tobytes()
            i = 0
            while i < w:
                j = i
                while j + 1 < w and plane[j + 1] == plane[i] and j - i < 126:
                    j += 1
                if j - i >= 2:
                    out += bytes([128 + (j - i + 1), plane[i]])
                    i = j + 1
                else:
                    k2 = i
                    while k2 < w and k2 - i < 128 and not (k2 + 2 < w and plane[k2] == plane[k2 + 1] == plane[k2 + 2]):
                        k2 += 1
                    k2 = max(k2, i + 1)
                    out += bytes([k2 - i]) + plane[i:k2]
                    i = k2
    return out


@pytest.mark.parametrize("w,h,rle", [(8, 3, True), (40, 11, True), (300, 4, True), (7, 5, False), (40, 11, False), (1, 1, False), (1, 200, False), (2, 90, False)])   # (tall 1- and 2-pixel-wide flat files: 4 w bytes per line is less than any run-length line)
def test_radiance_hdr_as_stbi_load_sees_it(tmp_path, w, h, rle):
    """.hdr goes through stbi_load(.., 4) in the reference's loader (src/nerf_loader.cu:581): RGBE -> float -> (float)pow(v, 1 / 2.2) * 255 + 0.5, clamped, truncated;
    alpha 255.  Run-length and flat files, both magic lines; bit-identical to the reference's stb_image where it is built."""
    import pyngp
    rs = np.random.RandomState(w + h)
    rgbe = rs.randint(0, 256, (h, w, 4)).astype(np.uint8)
    rgbe[..., 3] = rs.randint(118, 134, (h, w))        # exponents around 2^-10 .. 2^6: dark to far above 1
    rgbe[0, 0] = (0, 0, 0, 0)                          # e = 0: black
    rgbe[h // 2, : max(1, w // 3)] = rgbe[h // 2, 0]   # runs
    p = tmp_path / "t.hdr"
    p.write_bytes(_hdr_bytes(rgbe, rle, b"#?RGBE" if (w + h) % 2 else b"#?RADIANCE"))
    got = pyngp.decode_image(str(p))
    assert got.shape == (h, w, 4)
    f = np.where(rgbe[..., 3:4] > 0, np.ldexp(np.float32(1.0), rgbe[..., 3:4].astype(np.int32) - 136), np.float32(0)).astype(np.float32)
    lin = (rgbe[..., :3].astype(np.float32) * f).astype(np.float32)
    z = (np.power(lin.astype(np.float64), np.float64(np.float32(1.0 / 2.2))).astype(np.float32) * np.float32(255) + np.float32(0.5)).astype(np.float32)
    want = np.clip(z, 0, 255).astype(np.int32)
    assert np.abs(got[..., :3].astype(np.int32) - want).max() <= 1      # numpy's pow against libm's: the last place may differ at a rounding boundary
    assert (got[..., 3] == 255).all()
    if os.path.exists(REF_SO):
        import ctypes
        _, grab = _ref()
        np.testing.assert_array_equal(got, grab("ref_stbi_load_rgba8", p, ctypes.c_ubyte, 4))


def test_radiance_hdr_errors(tmp_path):
    import pyngp
    p = tmp_path / "bad.hdr"
    good = _hdr_bytes(np.full((4, 16, 4), 128, np.uint8), True)
    for blob, what in ((good.replace(b"FORMAT=32-bit_rle_rgbe", b"FORMAT=32-bit_rle_xyze"), "format"), (good.replace(b"-Y 4 +X 16", b"+Y 4 +X 16"), "layout"),
                       (good[:-5], "truncated"), (good.replace(b"-Y 4 +X 16", b"-Y 4 +X 17"), "scanline length")):
        p.write_bytes(blob)
        with pytest.raises(RuntimeError, match="HDR"):
            pyngp.decode_image(str(p))


def test_tiny_files_cannot_promise_huge_images(tmp_path):
    """a header of a few bytes must not force a multi-GiB allocation — EXR (data window / 1 x 1 tiles), PNG (IHDR) and Radiance .hdr (resolution line)
    are checked against what the file could encode before anything is sized by them."""
    import zlib
    import pyngp
    # ---- EXR: a valid 8 x 8 file whose data window is rewritten to 16384 x 16384 (2^28 pixels: inside the pixel cap, 4 GiB of floats)
    img = np.random.RandomState(0).rand(8, 8, 4).astype(np.float32)
    p = str(tmp_path / "src.exr")
    _write_exr(p, img, 1, 2)
    raw = open(p, "rb").read()
    key = b"dataWindow\0box2i\0" + struct.pack("<I", 16)
    i = raw.index(key) + len(key)
    q = str(tmp_path / "huge.exr")
    open(q, "wb").write(raw[:i] + struct.pack("<iiii", 0, 0, 16383, 16383) + raw[i + 16:])
    with pytest.raises(RuntimeError, match="too short|could encode"):
        pyngp.decode_exr(q)
    # ---- PNG: IHDR 16384 x 16384 RGBA with a 30-byte IDAT
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 16384, 16384, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
    q = str(tmp_path / "huge.png")
    open(q, "wb").write(png)
    with pytest.raises(RuntimeError, match="could encode"):
        pyngp.decode_image(q)
    # ---- Radiance: a 16384 x 16384 resolution line over a few bytes of data
    q = str(tmp_path / "huge.hdr")
    open(q, "wb").write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 16384 +X 16384\n" + b"\x02\x02\x40\x00" + b"\x81\x10" * 8)
    with pytest.raises(RuntimeError, match="could encode"):
        pyngp.decode_image(q)
