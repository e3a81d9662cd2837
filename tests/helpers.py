"""Shared test scaffolding: oracle loader, synthetic scene pieces, device-buffer plumbing (torch = memory only)."""
import ctypes
import os

import numpy as np

import capi
from capi import AABB, COORD, IMAGE_META, NET_DESC, PAYLOAD, RAY, XFORM, ptr  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_oracle = None


def load_oracle():
    global _oracle
    if _oracle is None:
        _oracle = capi.CLib(os.path.join(ROOT, "oracle", "_build", "libngp_oracle.so"), os.path.join(ROOT, "oracle", "ngp_oracle.h"), "orc_")
    return _oracle


def load_oracle_native():
    """The oracle built -O3 -march=native ON THIS MACHINE (bench.py's cpu_baseline leg; `make -C oracle native` into a temporary directory, so a library built for
    another host's CPU is never picked up).  Returns (library, "native") or, when the compiler is missing or fails, (the portable build, "portable: <why>")."""
    import subprocess, tempfile
    try:
        out = os.path.join(tempfile.mkdtemp(prefix="ngp_oracle_native_"), "libngp_oracle_native.so")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "native", "NATIVE_OUT=" + out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return capi.CLib(out, os.path.join(ROOT, "oracle", "ngp_oracle.h"), "orc_"), "native"
    except Exception as e:   # noqa: BLE001 - any build problem falls back to the checked-in recipe's portable library
        return load_oracle(), "portable: %s" % type(e).__name__


def f32(x):
    return ctypes.c_float(float(x))


def pcg32_state(seed):
    """default_rng_t{seed}: pcg32(initstate=seed, initseq=1) -> (state, inc)."""
    mult, mask = 0x5851f42d4c957f2d, (1 << 64) - 1
    inc = (1 << 1) | 1
    state = 0
    state = (state * mult + inc) & mask
    state = (state + seed) & mask
    state = (state * mult + inc) & mask
    return state, inc


def pcg32_advance(state, inc, delta):
    mult, mask = 0x5851f42d4c957f2d, (1 << 64) - 1
    cur_mult, cur_plus, acc_mult, acc_plus = mult, inc, 1, 0
    delta &= mask
    while delta > 0:
        if delta & 1:
            acc_mult = (acc_mult * cur_mult) & mask
            acc_plus = (acc_plus * cur_mult + cur_plus) & mask
        cur_plus = ((cur_mult + 1) * cur_plus) & mask
        cur_mult = (cur_mult * cur_mult) & mask
        delta >>= 1
    return (acc_mult * state + acc_plus) & mask, inc


def unit_aabb(scale=1):
    a = np.zeros(1, dtype=AABB)
    a["min"][0] = 0.5 - 0.5 * scale
    a["max"][0] = 0.5 + 0.5 * scale
    return a


def make_desc(ngp, log2_hashmap_size=19, base_resolution=16, aabb_scale=1, n_levels=16):
    desc = np.zeros(1, dtype=NET_DESC)
    pls = float(np.exp(np.log(2048.0 * aabb_scale / base_resolution) / (n_levels - 1)).astype(np.float32))
    rc = ngp.ngp_hip_net_make_desc_host(n_levels, log2_hashmap_size, base_resolution, f32(pls), desc.ctypes.data)
    assert rc == 0
    return desc


def n_params(desc):
    return 10240 + 2 * int(desc["n_grid_entries"][0])


def random_params(desc, seed=0, grid_amp=1.0, mlp_gain=1.0):
    """fp16 parameter vector with Xavier-ish MLP weights and a grid large enough that outputs are non-trivial."""
    rs = np.random.RandomState(seed)
    dims = [(64, 32), (16, 64), (64, 32), (64, 64), (16, 64)]
    parts = []
    for o, i in dims:
        s = mlp_gain * np.sqrt(6.0 / (o + i))
        parts.append(rs.uniform(-s, s, size=o * i))
    parts.append(rs.uniform(-grid_amp, grid_amp, size=2 * int(desc["n_grid_entries"][0])))
    return np.concatenate(parts).astype(np.float16)


def random_coords(n, seed=0):
    rs = np.random.RandomState(seed)
    c = np.zeros(n, dtype=COORD)
    c["pos"] = rs.rand(n, 3).astype(np.float32)
    d = rs.randn(n, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c["dir"] = ((d + 1.0) * 0.5).astype(np.float32)
    c["dt"] = rs.rand(n).astype(np.float32) * 0.01
    return c


def look_at_xform(pos, target=(0.5, 0.5, 0.5), up=(0.0, 0.0, 1.0)):
    """3x4 column-major camera matrix in the reference's convention: columns = right, down, forward, origin."""
    pos = np.asarray(pos, dtype=np.float64)
    fwd = np.asarray(target) - pos
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, dtype=np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    m = np.stack([right, down, fwd, pos], axis=1)  # 3x4
    return m.astype(np.float32).T.reshape(-1).copy()  # column-major flat


def hemisphere_cameras(n, radius=1.3, seed=1):
    rs = np.random.RandomState(seed)
    xf = np.zeros(n, dtype=XFORM)
    for i in range(n):
        th = rs.uniform(0, 2 * np.pi)
        ph = rs.uniform(0.15, 1.2)
        p = np.array([0.5 + radius * np.cos(th) * np.cos(ph), 0.5 + radius * np.sin(th) * np.cos(ph), 0.5 + radius * np.sin(ph)])
        m = look_at_xform(p)
        xf["start"][i] = m
        xf["end"][i] = m
    return xf


def make_images(n, w, h, seed=2, masked_fraction=0.01):
    """RGBA8 training images (n, h, w) as uint32 little-endian R | G<<8 | B<<16 | A<<24, some pixels set to the mask colour."""
    rs = np.random.RandomState(seed)
    px = rs.randint(0, 256, size=(n, h, w, 4), dtype=np.uint8)
    px[..., 3] = np.where(rs.rand(n, h, w) < 0.7, 255, px[..., 3])
    raw = px.view(np.uint32).reshape(n, h, w).copy()
    raw[rs.rand(n, h, w) < masked_fraction] = 0x00FF00FF
    return raw


def make_metadata(pixel_ptrs, w, h, focal, lens_mode=0, lens_params=None):
    n = len(pixel_ptrs)
    md = np.zeros(n, dtype=IMAGE_META)
    md["pixels"] = np.asarray(pixel_ptrs, dtype=np.uint64)
    md["image_data_type"] = 1
    md["res"] = (w, h)
    md["focal_length"] = (focal, focal)
    md["principal_point"] = (0.5, 0.5)
    md["lens_mode"] = lens_mode
    if lens_params is not None:
        md["lens_params"][:, : len(lens_params)] = lens_params
    return md


def blob_density_grid(n_cascades=1, seed=3):
    """fp32 density grid (Morton order) with a few occupied blobs; negative cells mark 'never seen'."""
    G = 128
    x, y, z = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")

    def expand(v):
        v = v.astype(np.uint32)
        v = (v * 0x00010001) & 0xFF0000FF
        v = (v * 0x00000101) & 0x0F00F00F
        v = (v * 0x00000011) & 0xC30C30C3
        v = (v * 0x00000005) & 0x49249249
        return v

    morton = (expand(x) | (expand(y) << 1) | (expand(z) << 2)).reshape(-1)
    rs = np.random.RandomState(seed)
    grid = np.zeros((n_cascades, G ** 3), dtype=np.float32)
    for c in range(n_cascades):
        cx, cy, cz = (x + 0.5) / G - 0.5, (y + 0.5) / G - 0.5, (z + 0.5) / G - 0.5
        dens = np.zeros((G, G, G), dtype=np.float32)
        for _ in range(6):
            o = rs.uniform(-0.3, 0.3, size=3)
            r = rs.uniform(0.05, 0.18)
            dens += np.where((cx - o[0]) ** 2 + (cy - o[1]) ** 2 + (cz - o[2]) ** 2 < r * r, rs.uniform(0.02, 1.0), 0.0).astype(np.float32)
        dens[rs.rand(G, G, G) < 0.002] = 0.5
        flat = np.zeros(G ** 3, dtype=np.float32)
        flat[morton] = dens.reshape(-1)
        flat[rs.rand(G ** 3) < 0.01] = -1.0
        grid[c] = flat
    return grid.reshape(-1)


def oracle_bitfield(orc, grid, n_cascades):
    mean = orc.orc_density_grid_mean(grid.ctypes.data, 128 ** 3)
    bf = np.zeros(128 ** 3, dtype=np.uint8)  # 8 cascades * G/8 bytes
    orc.orc_update_bitfield(grid.ctypes.data, n_cascades, f32(mean), bf.ctypes.data)
    return bf, mean


def to_dev(arr, device):
    import torch
    a = np.ascontiguousarray(arr)
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(device)


def dev_zeros(nbytes, device):
    import torch
    return torch.zeros(int(nbytes), dtype=torch.uint8, device=device)


def to_host(t, dtype, count=None):
    a = t.cpu().numpy().view(np.uint8)
    out = a.view(dtype)
    return out if count is None else out[:count]


def make_error_map_cdfs(oracle, n_img, w, h, seed=7):
    """random error map -> the three CDFs through the oracle (construct_cdf_2d / _1d + host image CDF).  Returns host arrays."""
    rs = np.random.RandomState(seed)
    em = (rs.rand(n_img, h, w) ** 4).astype(np.float32)
    em[:, : h // 3] *= 20.0                        # make the distribution visibly non-uniform
    em[0] *= 5.0
    x = np.zeros((n_img, h, w), np.float32)
    y = np.zeros((n_img, h), np.float32)
    im = np.zeros(n_img, np.float32)
    oracle.orc_construct_cdf_2d(n_img, h, w, em.ctypes.data, x.ctypes.data, y.ctypes.data)
    oracle.orc_construct_cdf_1d(n_img, h, y.ctypes.data, im.ctypes.data)
    pmf, cdf = np.zeros(n_img, np.float32), np.zeros(n_img, np.float32)
    oracle.orc_image_cdf_host(n_img, im.ctypes.data, pmf.ctypes.data, cdf.ctypes.data)
    return dict(em=em, x=x, y=y, img_sums=im, pmf=pmf, img=cdf, res=(w, h))


def error_map_cdf_struct(x_ptr, y_ptr, img_ptr, res):
    c = np.zeros(1, dtype=capi.ERROR_MAP_CDF)
    c["cdf_x_cond_y"], c["cdf_y"], c["cdf_img"] = x_ptr or 0, y_ptr or 0, img_ptr or 0
    c["res"][0] = res
    return c


def morton3d_invert(x):
    """compact every third bit of a Morton code (common_device.cuh morton3D_invert), vectorised"""
    x = np.asarray(x, np.uint32) & np.uint32(0x49249249)
    x = (x | (x >> np.uint32(2))) & np.uint32(0xC30C30C3)
    x = (x | (x >> np.uint32(4))) & np.uint32(0x0F00F00F)
    x = (x | (x >> np.uint32(8))) & np.uint32(0xFF0000FF)
    x = (x | (x >> np.uint32(16))) & np.uint32(0x0000FFFF)
    return x
