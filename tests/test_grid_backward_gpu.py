"""Hash-grid backward on its own (ngp_hip_grid_backward = tcnn kernel_grid_backward, EGradientMode::Overwrite) against the oracle's EXACT sum.

Dense levels and hashed levels with a power-of-two table are accumulated in 64-bit fixed point on the device: the result is the exact sum of the
fp16 terms half(w * dL/dx), rounded to fp16 once — the comparison is BIT-EXACT and does not depend on the scheduling of the adds.  Only tables with more
than 2^20 entries per level (big.json) take the float fallback (fp16 LDS atomics) and are compared with a tolerance."""
import numpy as np
import pytest

import capi
import helpers as H
from capi import check

pytestmark = pytest.mark.gpu


def _gm_desc(ngp, n_dims, log2, desired):
    desc = np.zeros(1, dtype=capi.NET_DESC)
    pls = float(np.exp(np.log(desired / 16.0) / 15).astype(np.float32))
    check(ngp.ngp_hip_gridmlp_make_desc_host(n_dims, 16, log2, 16, H.f32(pls), desc.ctypes.data))
    return desc


def _ray_positions(n, n_dims, rs, run=40):
    """consecutive fixed-step samples along random rays (what a compacted training batch looks like), clipped to the unit cube"""
    n_rays = (n + run - 1) // run
    o = rs.rand(n_rays, n_dims)
    d = rs.randn(n_rays, n_dims)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = (np.arange(run) * (np.sqrt(3.0) / 1024.0))[None, :, None]
    pos = (o[:, None, :] + d[:, None, :] * t).reshape(-1, n_dims)[:n]
    return np.clip(pos, 0.0, 1.0).astype(np.float32)


def _run(ngp, cuda, n_dims, desc, pos, planes, unordered=False):
    n = pos.shape[0]
    d_desc, d_pos, d_pl = H.to_dev(desc, cuda), H.to_dev(pos, cuda), H.to_dev(planes, cuda)
    n_entries = int(desc["n_grid_entries"][0])
    sb = ngp.ngp_hip_grid_backward_scratch_bytes(n)
    scratch, grad = H.dev_zeros(sb, cuda), H.dev_zeros(n_entries * 4, cuda)
    grad[:] = 0x3c                                      # poison: Overwrite mode must write every entry
    fn = ngp.ngp_hip_grid_backward_unordered if unordered else ngp.ngp_hip_grid_backward   # unordered: dense levels binned as pair records (batches without ray order: image, SDF)
    check(fn(None, n_dims, d_desc.data_ptr(), d_pos.data_ptr(), pos.shape[1], n, d_pl.data_ptr(), grad.data_ptr(), scratch.data_ptr(), sb))
    return H.to_host(grad, np.uint16)


def _planes(n, rs, amp=0.05, special=True):
    pl = (rs.randn(16, n, 2) * amp).astype(np.float16)
    pl[:, rs.rand(n) < 0.1] = 0                         # samples the loss masked out
    if special:
        pl[3, 5, 0] = np.float16(np.inf)                # dropped terms (a step the loss scaler skips)
        pl[9, 7, 1] = np.float16(np.nan)
        pl[12, 11] = np.float16(6.0e-8)                 # subnormal gradients: terms underflow to exact zeros
        pl[1, 13] = np.float16(60000.0)                 # near the top of the fp16 range
    return pl


CASES = [("nerf", 3, 19, 2048.0), ("image", 2, 24, 512.0), ("sdf", 3, 19, 2048.0), ("hashed2d", 2, 14, 2048.0), ("t20", 3, 20, 2048.0)]


@pytest.mark.parametrize("name,n_dims,log2,desired", CASES)
@pytest.mark.parametrize("coherent", [True, False])
@pytest.mark.parametrize("unordered", [False, True])
def test_grid_backward_is_the_exact_sum(ngp, oracle, cuda, name, n_dims, log2, desired, coherent, unordered):
    n = 8192
    desc = H.make_desc(ngp, log2) if name == "nerf" else _gm_desc(ngp, n_dims, log2, desired)
    rs = np.random.RandomState(n_dims * 100 + log2 + coherent)
    pos = _ray_positions(n, n_dims, rs) if coherent else rs.rand(n, n_dims).astype(np.float32)
    pl = _planes(n, rs)
    got = _run(ngp, cuda, n_dims, desc, pos, pl.view(np.uint16), unordered)
    ref = np.zeros(got.size, np.uint16)
    oracle.orc_grid_backward_exact(n_dims, desc.ctypes.data, pos.ctypes.data, n_dims, n, pl.view(np.uint16).ctypes.data, ref.ctypes.data)
    assert (ref != 0).sum() > 1000
    # -0 vs +0: an entry nobody touched is written as +0 by both; sums that cancel exactly are +0 in both (integer zero)
    np.testing.assert_array_equal(got, ref)


def test_grid_backward_schedule_independent(ngp, cuda):
    """two runs over the same inputs give the same bits (the sums are integer; the order of the LDS atomics does not matter)"""
    n = 16384
    desc = H.make_desc(ngp, 19)
    rs = np.random.RandomState(5)
    pos = _ray_positions(n, 3, rs)
    pl = _planes(n, rs, special=False)
    a = _run(ngp, cuda, 3, desc, pos, pl.view(np.uint16))
    b = _run(ngp, cuda, 3, desc, pos, pl.view(np.uint16))
    np.testing.assert_array_equal(a, b)
    # ... and a permutation of the samples changes nothing either
    perm = rs.permutation(n)
    c = _run(ngp, cuda, 3, desc, np.ascontiguousarray(pos[perm]), np.ascontiguousarray(pl[:, perm]).view(np.uint16))
    np.testing.assert_array_equal(a, c)


@pytest.mark.parametrize("aabb_scale", [4, 16])
def test_grid_backward_fine_levels_straddle_slices(ngp, oracle, cuda, aabb_scale):
    """aabb_scale >= 4 (4: the fox scene): the finest levels have resolution >= 4096, so the x term of the hash reaches the slice bits and the two x corners of a cell
    with x + 1 a multiple of 4096 fall into different slices (two records per pair).  Rounds 1-3 sent such levels to an fp16 LDS-atomic fallback; they are exact now.
    A third of the samples sit ON such cells of every fine level."""
    n = 6144
    desc = H.make_desc(ngp, 19, aabb_scale=aabb_scale)
    rs = np.random.RandomState(9)
    pos = rs.rand(n, 3).astype(np.float32)
    lv = desc["levels"][0]
    fine = [l for l in range(16) if int(lv[l]["resolution"]) >= 4096]
    assert fine                                          # the case exists in this configuration
    k = 2048
    for q, l in enumerate(fine):
        sc, res = float(lv[l]["scale"]), int(lv[l]["resolution"])
        sel = slice(k + q * (2048 // len(fine)), k + (q + 1) * (2048 // len(fine)))
        m = pos[sel].shape[0]
        bx = 4096 * rs.randint(1, max(2, res // 4096), size=m) - 1     # cell x = 4095, 8191, ...: x + 1 crosses a slice boundary
        pos[sel, 0] = ((bx + rs.rand(m) * 0.98 + 0.01) - 0.5) / sc     # level_pos: floor(x * scale + 0.5)
    pos = np.clip(pos, 0.0, 1.0).astype(np.float32)
    pl = _planes(n, rs, special=False)
    got = _run(ngp, cuda, 3, desc, pos, pl.view(np.uint16))
    ref = np.zeros(got.size, np.uint16)
    oracle.orc_grid_backward_exact(3, desc.ctypes.data, pos.ctypes.data, 3, n, pl.view(np.uint16).ctypes.data, ref.ctypes.data)
    for l in fine:   # the crafted samples do straddle: cell x = 4095 mod 4096 on their level
        gx = np.floor(pos[:, 0].astype(np.float32) * np.float32(lv[l]["scale"]) + np.float32(0.5)).astype(np.int64)
        assert ((gx % 4096) == 4095).sum() > 100
    np.testing.assert_array_equal(got, ref)


def test_grid_backward_rejects_bad_arguments(ngp, cuda):
    desc = H.make_desc(ngp, 19)
    d_desc = H.to_dev(desc, cuda)
    buf = H.dev_zeros(1024, cuda)
    assert ngp.ngp_hip_grid_backward(None, 3, d_desc.data_ptr(), buf.data_ptr(), 3, 100, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 1 << 40) != 0
    assert ngp.ngp_hip_grid_backward(None, 4, d_desc.data_ptr(), buf.data_ptr(), 3, 256, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 1 << 40) != 0
    assert ngp.ngp_hip_grid_backward(None, 3, d_desc.data_ptr(), buf.data_ptr(), 3, 256, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 16) != 0
    assert b"scratch" in ngp.ngp_hip_last_error()


def test_grid_backward_big_table(ngp, oracle, cuda):
    """configs/nerf/big.json (log2_hashmap_size 21): level 6 (res 112) is dense with 343 slices of 4096 entries — more than the 256 bins of the counting sort — and
    the hashed levels have 512: both take the owners' float path (fp16 LDS atomics), the smaller dense levels stay exact"""
    n = 4096
    desc = H.make_desc(ngp, 21)
    rs = np.random.RandomState(21)
    pos = _ray_positions(n, 3, rs)
    pl = _planes(n, rs, special=False)
    got = _run(ngp, cuda, 3, desc, pos, pl.view(np.uint16)).view(np.float16).astype(np.float64)
    ref = np.zeros(got.size, np.uint16)
    oracle.orc_grid_backward_exact(3, desc.ctypes.data, pos.ctypes.data, 3, n, pl.view(np.uint16).ctypes.data, ref.ctypes.data)
    ref = ref.view(np.float16).astype(np.float64)
    assert np.isfinite(got).all()
    lv = desc["levels"][0]
    n_float = 0
    for l in range(16):
        o, sz, res = int(lv[l]["offset"]) * 2, int(lv[l]["size"]) * 2, int(lv[l]["resolution"])
        exact = sz // 2 <= 256 * 4096 and (res ** 3 <= sz // 2)
        if exact:
            np.testing.assert_array_equal(got[o:o + sz], ref[o:o + sz])
        else:
            n_float += 1
            assert np.linalg.norm(ref[o:o + sz]) > 0
            assert np.linalg.norm(got[o:o + sz] - ref[o:o + sz]) < 2e-3 * np.linalg.norm(ref[o:o + sz]), l
    assert n_float >= 2
