"""GPU parity: fused hash-grid + MFMA MLP kernels (network.hip) vs the CPU oracle, through the C ABI.

Tolerances (fp16 storage, fp32 accumulate; the MFMA sums K in a different order than the oracle's scalar loop):
  * encoded features / activations: 2 fp16 ulp of the value's magnitude (rtol 2e-3, atol 2e-3 on O(1) values)
  * network outputs: rtol 1e-2 / atol 1e-2 (5 layers of fp16 rounding)
  * gradients: fp16 atomic accumulation order is not deterministic -> rtol 3e-2, atol scaled to the gradient norm
"""
import numpy as np
import pytest
import torch

import helpers as H
from capi import check

pytestmark = pytest.mark.gpu


def _setup(ngp, cuda, log2=15, seed=0, n=4096, grid_amp=1.0):
    desc = H.make_desc(ngp, log2_hashmap_size=log2)
    params = H.random_params(desc, seed=seed, grid_amp=grid_amp)
    coords = H.random_coords(n, seed=seed + 1)
    return desc, params, coords, H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(coords, cuda)


def test_mfma_operand_layout_asymmetric(ngp, oracle, cuda):
    """Transpose-detecting check: identity-like W1 rows with an asymmetric grid; any row/col swap in the MFMA operand maps fails."""
    desc = H.make_desc(ngp, log2_hashmap_size=12)
    P = np.zeros(H.n_params(desc), dtype=np.float16)
    W1 = np.zeros((64, 32), dtype=np.float16)
    for o in range(64):
        W1[o, (o * 7 + 3) % 32] = 1.0 + o / 64.0  # asymmetric permutation-like matrix
    W2 = np.zeros((16, 64), dtype=np.float16)
    W2[0, 5] = 1.0; W2[0, 41] = 0.5; W2[3, 17] = 2.0
    P[0:2048] = W1.reshape(-1)
    P[2048:3072] = W2.reshape(-1)
    rs = np.random.RandomState(5)
    P[10240:] = rs.uniform(0.1, 1.0, size=P.size - 10240).astype(np.float16)  # positive -> ReLU keeps everything
    coords = H.random_coords(256, seed=9)
    d_desc, d_P, d_c = H.to_dev(desc, cuda), H.to_dev(P, cuda), H.to_dev(coords, cuda)
    out = H.dev_zeros(256 * 2, cuda)
    check(ngp.ngp_hip_nerf_density(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, 256, out.data_ptr()))
    got = H.to_host(out, np.float16).astype(np.float32)
    ref = np.zeros(256, dtype=np.uint16)
    oracle.orc_nerf_density(desc.ctypes.data, P.ctypes.data, coords.ctypes.data, 7, 256, ref.ctypes.data)
    ref = ref.view(np.float16).astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("log2,n", [(15, 4096), (19, 2048), (12, 1000)])
def test_inference_matches_oracle(ngp, oracle, cuda, log2, n):
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=log2, n=n)
    out = H.dev_zeros(n * 4 * 2, cuda)
    check(ngp.ngp_hip_nerf_inference(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4))
    got = H.to_host(out, np.float16).reshape(n, 4).astype(np.float32)
    ref = np.zeros((n, 4), dtype=np.uint16)
    oracle.orc_nerf_inference(desc.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, ref.ctypes.data, 4)
    ref = ref.view(np.float16).astype(np.float32)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-2)
    assert np.abs(ref).max() > 0.05  # the comparison is not vacuous


def test_parameters_at_a_4_byte_aligned_address(ngp, cuda):
    """The kernels stage the network's matrices into LDS with 16-byte loads when the parameter block allows it and half by half otherwise (stage_tiles): a block that
    starts 4 bytes into an allocation (the hash tables are read as 4-byte pairs: that much alignment is required) gives the same bits, forward and backward."""
    n = 4096
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=15, n=n, grid_amp=0.5)
    shifted = np.zeros(params.size + 8, dtype=np.float16)
    shifted[2:2 + params.size] = params
    d_S = H.to_dev(shifted, cuda)
    assert d_P.data_ptr() % 16 == 0 and (d_S.data_ptr() + 4) % 16 == 4
    res = []
    dl = H.to_dev((np.random.RandomState(3).randn(n, 4) * 0.01).astype(np.float16), cuda)
    for ptr in (d_P.data_ptr(), d_S.data_ptr() + 4):
        out, xs = H.dev_zeros(n * 4 * 2, cuda), H.dev_zeros(n * 32 * 2, cuda)
        check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), ptr, d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
        sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
        scratch, grads = H.dev_zeros(sb, cuda), H.dev_zeros(H.n_params(desc) * 2, cuda)
        check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, ptr, d_c.data_ptr(), 7, n, xs.data_ptr(), dl.data_ptr(), 4, grads.data_ptr(), scratch.data_ptr(), sb))
        res.append((H.to_host(out, np.uint16), H.to_host(xs, np.uint16), H.to_host(grads, np.uint16)))
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)
    assert np.any(res[0][0] != 0) and np.any(res[0][2][:10240] != 0)


@pytest.mark.parametrize("log2,aabb_scale", [(19, 1), (19, 4), (15, 1)])
def test_backward_in_a_scratch_sized_for_its_level_table(ngp, cuda, log2, aabb_scale):
    """`ngp_hip_nerf_backward_scratch_bytes_for(desc_host, n)` packs the sort records of the 16 levels by kind (48 bytes per sample for a hashed level, 160 for a dense one)
    instead of pricing every level as dense: the same gradients, bit for bit, as in the any-table scratch, nothing written behind the buffer (a guard of 1 MiB of 0xA5
    follows it), and a scratch one byte short is refused."""
    n = 8192
    desc = H.make_desc(ngp, log2_hashmap_size=log2, aabb_scale=aabb_scale)
    params = H.random_params(desc, seed=5, grid_amp=0.5)
    coords = H.random_coords(n, seed=9)
    if aabb_scale == 4:
        # the worst case of a fine level: EVERY sample in a cell whose x corners straddle a 4096-entry slice of the finest level (two records per pair: 96 bytes per sample,
        # exactly what the level's space holds; it is the last level, an overflow would land in the guard)
        lv = desc["levels"][0][15]
        sc, res = float(lv["scale"]), int(lv["resolution"])
        assert res >= 4096
        rs = np.random.RandomState(4)
        bx = 4096 * rs.randint(1, max(2, res // 4096), size=n) - 1
        coords["pos"][:, 0] = np.clip(((bx + rs.rand(n) * 0.98 + 0.01) - 0.5) / sc, 0.0, 1.0).astype(np.float32)
        gx = np.floor(coords["pos"][:, 0] * np.float32(sc) + np.float32(0.5)).astype(np.int64)
        assert ((gx % 4096) == 4095).mean() > 0.99
    d_desc, d_P, d_c = H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(coords, cuda)
    dl = H.to_dev((np.random.RandomState(3).randn(n, 4) * 0.01).astype(np.float16), cuda)
    out, xs = H.dev_zeros(n * 4 * 2, cuda), H.dev_zeros(n * 32 * 2, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    any_table = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    packed = ngp.ngp_hip_nerf_backward_scratch_bytes_for(desc.ctypes.data, n)
    assert packed < any_table and ngp.ngp_hip_nerf_backward_scratch_bytes_for(None, n) == any_table
    if log2 == 19:
        assert packed < 0.56 * any_table
    guard = 1 << 20
    res = []
    for sb, host_desc in ((any_table, None), (packed, desc.ctypes.data)):
        buf = torch.full((sb + guard,), 0xA5, dtype=torch.uint8, device=cuda)
        grads = H.dev_zeros(H.n_params(desc) * 2, cuda)
        check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), host_desc, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), dl.data_ptr(), 4, grads.data_ptr(), buf.data_ptr(), sb))
        torch.cuda.synchronize()
        assert bool((buf[sb:] == 0xA5).all()), "the backward pass wrote behind its scratch"
        res.append(H.to_host(grads, np.uint16))
    np.testing.assert_array_equal(res[0], res[1])
    assert np.any(res[0][:10240] != 0) and np.any(res[0][10240:] != 0)
    buf = torch.zeros(packed, dtype=torch.uint8, device=cuda)
    assert ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), dl.data_ptr(), 4, grads.data_ptr(), buf.data_ptr(), packed - 1) != 0
    assert b"scratch too small" in ngp.ngp_hip_last_error()


def test_backward_refuses_a_host_level_table_that_is_not_the_device_one(ngp, cuda):
    """the host side of the backward pass (record offsets, owners' grid) is laid out from `desc_host`, the kernels read `desc_dev`.  A desc_host that describes another
    table (here: aabb_scale 4 against a device table of aabb_scale 1) is refused on the first call instead of writing gradients to the wrong places; NULL and the true copy pass."""
    n = 1024
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=15, n=n, grid_amp=0.5)
    other = H.make_desc(ngp, log2_hashmap_size=15, aabb_scale=4)
    assert other.tobytes() != desc.tobytes()
    dl = H.to_dev((np.random.RandomState(3).randn(n, 4) * 0.01).astype(np.float16), cuda)
    out, xs = H.dev_zeros(n * 4 * 2, cuda), H.dev_zeros(n * 32 * 2, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch, grads = H.dev_zeros(sb, cuda), H.dev_zeros(H.n_params(desc) * 2, cuda)
    args = (d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), dl.data_ptr(), 4, grads.data_ptr(), scratch.data_ptr(), sb)
    assert ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), other.ctypes.data, *args) != 0
    assert b"desc_host is not the level table" in ngp.ngp_hip_last_error()
    check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, *args))
    a = H.to_host(grads, np.uint16).copy()
    check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), None, *args))
    np.testing.assert_array_equal(a, H.to_host(grads, np.uint16))
    assert ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), other.ctypes.data, *args) != 0     # still refused after the good pair was remembered


@pytest.mark.parametrize("zero_fraction", [0.45, 0.0, 1.0, 0.999])
@pytest.mark.parametrize("aabb_scale", [1, 4])
def test_backward_over_the_live_samples_is_the_backward_over_the_batch(ngp, cuda, zero_fraction, aabb_scale):
    """ngp_hip_compact_live_samples + ngp_hip_nerf_backward_live against ngp_hip_nerf_backward on the whole batch.  A third to a half of a training batch carries a loss
    gradient that is zero in all four channels (ray tails, fp16): the live entry runs the MFMA kernel and the hash-grid binning over the other samples only.
      * the count is the number of rows with a non-zero channel, the list holds every such row once (256-row workgroups keep their order), the coordinate rows sit beside it;
      * hash-grid gradients: bit for bit those of the whole batch (exact sums: a zero term changes nothing) — with poison in every list slot behind the live samples;
      * MLP weight gradients: the same sums in another association: |difference| <= 2e-3 of the largest weight gradient (fp32 accumulation of fp16 products)."""
    n = 16384
    desc = H.make_desc(ngp, log2_hashmap_size=17, aabb_scale=aabb_scale)
    params = H.random_params(desc, seed=5, grid_amp=0.5)
    coords = H.random_coords(n, seed=9)
    rs = np.random.RandomState(int(zero_fraction * 1000) + aabb_scale)
    dl = (rs.randn(n, 4) * 0.02).astype(np.float16)
    # zero rows in runs, like ray tails; some rows with a single live channel, some with -0.0 only (dead)
    dead = np.zeros(n, bool)
    if zero_fraction >= 1.0:
        dead[:] = True
    elif zero_fraction > 0:
        starts = rs.randint(0, n, size=int(n * zero_fraction / 12))
        for st in starts:
            dead[st:st + rs.randint(1, 40)] = True
        if zero_fraction > 0.99:
            dead[:] = True; dead[rs.randint(0, n, size=5)] = False
    dl[dead] = 0
    dl[dead & (rs.rand(n) < 0.1), 3] = np.float16(-0.0)
    one = (~dead) & (rs.rand(n) < 0.05)
    dl[one, :3] = 0
    live_rows = np.flatnonzero((dl.view(np.uint16) & 0x7fff).any(axis=1))
    d_desc, d_P, d_c, d_dl = H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(coords, cuda), H.to_dev(dl, cuda)
    out, xs = H.dev_zeros(n * 4 * 2, cuda), H.dev_zeros(n * 32 * 2, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes_for(desc.ctypes.data, n)
    scratch, g_full = H.dev_zeros(sb, cuda), H.dev_zeros(H.n_params(desc) * 2, cuda)
    check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4, g_full.data_ptr(), scratch.data_ptr(), sb))
    full = H.to_host(g_full, np.uint16).copy()
    # ---- the live list, into poisoned buffers
    p_idx = torch.full((n,), -1, dtype=torch.int32, device=cuda)
    p_c = torch.full((n * 7,), float("nan"), dtype=torch.float32, device=cuda)
    counters = torch.tensor([0, 12345], dtype=torch.int32, device=cuda)
    check(ngp.ngp_hip_compact_live_samples(None, n, d_dl.data_ptr(), 4, d_c.data_ptr(), 7, p_idx.data_ptr(), p_c.data_ptr(), counters.data_ptr()))
    torch.cuda.synchronize()
    n_live = int(counters[0])
    assert n_live == live_rows.size
    got_idx = p_idx.cpu().numpy()[:n_live].astype(np.int64)
    got_c = p_c.cpu().numpy().reshape(n, 7)[:n_live]
    c_host = np.frombuffer(coords.tobytes(), np.float32).reshape(n, 7)
    np.testing.assert_array_equal(np.sort(got_idx), live_rows)                       # every live row once
    np.testing.assert_array_equal(got_c.view(np.uint32), c_host[got_idx].view(np.uint32))   # its coordinate row beside it
    blocks = got_idx // 256
    assert (np.diff(got_idx)[np.diff(blocks) == 0] > 0).all()                        # inside a workgroup of 256 the rows keep their order ...
    assert np.unique(blocks[np.flatnonzero(np.diff(blocks) != 0) + 1]).size == np.flatnonzero(np.diff(blocks) != 0).size if n_live else True   # ... and a workgroup's rows are one run
    # ---- the live backward
    g_live = H.dev_zeros(H.n_params(desc) * 2, cuda)
    scratch2 = torch.full((sb,), 0xA5, dtype=torch.uint8, device=cuda)
    check(ngp.ngp_hip_nerf_backward_live(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4, g_live.data_ptr(), scratch2.data_ptr(), sb,
                                         None, None, p_idx.data_ptr(), p_c.data_ptr(), counters.data_ptr(), counters.data_ptr() + 4))
    torch.cuda.synchronize()
    assert int(counters[1]) == 0 and int(counters[0]) == n_live    # the next step's word is cleared, this step's stays
    live = H.to_host(g_live, np.uint16)
    np.testing.assert_array_equal(live[10240:], full[10240:])
    a, b = live[:10240].view(np.float16).astype(np.float32), full[:10240].view(np.float16).astype(np.float32)
    assert np.isfinite(a).all()
    assert np.abs(a - b).max() <= 2e-3 * max(np.abs(b).max(), 1e-6) + 1e-7, (np.abs(a - b).max(), np.abs(b).max())
    if n_live:
        assert np.any(full[10240:] != 0) and np.any(b != 0)
    else:
        assert not np.any(live & 0x7fff)


def test_inference_ragged_and_empty(ngp, oracle, cuda):
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=14, n=33)
    for n in (0, 1, 31, 33):
        out = H.dev_zeros(max(n, 1) * 16 * 2, cuda)
        check(ngp.ngp_hip_nerf_inference(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 16))
        if n == 0:
            continue
        got = H.to_host(out, np.float16).reshape(n, 16)[:, :4].astype(np.float32)
        ref = np.zeros((n, 4), dtype=np.uint16)
        oracle.orc_nerf_inference(desc.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, ref.ctypes.data, 4)
        np.testing.assert_allclose(got, ref.view(np.float16).astype(np.float32), rtol=1e-2, atol=1e-2)


def test_density_matches_oracle(ngp, oracle, cuda):
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=16, n=3000)
    pos = np.ascontiguousarray(coords["pos"])
    d_pos = H.to_dev(pos, cuda)
    out = H.dev_zeros(3000 * 2, cuda)
    check(ngp.ngp_hip_nerf_density(None, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), 3, 3000, out.data_ptr()))
    got = H.to_host(out, np.float16).astype(np.float32)
    ref = np.zeros(3000, dtype=np.uint16)
    oracle.orc_nerf_density(desc.ctypes.data, params.ctypes.data, pos.ctypes.data, 3, 3000, ref.ctypes.data)
    np.testing.assert_allclose(got, ref.view(np.float16).astype(np.float32), rtol=5e-3, atol=5e-3)


def test_forward_saves_encoding(ngp, oracle, cuda):
    n = 2048
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=15, n=n)
    out = H.dev_zeros(n * 4 * 2, cuda)
    xs = H.dev_zeros(n * 32 * 2, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    got = H.to_host(xs, np.float16).reshape(n, 32).astype(np.float32)
    ref = np.zeros((n, 32), dtype=np.uint16)
    for i in range(n):
        oracle.orc_grid_encode_one(desc.ctypes.data, params[10240:].ctypes.data, coords["pos"][i].ctypes.data, ref[i].ctypes.data)
    # the encoding has no MFMA in it: identical fp32 op order => bit-exact up to FMA contraction (1 fp16 ulp)
    np.testing.assert_allclose(got, ref.view(np.float16).astype(np.float32), rtol=1.5e-3, atol=1e-4)


def test_backward_matches_oracle(ngp, oracle, cuda):
    n = 2048
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=14, n=n, grid_amp=0.5)
    rs = np.random.RandomState(11)
    dl = (rs.randn(n, 4) * 0.05).astype(np.float16)
    d_dl = H.to_dev(dl, cuda)
    out = H.dev_zeros(n * 4 * 2, cuda)
    xs = H.dev_zeros(n * 32 * 2, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    np_ = H.n_params(desc)
    grads = H.to_dev(np.full(np_, 7.0, dtype=np.float16), cuda)  # poison: Overwrite mode must not accumulate into it
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch = H.dev_zeros(sb, cuda)
    check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4,
                                    grads.data_ptr(), scratch.data_ptr(), sb))
    got = H.to_host(grads, np.float16).astype(np.float64)
    ref = np.zeros(np_, dtype=np.float64)
    oracle.orc_nerf_forward_backward(desc.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, dl.ctypes.data, None, ref.ctypes.data, None)
    assert np.isfinite(got).all()
    # MLP weight gradients: fp32 accumulate on device
    gm, rm = got[:10240], ref[:10240]
    scale = np.abs(rm).max()
    assert scale > 1e-3
    np.testing.assert_allclose(gm, rm, rtol=3e-2, atol=3e-3 * scale)
    # grid gradients: fp16 atomics; compare per level with a norm-relative tolerance + check zero pattern
    gg, rg = got[10240:], ref[10240:]
    assert np.count_nonzero(gg[rg == 0]) == 0
    err = np.linalg.norm(gg - rg) / np.linalg.norm(rg)
    assert err < 2e-2, err
    np.testing.assert_allclose(gg, rg, rtol=5e-2, atol=5e-3 * np.abs(rg).max())


def test_network_pass_organisations_agree_bit_for_bit(ngp, cuda):
    """The fused kernel (4-byte corner gathers) and the two-kernel pass (x-pair gathers into level planes, then the MLP kernel) end in ONE spelled-out trilinear blend
    (blend_corners: no implicit contraction), so their outputs are the same bits — round 6 found a third instantiation of the encoder rounding 23 of 20 000 samples'
    features differently while the blend was left to `fp contract(fast)`."""
    n = 20000
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=15, n=n, grid_amp=1.0)
    coords["pos"][:64] = 1.0
    coords["pos"][64:128] = 0.0
    d_c = H.to_dev(coords, cuda)
    ws_bytes = int(ngp.ngp_hip_nerf_encode_workspace_bytes(n))
    ws = H.dev_zeros(ws_bytes, cuda)
    a, b = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 8, cuda)
    check(ngp.ngp_hip_nerf_inference_ws(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, a.data_ptr(), 4, ws.data_ptr(), ws_bytes))
    check(ngp.ngp_hip_nerf_inference(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, b.data_ptr(), 4))
    np.testing.assert_array_equal(H.to_host(a, np.uint16), H.to_host(b, np.uint16))


def test_backward_through_a_row_index_is_the_backward_over_the_rows(ngp, cuda):
    """round 6: ngp_hip_nerf_backward(..., x_row_index): sample k's encoding is row x_row_index[k] of a LARGER x_saved (the uncompacted batch's rows, which the loss
    kernel's compaction no longer copies).  Same kernels, same sums: the gradients of the indexed call are bit for bit those of the call over the gathered rows."""
    n = 4096
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=14, n=n, grid_amp=0.5)
    rs = np.random.RandomState(21)
    dl = (rs.randn(n, 4) * 0.05).astype(np.float16)
    d_dl = H.to_dev(dl, cuda)
    out, xs = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    rows = H.to_host(xs, np.uint16).reshape(n, 32)
    n_all = 3 * n
    where = rs.permutation(n_all)[:n].astype(np.uint32)              # row k lives at where[k] of the big buffer; the rest is poison
    big = np.full((n_all, 32), 0x7E00, np.uint16)                     # fp16 NaN
    big[where] = rows
    d_big, d_where = H.to_dev(big, cuda), H.to_dev(where, cuda)
    np_ = H.n_params(desc)
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch = H.dev_zeros(sb, cuda)
    g_rows, g_index = H.dev_zeros(np_ * 2, cuda), H.dev_zeros(np_ * 2, cuda)
    check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4,
                                    g_rows.data_ptr(), scratch.data_ptr(), sb))
    check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, d_big.data_ptr(), d_dl.data_ptr(), 4,
                                    g_index.data_ptr(), scratch.data_ptr(), sb, None, None, None, None, d_where.data_ptr()))
    a, b = H.to_host(g_rows, np.uint16), H.to_host(g_index, np.uint16)
    assert np.isfinite(a.view(np.float16).astype(np.float32)).all() and (a != 0).any()
    np.testing.assert_array_equal(a, b)


def test_backward_input_gradient_matches_oracle(ngp, oracle, cuda):
    """ngp_hip_nerf_backward with dL_dinput: the same parameter gradients as ngp_hip_nerf_backward plus dL/d(pos, dir) per sample (what tcnn's backward writes when the
    caller passes dL_dinput, testbed_nerf.cu:3329-3330).  The position gradient runs through 16 levels of fp16 feature gradients, the direction gradient
    through the SH polynomials' derivatives: norm-relative 2e-2 like the grid gradients, plus a per-element bound."""
    n = 2048
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=14, n=n, grid_amp=0.5)
    rs = np.random.RandomState(12)
    dl = (rs.randn(n, 4) * 0.05).astype(np.float16)
    d_dl = H.to_dev(dl, cuda)
    out = H.dev_zeros(n * 4 * 2, cuda)
    xs = H.dev_zeros(n * 32 * 2, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    np_ = H.n_params(desc)
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch = H.dev_zeros(sb, cuda)
    g_plain, g_input = H.dev_zeros(np_ * 2, cuda), H.dev_zeros(np_ * 2, cuda)
    d_in = H.to_dev(np.full((n, 6), 7.0, np.float32), cuda)                   # poison: every element is overwritten
    check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4,
                                    g_plain.data_ptr(), scratch.data_ptr(), sb))
    check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4,
                                          g_input.data_ptr(), scratch.data_ptr(), sb, None, None, d_in.data_ptr()))
    a, b = H.to_host(g_plain, np.float16).astype(np.float64), H.to_host(g_input, np.float16).astype(np.float64)
    np.testing.assert_array_equal(a[:10240], b[:10240])                       # MLP part: same arithmetic, fixed reduction order
    assert np.linalg.norm(a[10240:] - b[10240:]) <= 2e-3 * np.linalg.norm(a[10240:])   # grid part: fp16 atomics, order varies between launches
    got = H.to_host(d_in, np.float32).reshape(n, 6).astype(np.float64)
    ref = np.zeros((n, 6), np.float32)
    oracle.orc_nerf_input_gradient(desc.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, dl.ctypes.data, ref.ctypes.data)
    ref = ref.astype(np.float64)
    assert np.isfinite(got).all()
    for name, sl in (("pos", slice(0, 3)), ("dir", slice(3, 6))):
        g, r = got[:, sl], ref[:, sl]
        assert np.linalg.norm(r) > 0
        rel = np.linalg.norm(g - r) / np.linalg.norm(r)
        assert rel < 2e-2, (name, rel)
        np.testing.assert_allclose(g, r, rtol=5e-2, atol=2e-2 * np.abs(r).max(), err_msg=name)
    # a prefix of the batch gives the prefix of the result (n is a multiple of 256, like ngp_hip_nerf_backward)
    for m_ in (256, 768):
        d_in2 = H.dev_zeros(m_ * 24, cuda)
        check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, m_, xs.data_ptr(), d_dl.data_ptr(), 4,
                                              g_input.data_ptr(), scratch.data_ptr(), sb, None, None, d_in2.data_ptr()))
        g2 = H.to_host(d_in2, np.float32).reshape(m_, 6)
        np.testing.assert_allclose(g2, got[:m_], rtol=1e-3, atol=1e-3 * np.abs(got).max())


def test_backward_linearity(ngp, cuda):
    """size-independent property at the full batch size B = 2^18 and the real T = 2^19 table: grads(2*dL) ~= 2*grads(dL)."""
    n = 1 << 18
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=19, n=n, grid_amp=0.5)
    rs = np.random.RandomState(3)
    dl = (rs.randn(n, 4) * 0.01).astype(np.float16)
    res = []
    out = H.dev_zeros(n * 4 * 2, cuda)
    xs = H.dev_zeros(n * 32 * 2, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch = H.dev_zeros(sb, cuda)
    for k in (1.0, 2.0):
        d_dl = H.to_dev((dl.astype(np.float32) * k).astype(np.float16), cuda)
        grads = H.dev_zeros(H.n_params(desc) * 2, cuda)
        check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4,
                                        grads.data_ptr(), scratch.data_ptr(), sb))
        res.append(H.to_host(grads, np.float16).astype(np.float64))
    assert np.isfinite(res[0]).all() and np.isfinite(res[1]).all()
    num = np.linalg.norm(res[1] - 2.0 * res[0])
    den = np.linalg.norm(2.0 * res[0])
    assert den > 0 and num / den < 2e-2, (num, den)


def test_backward_back_to_back_calls_and_events(ngp, cuda):
    """ngp_hip_nerf_backward with its optional events: calls that follow each other without a host sync share one scratch (the partials / planes of call k + 1 must not
    overtake the reduce of call k — everything is stream-ordered), with and without the two optional events: same bits."""
    import torch
    n = 1 << 16
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=19, n=n, grid_amp=0.5)
    rs = np.random.RandomState(5)
    out, xs = H.dev_zeros(n * 4 * 2, cuda), H.dev_zeros(n * 32 * 2, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr()))
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch = H.dev_zeros(sb, cuda)
    dls = [H.to_dev((rs.randn(n, 4) * 0.01).astype(np.float16), cuda) for _ in range(3)]
    res = {}
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    for e in evs:
        e.record()   # materialises the hipEvent_t
    for with_events in (0, 1):
        gs = [H.dev_zeros(H.n_params(desc) * 2, cuda) for _ in dls]
        for d_dl, g in zip(dls, gs):   # back to back, no sync in between
            check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4,
                                               g.data_ptr(), scratch.data_ptr(), sb, evs[0].cuda_event if with_events else None, evs[1].cuda_event if with_events else None))
        if with_events:
            evs[1].synchronize()   # "all of grads final" of the last call
            res[with_events] = [H.to_host(g, np.uint16) for g in gs]
        else:
            res[with_events] = [H.to_host(g, np.uint16) for g in gs]
    for a, b in zip(res[0], res[1]):
        assert np.any(a[:10240] != 0) and np.any(a[10240:] != 0)
        np.testing.assert_array_equal(a, b)
    assert np.any(res[0][0] != res[0][1])


def test_init_params_bit_exact(ngp, oracle, cuda):
    desc = H.make_desc(ngp, log2_hashmap_size=14)
    np_ = H.n_params(desc)
    master, p16, inf16 = H.dev_zeros(np_ * 4, cuda), H.dev_zeros(np_ * 2, cuda), H.dev_zeros(np_ * 2, cuda)
    check(ngp.ngp_hip_nerf_init_params(None, desc.ctypes.data, 1337, master.data_ptr(), p16.data_ptr(), inf16.data_ptr()))
    ref = np.zeros(np_, dtype=np.float32)
    oracle.orc_nerf_init_params(desc.ctypes.data, 1337, ref.ctypes.data)
    np.testing.assert_array_equal(H.to_host(master, np.float32), ref)
    np.testing.assert_array_equal(H.to_host(p16, np.float16), ref.astype(np.float16))
    np.testing.assert_array_equal(H.to_host(inf16, np.float16), ref.astype(np.float16))


@pytest.mark.parametrize("n,nm,off", [(50000, 10240, 0), (50003, 10242, 0), (50000, 10240, 1), (3, 2, 0)])
def test_optimizer_step_bit_exact(ngp, oracle, cuda, n, nm, off):
    """groups of four with a scalar tail when the arrays are 16-byte aligned (off = 0), the scalar kernel otherwise (off = 1 element)"""
    rs = np.random.RandomState(4)
    grads = (rs.randn(n) * 0.3).astype(np.float16)
    grads[rs.rand(n) < 0.3] = 0  # untouched hash slots are skipped
    master = rs.randn(n).astype(np.float32) * 0.1
    p16 = master.astype(np.float16)
    m1 = (rs.randn(n) * 1e-3).astype(np.float32)
    m2 = (rs.rand(n) * 1e-5).astype(np.float32)
    ema = master.copy()
    inf = p16.copy()
    d_all = [H.to_dev(np.concatenate([np.zeros(off, a.dtype), a]), cuda) for a in (grads, master, p16, m1, m2, ema, inf)]
    ptr = [t.data_ptr() + off * a.dtype.itemsize for t, a in zip(d_all, (grads, master, p16, m1, m2, ema, inf))]
    step = 7
    check(ngp.ngp_hip_optimizer_step(None, n, nm, step, H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-15), H.f32(1e-6), H.f32(128.0), H.f32(0.95), *ptr, 3))
    d = [None] + [H.to_host(t, a.dtype)[off:] for t, a in zip(d_all[1:], (master, p16, m1, m2, ema, inf))]
    oracle.orc_adam_ema_step(n, nm, step, H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-15), H.f32(1e-6), H.f32(128.0), H.f32(0.95),
                             grads.ctypes.data, master.ctypes.data, p16.ctypes.data, m1.ctypes.data, m2.ctypes.data, ema.ctypes.data, inf.ctypes.data)
    for name, t, ref, dt in zip(("master", "params", "m1", "m2", "ema", "inference"), d[1:], (master, p16, m1, m2, ema, inf), (np.float32, np.float16, np.float32, np.float32, np.float32, np.float16)):
        got = t
        bad = np.nonzero(got != ref)[0]
        assert bad.size == 0, (name, bad[:5], got[bad[:5]], ref[bad[:5]], grads[bad[:5]])


# ---- two-kernel (XCD-affine encode + MLP) variants: bit-identical to the single-kernel entry points -------------------------
@pytest.mark.parametrize("log2,n", [(19, 70001), (15, 4096), (12, 1), (14, 1025)])
def test_ws_variants_bit_identical(ngp, cuda, log2, n):
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=log2, n=n)
    ws_bytes = ngp.ngp_hip_nerf_encode_workspace_bytes(n)
    assert ws_bytes >= 256 + 64 * n
    ws = H.dev_zeros(ws_bytes, cuda)
    # inference
    a, b = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 8, cuda)
    check(ngp.ngp_hip_nerf_inference(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, a.data_ptr(), 4))
    check(ngp.ngp_hip_nerf_inference_ws(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, b.data_ptr(), 4, ws.data_ptr(), ws_bytes))
    np.testing.assert_array_equal(H.to_host(a, np.uint16), H.to_host(b, np.uint16))
    assert H.to_host(a, np.uint16).any()
    # density only
    a, b = H.dev_zeros(n * 2 + 2, cuda), H.dev_zeros(n * 2 + 2, cuda)
    check(ngp.ngp_hip_nerf_density(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, a.data_ptr()))
    check(ngp.ngp_hip_nerf_density_ws(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, b.data_ptr(), ws.data_ptr(), ws_bytes))
    np.testing.assert_array_equal(H.to_host(a, np.uint16), H.to_host(b, np.uint16))
    # training forward: outputs and the saved encoded features
    a, b = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 8, cuda)
    xa, xb = H.dev_zeros(n * 64, cuda), H.dev_zeros(n * 64, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, a.data_ptr(), 4, xa.data_ptr()))
    check(ngp.ngp_hip_nerf_forward_ws(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, b.data_ptr(), 4, xb.data_ptr(), ws.data_ptr(), ws_bytes))
    np.testing.assert_array_equal(H.to_host(a, np.uint16), H.to_host(b, np.uint16))
    np.testing.assert_array_equal(H.to_host(xa, np.uint16), H.to_host(xb, np.uint16))


def test_ws_rejects_small_workspace(ngp, cuda):
    desc, params, coords, d_desc, d_P, d_c = _setup(ngp, cuda, log2=12, n=2048)
    ws = H.dev_zeros(1024, cuda)
    out = H.dev_zeros(2048 * 8, cuda)
    assert ngp.ngp_hip_nerf_inference_ws(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, 2048, out.data_ptr(), 4, ws.data_ptr(), 1024) != 0
    assert b"workspace" in ngp.ngp_hip_last_error()
