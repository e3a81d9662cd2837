"""GPU parity tests of the plumbing-config kernels: grid encoding (2-D / 3-D) -> one MLP 32 -> 64 -> 64 -> 16,
tcnn losses, the image-fitting batch generator — `ngp_hip_gridmlp_*`, `ngp_hip_loss_and_gradient`, `ngp_hip_image_*` against the oracle.
Tolerances as for the NeRF network (tests/test_network_gpu.py): MFMA summation order differs from the oracle's sequential fp32 sums."""
import ctypes

import numpy as np
import pytest

import capi
import helpers as H

pytestmark = pytest.mark.gpu
check = capi.check


def _desc(ngp, n_dims, log2, desired):
    desc = np.zeros(1, dtype=capi.NET_DESC)
    pls = float(np.exp(np.log(desired / 16.0) / 15).astype(np.float32))
    check(ngp.ngp_hip_gridmlp_make_desc_host(n_dims, 16, log2, 16, H.f32(pls), desc.ctypes.data))
    return desc


def _params(desc, seed, grid_amp=1.0):
    rs = np.random.RandomState(seed)
    parts = []
    for o, i in [(64, 32), (64, 64), (16, 64)]:
        s = np.sqrt(6.0 / (o + i))
        parts.append(rs.uniform(-s, s, size=o * i))
    parts.append(rs.uniform(-grid_amp, grid_amp, size=2 * int(desc["n_grid_entries"][0])))
    return np.concatenate(parts).astype(np.float16)


CASES = [(2, 24, 512.0), (2, 12, 512.0), (3, 19, 2048.0), (3, 12, 2048.0)]   # image config (all dense), hashed 2-D, sdf config, small hashed 3-D


@pytest.mark.parametrize("n_dims,log2,desired", CASES)
def test_desc_matches_oracle(ngp, oracle, n_dims, log2, desired):
    desc = _desc(ngp, n_dims, log2, desired)
    ref = np.zeros(1, dtype=capi.NET_DESC)
    pls = float(np.exp(np.log(desired / 16.0) / 15).astype(np.float32))
    n_entries = np.zeros(1, np.uint32)
    oracle.orc_gridmlp_make_levels(n_dims, 16, log2, 16, H.f32(pls), ref["levels"].ctypes.data, n_entries.ctypes.data)
    ref["n_levels"], ref["n_grid_entries"] = 16, n_entries[0]
    assert desc.tobytes() == ref.tobytes()
    assert ngp.ngp_hip_gridmlp_n_params_host(desc.ctypes.data) == 7168 + 2 * int(n_entries[0]) == oracle.orc_gridmlp_n_params(desc.ctypes.data)
    if (n_dims, log2) == (2, 24):
        assert (desc["levels"][0]["size"] >= desc["levels"][0]["resolution"].astype(np.uint64) ** 2).all()      # 1024^2 albert: every level dense


@pytest.mark.parametrize("n_dims,log2,desired", CASES)
@pytest.mark.parametrize("n", [1, 1000, 4096])
def test_forward_matches_oracle(ngp, oracle, cuda, n_dims, log2, desired, n):
    desc = _desc(ngp, n_dims, log2, desired)
    P = _params(desc, 3)
    rs = np.random.RandomState(n)
    pos = rs.rand(n, n_dims).astype(np.float32)
    d_desc, d_P, d_pos = H.to_dev(desc, cuda), H.to_dev(P, cuda), H.to_dev(pos, cuda)
    out, xs = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    check(ngp.ngp_hip_gridmlp_forward(None, n_dims, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), n_dims, n, out.data_ptr(), 4, xs.data_ptr()))
    got = H.to_host(out, np.float16).reshape(n, 4).astype(np.float32)
    ref = np.zeros((n, 4), np.uint16)
    oracle.orc_gridmlp_inference(n_dims, desc.ctypes.data, P.view(np.uint16).ctypes.data, pos.ctypes.data, n_dims, n, ref.ctypes.data, 4)
    ref = ref.view(np.float16).astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-2)
    assert np.abs(ref).max() > 0.05
    # saved encoding: exactly the oracle's (one fp16 rounding of an fp32 sum of <= 8 products in the same order)
    enc = np.zeros((n, 32), np.uint16)
    for i in range(min(n, 64)):
        oracle.orc_grid_encode_nd(n_dims, desc.ctypes.data, P.view(np.uint16)[7168:].ctypes.data, pos[i].ctypes.data, enc[i].ctypes.data)
    gx = H.to_host(xs, np.float16).reshape(n, 32).astype(np.float32)[:min(n, 64)]
    np.testing.assert_allclose(gx, enc[:min(n, 64)].view(np.float16).astype(np.float32), rtol=2e-3, atol=1e-6)
    # inference entry (no x_saved) gives the same outputs
    out2 = H.dev_zeros(n * 8, cuda)
    check(ngp.ngp_hip_gridmlp_forward(None, n_dims, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), n_dims, n, out2.data_ptr(), 4, None))
    np.testing.assert_array_equal(H.to_host(out2, np.uint16), H.to_host(out, np.uint16))


@pytest.mark.parametrize("n_dims,log2,desired", CASES)
@pytest.mark.parametrize("n", [1, 1000, 5000, 1 << 16])
def test_two_kernel_forward_is_the_fused_forward_bit_for_bit(ngp, cuda, n_dims, log2, desired, n):
    """ngp_hip_gridmlp_forward_ws (XCD-affine encode into level planes + MLP kernel over them) against ngp_hip_gridmlp_forward: the same fp32 sums in the same order,
    one fp16 rounding each — outputs AND saved encodings identical, ragged sizes included (the planes are padded to 1024-sample chunks).  The host chooses between
    the two by measurement (Testbed.network_pass), so they must be interchangeable at any step."""
    desc = _desc(ngp, n_dims, log2, desired)
    P = _params(desc, 5)
    pos = np.random.RandomState(n + n_dims).rand(n, n_dims).astype(np.float32)
    d_desc, d_P, d_pos = H.to_dev(desc, cuda), H.to_dev(P, cuda), H.to_dev(pos, cuda)
    out_a, xs_a = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    out_b, xs_b = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    ws_bytes = int(ngp.ngp_hip_nerf_encode_workspace_bytes(n))
    ws = H.dev_zeros(ws_bytes, cuda)
    check(ngp.ngp_hip_gridmlp_forward(None, n_dims, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), n_dims, n, out_a.data_ptr(), 4, xs_a.data_ptr()))
    check(ngp.ngp_hip_gridmlp_forward_ws(None, n_dims, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), n_dims, n, out_b.data_ptr(), 4, xs_b.data_ptr(), ws.data_ptr(), ws_bytes))
    np.testing.assert_array_equal(H.to_host(out_b, np.uint16), H.to_host(out_a, np.uint16))
    np.testing.assert_array_equal(H.to_host(xs_b, np.uint16), H.to_host(xs_a, np.uint16))
    assert n < 100 or np.abs(H.to_host(out_a, np.float16).astype(np.float32)).max() > 0.05
    # inference form (no saved encoding), and a workspace that is too small is refused, not overrun
    out_c = H.dev_zeros(n * 8, cuda)
    check(ngp.ngp_hip_gridmlp_forward_ws(None, n_dims, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), n_dims, n, out_c.data_ptr(), 4, None, ws.data_ptr(), ws_bytes))
    np.testing.assert_array_equal(H.to_host(out_c, np.uint16), H.to_host(out_a, np.uint16))
    assert ngp.ngp_hip_gridmlp_forward_ws(None, n_dims, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), n_dims, n, out_c.data_ptr(), 4, None, ws.data_ptr(), ws_bytes - 1) != 0
    assert ngp.ngp_hip_gridmlp_forward_ws(None, 4, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), n_dims, n, out_c.data_ptr(), 4, None, ws.data_ptr(), ws_bytes) != 0


@pytest.mark.parametrize("n_dims,log2,desired", CASES)
def test_backward_matches_oracle(ngp, oracle, cuda, n_dims, log2, desired):
    n = 2048
    desc = _desc(ngp, n_dims, log2, desired)
    P = _params(desc, 5)
    rs = np.random.RandomState(11)
    pos = rs.rand(n, n_dims).astype(np.float32)
    dl = np.zeros((n, 4), np.float16)
    dl[:, :3] = rs.uniform(-1, 1, (n, 3)) * 0.05
    d_desc, d_P, d_pos, d_dl = H.to_dev(desc, cuda), H.to_dev(P, cuda), H.to_dev(pos, cuda), H.to_dev(dl, cuda)
    out, xs = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    check(ngp.ngp_hip_gridmlp_forward(None, n_dims, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), n_dims, n, out.data_ptr(), 4, xs.data_ptr()))
    sb = ngp.ngp_hip_gridmlp_backward_scratch_bytes(n)
    scratch, grads = H.dev_zeros(sb, cuda), H.dev_zeros(P.size * 2, cuda)
    grads[:] = 0x3c                                     # poison: the whole gradient vector must be overwritten
    check(ngp.ngp_hip_gridmlp_backward(None, n_dims, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), n_dims, n, xs.data_ptr(), d_dl.data_ptr(), 4, grads.data_ptr(), scratch.data_ptr(), sb))
    got = H.to_host(grads, np.float16).astype(np.float64)
    ref = np.zeros(P.size, np.float64)
    oracle.orc_gridmlp_forward_backward(n_dims, desc.ctypes.data, P.view(np.uint16).ctypes.data, pos.ctypes.data, n_dims, n, dl.view(np.uint16).ctypes.data, None, ref.ctypes.data, None)
    assert np.isfinite(got).all()
    gm, rm = got[:7168], ref[:7168]
    assert np.abs(rm).max() > 1e-3
    np.testing.assert_allclose(gm, rm, rtol=3e-2, atol=3e-2 * np.abs(rm).max())
    gg, rg = got[7168:], ref[7168:]
    assert np.linalg.norm(rg) > 0
    assert np.linalg.norm(gg - rg) < 2e-2 * np.linalg.norm(rg)
    assert (gg[rg == 0] == 0).all()                     # untouched entries are written as exact zeros


@pytest.mark.parametrize("loss_type,dims", [(0, 3), (6, 3), (1, 3), (2, 1), (0, 4)])
def test_loss_and_gradient(ngp, oracle, cuda, loss_type, dims):
    n = 3000
    rs = np.random.RandomState(loss_type + dims)
    pred = rs.uniform(-2, 2, (n, 4)).astype(np.float16)
    tgt = rs.uniform(-2, 2, (n, dims)).astype(np.float32)
    vals, grad = np.zeros((n, dims), np.float32), np.full((n, 4), 0x3c00, np.uint16)
    oracle.orc_tcnn_loss_and_gradient(loss_type, n, dims, H.f32(128.0), pred.view(np.uint16).ctypes.data, 4, tgt.ctypes.data, vals.ctypes.data, grad.ctypes.data, 4)
    d_pred, d_tgt = H.to_dev(pred, cuda), H.to_dev(tgt, cuda)
    d_vals, d_grad = H.dev_zeros(vals.nbytes, cuda), H.to_dev(np.full((n, 4), 0x3c00, np.uint16), cuda)
    check(ngp.ngp_hip_loss_and_gradient(None, loss_type, n, dims, H.f32(128.0), d_pred.data_ptr(), 4, d_tgt.data_ptr(), d_vals.data_ptr(), d_grad.data_ptr(), 4))
    np.testing.assert_array_equal(H.to_host(d_grad, np.uint16).reshape(n, 4), grad)        # one fp16 rounding of identical fp32 arithmetic
    np.testing.assert_array_equal(H.to_host(d_vals, np.float32).reshape(n, dims), vals)
    assert (grad[:, dims:] == 0).all() and np.abs(vals).sum() > 0
    assert ngp.ngp_hip_loss_and_gradient(None, 4, n, dims, H.f32(128.0), d_pred.data_ptr(), 4, d_tgt.data_ptr(), d_vals.data_ptr(), d_grad.data_ptr(), 4) != 0   # Huber: not a tcnn-path loss here


@pytest.mark.parametrize("dtype,snap,linear", [(np.float16, 0, 1), (np.float32, 0, 0), (np.float16, 1, 0), (np.float32, 1, 1)])
def test_image_training_batch(ngp, oracle, cuda, dtype, snap, linear):
    """generate_random_uniform + stratify2 + eval_image_kernel_and_snap (src/testbed_image.cu:62-77, 172-218, 236-262)"""
    n = 1 << 12
    st, inc = H.pcg32_state(1337)
    ref = np.zeros(2 * n, np.float32)
    oracle.orc_generate_random_uniform(st, inc, 2 * n, ref.ctypes.data)
    d_xy = H.dev_zeros(2 * n * 4, cuda)
    check(ngp.ngp_hip_generate_random_uniform(None, st, inc, 2 * n, d_xy.data_ptr()))
    np.testing.assert_array_equal(H.to_host(d_xy, np.float32), ref)
    assert 0.0 <= ref.min() and ref.max() < 1.0 and abs(ref.mean() - 0.5) < 0.02
    oracle.orc_image_stratify2(n, 12, ref.ctypes.data)
    check(ngp.ngp_hip_image_stratify2(None, n, 12, d_xy.data_ptr()))
    np.testing.assert_array_equal(H.to_host(d_xy, np.float32), ref)
    cells = (ref.reshape(n, 2) * 64).astype(int)
    assert len({(a, b) for a, b in cells}) == n                                  # one sample per stratum of the 64 x 64 lattice
    rs = np.random.RandomState(4)
    w, h = 37, 29
    img = rs.rand(h, w, 4).astype(dtype)
    res = np.array([w, h], np.int32)
    tgt = np.zeros((n, 3), np.float32)
    xy_ref = ref.copy()
    oracle.orc_image_eval_and_snap(n, img.ctypes.data, 2 if dtype == np.float16 else 3, xy_ref.ctypes.data, res.ctypes.data, tgt.ctypes.data, 3, snap, linear)
    d_img, d_tgt = H.to_dev(img, cuda), H.dev_zeros(tgt.nbytes, cuda)
    check(ngp.ngp_hip_image_eval_and_snap(None, n, d_img.data_ptr(), 2 if dtype == np.float16 else 3, d_xy.data_ptr(), res.ctypes.data, d_tgt.data_ptr(), 3, snap, linear))
    np.testing.assert_array_equal(H.to_host(d_xy, np.float32), xy_ref)
    got = H.to_host(d_tgt, np.float32).reshape(n, 3)
    np.testing.assert_allclose(got, tgt, rtol=0, atol=0 if linear else 2e-6)     # powf in linear_to_srgb
    assert tgt.std() > 0.1
