"""The network variants (per-image extra dims, 0 / 1 / 3 hidden colour layers) against oracle/orc_netx.c, on both kernel families:
  "mfma"    csrc/network_netx_mfma.cuh — the product path: the base family's MFMA scheme with the layer list as a template parameter (fused dgrad + wgrad backward); sums in
            MFMA accumulation order, so outputs and gradients are held to the base family's tolerances;
  "scalar"  csrc/network_generic.cuh (NgpNetVariant.flags = NGP_NETX_SCALAR) — the checker, described next.
The scalar kernels sum in the oracle's order with separate multiply and add: per sample the results are the oracle's bits except where an
interpolated feature sits on an fp16 rounding tie (about 3 in 10 000 encoded features come out one ulp apart — the same bound the fused kernels' encoding has), so
outputs are held to "almost all bit-identical, the rest within fp16 noise"; weight gradients (sums over samples in another order) and the hash-grid gradient to the
tolerances of the base family's tests."""
import os
import sys

import numpy as np
import pytest

import capi
import helpers as H
from capi import check
from test_netx_cpu import netx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
VARIANTS = [(0, 0), (0, 1), (0, 3), (4, 2), (16, 2), (3, 1), (8, 3), (5, 0)]


KERNELS = ["mfma", "scalar"]


def _variant(n_extra, n_hidden, extra=None, slot=None, dl_dextra=None, kernels="mfma"):
    v = np.zeros(1, capi.NET_VARIANT)
    v["flags"] = capi.NETX_SCALAR if kernels == "scalar" else 0
    v["n_extra_dims"], v["n_rgb_hidden_layers"] = n_extra, n_hidden
    v["extra_dims"] = extra.data_ptr() if extra is not None else 0
    v["sample_slot"] = slot.data_ptr() if slot is not None else 0
    v["dL_dextra"] = dl_dextra.data_ptr() if dl_dextra is not None else 0
    return v


def _params(orc, desc, x, seed):
    npar = orc.orc_netx_n_params(desc.ctypes.data, x.ctypes.data)
    n_mlp = orc.orc_netx_mlp_params(x.ctypes.data)
    p32 = np.zeros(npar, np.float32)
    orc.orc_nerf_init_params_x(desc.ctypes.data, x.ctypes.data, seed, p32.ctypes.data)
    p32[:n_mlp] *= 2.5
    p32[n_mlp:] *= 3000.0
    return p32.astype(np.float16).view(np.uint16), npar, n_mlp


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("n_extra,n_hidden", VARIANTS)
def test_init_inference_forward_bit_exact(ngp, oracle, cuda, n_extra, n_hidden, kernels):
    n = 1000 + 3 * n_extra + n_hidden                     # not a multiple of the 4-sample groups
    desc = H.make_desc(ngp, log2_hashmap_size=12)
    coords = H.random_coords(n, seed=4)
    rs = np.random.RandomState(n_extra * 7 + n_hidden)
    table = (rs.rand(6, max(n_extra, 1)) * 2 - 1).astype(np.float32)
    slot = rs.randint(0, 6, n).astype(np.uint32)
    x = netx(n_extra, n_hidden, table if n_extra else None, slot if n_extra else None)
    params, npar, n_mlp = _params(oracle, desc, x, 11)
    # ---- parameter init: element for element
    d_master, d_p, d_i = H.dev_zeros(npar * 4, cuda), H.dev_zeros(npar * 2, cuda), H.dev_zeros(npar * 2, cuda)
    v0 = _variant(n_extra, n_hidden, kernels=kernels)
    check(ngp.ngp_hip_nerf_init_params(None, desc.ctypes.data, 11, d_master.data_ptr(), d_p.data_ptr(), d_i.data_ptr(), v0.ctypes.data))
    want = np.zeros(npar, np.float32)
    oracle.orc_nerf_init_params_x(desc.ctypes.data, x.ctypes.data, 11, want.ctypes.data)
    np.testing.assert_array_equal(H.to_host(d_master, np.float32), want)
    np.testing.assert_array_equal(H.to_host(d_p, np.uint16), want.astype(np.float16).view(np.uint16))
    # ---- inference / forward
    d_desc, d_P, d_c = H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(coords, cuda)
    d_tab, d_slot = H.to_dev(table, cuda), H.to_dev(slot, cuda)
    v = _variant(n_extra, n_hidden, d_tab if n_extra else None, d_slot if n_extra else None, kernels=kernels)
    out, out2, xs = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    check(ngp.ngp_hip_nerf_inference(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, v.ctypes.data))
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out2.data_ptr(), 4, xs.data_ptr(), v.ctypes.data))
    ref = np.zeros((n, 4), np.uint16)
    oracle.orc_nerf_inference_x(desc.ctypes.data, x.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, ref.ctypes.data, 4)
    got = H.to_host(out, np.uint16).reshape(n, 4)
    assert np.abs(ref.view(np.float16).astype(np.float32)[:, :3]).max() > 1e-2            # a live network
    if kernels == "scalar":
        assert (got == ref).mean() > 0.98                                                 # (a one-ulp feature moves its sample's four outputs)
    np.testing.assert_allclose(got.view(np.float16).astype(np.float32), ref.view(np.float16).astype(np.float32), rtol=2e-2, atol=3e-3)
    np.testing.assert_array_equal(H.to_host(out2, np.uint16).reshape(n, 4), got)         # forward == inference, bit for bit
    enc = np.zeros((n, 32), np.uint16)
    for i in range(n):
        oracle.orc_grid_encode_one(desc.ctypes.data, params[n_mlp:].ctypes.data, coords[i:i + 1].ctypes.data, enc[i].ctypes.data)
    gx = H.to_host(xs, np.uint16).reshape(n, 32)
    assert (gx != enc).mean() < 2e-3 and np.abs(gx.astype(np.int32) - enc.astype(np.int32)).max() <= 1
    # ---- density(): channel 0 of the density network, positions only
    pos = np.ascontiguousarray(coords["pos"])
    d_pos, d0 = H.to_dev(pos, cuda), H.dev_zeros(n * 2, cuda)
    check(ngp.ngp_hip_nerf_density(None, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), 3, n, d0.data_ptr(), v0.ctypes.data))
    np.testing.assert_array_equal(H.to_host(d0, np.uint16), got[:, 3])
    # ---- the two-kernel path of the renderers (XCD-affine encode into level planes, then the variant's MLP kernel): the same outputs
    wsb = ngp.ngp_hip_nerf_encode_workspace_bytes(n)
    ws, out3, d1 = H.dev_zeros(wsb, cuda), H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 2, cuda)
    check(ngp.ngp_hip_nerf_inference_ws(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out3.data_ptr(), 4, ws.data_ptr(), wsb, v.ctypes.data))
    check(ngp.ngp_hip_nerf_density_ws(None, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), 3, n, d1.data_ptr(), ws.data_ptr(), wsb, v0.ctypes.data))
    got3 = H.to_host(out3, np.uint16).reshape(n, 4)
    np.testing.assert_allclose(got3.view(np.float16).astype(np.float32), got.view(np.float16).astype(np.float32), rtol=1e-2, atol=2e-3)   # (the plane encoder's pair loads: one fp16 ulp in a few features)
    np.testing.assert_array_equal(H.to_host(d1, np.uint16), got3[:, 3])


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("n_extra,n_hidden", VARIANTS)
def test_backward_against_the_oracle(ngp, oracle, cuda, n_extra, n_hidden, kernels):
    import torch
    n = 2048
    desc = H.make_desc(ngp, log2_hashmap_size=12)
    coords = H.random_coords(n, seed=5)
    rs = np.random.RandomState(50 + n_extra + 3 * n_hidden)
    table = (rs.rand(5, max(n_extra, 1)) * 2 - 1).astype(np.float32)
    slot = rs.randint(0, 5, n).astype(np.uint32)
    x = netx(n_extra, n_hidden, table if n_extra else None, slot if n_extra else None)
    params, npar, n_mlp = _params(oracle, desc, x, 13)
    dl = (rs.randn(n, 4) * 0.05).astype(np.float16)
    d_desc, d_P, d_c, d_dl = H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(coords, cuda), H.to_dev(dl, cuda)
    d_tab, d_slot = H.to_dev(table, cuda), H.to_dev(slot, cuda)
    d_dext = torch.full((n * max(n_extra, 1),), 7.0, device=cuda, dtype=torch.float32)
    v = _variant(n_extra, n_hidden, d_tab if n_extra else None, d_slot if n_extra else None, d_dext if n_extra else None, kernels=kernels)
    out, xs = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr(), v.ctypes.data))
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch, grads = H.dev_zeros(sb, cuda), H.dev_zeros(npar * 2, cuda)
    for _ in range(2):   # twice into the same scratch: stream-ordered, same bits
        check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), d_c.data_ptr(), 7, n, xs.data_ptr(), d_dl.data_ptr(), 4, grads.data_ptr(), scratch.data_ptr(), sb,
                                        None, None, None, v.ctypes.data))
    torch.cuda.synchronize()
    gref = np.zeros(npar, np.float64)
    dx = np.zeros((n, 32), np.uint16)
    dext = np.zeros((n, max(n_extra, 1)), np.float32)
    oracle.orc_nerf_forward_backward_x(desc.ctypes.data, x.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, dl.ctypes.data, None, gref.ctypes.data, dx.ctypes.data, dext.ctypes.data if n_extra else None)
    got = H.to_host(grads, np.float16).astype(np.float64)
    assert np.isfinite(got).all()
    gm, rm = got[:n_mlp], gref[:n_mlp]
    assert np.abs(rm).max() > 1e-3
    np.testing.assert_allclose(gm, rm, rtol=2e-2, atol=2e-3 * np.abs(rm).max())               # weight gradients: fp32 partial sums in another order, one fp16 rounding
    gg, rg = got[n_mlp:], gref[n_mlp:]
    assert np.linalg.norm(gg - rg) / np.linalg.norm(rg) < 2e-2                               # hash-grid gradient, relative to the norm (as for the base family)
    if n_extra:
        ge = H.to_host(d_dext, np.float32).reshape(n, n_extra)
        if kernels == "scalar":
            assert (ge == dext).mean() > 0.97                                                 # Identity backward: the fp16 delta as float
        np.testing.assert_allclose(ge, dext, rtol=2e-2, atol=2e-3 * np.abs(dext).max())


@pytest.mark.parametrize("n_extra,n_hidden", [(0, 1), (6, 3), (4, 0)])
def test_input_gradient_of_a_variant(ngp, oracle, cuda, n_extra, n_hidden):
    """ngp_hip_nerf_input_gradient on the variants' kernels ([tcnn] input_gradient: the Normals render mode; the same backward path feeds the camera-side trainables).
    The density network and the grid are the base family's, so d(sigma)/d(position) of a variant must be the base family's for the same W1, W2 and grid — checked
    against the oracle's orc_nerf_input_gradient on a base-family parameter vector that shares them; d(red)/d(direction) must be non-zero and finite."""
    n = 1024
    desc = H.make_desc(ngp, log2_hashmap_size=12)
    coords = H.random_coords(n, seed=21)
    rs = np.random.RandomState(3)
    table = (rs.rand(4, max(n_extra, 1)) * 2 - 1).astype(np.float32)
    slot = rs.randint(0, 4, n).astype(np.uint32)
    x = netx(n_extra, n_hidden, table if n_extra else None, slot if n_extra else None)
    params, npar, n_mlp = _params(oracle, desc, x, 17)
    base = H.random_params(desc, seed=2)                                     # base family: [W1 | W2 | W3 W4 W5 | grid]
    base = base.view(np.uint16).copy()
    base[:3072] = params[:3072]                                              # the density network ...
    base[10240:] = params[n_mlp:]                                            # ... and the grid
    flat = coords.view(np.float32).reshape(n, 7).copy()
    dl = np.zeros((n, 4), np.float16); dl[:, 3] = 128.0
    din = np.zeros((n, 6), np.float32)
    oracle.orc_nerf_input_gradient(desc.ctypes.data, base.ctypes.data, flat.ctypes.data, 7, n, dl.ctypes.data, din.ctypes.data)
    ref = din[:, 0:3] / 128.0
    d_desc, d_P, d_c = H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(flat, cuda)
    d_tab, d_slot = H.to_dev(table, cuda), H.to_dev(slot, cuda)
    v = _variant(n_extra, n_hidden, d_tab if n_extra else None, d_slot if n_extra else None)
    sb = ngp.ngp_hip_nerf_input_gradient_scratch_bytes(n)
    d_s = H.dev_zeros(sb, cuda)
    check(ngp.ngp_hip_nerf_input_gradient(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), 3, d_c.data_ptr(), 7, n, d_s.data_ptr(), sb, v.ctypes.data))
    got = H.to_host(d_c, np.float32).reshape(n, 7)
    scale = np.abs(ref).max()
    assert scale > 1e-3
    assert np.linalg.norm(got[:, 0:3] - ref) < 1e-2 * np.linalg.norm(ref) and np.abs(got[:, 0:3] - ref).max() < 3e-2 * scale
    assert np.abs(got[:, 4:7]).max() == 0.0                                  # sigma does not depend on the direction
    d_c2 = H.to_dev(flat, cuda)
    check(ngp.ngp_hip_nerf_input_gradient(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), 0, d_c2.data_ptr(), 7, n, d_s.data_ptr(), sb, v.ctypes.data))
    got0 = H.to_host(d_c2, np.float32).reshape(n, 7)
    assert np.isfinite(got0).all() and np.abs(got0[:, 4:7]).max() > 0        # red does
    vs = _variant(n_extra, n_hidden, d_tab if n_extra else None, d_slot if n_extra else None, kernels="scalar")
    assert ngp.ngp_hip_nerf_input_gradient(None, d_desc.data_ptr(), desc.ctypes.data, d_P.data_ptr(), 3, d_c.data_ptr(), 7, n, d_s.data_ptr(), sb, vs.ctypes.data) != 0   # the checker kernels have none


@pytest.mark.parametrize("n_extra,n_hidden", [(0, 1), (5, 3), (4, 0), (8, 2)])
def test_visualize_activation_of_a_variant(ngp, oracle, cuda, n_extra, n_hidden):
    """ngp_hip_nerf_visualize_activation with an NgpNetVariant ([tcnn] visualize_activation: the EncodingVis render mode).  Layer map of NerfNetwork::width
    (nerf_network.h:474-484): 0 encoding, 1 density hidden, 2 colour input, 3.. colour hidden layers.  Self-consistency with the inference kernels: the encoding read-out is
    the saved encoding of the training forward; the LAST layer's read-out, pushed through the network's last matrix on the host, is the inference output's colour."""
    import torch
    n = 512
    desc = H.make_desc(ngp, log2_hashmap_size=12)
    coords = H.random_coords(n, seed=31)
    rs = np.random.RandomState(8)
    table = (rs.rand(3, max(n_extra, 1)) * 2 - 1).astype(np.float32)
    x = netx(n_extra, n_hidden, table if n_extra else None, None)
    params, npar, n_mlp = _params(oracle, desc, x, 23)
    d_desc, d_P, d_c, d_tab = H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(coords, cuda), H.to_dev(table, cuda)
    v = _variant(n_extra, n_hidden, d_tab if n_extra else None, None)            # sample_slot NULL: row 0 for every sample (rendering)

    def read(layer, width):
        cols = []
        for dim in range(width):
            o = torch.zeros(n, device=cuda, dtype=torch.float32)
            check(ngp.ngp_hip_nerf_visualize_activation(None, d_desc.data_ptr(), d_P.data_ptr(), layer, dim, d_c.data_ptr(), 7, n, o.data_ptr(), 1, v.ctypes.data))
            cols.append(H.to_host(o, np.float32))
        return np.stack(cols, 1)

    out, xs = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr(), v.ctypes.data))
    enc = H.to_host(xs, np.float16).reshape(n, 32).astype(np.float32)
    got_enc = read(0, 32)
    assert (np.abs(got_enc - enc) <= np.abs(enc) * 2e-3 + 1e-6).mean() > 0.995          # (contraction mode of the two encoders: one fp16 ulp in a few features)
    rgb_in = 32 + (16 if n_extra else 0)
    last_layer, last_width = (2, rgb_in) if n_hidden == 0 else (2 + n_hidden, 64)
    act = read(last_layer, last_width)
    P = params.view(np.float16).astype(np.float32)
    w_last = P[n_mlp - 16 * last_width:n_mlp].reshape(16, last_width)             # the network's last matrix: [16][64], or [16][rgb_in] without hidden layers
    want = act @ w_last.T
    got = H.to_host(out, np.float16).reshape(n, 4).astype(np.float32)
    assert np.abs(want[:, :3]).max() > 1e-2
    np.testing.assert_allclose(got[:, :3], want[:, :3], rtol=2e-2, atol=4e-3)
    if n_hidden:
        assert (act >= 0).all()                                                   # behind a ReLU
    if n_extra:                                                                   # the colour input's extra block is the table's row 0, padded with zeros
        rin = read(2, rgb_in)
        np.testing.assert_allclose(rin[:, 32:32 + n_extra], np.tile(table[0].astype(np.float16).astype(np.float32), (n, 1)), rtol=0, atol=0)
        assert (rin[:, 32 + n_extra:] == 0).all()
    o = torch.zeros(n, device=cuda, dtype=torch.float32)
    assert ngp.ngp_hip_nerf_visualize_activation(None, d_desc.data_ptr(), d_P.data_ptr(), 3 + n_hidden, 0, d_c.data_ptr(), 7, n, o.data_ptr(), 1, v.ctypes.data) != 0   # no such layer


def test_slot_expansion_rollover_and_latent_code_gradient(ngp, oracle, cuda):
    """ngp_hip_ray_images / _expand_ray_slots / _rollover_slots / _extra_dims_gradient against their definitions"""
    import torch
    rs = np.random.RandomState(9)
    R, n_rays, n_img, ne, B = 300, 211, 7, 5, 4096
    counts = rs.randint(0, 30, R).astype(np.uint32)
    counts[n_rays:] = 0
    bases = np.zeros(R, np.uint32); bases[1:n_rays] = np.cumsum(counts[:n_rays - 1])
    n_kept = int(counts[:n_rays].sum())
    numsteps = np.stack([counts, bases], 1).reshape(-1).astype(np.uint32)
    ray_idx = rs.permutation(4 * R)[:R].astype(np.uint32)
    n_rays_global = 4 * R
    d_cnt = H.to_dev(np.array([n_rays], np.uint32), cuda)
    d_ns, d_ri = H.to_dev(numsteps, cuda), H.to_dev(ray_idx, cuda)
    d_img = torch.full((R,), 99, device=cuda, dtype=torch.int32)
    check(ngp.ngp_hip_ray_images(None, R, d_cnt.data_ptr(), d_ri.data_ptr(), n_rays_global, n_img, None, d_img.data_ptr()))
    want_img = ((ray_idx.astype(np.uint64) * n_img // n_rays_global) % n_img).astype(np.uint32)          # image_idx without a CDF (testbed_nerf.cu:1082)
    got_img = H.to_host(d_img, np.uint32)
    np.testing.assert_array_equal(got_img[:n_rays], want_img[:n_rays])
    assert (got_img[n_rays:] == 99).all()
    d_slot = torch.zeros(B, device=cuda, dtype=torch.int32)
    check(ngp.ngp_hip_expand_ray_slots(None, R, d_cnt.data_ptr(), d_img.data_ptr(), d_ns.data_ptr(), B, d_slot.data_ptr()))
    d_kept = H.to_dev(np.array([n_kept], np.uint32), cuda)
    check(ngp.ngp_hip_rollover_slots(None, B, d_kept.data_ptr(), d_slot.data_ptr()))
    want = np.zeros(B, np.uint32)
    for i in range(n_rays):
        want[bases[i]:bases[i] + counts[i]] = want_img[i]
    want[n_kept:] = want[np.arange(n_kept, B) % n_kept]
    np.testing.assert_array_equal(H.to_host(d_slot, np.uint32), want)
    dext = rs.randn(B, ne).astype(np.float32)
    d_dext, d_grad = H.to_dev(dext, cuda), torch.zeros(n_img * ne, device=cuda, dtype=torch.float32)
    check(ngp.ngp_hip_extra_dims_gradient(None, R, d_cnt.data_ptr(), d_img.data_ptr(), d_ns.data_ptr(), d_dext.data_ptr(), ne, d_grad.data_ptr()))
    ref = np.zeros((n_img, ne), np.float32)
    oracle.orc_compute_extra_dims_gradient(n_rays, want_img.ctypes.data, numsteps.ctypes.data, dext.ctypes.data, ne, ref.ctypes.data)
    np.testing.assert_allclose(H.to_host(d_grad, np.float32).reshape(n_img, ne), ref, rtol=1e-4, atol=1e-4)
