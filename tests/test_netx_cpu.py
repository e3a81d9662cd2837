"""oracle/orc_netx.c — NerfNetwork for configs other than base.json (per-image extra dims, 0..3 hidden colour layers), CPU side: it must reduce to orc_network.c
for the base family bit for bit, have the parameter counts the constructor's arithmetic gives (nerf_network.h:82-99), and its hand-written backward must agree with
finite differences of its own forward where the fp16 storage allows (the extra-dim inputs: smooth, not quantised)."""
import numpy as np
import pytest

import capi
import helpers as H

NETX = np.dtype([("n_extra_dims", "<u4"), ("n_rgb_hidden_layers", "<u4"), ("extra_dims", "<u8"), ("sample_slot", "<u8")], align=True)


def netx(n_extra, n_hidden, extra=None, slot=None):
    x = np.zeros(1, NETX)
    x["n_extra_dims"], x["n_rgb_hidden_layers"] = n_extra, n_hidden
    x["extra_dims"] = extra.ctypes.data if extra is not None else 0
    x["sample_slot"] = slot.ctypes.data if slot is not None else 0
    return x


def _setup(n=64, log2=10):
    ngp = capi.load_ngp_hip()
    orc = H.load_oracle()
    desc = H.make_desc(ngp, log2_hashmap_size=log2)
    coords = H.random_coords(n, seed=2)
    return ngp, orc, desc, coords


def test_base_family_reduces_to_orc_network_bit_for_bit():
    ngp, orc, desc, coords = _setup()
    n = len(coords)
    params = H.random_params(desc, seed=1, grid_amp=0.5)
    x = netx(0, 2)
    assert orc.orc_netx_mlp_params(x.ctypes.data) == 10240 and orc.orc_netx_n_params(desc.ctypes.data, x.ctypes.data) == H.n_params(desc)
    a, b = np.zeros((n, 4), np.uint16), np.zeros((n, 4), np.uint16)
    orc.orc_nerf_inference(desc.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, a.ctypes.data, 4)
    orc.orc_nerf_inference_x(desc.ctypes.data, x.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, b.ctypes.data, 4)
    np.testing.assert_array_equal(a, b)
    dl = (np.random.RandomState(3).randn(n, 4) * 0.05).astype(np.float16)
    npar = H.n_params(desc)
    g0, g1 = np.zeros(npar, np.float64), np.zeros(npar, np.float64)
    dx0, dx1 = np.zeros((n, 32), np.uint16), np.zeros((n, 32), np.uint16)
    orc.orc_nerf_forward_backward(desc.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, dl.ctypes.data, None, g0.ctypes.data, dx0.ctypes.data)
    orc.orc_nerf_forward_backward_x(desc.ctypes.data, x.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, dl.ctypes.data, None, g1.ctypes.data, dx1.ctypes.data, None)
    np.testing.assert_array_equal(dx0, dx1)
    np.testing.assert_allclose(g1, g0, rtol=1e-12, atol=1e-15)   # double sums, same terms (thread partition may differ)
    p0, p1 = np.zeros(npar, np.float32), np.zeros(npar, np.float32)
    orc.orc_nerf_init_params(desc.ctypes.data, 1337, p0.ctypes.data)
    orc.orc_nerf_init_params_x(desc.ctypes.data, x.ctypes.data, 1337, p1.ctypes.data)
    np.testing.assert_array_equal(p0, p1)


@pytest.mark.parametrize("n_extra,n_hidden,want", [(0, 0, 3072 + 16 * 32), (0, 1, 3072 + 64 * 32 + 16 * 64), (0, 3, 3072 + 64 * 32 + 2 * 4096 + 1024), (4, 2, 3072 + 64 * 48 + 4096 + 1024),
                                                   (16, 2, 3072 + 64 * 48 + 4096 + 1024), (3, 0, 3072 + 16 * 48)])
def test_parameter_counts(n_extra, n_hidden, want):
    orc = H.load_oracle()
    ngp = capi.load_ngp_hip()
    x = netx(n_extra, n_hidden)
    assert orc.orc_netx_mlp_params(x.ctypes.data) == want
    v = np.zeros(1, capi.NET_VARIANT)
    v["n_extra_dims"], v["n_rgb_hidden_layers"] = n_extra, n_hidden
    assert ngp.ngp_hip_net_mlp_params_host(v.ctypes.data) == want            # the product's layout agrees
    assert ngp.ngp_hip_net_mlp_params_host(None) == 10240


def test_extra_dims_enter_the_colour_network_and_their_gradient_matches_finite_differences():
    ngp, orc, desc, coords = _setup(n=24)
    n, ne = len(coords), 5
    x0 = netx(ne, 2)
    npar = orc.orc_netx_n_params(desc.ctypes.data, x0.ctypes.data)
    p32 = np.zeros(npar, np.float32)
    orc.orc_nerf_init_params_x(desc.ctypes.data, x0.ctypes.data, 7, p32.ctypes.data)
    p32[:orc.orc_netx_mlp_params(x0.ctypes.data)] *= 3.0                      # livelier than Xavier: the colour outputs react to the inputs
    p32[orc.orc_netx_mlp_params(x0.ctypes.data):] *= 3000.0
    params = p32.astype(np.float16).view(np.uint16)
    rs = np.random.RandomState(0)
    table = (rs.rand(3, ne) * 2 - 1).astype(np.float32)
    slot = rs.randint(0, 3, n).astype(np.uint32)
    x = netx(ne, 2, table, slot)
    out = np.zeros((n, 4), np.uint16)
    orc.orc_nerf_inference_x(desc.ctypes.data, x.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, out.ctypes.data, 4)
    table2 = table.copy(); table2[:, 0] += 0.5
    x2 = netx(ne, 2, table2, slot)
    out2 = np.zeros((n, 4), np.uint16)
    orc.orc_nerf_inference_x(desc.ctypes.data, x2.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, out2.ctypes.data, 4)
    f, f2 = out.view(np.float16).astype(np.float32), out2.view(np.float16).astype(np.float32)
    assert np.abs(f[:, :3] - f2[:, :3]).max() > 1e-3 and np.array_equal(f[:, 3], f2[:, 3])     # colour moves, density does not
    # dL/d(extra) from the backward vs central differences of the colour outputs (loss = sum of w * rgb)
    w = rs.randn(n, 4).astype(np.float16); w[:, 3] = 0
    grads = np.zeros(npar, np.float64)
    dext = np.zeros((n, ne), np.float32)
    orc.orc_nerf_forward_backward_x(desc.ctypes.data, x.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, w.ctypes.data, None, grads.ctypes.data, None, dext.ctypes.data)

    def loss_rows(tab):   # per-sample loss with a table of one row per sample
        xx = netx(ne, 2, tab, np.arange(n, dtype=np.uint32))
        o = np.zeros((n, 4), np.uint16)
        orc.orc_nerf_inference_x(desc.ctypes.data, xx.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, o.ctypes.data, 4)
        return (o.view(np.float16).astype(np.float64)[:, :3] * w.astype(np.float64)[:, :3]).sum(1)
    per = table[slot].copy()
    eps = 0.06                                                               # several fp16 steps of the inputs: the quantisation averages out
    num = np.zeros((n, ne))
    for e in range(ne):
        a, b = per.copy(), per.copy()
        a[:, e] += eps; b[:, e] -= eps
        num[:, e] = (loss_rows(a) - loss_rows(b)) / (2 * eps)
    rel = np.linalg.norm(dext - num) / max(np.linalg.norm(num), 1e-9)
    assert rel < 0.15, rel                                                   # piecewise linear network, ReLU kinks inside the stencil on a few samples
    # and the matrix that reads them gets gradient in its extra columns only where extra dims exist
    g3 = grads[3072:3072 + 64 * 48].reshape(64, 48)
    assert np.abs(g3[:, 32:32 + ne]).max() > 0 and not g3[:, 32 + ne:].any()
