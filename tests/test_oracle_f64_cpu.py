"""The oracle's network arithmetic against an INDEPENDENT float64 statement of the same model (tests/golden/network_f64.npz, written by
tests/golden/make_network_f64.py: torch float64 + autograd from the Instant-NGP paper's formulas, the SH basis from scipy, Adam from its update
rules).  The oracle restates tiny-cuda-nn from recall and cannot be pinned against the reference; this bounds how far its forward,
its hand-derived backward, its SH constants and its optimizer can be from a second derivation.  Tolerances are the fp16 storage of activations /
parameters that the oracle (like tcnn) applies and the float64 model does not."""
import os

import numpy as np

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "network_f64.npz"))


def _desc():
    return np.frombuffer(G["desc"].tobytes(), dtype=H.NET_DESC).copy()


def test_sh4_constants_against_scipy(oracle):
    coords = G["coords"]
    got = np.zeros((len(coords), 16), np.float32)
    for i, c in enumerate(coords):
        d = np.ascontiguousarray(c[4:7], np.float32)
        oracle.orc_sh4(d.ctypes.data, got[i].ctypes.data)
    np.testing.assert_allclose(got, G["sh"], rtol=0, atol=3e-6)


def test_forward_against_float64(oracle):
    desc, coords, p16 = _desc(), np.ascontiguousarray(G["coords"]), np.ascontiguousarray(G["params16"])
    n = len(coords)
    out = np.zeros((n, 4), np.uint16)
    oracle.orc_nerf_inference(desc.ctypes.data, p16.ctypes.data, coords.ctypes.data, 7, n, out.ctypes.data, 4)
    got = out.view(np.float16).astype(np.float64)
    want = G["out"]
    # fp16 rounding of 32 + 64 + 16 + 64 + 64 activations per sample on the way: a few 1e-3 of the output scale
    assert np.abs(got - want).max() < 6e-3 * max(1.0, np.abs(want).max())
    assert np.corrcoef(got.ravel(), want.ravel())[0, 1] > 0.99999


def test_backward_against_autograd(oracle):
    desc, coords, p16, dl = _desc(), np.ascontiguousarray(G["coords"]), np.ascontiguousarray(G["params16"]), np.ascontiguousarray(G["dL_dout"])
    n, n_params = len(coords), len(p16)
    out = np.zeros((n, 4), np.uint16)
    grads = np.zeros(n_params, np.float64)
    dx = np.zeros((n, 32), np.uint16)
    oracle.orc_nerf_forward_backward(desc.ctypes.data, p16.ctypes.data, coords.ctypes.data, 7, n, dl.ctypes.data, out.ctypes.data, grads.ctypes.data, dx.ctypes.data)
    want = G["grads"]
    n_mlp = 10240
    # The output layer sees no ReLU decision behind it: fp16 storage noise only.  Further back a hidden unit that sits within fp16 rounding of zero
    # is switched on in one model and off in the other, which flips that sample's whole contribution: a few 1e-2 of the norm at 384 samples
    # (measured 8e-3 / 2e-2 / 1.4e-2 / 1.9e-2 for W4 / W3 / W2 / W1, row errors between 4e-4 and 3e-2 — scattered, not systematic).
    tol = {"W5": 2e-3, "W4": 2.5e-2, "W3": 4e-2, "W2": 4e-2, "W1": 4e-2}
    off = 0
    for name, o, i in (("W1", 64, 32), ("W2", 16, 64), ("W3", 64, 32), ("W4", 64, 64), ("W5", 16, 64)):
        a, b = grads[off:off + o * i], want[off:off + o * i]
        off += o * i
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < tol[name], (name, np.linalg.norm(a - b) / np.linalg.norm(b))
        assert np.corrcoef(a, b)[0, 1] > 0.999
    num = np.linalg.norm(grads[n_mlp:] - want[n_mlp:]); den = np.linalg.norm(want[n_mlp:])
    assert den > 0 and num / den < 4e-2, num / den
    # the same entries are reached (index rule: dense strides / spatial hash, level offsets)
    never = want[n_mlp:] == 0.0            # table entries no sample's 8 corners reach at any level: autograd leaves them exactly zero
    assert never.sum() > 100 and (grads[n_mlp:][never] == 0.0).all()


def test_adam_ema_against_update_rules(oracle):
    step, lr, b1, b2, eps, l2, scale, decay, n_matrix = G["adam_hyper"]
    g16 = np.ascontiguousarray(G["adam_grads16"])
    n = len(g16)
    master, m1, m2, ema = (np.ascontiguousarray(G[k]).copy() for k in ("adam_master", "adam_m1", "adam_m2", "adam_ema"))
    p16 = np.ascontiguousarray(G["adam_params16"]).copy()
    inf16 = np.zeros(n, np.uint16)
    oracle.orc_adam_ema_step(n, int(n_matrix), int(step), H.f32(lr), H.f32(b1), H.f32(b2), H.f32(eps), H.f32(l2), H.f32(scale), H.f32(decay), g16.ctypes.data, master.ctypes.data,
                             p16.ctypes.data, m1.ctypes.data, m2.ctypes.data, ema.ctypes.data, inf16.ctypes.data)
    # fp32 on the oracle's side: errors are relative to the OPERANDS (moments ~1e-3 / 1e-4, gradients up to ~0.1 / loss scale), not to a result that cancels
    np.testing.assert_allclose(m1, G["adam_new_m1"], rtol=2e-6, atol=3e-9)
    np.testing.assert_allclose(m2, G["adam_new_m2"], rtol=2e-6, atol=1e-11)
    np.testing.assert_allclose(master, G["adam_new_master"], rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(ema, G["adam_new_ema"], rtol=5e-6, atol=1e-8)
    np.testing.assert_array_equal(p16.view(np.float16), master.astype(np.float16))          # fp16 copy = rounded master weight (skipped entries unchanged on both)
    np.testing.assert_array_equal(inf16.view(np.float16), ema.astype(np.float16))


def test_input_gradient_against_autograd(oracle):
    """dL/d(position) through the hash encoding and dL/d(direction) through the SH basis (what camera optimisation consumes: nerf_network.h:187-266
    with a dL_dinput matrix) against the float64 model: autograd for the position, central differences of the whole model for the direction"""
    desc, coords, p16, dl = _desc(), np.ascontiguousarray(G["coords"]), np.ascontiguousarray(G["params16"]), np.ascontiguousarray(G["dL_dout"])
    n = len(coords)
    got = np.zeros((n, 6), np.float32)
    oracle.orc_nerf_input_gradient(desc.ctypes.data, p16.ctypes.data, coords.ctypes.data, 7, n, dl.ctypes.data, got.ctypes.data)
    # The float64 model evaluates the SH basis on the NORMALISED direction, tcnn's polynomials are defined off the sphere as well: the two agree on
    # the unit sphere and so do their TANGENTIAL derivatives; the radial component of the polynomial gradient has no counterpart (and no consumer:
    # the camera gradient takes ray.d x gradient, testbed_nerf.cu:1700-1706).
    r = coords[:, 4:7].astype(np.float64) * 2.0 - 1.0
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    tangential = lambda v: v - (v * r).sum(1, keepdims=True) * r
    for name, a, b, tol in (("pos", got[:, 0:3].astype(np.float64), G["dL_dpos"], 5e-2), ("dir", tangential(got[:, 3:6].astype(np.float64)), tangential(G["dL_ddir"]), 5e-2)):
        rel = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert np.linalg.norm(b) > 0 and rel < tol, (name, rel)       # fp16 deltas + ReLU decisions flipped by fp16 rounding, as for the weight gradients
        assert np.corrcoef(a.ravel(), b.ravel())[0, 1] > 0.998, name
