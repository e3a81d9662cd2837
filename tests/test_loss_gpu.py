"""GPU parity: loss / compositing / compaction kernel (loss.hip) vs the CPU oracle, through the C ABI.

Counts are integers but depend on `T < 1e-4` where T is a product of (1 - alpha), alpha = 1 - exp(-sigma dt): the oracle uses
libm expf, the kernel v_exp_f32.  Rays whose transmittance passes within 2e-3 (relative) of the threshold are excluded from the
exact count comparison (reported, must be rare); everything else must match exactly.  Float outputs: rtol 2e-3 (fp16 stores).
"""
import numpy as np
import pytest

import capi
import helpers as H
from capi import check
from test_sampling_gpu import _cameras

pytestmark = pytest.mark.gpu


def _inputs(oracle, cuda, n_rays=4096, seed=0, sigma_gain=3.0, cdf_mode=0):
    imgs, d_imgs, md_host, md_dev, xf = _cameras(cuda)
    grid = H.blob_density_grid(1)
    bf, mean = H.oracle_bitfield(oracle, grid, 1)
    aabb = H.unit_aabb()
    st, inc = H.pcg32_state(1337)
    max_samples = n_rays * 64
    r = dict(rc=np.zeros(1, np.uint32), nc=np.zeros(1, np.uint32), idx=np.zeros(n_rays, np.uint32), rays=np.zeros(n_rays, H.RAY),
             ns=np.zeros(n_rays * 2, np.uint32), co=np.zeros(max_samples, H.COORD))
    dres = np.array([32, 32], np.int32)
    C = H.make_error_map_cdfs(oracle, len(xf), 20, 14) if cdf_mode else None
    c_host = H.error_map_cdf_struct(C["x"].ctypes.data if cdf_mode & 1 else 0, C["y"].ctypes.data if cdf_mode & 1 else 0, C["img"].ctypes.data if cdf_mode & 2 else 0, C["res"]) if cdf_mode else None
    oracle.orc_generate_training_samples(n_rays, aabb.ctypes.data, max_samples, st, inc, r["rc"].ctypes.data, r["nc"].ctypes.data, r["idx"].ctypes.data,
                                         r["rays"].ctypes.data, r["ns"].ctypes.data, r["co"].ctypes.data, len(xf), md_host.ctypes.data, xf.ctypes.data,
                                         bf.ctypes.data, 0, None, 0, 0, H.f32(0.0), None, dres.ctypes.data, 0, n_rays, c_host.ctypes.data if cdf_mode else None)
    n_samples = int(r["nc"][0])
    rs = np.random.RandomState(seed)
    mlp = np.zeros((n_samples, 4), np.float16)
    mlp[:, :3] = rs.randn(n_samples, 3).astype(np.float16)
    mlp[:, 3] = (rs.randn(n_samples) * sigma_gain + 2.0).astype(np.float16)
    return dict(r=r, mlp=mlp, md_host=md_host, md_dev=md_dev, xf=xf, aabb=aabb, st=st, inc=inc, n_rays=n_rays, n_alive=int(r["rc"][0]), n_samples=n_samples,
                mean=mean, keep=(imgs, d_imgs), cdf_mode=cdf_mode, C=C, c_host=c_host)


def _run(ngp, oracle, cuda, I, loss_type, B, random_bg=1, color_space=0, linear=0, rgb_act=2):
    n_rays, n_alive, ns = I["n_rays"], I["n_alive"], I["n_samples"]
    bg = np.array([0.2, 0.4, 0.7], np.float32)
    em_res = np.array([16, 12], np.int32)
    n_img = len(I["xf"])
    exposure = I.get("exposure", np.zeros((n_img, 3), np.float32))
    # ---- oracle
    o = dict(cnt=np.zeros(1, np.uint32), ns=I["r"]["ns"].copy(), co=np.zeros(B, H.COORD), dl=np.zeros((B, 4), np.float16), loss=np.zeros(n_rays, np.float32),
             em=np.zeros(n_img * 16 * 12, np.float32), enc=np.zeros((B, 32), np.uint16), expg=np.zeros((n_img, 3), np.float32))
    env = I.get("envmap")      # dict(data fp32 [h][w][4], res (w, h), loss_type, train): the *_ex entry points (environment map in front of the background)
    sharp = I.get("sharp")     # dict(data fp32 [n_img][sry][srx], res (srx, sry)): include_sharpness_in_error
    o_loss_fn, o_tail = oracle.orc_compute_loss, ()
    if env is not None or sharp is not None:
        ex_o = np.zeros(1, capi.LOSS_EXTRAS)
        if env is not None:
            o["envg"] = np.zeros_like(env["data"])
            ex_o["envmap_data"], ex_o["envmap_gradient"], ex_o["envmap_res"][0], ex_o["envmap_loss_type"] = env["data"].ctypes.data, o["envg"].ctypes.data if env["train"] else 0, env["res"], env["loss_type"]
        if sharp is not None:
            o["sgrid"] = sharp["grid0"].copy()
            ex_o["sharpness_data"], ex_o["sharpness_res"][0], ex_o["sharpness_grid"] = sharp["data"].ctypes.data, sharp["res"], o["sgrid"].ctypes.data
        o_loss_fn, o_tail = oracle.orc_compute_loss_ex, (ex_o.ctypes.data,)
    o_loss_fn(n_rays, I["aabb"].ctypes.data, I["st"], I["inc"], B, n_alive, H.f32(128.0), 4, bg.ctypes.data, color_space, random_bg, linear, n_img,
                            I["md_host"].ctypes.data, I["mlp"].ctypes.data, o["cnt"].ctypes.data, I["r"]["idx"].ctypes.data, I["r"]["rays"].ctypes.data, o["ns"].ctypes.data,
                            I["r"]["co"].ctypes.data, o["co"].ctypes.data, o["dl"].ctypes.data, loss_type, o["loss"].ctypes.data, 0, None, rgb_act, 3, 0,
                            o["em"].ctypes.data, em_res.ctypes.data, H.f32(I["mean"]), exposure.ctypes.data, H.f32(0.2), I["c_host"].ctypes.data if I.get("cdf_mode") else None,
                            I["enc"].ctypes.data if "enc" in I else None, o["enc"].ctypes.data if "enc" in I else None, H.f32(I.get("depth_lambda", 0.0)), I.get("depth_loss", 2), o["expg"].ctypes.data if I.get("exposure_grad") else None, *o_tail)
    # ---- device
    d = dict(cnt=H.dev_zeros(4, cuda), ns=H.to_dev(I["r"]["ns"], cuda), co=H.dev_zeros(B * 28, cuda), dl=H.dev_zeros(B * 8, cuda), loss=H.dev_zeros(n_rays * 4, cuda),
             em=H.dev_zeros(n_img * 16 * 12 * 4, cuda), enc=H.dev_zeros(B * 64, cuda), expg=H.dev_zeros(n_img * 12, cuda))
    d_enc_in = H.to_dev(I["enc"], cuda) if "enc" in I else None
    d_rc = H.to_dev(np.array([n_alive], np.uint32), cuda)
    d_md, d_mlp, d_idx, d_rays, d_co = (H.to_dev(a, cuda) for a in (I["md_dev"], I["mlp"], I["r"]["idx"], I["r"]["rays"], I["r"]["co"]))
    d_mean, d_exp = H.to_dev(np.array([I["mean"]], np.float32), cuda), H.to_dev(exposure, cuda)
    c_dev = None
    if I.get("cdf_mode"):
        m, C = I["cdf_mode"], I["C"]
        d_cx, d_cy, d_ci = H.to_dev(C["x"], cuda), H.to_dev(C["y"], cuda), H.to_dev(C["img"], cuda)
        c_dev = H.error_map_cdf_struct(d_cx.data_ptr() if m & 1 else 0, d_cy.data_ptr() if m & 1 else 0, d_ci.data_ptr() if m & 2 else 0, C["res"])
    d_loss_fn, d_tail = ngp.ngp_hip_compute_loss, ()
    if env is not None or sharp is not None or I.get("x_index"):
        ex_d = np.zeros(1, capi.LOSS_EXTRAS)
        if I.get("x_index"):    # round 6: the compaction leaves, per kept sample, the index of the uncompacted sample it came from (device only: the oracle copies rows)
            d["xidx"] = H.to_dev(np.full(B, 0xFFFFFFFF, np.uint32), cuda)
            ex_d["x_row_index_out"] = d["xidx"].data_ptr()
        if env is not None:
            d_env, d["envg"] = H.to_dev(env["data"], cuda), H.dev_zeros(env["data"].nbytes, cuda)
            ex_d["envmap_data"], ex_d["envmap_gradient"], ex_d["envmap_res"][0], ex_d["envmap_loss_type"] = d_env.data_ptr(), d["envg"].data_ptr() if env["train"] else 0, env["res"], env["loss_type"]
        if sharp is not None:
            d_sd, d["sgrid"] = H.to_dev(sharp["data"], cuda), H.to_dev(sharp["grid0"], cuda)
            ex_d["sharpness_data"], ex_d["sharpness_res"][0], ex_d["sharpness_grid"] = d_sd.data_ptr(), sharp["res"], d["sgrid"].data_ptr()
        d_loss_fn, d_tail = ngp.ngp_hip_compute_loss, (ex_d.ctypes.data,)
    check(d_loss_fn(None, n_rays, I["aabb"].ctypes.data, I["st"], I["inc"], B, d_rc.data_ptr(), H.f32(128.0), 4, bg.ctypes.data, color_space, random_bg, linear,
                                   n_img, d_md.data_ptr(), d_mlp.data_ptr(), d["cnt"].data_ptr(), d_idx.data_ptr(), d_rays.data_ptr(), d["ns"].data_ptr(), d_co.data_ptr(),
                                   d["co"].data_ptr(), d["dl"].data_ptr(), 4, loss_type, d["loss"].data_ptr(), 0, None, rgb_act, 3, 0, d["em"].data_ptr(), em_res.ctypes.data,
                                   d_mean.data_ptr(), d_exp.data_ptr(), H.f32(0.2), c_dev.ctypes.data if c_dev is not None else None,
                                   d_enc_in.data_ptr() if "enc" in I else None, d["enc"].data_ptr() if "enc" in I else None, H.f32(I.get("depth_lambda", 0.0)), I.get("depth_loss", 2), d["expg"].data_ptr() if I.get("exposure_grad") else None, *d_tail))
    g = dict(cnt=H.to_host(d["cnt"], np.uint32), ns=H.to_host(d["ns"], np.uint32), co=H.to_host(d["co"], H.COORD), dl=H.to_host(d["dl"], np.float16).reshape(B, 4),
             loss=H.to_host(d["loss"], np.float32), em=H.to_host(d["em"], np.float32), enc=H.to_host(d["enc"], np.uint16).reshape(B, 32), d_co=d["co"], expg=H.to_host(d["expg"], np.float32).reshape(n_img, 3))
    if env is not None:
        g["envg"] = H.to_host(d["envg"], np.float32).reshape(env["data"].shape)
    if sharp is not None:
        g["sgrid"] = H.to_host(d["sgrid"], np.float32)
    if I.get("x_index"):
        g["xidx"] = H.to_host(d["xidx"], np.uint32)
    return o, g


def _borderline_rays(I):
    """rays whose running transmittance comes within 2e-3 (relative) of EPSILON = 1e-4 at any step (float64 replay)."""
    ns, mlp, co = I["r"]["ns"], I["mlp"], I["r"]["co"]
    min_step = np.float64(1.73205080757 / 1024)
    out = set()
    for i in range(I["n_alive"]):
        n, b = int(ns[2 * i]), int(ns[2 * i + 1])
        dt = co["dt"][b:b + n].astype(np.float64) * (min_step * 128 - min_step) + min_step
        sigma = np.exp(mlp[b:b + n, 3].astype(np.float64))
        T = np.cumprod(np.exp(-sigma * dt))
        if np.any(np.abs(T - 1e-4) < 2e-7):
            out.add(i)
    return out


@pytest.mark.parametrize("loss_type", [4, 0, 6, 1, 2, 3, 5])  # Huber (configs/nerf/base.json:2-4) first, then the rest of ELossType
def test_loss_and_compaction_match_oracle(ngp, oracle, cuda, loss_type):
    I = _inputs(oracle, cuda)
    B = I["n_samples"] + 128  # no truncation
    o, g = _run(ngp, oracle, cuda, I, loss_type, B)
    border = _borderline_rays(I)
    assert len(border) < 0.01 * I["n_alive"]
    mism = [i for i in range(I["n_alive"]) if int(o["ns"][2 * i]) != int(g["ns"][2 * i]) and i not in border]
    assert not mism, mism[:10]
    if not border:
        assert int(o["cnt"][0]) == int(g["cnt"][0])
    # compacted slots tile [0, total)
    gb = sorted((int(g["ns"][2 * i + 1]), int(g["ns"][2 * i])) for i in range(I["n_alive"]) if int(g["ns"][2 * i]) > 0)
    pos = 0
    for b, c in gb:
        assert b == pos
        pos += c
    assert pos == int(g["cnt"][0]) and pos < I["n_samples"]  # compaction actually removed samples
    worst = 0.0
    for i in range(I["n_alive"]):
        if i in border:
            continue
        n, bo, bg_ = int(o["ns"][2 * i]), int(o["ns"][2 * i + 1]), int(g["ns"][2 * i + 1])
        if n == 0:
            continue
        assert o["co"][bo:bo + n].tobytes() == g["co"][bg_:bg_ + n].tobytes()
        a, b = g["dl"][bg_:bg_ + n].astype(np.float32), o["dl"][bo:bo + n].astype(np.float32)
        # atol: the kernel sums colour / transmittance as wave scans (tree order), the oracle left to right; the `rgb_ray - rgb_ray2`
        # suffix cancels to ~1e-7 differently and is amplified by d(sigma)/d(logit) -> a few 1e-6 on gradients of magnitude ~1e-3
        np.testing.assert_allclose(a, b, rtol=4e-3, atol=6e-6)
        worst = max(worst, float(np.abs(a - b).max()))
    keep = np.array([i for i in range(I["n_alive"]) if i not in border])
    np.testing.assert_allclose(g["loss"][keep], o["loss"][keep], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(g["em"], o["em"], rtol=2e-3, atol=1e-6 * max(1.0, float(o["em"].max())))
    assert np.abs(o["dl"].astype(np.float32)).max() > 1e-4


def test_compaction_leaves_the_index_of_every_kept_sample(ngp, oracle, cuda):
    """NgpLossExtras.x_row_index_out (round 6): slot k of the compacted batch <- the uncompacted sample it came from, so that the backward pass can read the encoding
    row where the network pass wrote it instead of a copy.  Every compacted coordinate row must be the row of the sample its index names, the indices of a ray run up
    by one from the ray's first sample, slots behind the compacted samples stay untouched, and everything else the kernel produces is what it produces without the index."""
    I = _inputs(oracle, cuda)
    B = I["n_samples"] + 128
    o0, g0 = _run(ngp, oracle, cuda, I, 4, B)
    I["x_index"] = True
    o, g = _run(ngp, oracle, cuda, I, 4, B)
    n_c = int(g["cnt"][0])
    assert 0 < n_c < I["n_samples"] and n_c == int(g0["cnt"][0])
    idx = g["xidx"]
    assert (idx[n_c:] == 0xFFFFFFFF).all() and (idx[:n_c] < I["n_samples"]).all()
    assert g["co"][:n_c].tobytes() == I["r"]["co"][idx[:n_c]].tobytes()
    for i in range(I["n_alive"]):
        n, b_out, b_in = int(g["ns"][2 * i]), int(g["ns"][2 * i + 1]), int(I["r"]["ns"][2 * i + 1])
        if n:
            np.testing.assert_array_equal(idx[b_out:b_out + n], np.arange(b_in, b_in + n, dtype=np.uint32))
    # slot order is decided by an atomic: compare the two runs ray by ray
    for i in range(I["n_alive"]):
        n = int(g["ns"][2 * i])
        assert n == int(g0["ns"][2 * i])
        a, b = int(g["ns"][2 * i + 1]), int(g0["ns"][2 * i + 1])
        assert g["dl"][a:a + n].tobytes() == g0["dl"][b:b + n].tobytes()
    np.testing.assert_array_equal(g["loss"], g0["loss"])


def test_compaction_carries_the_saved_encoding(ngp, oracle, cuda):
    """ngp_hip.h "Forward pass": the encoding rows that ngp_hip_nerf_forward wrote for the uncompacted samples, carried through the
    compaction, are bit for bit what a second ngp_hip_nerf_forward over the compacted coordinates writes (testbed_nerf.cu:3330)."""
    I = _inputs(oracle, cuda)
    n_s = I["n_samples"]
    n_pad = (n_s + 255) // 256 * 256
    desc = H.make_desc(ngp, 19)
    P = H.random_params(desc, 3, grid_amp=1.0)
    d_desc, d_P = H.to_dev(desc, cuda), H.to_dev(P, cuda)
    co_pad = np.zeros(n_pad, H.COORD); co_pad[:n_s] = I["r"]["co"][:n_s]
    d_co = H.to_dev(co_pad, cuda)
    out, xs = H.dev_zeros(n_pad * 8, cuda), H.dev_zeros(n_pad * 64, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), d_co.data_ptr(), 7, n_pad, out.data_ptr(), 4, xs.data_ptr()))
    I["enc"] = H.to_host(xs, np.uint16).reshape(n_pad, 32)
    B = n_pad + 256
    o, g = _run(ngp, oracle, cuda, I, 0, B)
    n_c = int(g["cnt"][0])
    assert 0 < n_c < n_s
    # device: carried rows == second pass over the compacted coordinates
    n_c_pad = (n_c + 255) // 256 * 256
    out2, xs2 = H.dev_zeros(n_c_pad * 8, cuda), H.dev_zeros(n_c_pad * 64, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_P.data_ptr(), g["d_co"].data_ptr(), 7, n_c_pad, out2.data_ptr(), 4, xs2.data_ptr()))
    np.testing.assert_array_equal(g["enc"][:n_c], H.to_host(xs2, np.uint16).reshape(n_c_pad, 32)[:n_c])
    assert (g["enc"][:n_c] != 0).any()
    # oracle: the same rows, ray by ray (slot order differs between the two)
    border = _borderline_rays(I)
    for i in range(I["n_alive"]):
        n, bo, bg_ = int(o["ns"][2 * i]), int(o["ns"][2 * i + 1]), int(g["ns"][2 * i + 1])
        if i in border or n == 0 or n != int(g["ns"][2 * i]):
            continue
        np.testing.assert_array_equal(g["enc"][bg_:bg_ + n], o["enc"][bo:bo + n])
    # the two pointers go together
    assert ngp.ngp_hip_compute_loss(None, 1, I["aabb"].ctypes.data, 0, 1, 256, out.data_ptr(), H.f32(128.0), 4, np.zeros(3, np.float32).ctypes.data, 0, 0, 0, 1, out.data_ptr(),
                                    out.data_ptr(), out.data_ptr(), out.data_ptr(), out.data_ptr(), out.data_ptr(), out.data_ptr(), out.data_ptr(), out.data_ptr(), 4, 0, None, 0, None,
                                    2, 3, 0, None, None, out.data_ptr(), out.data_ptr(), H.f32(0.2), None, xs.data_ptr(), None, H.f32(0.0), 2, None) != 0


@pytest.mark.parametrize("linear,color_space", [(0, 0), (1, 1)])
def test_exposure_gradient(ngp, oracle, cuda, linear, color_space):
    """optimize_exposure (testbed_nerf.cu:1558-1572): per-image sums of loss_scale * (-dL/drgb [/ srgb'(target)]) * 2^exposure * ln 2"""
    I = _inputs(oracle, cuda)
    n_img = len(I["xf"])
    I["exposure"] = (np.random.RandomState(2).randn(n_img, 3) * 0.3).astype(np.float32)
    I["exposure_grad"] = True
    B = I["n_samples"] + 128
    o, g = _run(ngp, oracle, cuda, I, 0, B, random_bg=1, color_space=color_space, linear=linear)
    assert np.abs(o["expg"]).min() > 0
    # sums over ~700 rays per image in different orders (atomics), a few borderline rays may differ in their compaction
    np.testing.assert_allclose(g["expg"], o["expg"], rtol=2e-2, atol=2e-2 * np.abs(o["expg"]).max())
    # the exposures themselves reach the targets: the loss differs from the unexposed run
    I0 = dict(I); I0["exposure"] = np.zeros((n_img, 3), np.float32)
    o0 = _run(ngp, oracle, cuda, I0, 0, B, random_bg=1, color_space=color_space, linear=linear)[0]
    assert not np.allclose(o0["loss"], o["loss"], rtol=1e-3)


@pytest.mark.parametrize("depth_loss", [1, 0])
def test_loss_with_depth_supervision(ngp, oracle, cuda, depth_loss):
    """depth_supervision_lambda > 0 (testbed_nerf.cu:1450-1452, 1536-1541): extra density gradient from the expected termination depth vs
    the depth image; images without depth (pointer NULL) are unaffected"""
    I = _inputs(oracle, cuda)
    n_img = len(I["xf"])
    w, h = int(I["md_host"]["res"][0][0]), int(I["md_host"]["res"][0][1])
    rs = np.random.RandomState(4)
    depth = (0.4 + 1.5 * rs.rand(n_img, h, w)).astype(np.float32)
    d_depth = H.to_dev(depth, cuda)
    mh, mdv = I["md_host"].copy(), I["md_dev"].copy()
    for k in range(n_img - 1):                       # the last image has no depth
        mh["depth"][k] = depth[k].ctypes.data
        mdv["depth"][k] = d_depth.data_ptr() + k * w * h * 4
    I["md_host"], I["md_dev"] = mh, mdv
    B = I["n_samples"] + 128
    I["depth_lambda"], I["depth_loss"] = 0.0, depth_loss
    o0, g0 = _run(ngp, oracle, cuda, I, 0, B)
    I["depth_lambda"] = 0.7
    o, g = _run(ngp, oracle, cuda, I, 0, B)
    border = _borderline_rays(I)
    keep = [i for i in range(I["n_alive"]) if i not in border and int(o["ns"][2 * i]) == int(g["ns"][2 * i]) and int(o["ns"][2 * i]) > 0]
    changed = unchanged = 0
    for i in keep[::max(1, len(keep) // 600)]:            # rays are ordered by image: sample all of them
        n, bo, bg_ = int(o["ns"][2 * i]), int(o["ns"][2 * i + 1]), int(g["ns"][2 * i + 1])
        a, b = g["dl"][bg_:bg_ + n].astype(np.float32), o["dl"][bo:bo + n].astype(np.float32)
        np.testing.assert_allclose(a, b, rtol=4e-3, atol=1e-5)
        b0 = o0["dl"][int(o0["ns"][2 * i + 1]):int(o0["ns"][2 * i + 1]) + n].astype(np.float32)
        np.testing.assert_array_equal(b[:, :3], b0[:, :3])          # colour gradients do not see the depth term
        img = (int(I["r"]["idx"][i]) * n_img // I["n_rays"]) % n_img
        if img == n_img - 1:
            np.testing.assert_array_equal(b[:, 3], b0[:, 3]); unchanged += 1
        elif not np.array_equal(b[:, 3], b0[:, 3]):
            changed += 1
    assert changed > 100 and unchanged > 10


@pytest.mark.parametrize("cdf_mode", [1, 3])
def test_loss_with_error_map_importance_sampling(ngp, oracle, cuda, cdf_mode):
    """the loss kernel replays image_idx / nerf_random_image_pos_training with the CDFs and divides the reported loss (and the error-map
    deposit) by img_pdf * xy_pdf, not the gradient (testbed_nerf.cu:1381-1386, 1448-1458)"""
    I = _inputs(oracle, cuda, cdf_mode=cdf_mode)
    B = I["n_samples"] + 128
    o, g = _run(ngp, oracle, cuda, I, 0, B)
    border = _borderline_rays(I)
    keep = np.array([i for i in range(I["n_alive"]) if i not in border])
    np.testing.assert_array_equal(o["ns"].reshape(-1, 2)[keep, 0], g["ns"].reshape(-1, 2)[keep, 0])
    np.testing.assert_allclose(g["loss"][keep], o["loss"][keep], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(g["em"], o["em"], rtol=2e-3, atol=1e-6 * max(1.0, float(o["em"].max())))
    for i in keep[:400]:
        n, bo, bg_ = int(o["ns"][2 * i]), int(o["ns"][2 * i + 1]), int(g["ns"][2 * i + 1])
        if n:
            # same summation-order noise as in test_loss_and_compaction_match_oracle (a cancelling suffix sum), other rays: atol 1e-5
            np.testing.assert_allclose(g["dl"][bg_:bg_ + n].astype(np.float32), o["dl"][bo:bo + n].astype(np.float32), rtol=4e-3, atol=1e-5)
    # the weighting is really there: the same rays without the CDFs report different losses
    I0 = dict(I); I0["cdf_mode"] = 0
    o0 = _run(ngp, oracle, cuda, I0, 0, B)[0]
    assert not np.allclose(o0["loss"][keep], o["loss"][keep], rtol=1e-3)


@pytest.mark.parametrize("color_space,linear,random_bg,rgb_act", [(1, 0, 0, 2), (0, 1, 1, 3), (1, 1, 0, 0)])
def test_loss_colour_space_variants(ngp, oracle, cuda, color_space, linear, random_bg, rgb_act):
    I = _inputs(oracle, cuda, n_rays=1024, seed=2)
    B = I["n_samples"] + 128
    o, g = _run(ngp, oracle, cuda, I, 0, B, random_bg=random_bg, color_space=color_space, linear=linear, rgb_act=rgb_act)
    border = _borderline_rays(I)
    keep = np.array([i for i in range(I["n_alive"]) if i not in border])
    np.testing.assert_array_equal(g["ns"][2 * keep], o["ns"][2 * keep])
    np.testing.assert_allclose(g["loss"][keep], o["loss"][keep], rtol=2e-4, atol=1e-9)


def test_compaction_truncates_at_batch_size(ngp, oracle, cuda):
    """testbed_nerf.cu:1434-1440: a ray straddling B is truncated, rays past B contribute nothing; the counter still counts all."""
    I = _inputs(oracle, cuda)
    B = 4096
    o, g = _run(ngp, oracle, cuda, I, 4, B)
    total = int(g["cnt"][0])
    assert total > B
    kept = 0
    for i in range(I["n_alive"]):
        n, b = int(g["ns"][2 * i]), int(g["ns"][2 * i + 1])
        assert b + n <= B or n == 0
        kept += n
    assert kept == B


def test_fill_rollover(ngp, oracle, cuda):
    rs = np.random.RandomState(0)
    B = 1024
    for n_in in (0, 1, 300, 1023, 1024, 5000):
        a = rs.randn(B, 4).astype(np.float16)
        c = rs.rand(B, 7).astype(np.float32)
        d_a, d_c, d_n = H.to_dev(a, cuda), H.to_dev(c, cuda), H.to_dev(np.array([n_in], np.uint32), cuda)
        check(ngp.ngp_hip_fill_rollover_and_rescale_f16(None, B, 4, d_n.data_ptr(), d_a.data_ptr()))
        check(ngp.ngp_hip_fill_rollover_f32(None, B, 7, d_n.data_ptr(), d_c.data_ptr()))
        oracle.orc_fill_rollover_and_rescale_f16(B, 4, n_in, a.ctypes.data)
        oracle.orc_fill_rollover_f32(B, 7, n_in, c.ctypes.data)
        np.testing.assert_array_equal(H.to_host(d_a, np.float16).reshape(B, 4), a)
        np.testing.assert_array_equal(H.to_host(d_c, np.float32).reshape(B, 7), c)


def test_fill_rollover_training_equals_the_separate_launches(ngp, cuda):
    import torch
    n, n_in = 1024, 700
    rs = np.random.RandomState(3)
    dl = rs.randn(n, 4).astype(np.float16); co = rs.rand(n, 7).astype(np.float32); en = rs.rand(n, 16).astype(np.float32)
    cnt = H.to_dev(np.array([n_in], np.uint32), cuda)
    a = [H.to_dev(x, cuda) for x in (dl, co, en)]
    b = [H.to_dev(x, cuda) for x in (dl, co, en)]
    check(ngp.ngp_hip_fill_rollover_and_rescale_f16(None, n, 4, cnt.data_ptr(), a[0].data_ptr()))
    check(ngp.ngp_hip_fill_rollover_f32(None, n, 7, cnt.data_ptr(), a[1].data_ptr()))
    check(ngp.ngp_hip_fill_rollover_f32(None, n, 16, cnt.data_ptr(), a[2].data_ptr()))
    check(ngp.ngp_hip_fill_rollover_training(None, n, cnt.data_ptr(), b[0].data_ptr(), 4, b[1].data_ptr(), 7, b[2].data_ptr(), 16))
    for x, y, dt in zip(a, b, (np.uint16, np.uint32, np.uint32)):
        np.testing.assert_array_equal(H.to_host(x, dt), H.to_host(y, dt))
    assert not np.array_equal(H.to_host(b[1], np.float32).reshape(n, 7)[n_in:], co[n_in:])       # it did roll over
    c2 = H.to_dev(co, cuda)
    check(ngp.ngp_hip_fill_rollover_training(None, n, cnt.data_ptr(), b[0].data_ptr(), 4, c2.data_ptr(), 7, None, 0))   # without encoding rows
    np.testing.assert_array_equal(H.to_host(c2, np.uint32), H.to_host(a[1], np.uint32))
    # the step's merged launch: the counter post of ngp_hip_post_words, then the same three roll-overs
    m = [H.to_dev(x, cuda) for x in (dl, co, en)]
    wa, wb = torch.tensor([4242], dtype=torch.int32, device=cuda), torch.tensor([0.625], dtype=torch.float32, device=cuda)
    dst, zero, sums = torch.full((4,), -1, dtype=torch.int32, device=cuda), torch.full((4,), 9, dtype=torch.int32, device=cuda), torch.zeros(3, dtype=torch.float64, device=cuda)
    check(ngp.ngp_hip_post_words_and_fill_rollover_training(None, wa.data_ptr(), cnt.data_ptr(), wb.data_ptr(), 77, dst.data_ptr(), zero.data_ptr(), 4, sums.data_ptr(), n, cnt.data_ptr(),
                                                            m[0].data_ptr(), 4, m[1].data_ptr(), 7, m[2].data_ptr(), 16))
    for x, y, dt in zip(a, m, (np.uint16, np.uint32, np.uint32)):
        np.testing.assert_array_equal(H.to_host(x, dt), H.to_host(y, dt))
    d = dst.cpu().numpy()
    assert d[0] == 4242 and d[1] == n_in and d[2:3].view(np.float32)[0] == 0.625 and d[3] == 77 and (zero.cpu().numpy() == 0).all()
    assert sums.cpu().numpy().tolist() == [4242.0, float(n_in), 0.625]
    for n_in2 in (0, n, n + 5):                                            # nothing to fill: untouched
        m2, cnt2 = H.to_dev(co, cuda), H.to_dev(np.array([n_in2], np.uint32), cuda)
        check(ngp.ngp_hip_post_words_and_fill_rollover_training(None, None, None, None, 1, None, None, 0, None, n, cnt2.data_ptr(), m[0].data_ptr(), 4, m2.data_ptr(), 7, None, 0))
        np.testing.assert_array_equal(H.to_host(m2, np.float32).reshape(n, 7), co)


def test_reduce_sum(ngp, cuda):
    rs = np.random.RandomState(0)
    for n in (1, 255, 4096, 262144):
        x = rs.rand(n).astype(np.float32)
        d_o, d_x = H.dev_zeros(4, cuda), H.to_dev(x, cuda)
        check(ngp.ngp_hip_reduce_sum_f32(None, d_x.data_ptr(), n, d_o.data_ptr()))
        assert abs(float(H.to_host(d_o, np.float32)[0]) - float(x.astype(np.float64).sum())) <= 1e-5 * n


def test_counter_posting_kernels(ngp, cuda):
    """ngp_hip_gather_words / ngp_hip_post_words: the step's counters leave the device in stream order (pinned host memory in the
    Testbed; plain device memory here), tag last, optional clears and the double-precision copy for a data-parallel all-reduce"""
    import torch
    a = torch.tensor([123456], dtype=torch.int32, device=cuda)
    b = torch.tensor([7890], dtype=torch.int32, device=cuda)
    c = torch.tensor([0.375], dtype=torch.float32, device=cuda)
    dst = torch.full((4,), -1, dtype=torch.int32, device=cuda)
    check(ngp.ngp_hip_gather_words(None, a.data_ptr(), b.data_ptr(), None, c.data_ptr(), dst.data_ptr()))
    torch.cuda.synchronize()
    assert dst.cpu().tolist() == [123456, 7890, 0, int(np.float32(0.375).view(np.int32))]
    dst.fill_(-1)
    z = torch.tensor([5, 6, 7, 8], dtype=torch.int32, device=cuda)
    s3 = torch.zeros(3, dtype=torch.float64, device=cuda)
    check(ngp.ngp_hip_post_words(None, a.data_ptr(), b.data_ptr(), c.data_ptr(), 42, dst.data_ptr(), z.data_ptr(), 3, s3.data_ptr()))
    torch.cuda.synchronize()
    assert dst.cpu().tolist() == [123456, 7890, int(np.float32(0.375).view(np.int32)), 42]
    assert z.cpu().tolist() == [0, 0, 0, 8]
    assert s3.cpu().tolist() == [123456.0, 7890.0, 0.375]
    check(ngp.ngp_hip_post_words(None, None, None, None, 43, dst.data_ptr(), None, 0, None))
    torch.cuda.synchronize()
    assert dst.cpu().tolist() == [0, 0, 0, 43]


@pytest.mark.parametrize("cdf_mode", [0, 3])
def test_cam_gradient_matches_oracle(ngp, oracle, cuda, cdf_mode):
    """compute_cam_gradient_train_nerf (extrinsics outputs): both sides get the ORACLE's compacted batch and one random input gradient, so the
    comparison does not depend on the compaction's threshold decisions.  Per-ray sums are associated differently (16 lanes + shuffles vs a sequential
    loop) and the accumulation over rays is atomic: rtol 2e-4 of the per-image gradient's magnitude."""
    I = _inputs(oracle, cuda, n_rays=2048, seed=5, cdf_mode=cdf_mode)
    B = 1 << 13                                                               # smaller than what the rays keep: the tail gets no room (numsteps 0)
    o, _ = _run(ngp, oracle, cuda, I, 4, B)
    n_rays, n_alive, n_img = I["n_rays"], I["n_alive"], len(I["xf"])
    assert int(o["cnt"][0]) > B and (o["ns"][0::2][:n_alive] == 0).any() and (o["ns"][0::2][:n_alive] > 16).any()           # some rays were dropped: the "ray doesn't matter" branch runs
    rs = np.random.RandomState(9)
    cg = (rs.randn(B, 6) * 0.1).astype(np.float32)
    ref_pos, ref_rot = np.zeros((n_img, 3), np.float32), np.zeros((n_img, 3), np.float32)
    oracle.orc_compute_cam_gradient(n_rays, I["aabb"].ctypes.data, I["st"], I["inc"], n_alive, 0, ref_pos.ctypes.data, ref_rot.ctypes.data, n_img, I["md_host"].ctypes.data,
                                    I["r"]["idx"].ctypes.data, I["r"]["rays"].ctypes.data, o["ns"].ctypes.data, o["co"].ctypes.data, cg.ctypes.data,
                                    I["c_host"].ctypes.data if cdf_mode else None)
    d_pos, d_rot = H.dev_zeros(n_img * 12, cuda), H.dev_zeros(n_img * 12, cuda)
    d_rc = H.to_dev(np.array([n_alive], np.uint32), cuda)
    d_md, d_idx, d_rays, d_ns, d_co, d_cg = (H.to_dev(a, cuda) for a in (I["md_dev"], I["r"]["idx"], I["r"]["rays"], o["ns"], o["co"], cg))
    c_dev = None
    if cdf_mode:
        C = I["C"]
        d_cx, d_cy, d_ci = H.to_dev(C["x"], cuda), H.to_dev(C["y"], cuda), H.to_dev(C["img"], cuda)
        c_dev = H.error_map_cdf_struct(d_cx.data_ptr() if cdf_mode & 1 else 0, d_cy.data_ptr() if cdf_mode & 1 else 0, d_ci.data_ptr() if cdf_mode & 2 else 0, C["res"])
    for rep in range(2):                                                       # accumulates: the second call doubles the sums
        check(ngp.ngp_hip_compute_cam_gradient(None, n_rays, I["aabb"].ctypes.data, I["st"], I["inc"], d_rc.data_ptr(), 0, d_pos.data_ptr(), d_rot.data_ptr(), n_img,
                                               d_md.data_ptr(), d_idx.data_ptr(), d_rays.data_ptr(), d_ns.data_ptr(), d_co.data_ptr(), d_cg.data_ptr(),
                                               c_dev.ctypes.data if c_dev is not None else None))
        got_pos, got_rot = H.to_host(d_pos, np.float32).reshape(n_img, 3), H.to_host(d_rot, np.float32).reshape(n_img, 3)
        for got, ref in ((got_pos, ref_pos), (got_rot, ref_rot)):
            assert np.abs(ref).max() > 0
            np.testing.assert_allclose(got, ref * (rep + 1), rtol=0, atol=2e-4 * np.abs(ref).max() * (rep + 1))
    # either output may be left out
    d_pos2 = H.dev_zeros(n_img * 12, cuda)
    check(ngp.ngp_hip_compute_cam_gradient(None, n_rays, I["aabb"].ctypes.data, I["st"], I["inc"], d_rc.data_ptr(), 0, d_pos2.data_ptr(), None, n_img, d_md.data_ptr(), d_idx.data_ptr(),
                                           d_rays.data_ptr(), d_ns.data_ptr(), d_co.data_ptr(), d_cg.data_ptr(), c_dev.ctypes.data if c_dev is not None else None))
    np.testing.assert_allclose(H.to_host(d_pos2, np.float32).reshape(n_img, 3), ref_pos, rtol=0, atol=2e-4 * np.abs(ref_pos).max())


@pytest.mark.parametrize("linear,env_loss", [(0, 6), (1, 4)])
def test_loss_with_environment_map(ngp, oracle, cuda, linear, env_loss):
    """compute_loss_kernel_train_nerf with envmap_data / envmap_gradient (:1289-1292, 1394-1401, 1573-1596): the map in front of the background colour changes
    the targets' background term and the ray colours; rays whose every sample is kept deposit their background gradient (fp16 value x fp16 bilinear weight,
    fp32 sums).  The envmap loss (RelativeL2 in base.json:77-79) may differ from the NeRF loss (Huber)."""
    rs = np.random.RandomState(21)
    env = dict(data=(rs.rand(8, 16, 4) * np.array([1, 1, 1, 0.8])).astype(np.float32), res=(16, 8), loss_type=env_loss, train=True)
    I = _inputs(oracle, cuda, n_rays=4096, seed=8, sigma_gain=1.0)
    I["mlp"][:, 3] -= np.float16(4.0)                     # thin density: most rays keep every sample and see the background
    B = 1 << 18
    plain_o, _ = _run(ngp, oracle, cuda, I, 4, B, linear=linear)
    I["envmap"] = env
    o, g = _run(ngp, oracle, cuda, I, 4, B, linear=linear)
    n_alive = I["n_alive"]
    assert np.abs(o["loss"] - plain_o["loss"]).max() > 1e-7  # the map matters
    bad = _borderline_rays(I)
    ok = np.array([i not in bad for i in range(n_alive)])
    np.testing.assert_array_equal(g["ns"][0::2][:n_alive][ok], o["ns"][0::2][:n_alive][ok])
    np.testing.assert_allclose(g["loss"][:n_alive][ok], o["loss"][:n_alive][ok], rtol=4e-3, atol=1e-9)
    ref, got = o["envg"], g["envg"]
    assert np.abs(ref[..., :3]).max() > 0 and np.abs(ref[..., 3]).max() == 0 and np.abs(got[..., 3]).max() == 0     # alpha gets no gradient (:1592-1593)
    assert np.linalg.norm(got - ref) < 1e-2 * np.linalg.norm(ref)
    # without train_envmap the map is still composited, and nothing is deposited
    I["envmap"] = dict(env, train=False)
    o2, g2 = _run(ngp, oracle, cuda, I, 4, B, linear=linear)
    np.testing.assert_array_equal(o2["loss"], o["loss"])
    assert np.abs(g2["envg"]).max() == 0


def test_cam_gradient_distortion_branch(ngp, oracle, cuda):
    """compute_cam_gradient_train_nerf's distortion outputs (:1671-1685): image-plane gradient splatted into the distortion map, weights beside it; safe_divide."""
    I = _inputs(oracle, cuda, n_rays=2048, seed=5)
    B = 1 << 15
    o, _ = _run(ngp, oracle, cuda, I, 4, B)
    n_rays, n_alive, n_img = I["n_rays"], I["n_alive"], len(I["xf"])
    rs = np.random.RandomState(9)
    cg = (rs.randn(B, 6) * 0.1).astype(np.float32)
    dres = np.array([12, 10], np.int32)
    ref_g, ref_w = np.zeros((10, 12, 2), np.float32), np.zeros((10, 12, 2), np.float32)
    oracle.orc_compute_cam_gradient_ex(n_rays, I["aabb"].ctypes.data, I["st"], I["inc"], n_alive, 0, None, None, n_img, I["md_host"].ctypes.data, I["r"]["idx"].ctypes.data,
                                       I["r"]["rays"].ctypes.data, o["ns"].ctypes.data, o["co"].ctypes.data, cg.ctypes.data, None, I["xf"].ctypes.data, ref_g.ctypes.data, ref_w.ctypes.data,
                                       dres.ctypes.data)
    d_g, d_w = H.dev_zeros(ref_g.nbytes, cuda), H.dev_zeros(ref_w.nbytes, cuda)
    d_rc = H.to_dev(np.array([n_alive], np.uint32), cuda)
    d_md, d_idx, d_rays, d_ns, d_co, d_cg, d_xf = (H.to_dev(a, cuda) for a in (I["md_dev"], I["r"]["idx"], I["r"]["rays"], o["ns"], o["co"], cg, I["xf"]))
    check(ngp.ngp_hip_compute_cam_gradient(None, n_rays, I["aabb"].ctypes.data, I["st"], I["inc"], d_rc.data_ptr(), 0, None, None, n_img, d_md.data_ptr(), d_idx.data_ptr(), d_rays.data_ptr(),
                                              d_ns.data_ptr(), d_co.data_ptr(), d_cg.data_ptr(), None, d_xf.data_ptr(), d_g.data_ptr(), d_w.data_ptr(), dres.ctypes.data))
    got_g, got_w = H.to_host(d_g, np.float32).reshape(ref_g.shape), H.to_host(d_w, np.float32).reshape(ref_w.shape)
    assert ref_w.min() >= 0 and ref_w.sum() > 100 and np.abs(ref_g).max() > 0
    np.testing.assert_allclose(got_w, ref_w, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(got_g, ref_g, rtol=0, atol=3e-4 * np.abs(ref_g).max())
    # safe_divide: gradient / weight where the weight is positive, else 0
    ref_div = ref_g.copy().reshape(-1)
    ref_w_flat = ref_w.reshape(-1).copy(); ref_w_flat[::7] = 0.0
    oracle.orc_safe_divide(ref_div.size, ref_div.ctypes.data, ref_w_flat.ctypes.data)
    d_div, d_wz = H.to_dev(ref_g.reshape(-1), cuda), H.to_dev(ref_w_flat, cuda)
    check(ngp.ngp_hip_safe_divide(None, ref_div.size, d_div.data_ptr(), d_wz.data_ptr()))
    np.testing.assert_array_equal(H.to_host(d_div, np.float32), ref_div)
    assert (ref_div[::7] == 0).all()
    # the distortion outputs need the transforms and the weight buffer
    assert ngp.ngp_hip_compute_cam_gradient(None, n_rays, I["aabb"].ctypes.data, I["st"], I["inc"], d_rc.data_ptr(), 0, None, None, n_img, d_md.data_ptr(), d_idx.data_ptr(), d_rays.data_ptr(),
                                               d_ns.data_ptr(), d_co.data_ptr(), d_cg.data_ptr(), None, None, d_g.data_ptr(), d_w.data_ptr(), dres.ctypes.data) != 0


@pytest.mark.parametrize("use_ema", [0, 1])
def test_trainable_buffer_optimizer_step(ngp, oracle, cuda, use_ema):
    """[tcnn] Adam (+ Ema) over an all-fp32 TrainableBuffer: zero-gradient entries are skipped, the rest follow Adam's update rule; three steps in a row."""
    rs = np.random.RandomState(3)
    n = 5000
    p, m1, m2, ema = rs.randn(n).astype(np.float32) * 0.1, np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    d_p, d_m1, d_m2, d_ema = (H.to_dev(a, cuda) for a in (p, m1, m2, ema))
    for step in (1, 2, 3):
        g = (rs.randn(n) * 50.0).astype(np.float32)
        g[rs.rand(n) < 0.3] = 0.0
        before = p.copy()
        oracle.orc_optimizer_step_f32(n, step, H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-10), H.f32(128.0), H.f32(0.99), g.ctypes.data, p.ctypes.data, m1.ctypes.data, m2.ctypes.data,
                                      ema.ctypes.data if use_ema else None)
        d_g = H.to_dev(g, cuda)
        check(ngp.ngp_hip_optimizer_step_f32(None, n, step, H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-10), H.f32(128.0), H.f32(0.99), d_g.data_ptr(), d_p.data_ptr(),
                                             d_m1.data_ptr(), d_m2.data_ptr(), d_ema.data_ptr() if use_ema else None))
        assert (p[g == 0] == before[g == 0]).all() and (p[g != 0] != before[g != 0]).all()
        np.testing.assert_allclose(H.to_host(d_p, np.float32), p, rtol=2e-6, atol=1e-8)
        np.testing.assert_allclose(H.to_host(d_m2, np.float32), m2, rtol=2e-6, atol=1e-12)
        if use_ema:
            np.testing.assert_allclose(H.to_host(d_ema, np.float32), ema, rtol=4e-6, atol=1e-8)
    if use_ema:
        assert np.abs(ema - p).max() > 0 and np.abs(ema - p).max() < 0.2
    assert ngp.ngp_hip_optimizer_step_f32(None, n, 0, H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-10), H.f32(128.0), H.f32(0.99), d_p.data_ptr(), d_p.data_ptr(), d_m1.data_ptr(), d_m2.data_ptr(), None) != 0


def test_loss_with_sharpness_in_error(ngp, oracle, cuda):
    """include_sharpness_in_error (testbed_nerf.cu:1321-1323, 1346, 1367, 1374, 1476-1485): the error a ray deposits is scaled by its image tile's sharpness relative to the
    sharpest tile that has seen the cell of the ray's expected hit point (a cascaded grid, atomicMax on float bits); everything else of the kernel is untouched.  Rays race
    for the cells like in the reference: the grid's final content (a max) is order-independent, a ray's scale factor is not when two rays share a cell — so the error map is
    held to the oracle where the grid started at its final values (second pass), bit-level agreement of the grid itself in the first."""
    I = _inputs(oracle, cuda, n_rays=1024, seed=21)
    n_img = len(I["xf"])
    rs = np.random.RandomState(3)
    sharp = dict(data=(rs.rand(n_img, 9, 16).astype(np.float32) * 0.01), res=(16, 9), grid0=np.zeros(128 ** 3 * 8, np.float32))
    base_o, base_g = _run(ngp, oracle, cuda, dict(I), 0, B=I["n_samples"] + 64)
    I1 = dict(I); I1["sharp"] = sharp
    o, g = _run(ngp, oracle, cuda, I1, 0, B=I["n_samples"] + 64)
    # the max over the rays of each cell is order-independent: bit for bit, except where a ray's expected hit point (a wave reduction here, a sequential sum in the
    # oracle) sits on a cell boundary and lands next door
    assert (g["sgrid"] != o["sgrid"]).sum() <= 8 and (o["sgrid"] > 0).sum() > 100
    np.testing.assert_array_equal(g["loss"], base_g["loss"])                   # only the error-map deposit is scaled (per-ray losses: same bits)
    # second pass on the converged grid: every ray's factor is sharp / max(sharp, grid) whatever the order
    I2 = dict(I); I2["sharp"] = dict(sharp, grid0=o["sgrid"].copy())
    o2, g2 = _run(ngp, oracle, cuda, I2, 0, B=I["n_samples"] + 64)
    assert (g2["sgrid"] != o2["sgrid"]).sum() <= 8
    assert np.linalg.norm(g2["em"] - o2["em"]) < 5e-3 * np.linalg.norm(o2["em"])
    assert 0 < o2["em"].sum() < base_o["em"].sum()                             # rays that share a cell with a sharper tile's ray count for less, nobody for more


def test_compute_sharpness_matches_oracle(ngp, oracle, cuda):
    """compute_sharpness (nerf_loader.cu:129-169): variance of the Laplacian of the luma per tile, Byte and Half images"""
    rs = np.random.RandomState(8)
    w, h = 200, 120
    img8 = rs.randint(0, 256, (h, w, 4)).astype(np.uint8)
    img8[: h // 2] = 128                                                        # a flat half: zero sharpness
    img8[10:20, 10:20] = [255, 0, 255, 0]                                       # the "masked-away" colour reads as -1 (common_device.cuh:685-687)
    img16 = rs.rand(h, w, 4).astype(np.float16)
    for img, typ in ((img8, 1), (img16, 2)):
        sres, ires = np.array([16, 9], np.int32), np.array([w, h], np.int32)
        want = np.zeros((9, 16), np.float32)
        oracle.orc_compute_sharpness(sres.ctypes.data, ires.ctypes.data, img.ctypes.data, typ, want.ctypes.data)
        d_img, d_out = H.to_dev(img, cuda), H.dev_zeros(9 * 16 * 4, cuda)
        check(ngp.ngp_hip_compute_sharpness(None, sres.ctypes.data, ires.ctypes.data, d_img.data_ptr(), typ, d_out.data_ptr()))
        got = H.to_host(d_out, np.float32).reshape(9, 16)
        assert want.max() > 1e-3
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-7)
        if typ == 1:
            assert abs(got[1, 8]) < 1e-6                                        # inside the flat half
