"""GPU parity: the rest of the stock renderer vs the CPU oracle, through the C ABI.

init_rays with crop masks / quilting / envmap background / distortion map / Distortion mode (src/testbed_nerf.cu:1809-1978), composite with
crop masks, glow and the Normals / EncodingVis modes (:767-989), shade in Normals mode (:1748-1781), the Slice-mode kernels (:676-703), the
[tcnn] input_gradient and visualize_activation passes the tracer calls (:2225-2228).  Ray set-up is exact apart from the libm-vs-device
transcendentals (atan2 / acos of the envmap lookup, atan of the quilting parallax); colours rtol 2e-3 like the other render tests.
"""
import numpy as np
import pytest

import capi
import helpers as H
from capi import check

pytestmark = pytest.mark.gpu
W, Hh = 96, 64


def _mat4(m):
    return np.asarray(m, np.float32).T.reshape(-1).copy()


def _mask(shape, mode, transform, config, feather, opacity):
    m = np.zeros(1, capi.MASK3D)
    m["mode"], m["shape"] = mode, shape
    m["transform"][0] = _mat4(transform)
    m["itransform"][0] = _mat4(np.linalg.inv(transform))
    cfg = np.zeros(6, np.float32); cfg[:len(config)] = config
    m["config"][0] = cfg
    m["feather"], m["opacity"] = feather, opacity
    return m


def _translate(t):
    m = np.eye(4); m[:3, 3] = t
    return m


def _render_masks():
    """Testbed::prepare_nerf_masks (:2339-2352): an `All` mask of the opposite mode in front of a first mask that is not `All`."""
    box = _mask(0, 0, _translate([0.5, 0.5, 0.5]), [0.5, 0.4, 0.6], 0.05, 1.0)
    sph = _mask(2, 1, _translate([0.55, 0.5, 0.45]), [0.12], 0.02, 0.8)
    return np.concatenate([_mask(3, 1, np.eye(4), [], 0.0, 1.0), box, sph])


def _extras(**kw):
    e = np.zeros(1, capi.RENDER_EXTRAS)
    e["quilting_dims"][0] = (1, 1)
    e["render_mode"] = 1
    for k, v in kw.items():
        e[k][0] = v
    return e


def _camera():
    cam = H.look_at_xform([1.6, 1.3, 1.1])
    return cam, np.array([80.0, 80.0], np.float32), np.array([W, Hh], np.int32), np.array([0.5, 0.5], np.float32)


def _init(ngp, oracle, cuda, ex_host, ex_dev, spp=1, parallax=(0.0, 0.0, 0.0)):
    cam, focal, res, sc = _camera()
    aabb = H.unit_aabb(1)
    n = W * Hh
    ident = np.eye(3, dtype=np.float32).reshape(-1)
    zero4, par = np.zeros(4, np.float32), np.array(parallax, np.float32)
    pay, depth = np.zeros(n, H.PAYLOAD), np.zeros(n, np.float32)
    oracle.orc_init_rays_ex(spp, pay.ctypes.data, res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, zero4.ctypes.data, sc.ctypes.data, par.ctypes.data,
                            0, aabb.ctypes.data, ident.ctypes.data, H.f32(0.0), 0, None, depth.ctypes.data, H.f32(1.0), H.f32(0.0), None, ex_host.ctypes.data)
    d_pay, d_depth = H.dev_zeros(n * 40, cuda), H.dev_zeros(n * 4, cuda)
    check(ngp.ngp_hip_init_rays(None, spp, d_pay.data_ptr(), res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, zero4.ctypes.data, sc.ctypes.data, par.ctypes.data,
                                   0, aabb.ctypes.data, ident.ctypes.data, H.f32(0.0), 0, None, d_depth.data_ptr(), H.f32(1.0), H.f32(0.0), None, ex_dev.ctypes.data))
    return pay, depth, H.to_host(d_pay, H.PAYLOAD).copy(), H.to_host(d_depth, np.float32).copy()


def test_init_rays_crop_masks_kill_rays_bit_exact(ngp, oracle, cuda):
    masks = np.concatenate([_mask(0, 0, _translate([0.5, 0.5, 0.5]), [0.3, 0.3, 0.3], 0.0, 1.0)])   # one Add box: rays that miss it die
    d_masks = H.to_dev(masks, cuda)
    pay, depth, g, gd = _init(ngp, oracle, cuda, _extras(render_masks=masks.ctypes.data, n_render_masks=1), _extras(render_masks=d_masks.data_ptr(), n_render_masks=1))
    plain, _, _, _ = _init(ngp, oracle, cuda, _extras(), _extras())
    assert 0 < pay["alive"].sum() < plain["alive"].sum()
    np.testing.assert_array_equal(g["alive"], pay["alive"])
    al = pay["alive"] == 1
    for f in ("origin", "dir", "t", "idx"):
        np.testing.assert_array_equal(g[f][al], pay[f][al])
    np.testing.assert_array_equal(gd, depth)


@pytest.mark.parametrize("dims,parallax", [((2, 1), (0.06, 0.0, 0.0)), ((4, 2), (0.0, 0.0, 0.5))])
def test_init_rays_quilting(ngp, oracle, cuda, dims, parallax):
    pay, depth, g, gd = _init(ngp, oracle, cuda, _extras(quilting_dims=dims), _extras(quilting_dims=dims), parallax=parallax)
    np.testing.assert_array_equal(g["alive"], pay["alive"])
    al = pay["alive"] == 1
    assert al.sum() > 100
    # (2, 1): the two eyes' origins differ by the IPD, exact; the lenticular panels go through atanf (device vs libm)
    tol = 0 if dims == (2, 1) else 2e-6
    np.testing.assert_allclose(g["origin"][al], pay["origin"][al], rtol=0, atol=tol)
    np.testing.assert_allclose(g["dir"][al], pay["dir"][al], rtol=0, atol=tol)
    np.testing.assert_allclose(g["t"][al], pay["t"][al], rtol=0, atol=tol * 4)
    half = pay["origin"].reshape(Hh, W, 3)
    assert not np.array_equal(half[:, : W // dims[0]], half[:, -(W // dims[0]):])   # panels see the scene from different points


def test_init_rays_envmap_background_and_distortion(ngp, oracle, cuda):
    rs = np.random.RandomState(3)
    env = rs.rand(16, 32, 4).astype(np.float32)
    dist = (rs.randn(8, 8, 2) * 0.02).astype(np.float32)
    d_env, d_dist = H.to_dev(env, cuda), H.to_dev(dist, cuda)
    n = W * Hh
    fb = np.zeros((n, 4), np.float32)
    d_fb = H.dev_zeros(n * 16, cuda)
    kw = dict(envmap_res=(32, 16), distortion_res=(8, 8))
    pay, depth, g, gd = _init(ngp, oracle, cuda, _extras(envmap=env.ctypes.data, distortion=dist.ctypes.data, frame_buffer=fb.ctypes.data, **kw),
                              _extras(envmap=d_env.data_ptr(), distortion=d_dist.data_ptr(), frame_buffer=d_fb.data_ptr(), **kw))
    plain, _, _, _ = _init(ngp, oracle, cuda, _extras(), _extras())
    np.testing.assert_array_equal(g["alive"], pay["alive"])
    al = pay["alive"] == 1
    np.testing.assert_array_equal(g["dir"][al], pay["dir"][al])      # the distortion offset is plain fp32 arithmetic: exact
    assert np.abs(pay["dir"][al & (plain["alive"] == 1)] - plain["dir"][al & (plain["alive"] == 1)]).max() > 1e-3
    gfb = H.to_host(d_fb, np.float32).reshape(n, 4)
    assert fb.min() > 0.0                                              # every pixel got its background
    np.testing.assert_allclose(gfb, fb, rtol=0, atol=2e-4)             # acosf / atan2f: a texel-weight ulp, not a texel
    # Distortion render mode (:1959-1970): the offset painted as a colour, every ray done
    fb2 = np.zeros((n, 4), np.float32); d_fb2 = H.dev_zeros(n * 16, cuda)
    pay2, depth2, g2, gd2 = _init(ngp, oracle, cuda, _extras(distortion=dist.ctypes.data, frame_buffer=fb2.ctypes.data, render_mode=5, distortion_res=(8, 8)),
                                  _extras(distortion=d_dist.data_ptr(), frame_buffer=d_fb2.data_ptr(), render_mode=5, distortion_res=(8, 8)))
    assert pay2["alive"].sum() == 0 and g2["alive"].sum() == 0
    hit = depth2 == 1.0
    assert hit.sum() > 100
    np.testing.assert_array_equal(gd2, depth2)
    np.testing.assert_allclose(H.to_host(d_fb2, np.float32).reshape(n, 4)[hit], fb2[hit], rtol=0, atol=2e-4)
    assert (fb2[hit][:, 3] == 1.0).all() and fb2[hit][:, :3].max() > 0.3
    # an envmap / the Distortion mode without a frame buffer is refused
    assert ngp.ngp_hip_init_rays(None, 0, 0, np.array([4, 4], np.int32).ctypes.data, np.ones(2, np.float32).ctypes.data, np.eye(3, 4, dtype=np.float32).ctypes.data,
                                    np.eye(3, 4, dtype=np.float32).ctypes.data, None, np.ones(2, np.float32).ctypes.data, None, 0, H.unit_aabb(1).ctypes.data, None, H.f32(0.0), 0, None, 0,
                                    H.f32(1.0), H.f32(0.0), None, _extras(render_mode=5).ctypes.data) != 0


def _march_inputs(oracle, n_steps=4):
    """Oracle-side rays -> compacted -> next inputs: the state in front of composite."""
    cam, focal, res, sc = _camera()
    aabb = H.unit_aabb(1)
    grid = H.blob_density_grid(1)
    bf, _ = H.oracle_bitfield(oracle, grid, 1)
    n = W * Hh
    ident = np.eye(3, dtype=np.float32).reshape(-1)
    zero4, zero3 = np.zeros(4, np.float32), np.zeros(3, np.float32)
    pay, depth = np.zeros(n, H.PAYLOAD), np.zeros(n, np.float32)
    oracle.orc_init_rays(0, pay.ctypes.data, res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, zero4.ctypes.data, sc.ctypes.data, zero3.ctypes.data,
                         1, aabb.ctypes.data, ident.ctypes.data, H.f32(0.0), 0, None, depth.ctypes.data, H.f32(1.0), H.f32(0.0), None)
    oracle.orc_advance_pos(n, aabb.ctypes.data, ident.ctypes.data, 0, pay.ctypes.data, bf.ctypes.data, 0, H.f32(0.0))
    alive = pay[pay["alive"] == 1].copy()
    na = len(alive)
    coords = np.zeros(na * n_steps, H.COORD)
    oracle.orc_generate_next_inputs(na, aabb.ctypes.data, aabb.ctypes.data, alive.ctypes.data, coords.ctypes.data, n_steps, bf.ctypes.data, 0, H.f32(0.0))
    return dict(aabb=aabb, cam=cam, pay=alive, coords=coords, n=na, n_steps=n_steps)


def _composite_both(ngp, oracle, cuda, S, out, mode, ex_host, ex_dev, coords=None, accel=-1):
    na, n_steps = S["n"], S["n_steps"]
    coords = S["coords"] if coords is None else coords
    o_p, o_c, o_d = S["pay"].copy(), np.zeros((na, 4), np.float32), np.zeros(na, np.float32)
    oracle.orc_composite_ex(na, 1, S["aabb"].ctypes.data, S["cam"].ctypes.data, o_c.ctypes.data, o_d.ctypes.data, o_p.ctypes.data, coords.ctypes.data, out.ctypes.data, 4, n_steps, 2, 3,
                            H.f32(0.01), mode, H.f32(3.0), accel, ex_host.ctypes.data if ex_host is not None else None)
    d_p, d_c, d_d = H.to_dev(S["pay"], cuda), H.dev_zeros(na * 16, cuda), H.dev_zeros(na * 4, cuda)
    d_in, d_out = H.to_dev(coords, cuda), H.to_dev(out, cuda)
    check(ngp.ngp_hip_composite(None, na, 1, S["aabb"].ctypes.data, S["cam"].ctypes.data, d_c.data_ptr(), d_d.data_ptr(), d_p.data_ptr(), d_in.data_ptr(), d_out.data_ptr(), 4, n_steps, 2, 3,
                                   H.f32(0.01), mode, H.f32(3.0), accel, ex_dev.ctypes.data if ex_dev is not None else None))
    ok = H.to_host(d_p, H.PAYLOAD)["alive"] == o_p["alive"]
    assert ok.mean() > 0.99
    g_c = H.to_host(d_c, np.float32).reshape(na, 4)
    np.testing.assert_allclose(g_c[ok], o_c[ok], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(H.to_host(d_d, np.float32)[ok], o_d[ok], rtol=2e-3, atol=1e-5)
    return o_c, o_p


def _random_outputs(n, seed=0):
    rs = np.random.RandomState(seed)
    out = np.zeros((n, 4), np.float16)
    out[:, :3] = rs.randn(n, 3).astype(np.float16)
    out[:, 3] = (rs.randn(n) * 2 + 2).astype(np.float16)
    return out


def test_composite_crop_masks_and_glow(ngp, oracle, cuda):
    S = _march_inputs(oracle)
    out = _random_outputs(S["n"] * S["n_steps"])
    plain, _ = _composite_both(ngp, oracle, cuda, S, out, 1, None, None)
    masks = _render_masks()
    d_masks = H.to_dev(masks, cuda)
    masked, _ = _composite_both(ngp, oracle, cuda, S, out, 1, _extras(render_masks=masks.ctypes.data, n_render_masks=len(masks)), _extras(render_masks=d_masks.data_ptr(), n_render_masks=len(masks)))
    assert masked[:, 3].sum() < 0.9 * plain[:, 3].sum() and masked[:, 3].sum() > 0.05 * plain[:, 3].sum()
    ys = S["coords"]["pos"][:, 1]
    cutoff = float(np.median(ys[ys > 0])) + 0.05            # the glow band is the 0.26 below the cut-off: put the samples into it
    for glow_mode in (1, 2, 3, 4 | 1, 8 | 1, 16, 16 | 8):
        if glow_mode & 8 and not glow_mode & 16:            # radial: the distance from the camera (about 1.5 here), capped at (4.5 - y) / 3
            cutoff = 1.4
        glowing, _ = _composite_both(ngp, oracle, cuda, S, out, 1, _extras(glow_mode=glow_mode, glow_y_cutoff=cutoff), _extras(glow_mode=glow_mode, glow_y_cutoff=cutoff))
        assert np.abs(glowing - plain).max() > 1e-2, glow_mode
    # show_accel >= 0 makes every sample opaque (:827-829)
    opaque, p = _composite_both(ngp, oracle, cuda, S, out, 3, None, None, accel=0)
    assert (p["alive"] == 0).all() and np.allclose(opaque[:, 3], 1.0)


def test_input_gradient_and_normals_mode(ngp, oracle, cuda):
    """Tracer in Normals mode: input_gradient(dim 3) over the samples in place, then composite + shade."""
    S = _march_inputs(oracle)
    na, n_steps = S["n"], S["n_steps"]
    n = na * n_steps
    n_pad = (n + 255) // 256 * 256
    desc = H.make_desc(ngp, log2_hashmap_size=15)
    params = H.random_params(desc, seed=5)
    coords = np.zeros(n_pad, H.COORD)
    coords[:n] = S["coords"]
    coords["pos"][n:] = 0.5; coords["dir"][n:] = 0.5
    # samples past a ray's n_steps are never written by the march: give them valid positions
    junk = np.zeros(n_pad, bool); junk[:n] = (S["coords"]["pos"] == 0).all(axis=1)
    rs = np.random.RandomState(7)
    coords["pos"][junk] = rs.rand(int(junk.sum()), 3).astype(np.float32); coords["dir"][junk] = 0.5
    flat = coords.view(np.float32).reshape(n_pad, 7)
    # oracle: orc_nerf_input_gradient with the one-hot dL/doutput scaled by 128, result / 128 ([tcnn] input_gradient)
    dl = np.zeros((n_pad, 4), np.float16); dl[:, 3] = 128.0
    din = np.zeros((n_pad, 6), np.float32)
    oracle.orc_nerf_input_gradient(desc.ctypes.data, params.ctypes.data, flat.ctypes.data, 7, n_pad, dl.ctypes.data, din.ctypes.data)
    ref = flat.copy()
    ref[:, 0:3] = din[:, 0:3] / 128.0; ref[:, 3] = flat[:, 3] / 128.0; ref[:, 4:7] = din[:, 3:6] / 128.0
    d_desc, d_p, d_c = H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(flat, cuda)
    sb = ngp.ngp_hip_nerf_input_gradient_scratch_bytes(n_pad)
    d_s = H.dev_zeros(sb, cuda)
    check(ngp.ngp_hip_nerf_input_gradient(None, d_desc.data_ptr(), desc.ctypes.data, d_p.data_ptr(), 3, d_c.data_ptr(), 7, n_pad, d_s.data_ptr(), sb))
    got = H.to_host(d_c, np.float32).reshape(n_pad, 7)
    scale = np.abs(ref[:, 0:3]).max()
    assert scale > 1e-3
    assert np.abs(got[:, 0:3] - ref[:, 0:3]).max() < 3e-2 * scale          # fp16 backward through a 64-wide layer
    assert np.linalg.norm(got[:, 0:3] - ref[:, 0:3]) < 1e-2 * np.linalg.norm(ref[:, 0:3])
    np.testing.assert_array_equal(got[:, 3], ref[:, 3])
    assert np.abs(got[:, 4:7]).max() == 0.0 and np.abs(ref[:, 4:7]).max() == 0.0   # the density output does not depend on the direction
    # dim 0 (red) does depend on the direction
    d_c2 = H.to_dev(flat, cuda)
    check(ngp.ngp_hip_nerf_input_gradient(None, d_desc.data_ptr(), desc.ctypes.data, d_p.data_ptr(), 0, d_c2.data_ptr(), 7, n_pad, d_s.data_ptr(), sb))
    dl0 = np.zeros((n_pad, 4), np.float16); dl0[:, 0] = 128.0
    oracle.orc_nerf_input_gradient(desc.ctypes.data, params.ctypes.data, flat.ctypes.data, 7, n_pad, dl0.ctypes.data, din.ctypes.data)
    got0 = H.to_host(d_c2, np.float32).reshape(n_pad, 7)
    assert np.linalg.norm(got0[:, 4:7] - din[:, 3:6] / 128.0) < 2e-2 * np.linalg.norm(din[:, 3:6] / 128.0) and np.abs(din[:, 3:6]).max() > 0
    assert ngp.ngp_hip_nerf_input_gradient(None, d_desc.data_ptr(), desc.ctypes.data, d_p.data_ptr(), 3, d_c.data_ptr(), 7, n_pad - 1, d_s.data_ptr(), sb) != 0

    # composite in Normals mode on the ORACLE's gradients (both sides), then shade
    grad_coords = ref[:n].copy().view(H.COORD).reshape(-1)
    out = np.zeros((n, 4), np.float16)
    oracle.orc_nerf_inference(desc.ctypes.data, params.ctypes.data, flat.ctypes.data, 7, n, out.ctypes.data, 4)
    out[:, 3] += np.float16(3.0)   # random weights give little density: make the rays terminate
    nrm, p_after = _composite_both(ngp, oracle, cuda, S, out, 2, None, None, coords=grad_coords)
    shade_out = _random_outputs(n, 1)
    assert nrm[:, 3].max() > 0.05 and (np.linalg.norm(nrm[:, :3], axis=1) <= nrm[:, 3] + 1e-4).all()   # a sum of unit vectors with weights that add up to alpha
    res = np.array([W, Hh], np.int32)
    fb_o, db_o = np.zeros((W * Hh, 4), np.float32), np.zeros(W * Hh, np.float32)
    dep = np.linspace(1, 2, na).astype(np.float32)
    oracle.orc_shade_mode(na, nrm.ctypes.data, dep.ctypes.data, p_after.ctypes.data, 0, fb_o.ctypes.data, db_o.ctypes.data, 2)
    d_fb, d_db = H.dev_zeros(W * Hh * 16, cuda), H.dev_zeros(W * Hh * 4, cuda)
    d_nrm, d_dep, d_pa = H.to_dev(nrm, cuda), H.to_dev(dep, cuda), H.to_dev(p_after, cuda)
    check(ngp.ngp_hip_shade(None, na, d_nrm.data_ptr(), d_dep.data_ptr(), d_pa.data_ptr(), 0, d_fb.data_ptr(), d_db.data_ptr(), 2))
    np.testing.assert_allclose(H.to_host(d_fb, np.float32).reshape(-1, 4), fb_o, rtol=1e-4, atol=1e-6)
    hit = fb_o[:, 3] > 0.05
    assert hit.sum() > 50 and fb_o[hit][:, :3].min() >= -1e-6 and (fb_o[hit][:, :3] <= fb_o[hit][:, 3:4] + 1e-6).all()   # (0.5 n + 0.5) alpha


@pytest.mark.parametrize("layer,dim", [(0, 5), (0, 31), (1, 17), (2, 0), (2, 3), (2, 16), (2, 29), (3, 40), (4, 63)])
def test_visualize_activation(ngp, oracle, cuda, layer, dim):
    n = 3000
    desc = H.make_desc(ngp, log2_hashmap_size=15)
    params = H.random_params(desc, seed=2)
    coords = H.random_coords(n, seed=3)
    flat = coords.view(np.float32).reshape(n, 7)
    d_desc, d_p, d_c = H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(flat, cuda)
    for stride in (7, 4):
        ref = np.zeros((n, stride), np.float32)
        oracle.orc_nerf_visualize_activation(desc.ctypes.data, params.ctypes.data, layer, dim, flat.ctypes.data, 7, n, ref.ctypes.data, stride)
        d_o = H.dev_zeros(n * stride * 4, cuda) if stride == 4 else H.to_dev(flat, cuda)
        src = d_c if stride == 4 else d_o      # stride 7: in place over the coordinates, as the tracer calls it
        check(ngp.ngp_hip_nerf_visualize_activation(None, d_desc.data_ptr(), d_p.data_ptr(), layer, dim, src.data_ptr(), 7, n, d_o.data_ptr(), stride))
        got = H.to_host(d_o, np.float32).reshape(n, stride)
        scale = max(np.abs(ref[:, :2]).max(), 1e-3)
        assert np.abs(ref[:, :2]).max() > 0
        np.testing.assert_allclose(got[:, :2], ref[:, :2], rtol=0, atol=1e-2 * scale)
        assert (got[:, 2] == 0).all() and (got[:, 3:] == 1).all()
        assert (np.minimum(got[:, 0], got[:, 1]) == 0).all()        # negative part / positive part
    assert ngp.ngp_hip_nerf_visualize_activation(None, d_desc.data_ptr(), d_p.data_ptr(), 5, 0, d_c.data_ptr(), 7, n, d_o.data_ptr(), 4) != 0
    assert ngp.ngp_hip_nerf_visualize_activation(None, d_desc.data_ptr(), d_p.data_ptr(), 2, 32, d_c.data_ptr(), 7, n, d_o.data_ptr(), 4) != 0


def test_encoding_vis_composite_and_slice_kernels(ngp, oracle, cuda):
    S = _march_inputs(oracle)
    n = S["n"] * S["n_steps"]
    vis = S["coords"].copy()
    rs = np.random.RandomState(11)
    vis["pos"] = rs.rand(n, 3).astype(np.float32)
    vis["dt"] = 1.0                                                      # what extract_dimension_pos_neg leaves in the dt row
    out = _random_outputs(n, 4)
    col, _ = _composite_both(ngp, oracle, cuda, S, out, 8, None, None, coords=vis)
    assert col[:, :3].max() > 0.3
    # Slice mode kernels
    pay = S["pay"].copy()
    pay["t"] = np.linspace(0.5, 1.5, len(pay)).astype(np.float32)
    o_in = np.zeros(len(pay), H.COORD)
    oracle.orc_generate_inputs_at_current_position(len(pay), S["aabb"].ctypes.data, pay.ctypes.data, o_in.ctypes.data)
    d_in = H.dev_zeros(len(pay) * 28, cuda)
    d_pay = H.to_dev(pay, cuda)
    check(ngp.ngp_hip_generate_inputs_at_current_position(None, len(pay), S["aabb"].ctypes.data, d_pay.data_ptr(), d_in.data_ptr()))
    assert H.to_host(d_in, H.COORD).tobytes() == o_in.tobytes()
    assert (o_in["dt"] == 0).all()
    for density_as_alpha in (0, 1):
        o_rgba = np.zeros((n, 4), np.float32)
        oracle.orc_compute_nerf_rgba(n, out.ctypes.data, 4, o_rgba.ctypes.data, 2, 3, H.f32(0.01), density_as_alpha)
        d_rgba = H.dev_zeros(n * 16, cuda)
        d_out = H.to_dev(out, cuda)
        check(ngp.ngp_hip_compute_nerf_rgba(None, n, d_out.data_ptr(), 4, d_rgba.data_ptr(), 2, 3, H.f32(0.01), density_as_alpha))
        np.testing.assert_allclose(H.to_host(d_rgba, np.float32).reshape(n, 4), o_rgba, rtol=2e-3, atol=1e-6)
