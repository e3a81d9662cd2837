"""GPU parity tests of the Blender multi-NeRF renderer kernels (include/ngp_hip.h `ngp_hip_multi_*`, reference src/nerf_renderer.cu)
against the CPU oracle (oracle/orc_multi.c), kernel by kernel and for a whole two-NeRF frame.

Tolerances: index / flag / step-count fields are compared exactly; float fields that only go through + - * / sqrt exactly as well (both
sides are built with FP contraction off); fields that involve sinf / cosf / atan2f / expf use the stated absolute tolerance."""
import ctypes

import numpy as np
import pytest

import capi
import helpers as H

pytestmark = pytest.mark.gpu
check = capi.check
MIN_STEP = np.float32(np.sqrt(np.float32(3.0)) / np.float32(1024.0))
SQRT3 = np.float32(1.73205080757)


def _ds(orc, w, h, mip):
    ds = np.zeros(1, capi.DOWNSAMPLE_INFO)
    res = np.array([w, h], np.int32)
    orc.orc_downsample_info_from_mip(res.ctypes.data, mip, ds.ctypes.data)
    return ds


def _mat4(m):
    return np.asarray(m, np.float32).T.reshape(-1).copy()   # column-major flat


def _trs(t=(0, 0, 0), rot_z=0.0, s=1.0):
    c, sn = np.cos(rot_z), np.sin(rot_z)
    m = np.array([[c * s, -sn * s, 0, t[0]], [sn * s, c * s, 0, t[1]], [0, 0, s, t[2]], [0, 0, 0, 1]], np.float64)
    return m


def _mask(shape, mode, transform, config, feather, opacity):
    m = np.zeros(1, capi.MASK3D)
    m["mode"], m["shape"] = mode, shape
    m["transform"][0] = _mat4(transform)
    m["itransform"][0] = _mat4(np.linalg.inv(transform))
    cfg = np.zeros(6, np.float32); cfg[:len(config)] = config
    m["config"][0] = cfg
    m["feather"], m["opacity"] = feather, opacity
    return m


def _with_all_mask(masks):
    """RenderModifiers::copy_from_host (render_modifiers.cuh:53-62): an implicit `All` mask of the opposite mode is prepended."""
    if not masks:
        return np.zeros(0, capi.MASK3D)
    first = masks[0]
    mode = 1 if first["mode"][0] == 0 else 0
    return np.concatenate([_mask(3, mode, np.eye(4), [], 0.0, 1.0)] + masks)


def _props(transform, bitfield_ptr, masks_ptr, n_masks, aabb_scale=1, opacity=1.0, render_aabb=None):
    p = np.zeros(1, capi.NERF_PROPS)
    p["transform"][0] = _mat4(transform)
    p["itransform"][0] = _mat4(np.linalg.inv(transform))
    p["density_grid_bitfield"] = bitfield_ptr
    p["grid_size"], p["grid_volume"] = 128, 128 ** 3
    aabb = H.unit_aabb(aabb_scale)
    p["train_aabb"] = aabb
    p["render_aabb"] = aabb if render_aabb is None else render_aabb
    p["masks"], p["n_masks"] = masks_ptr, n_masks
    p["cone_angle"] = 0.0 if aabb_scale <= 1 else 1.0 / 256.0
    p["min_cone_stepsize"] = SQRT3 / np.float32(1024.0)
    p["max_cone_stepsize"] = SQRT3 / np.float32(1024.0) * np.float32(128 * 1024 / 128)
    p["nerf_cascades"] = 8
    p["opacity"] = opacity
    return p


def _camera(model=0, pos=(0.5, -1.6, 0.9), focal=60.0, aperture=0.0, focus_z=1.0, near=0.0, sq=(0, 0, 0), qh=None):
    c = np.zeros(1, capi.RENDER_CAMERA)
    c["transform"][0] = H.look_at_xform(pos)
    c["model"], c["focal_length"] = model, focal
    c["sq_width"], c["sq_height"], c["sq_curvature"] = sq
    if qh is not None:
        c["qh_front"][0], c["qh_back"][0] = qh
    c["near_distance"], c["aperture_size"], c["focus_z"] = near, aperture, focus_z
    return c


CAMERAS = {
    "perspective": dict(model=0),
    "perspective_dof": dict(model=0, aperture=0.05, focus_z=1.7, near=0.1),
    "spherical_quad_curved": dict(model=2, sq=(0.4, 0.3, 0.35), near=0.05),
    "spherical_quad_flat": dict(model=2, sq=(0.6, 0.4, 0.0)),
    "quad_hexahedron": dict(model=1, qh=(np.array([-.3, -.2, 1, .3, -.2, 1, -.3, .2, 1, .3, .2, 1], np.float32),
                                         np.array([-.1, -.1, 0, .1, -.1, 0, -.1, .1, 0, .1, .1, 0], np.float32)), aperture=0.02, focus_z=1.2),
}


@pytest.mark.parametrize("name", sorted(CAMERAS))
@pytest.mark.parametrize("mip", [0, 1])
def test_init_global_rays(ngp, oracle, cuda, name, mip):
    w, h = 50, 37
    ds = _ds(oracle, w, h, mip)
    n = int(ds["scaled_pixels"][0])
    cam = _camera(**CAMERAS[name])
    ref = np.zeros(n, capi.GLOBAL_RAY); ref_depth = np.zeros(n, np.float32)
    oracle.orc_multi_init_global_rays(0, ref.ctypes.data, ref_depth.ctypes.data, ds.ctypes.data, cam.ctypes.data)
    d_rays, d_depth = H.dev_zeros(ref.nbytes, cuda), H.dev_zeros(n * 4, cuda)
    check(ngp.ngp_hip_multi_init_global_rays(None, 0, d_rays.data_ptr(), d_depth.data_ptr(), ds.ctypes.data, cam.ctypes.data))
    got = H.to_host(d_rays, capi.GLOBAL_RAY)
    np.testing.assert_array_equal(got["idx"], ref["idx"])
    np.testing.assert_array_equal(got["alive"], ref["alive"])
    assert ref["alive"].all() and (ref["idx"] == np.arange(n)).all()
    # sinf / cosf / atan2f differ in the last ulps between libm and the device library
    tol = 0 if name == "perspective" else 3e-6
    np.testing.assert_allclose(got["origin"], ref["origin"], rtol=0, atol=tol)
    np.testing.assert_allclose(got["dir"], ref["dir"], rtol=0, atol=tol)
    np.testing.assert_array_equal(H.to_host(d_depth, np.float32), ref_depth)
    assert np.allclose(np.linalg.norm(ref["dir"], axis=1), 1.0, atol=1e-6)


def _scene_masks():
    box = _mask(0, 0, _trs((0.5, 0.5, 0.5), 0.4), [0.7, 0.5, 0.9], 0.1, 0.8)
    cyl = _mask(1, 1, _trs((0.45, 0.55, 0.5), 0.0), [0.12, 0.5], 0.05, 1.0)
    sph = _mask(2, 0, _trs((0.6, 0.4, 0.6), 0.0, 1.2), [0.2], 0.0, 0.6)
    return [box, cyl, sph]


@pytest.mark.parametrize("mask_set", ["none", "box_first", "sphere_only", "cylinder_add"])
def test_init_proxy_rays(ngp, oracle, cuda, mask_set):
    w, h = 64, 48
    ds = _ds(oracle, w, h, 0)
    n = w * h
    cam = _camera(pos=(1.9, -1.2, 1.1), focal=55.0)
    rays = np.zeros(n, capi.GLOBAL_RAY); depth = np.zeros(n, np.float32)
    oracle.orc_multi_init_global_rays(0, rays.ctypes.data, depth.ctypes.data, ds.ctypes.data, cam.ctypes.data)
    rays["alive"][::7] = 0
    box, cyl, sph = _scene_masks()
    masks = {"none": [], "box_first": [box, cyl, sph], "sphere_only": [sph],
             "cylinder_add": [_mask(1, 0, _trs((0.5, 0.5, 0.5), 0.3), [0.15, 0.6], 0.02, 1.0)]}[mask_set]
    m = _with_all_mask(masks)
    xf = _trs((0.3, -0.1, 0.05), 0.5, 1.3)
    d_masks = H.to_dev(m, cuda) if len(m) else None
    p_ref = _props(xf, 0, m.ctypes.data if len(m) else 0, len(m))
    p_dev = _props(xf, 0, d_masks.data_ptr() if len(m) else 0, len(m))
    ref = np.zeros(n, capi.PROXY_RAY)
    oracle.orc_multi_init_proxy_rays(n, rays.ctypes.data, ref.ctypes.data, p_ref.ctypes.data)
    d_rays, d_props, d_proxy = H.to_dev(rays, cuda), H.to_dev(p_dev, cuda), H.dev_zeros(ref.nbytes, cuda)
    check(ngp.ngp_hip_multi_init_proxy_rays(None, n, d_rays.data_ptr(), d_proxy.data_ptr(), d_props.data_ptr()))
    got = H.to_host(d_proxy, capi.PROXY_RAY)
    for f in ("alive", "active", "idx", "n_steps", "t", "origin", "dir"):
        np.testing.assert_array_equal(got[f], ref[f], err_msg=f)
    assert 0 < ref["alive"].sum() < n
    # note: with an Add mask first, the prepended All mask is a Subtract mask, and Subtract masks intersect every ray (mask_3D.cuh:215-217),
    # so the per-ray mask rejection only ever triggers for hand-built mask lists; the equality above covers whatever the lists produce


def _two_nerf_setup(oracle, cuda, n_rays_w=48, n_rays_h=36, with_masks=True):
    grid = H.blob_density_grid(1)
    bf, mean = H.oracle_bitfield(oracle, grid, 1)
    grid2 = H.blob_density_grid(1, seed=11)
    bf2, _ = H.oracle_bitfield(oracle, grid2, 1)
    d_bf, d_bf2 = H.to_dev(bf, cuda), H.to_dev(bf2, cuda)
    # Add box with a feathered edge on the first NeRF; on the second a Subtract cylinder, which in the reference's convention removes what
    # lies OUTSIDE the shape (the signed distance is negated for Subtract masks, mask_3D.cuh:180), plus a faint Add sphere
    keep_cyl = _mask(1, 1, _trs((0.5, 0.5, 0.5), 0.2), [0.33, 0.9], 0.06, 1.0)
    masks0 = _with_all_mask([_scene_masks()[0]]) if with_masks else np.zeros(0, capi.MASK3D)
    masks1 = _with_all_mask([keep_cyl, _scene_masks()[2]]) if with_masks else np.zeros(0, capi.MASK3D)
    d_m0 = H.to_dev(masks0, cuda) if len(masks0) else None
    d_m1 = H.to_dev(masks1, cuda) if len(masks1) else None
    xf0, xf1 = _trs((0.0, 0.0, 0.0), 0.0, 1.0), _trs((0.55, 0.2, 0.1), 0.7, 0.8)
    ref_props = np.concatenate([_props(xf0, bf.ctypes.data, masks0.ctypes.data if len(masks0) else 0, len(masks0), opacity=1.0),
                                _props(xf1, bf2.ctypes.data, masks1.ctypes.data if len(masks1) else 0, len(masks1), opacity=0.7)])
    dev_props = np.concatenate([_props(xf0, d_bf.data_ptr(), d_m0.data_ptr() if len(masks0) else 0, len(masks0), opacity=1.0),
                                _props(xf1, d_bf2.data_ptr(), d_m1.data_ptr() if len(masks1) else 0, len(masks1), opacity=0.7)])
    keep = (bf, bf2, d_bf, d_bf2, masks0, masks1, d_m0, d_m1)
    ds = _ds(oracle, n_rays_w, n_rays_h, 0)
    cam = _camera(pos=(1.7, -1.3, 1.0), focal=42.0)
    return ds, cam, ref_props, dev_props, keep


def test_march_cull_generate_composite_compact(ngp, oracle, cuda):
    """one full pass of march_rays_and_accumulate_colors (nerf_renderer.cu:652-785) with synthetic network outputs, stage by stage"""
    ds, cam, rp, dp, keep = _two_nerf_setup(oracle, cuda)
    n = int(ds["scaled_pixels"][0]); stride = (n + 127) // 128 * 128; n_nerfs = 2
    g = np.zeros(stride, capi.GLOBAL_RAY); depth = np.zeros(n, np.float32)
    oracle.orc_multi_init_global_rays(0, g.ctypes.data, depth.ctypes.data, ds.ctypes.data, cam.ctypes.data)
    px = np.zeros(stride * n_nerfs, capi.PROXY_RAY)
    for k in range(n_nerfs):
        oracle.orc_multi_init_proxy_rays(n, g.ctypes.data, px[k * stride:].ctypes.data, rp[k:k + 1].ctypes.data)
    d_g, d_px, d_props = H.to_dev(g, cuda), H.to_dev(px, cuda), H.to_dev(dp, cuda)
    cam_pos = np.ascontiguousarray(cam["transform"][0][9:12])

    # march + cull
    oracle.orc_multi_march_active_rays(n, n_nerfs, g.ctypes.data, px.ctypes.data, stride, rp.ctypes.data)
    check(ngp.ngp_hip_multi_march_active_rays(None, n, n_nerfs, d_g.data_ptr(), d_px.data_ptr(), stride, d_props.data_ptr()))
    oracle.orc_multi_cull_rays(n, n_nerfs, g.ctypes.data, px.ctypes.data, stride, cam_pos.ctypes.data, rp.ctypes.data)
    check(ngp.ngp_hip_multi_cull_rays(None, n, n_nerfs, d_g.data_ptr(), d_px.data_ptr(), stride, cam_pos.ctypes.data, d_props.data_ptr()))
    got_px, got_g = H.to_host(d_px, capi.PROXY_RAY), H.to_host(d_g, capi.GLOBAL_RAY)
    for f in ("alive", "active", "t"):
        np.testing.assert_array_equal(got_px[f], px[f], err_msg=f)
    np.testing.assert_array_equal(got_g["alive"], g["alive"])
    both = (px["alive"][:n] == 1) & (px["alive"][stride:stride + n] == 1)
    assert both.sum() > 20 and (px["active"][:n][both] + px["active"][stride:stride + n][both] == 1).all()   # exactly one proxy wins
    assert (px["active"][:n][both] == 1).any() and (px["active"][stride:stride + n][both] == 1).any()

    rs = np.random.RandomState(5)
    n_steps = 4
    for k in range(n_nerfs):
        # next inputs
        net_in = np.zeros(n * n_steps, capi.COORD); d_in = H.dev_zeros(net_in.nbytes, cuda)
        oracle.orc_multi_generate_next_inputs(n, g.ctypes.data, px[k * stride:].ctypes.data, net_in.ctypes.data, n_steps, rp[k:k + 1].ctypes.data)
        check(ngp.ngp_hip_multi_generate_next_inputs(None, n, d_g.data_ptr(), d_px.data_ptr() + k * stride * 40, d_in.data_ptr(), n_steps, d_props.data_ptr() + k * 224))
        got_px = H.to_host(d_px, capi.PROXY_RAY)
        sl = slice(k * stride, k * stride + n)
        for f in ("n_steps", "t", "alive", "active"):
            np.testing.assert_array_equal(got_px[f][sl], px[f][sl], err_msg="%s nerf %d" % (f, k))
        got_in = H.to_host(d_in, capi.COORD)
        used = np.zeros(n * n_steps, bool)
        act = (g["alive"][:n] == 1) & (px["active"][sl] == 1)
        for j in range(n_steps):
            used[j * n:(j + 1) * n] = act      # every step slot of an active ray is written before the march can fail
        for f in ("pos", "dt", "dir"):
            np.testing.assert_array_equal(got_in[f][used & (np.arange(n * n_steps) < n) ], net_in[f][used & (np.arange(n * n_steps) < n)], err_msg=f)
        wrote = act & (px["n_steps"][sl] > 0)
        assert wrote.sum() > 10
        # composite with synthetic raw outputs (fp16): sigma large enough that some rays saturate
        out = np.zeros((n * n_steps, 4), np.float16)
        out[:, :3] = rs.uniform(-2, 2, (n * n_steps, 3)); out[:, 3] = np.where(rs.rand(n * n_steps) < 0.25, 9.5, rs.uniform(-1, 5, n * n_steps))
        d_out = H.to_dev(out, cuda)
        oracle.orc_multi_composite(n, 1, g.ctypes.data, px[k * stride:].ctypes.data, net_in.ctypes.data, out.ctypes.data, 4, n_steps, 2, 3, ctypes.c_float(0.01), rp[k:k + 1].ctypes.data)
        d_in_ref = H.to_dev(net_in, cuda)   # identical inputs on both sides
        check(ngp.ngp_hip_multi_composite(None, n, 1, d_g.data_ptr(), d_px.data_ptr() + k * stride * 40, d_in_ref.data_ptr(), d_out.data_ptr(), 4, n_steps, 2, 3, 0.01, d_props.data_ptr() + k * 224))
        got_px, got_g = H.to_host(d_px, capi.PROXY_RAY), H.to_host(d_g, capi.GLOBAL_RAY)
        np.testing.assert_array_equal(got_px["alive"][sl], px["alive"][sl])
        np.testing.assert_array_equal(got_px["n_steps"][sl], px["n_steps"][sl])
        np.testing.assert_allclose(got_g["rgba"], g["rgba"], rtol=2e-5, atol=2e-6)   # __expf / logistic vs libm
        assert (g["rgba"][:, 3] > 0).sum() > 10
    assert (g["rgba"][:n, 3] >= 0.99).any() and ((g["rgba"][:n, 3] > 0) & (g["rgba"][:n, 3] < 0.9)).any()

    # compaction: same sets, order free (the reference's atomicAdd order is unspecified too)
    g["alive"][:n][g["rgba"][:n, 3] >= 0.99] = 0
    d_g = H.to_dev(g, cuda); d_px = H.to_dev(px, cuda)
    g2, px2, fin = np.zeros_like(g), np.zeros_like(px), np.zeros_like(g)
    ca, cf = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    oracle.orc_multi_compact_rays(n, g.ctypes.data, g2.ctypes.data, px.ctypes.data, px2.ctypes.data, n_nerfs, stride, fin.ctypes.data, ca.ctypes.data, cf.ctypes.data)
    d_g2, d_px2, d_fin, d_cnt = H.dev_zeros(g.nbytes, cuda), H.dev_zeros(px.nbytes, cuda), H.dev_zeros(g.nbytes, cuda), H.dev_zeros(8, cuda)
    check(ngp.ngp_hip_multi_compact_rays(None, n, d_g.data_ptr(), d_g2.data_ptr(), d_px.data_ptr(), d_px2.data_ptr(), n_nerfs, stride, d_fin.data_ptr(), d_cnt.data_ptr(), d_cnt.data_ptr() + 4))
    cnt = H.to_host(d_cnt, np.uint32)
    assert cnt[0] == ca[0] and cnt[1] == cf[0] and ca[0] > 0 and cf[0] > 0
    gg2, gpx2, gfin = H.to_host(d_g2, capi.GLOBAL_RAY), H.to_host(d_px2, capi.PROXY_RAY), H.to_host(d_fin, capi.GLOBAL_RAY)
    o_ref, o_got = np.argsort(g2["idx"][:ca[0]]), np.argsort(gg2["idx"][:ca[0]])
    np.testing.assert_array_equal(gg2[:ca[0]][o_got].tobytes(), g2[:ca[0]][o_ref].tobytes())
    for k in range(n_nerfs):
        a, b = gpx2[k * stride:k * stride + ca[0]][o_got], px2[k * stride:k * stride + ca[0]][o_ref]
        np.testing.assert_array_equal(a.tobytes(), b.tobytes())
    np.testing.assert_array_equal(np.sort(gfin["idx"][:cf[0]]), np.sort(fin["idx"][:cf[0]]))


@pytest.mark.parametrize("mip,flip_y", [(0, 0), (0, 1), (2, 0), (1, 1)])
def test_shade_downsampled(ngp, oracle, cuda, mip, flip_y):
    # resolution divisible by the skip: otherwise the blocks of the last column wrap into the next row (only idx >= max_pixels is
    # guarded, nerf_renderer.cu:548-553) and two rays blend into one pixel in thread order — unspecified in the reference as well
    w, h = 40, 32
    ds = _ds(oracle, w, h, mip)
    n = int(ds["scaled_pixels"][0])
    rs = np.random.RandomState(mip + 7)
    rays = np.zeros(n, capi.GLOBAL_RAY)
    rays["idx"] = rs.permutation(n)
    rays["rgba"] = rs.rand(n, 4); rays["depth"] = rs.rand(n)
    fb = rs.rand(w * h, 4).astype(np.float32); db = np.full(w * h, 1e10, np.float32)
    d_fb, d_db, d_r = H.to_dev(fb, cuda), H.to_dev(db, cuda), H.to_dev(rays, cuda)
    oracle.orc_multi_shade(n, rays.ctypes.data, 0, fb.ctypes.data, db.ctypes.data, ds.ctypes.data, flip_y)
    check(ngp.ngp_hip_multi_shade(None, n, d_r.data_ptr(), 0, d_fb.data_ptr(), d_db.data_ptr(), ds.ctypes.data, flip_y))
    np.testing.assert_allclose(H.to_host(d_fb, np.float32).reshape(-1, 4), fb, rtol=2e-6, atol=1e-7)   # powf in srgb_to_linear
    np.testing.assert_array_equal(H.to_host(d_db, np.float32), db)


def _gpu_frame(ngp, cuda, descs, params, dev_props, ds, cam, acts, flip_y):
    """NerfRenderer::render driven from Python through the C ABI (the C++ host does the same, see blender-ngp_amd/host/nerf_renderer.cpp)"""
    import torch
    n_nerfs = len(descs)
    n = int(ds["scaled_pixels"][0]); stride = (n + 127) // 128 * 128
    w, h = int(ds["max_res"][0][0]), int(ds["max_res"][0][1])
    d_g = [H.dev_zeros(stride * 52, cuda) for _ in range(2)]
    d_px = [H.dev_zeros(stride * n_nerfs * 40, cuda) for _ in range(2)]
    d_hit, d_cnt = H.dev_zeros(stride * 52, cuda), H.dev_zeros(8, cuda)
    d_in, d_out = H.dev_zeros(stride * 8 * 28, cuda), H.dev_zeros(stride * 8 * 8, cuda)
    d_fb, d_db = H.dev_zeros(w * h * 16, cuda), H.dev_zeros(w * h * 4, cuda)
    d_props = H.to_dev(dev_props, cuda)
    check(ngp.ngp_hip_multi_init_global_rays(None, 0, d_g[0].data_ptr(), d_db.data_ptr(), ds.ctypes.data, cam.ctypes.data))
    for k in range(n_nerfs):
        check(ngp.ngp_hip_multi_init_proxy_rays(None, n, d_g[0].data_ptr(), d_px[0].data_ptr() + k * stride * 40, d_props.data_ptr() + k * 224))
    cam_pos = np.ascontiguousarray(cam["transform"][0][9:12])
    n_alive, i, dbi, n_samples = n, 1, 0, 0
    while i < 10000:
        tmp, cur = dbi % 2, (dbi + 1) % 2
        dbi += 1
        d_cnt[:4] = 0
        check(ngp.ngp_hip_multi_compact_rays(None, n_alive, d_g[tmp].data_ptr(), d_g[cur].data_ptr(), d_px[tmp].data_ptr(), d_px[cur].data_ptr(), n_nerfs, stride,
                                             d_hit.data_ptr(), d_cnt.data_ptr(), d_cnt.data_ptr() + 4))
        n_alive = int(H.to_host(d_cnt, np.uint32)[0])
        if n_alive == 0:
            break
        check(ngp.ngp_hip_multi_march_active_rays(None, n_alive, n_nerfs, d_g[cur].data_ptr(), d_px[cur].data_ptr(), stride, d_props.data_ptr()))
        check(ngp.ngp_hip_multi_cull_rays(None, n_alive, n_nerfs, d_g[cur].data_ptr(), d_px[cur].data_ptr(), stride, cam_pos.ctypes.data, d_props.data_ptr()))
        n_steps = min(max(n // n_alive, 1), 8)
        n_el = (n_alive * n_steps + 127) // 128 * 128
        for k in range(n_nerfs):
            ppx = d_px[cur].data_ptr() + k * stride * 40
            check(ngp.ngp_hip_multi_generate_next_inputs(None, n_alive, d_g[cur].data_ptr(), ppx, d_in.data_ptr(), n_steps, d_props.data_ptr() + k * 224))
            check(ngp.ngp_hip_nerf_inference(None, descs[k].data_ptr(), params[k].data_ptr(), d_in.data_ptr(), 7, n_el, d_out.data_ptr(), 4))
            n_samples += n_el
            check(ngp.ngp_hip_multi_composite(None, n_alive, i, d_g[cur].data_ptr(), ppx, d_in.data_ptr(), d_out.data_ptr(), 4, n_steps, acts[k][0], acts[k][1], 0.01,
                                              d_props.data_ptr() + k * 224))
        i += n_steps
    n_hit = int(H.to_host(d_cnt, np.uint32)[1])
    check(ngp.ngp_hip_multi_shade(None, n_hit, d_hit.data_ptr(), 0, d_fb.data_ptr(), d_db.data_ptr(), ds.ctypes.data, flip_y))
    torch.cuda.synchronize()
    return H.to_host(d_fb, np.float32).reshape(h, w, 4), n_hit


@pytest.mark.parametrize("with_masks", [False, True])
def test_two_nerf_frame_matches_oracle(ngp, oracle, cuda, with_masks):
    ds, cam, rp, dp, keep = _two_nerf_setup(oracle, cuda, 40, 30, with_masks)
    desc = H.make_desc(ngp, log2_hashmap_size=12)
    nets, host_params, d_descs, d_params = [], [], [], []
    for k in range(2):
        P = H.random_params(desc, seed=20 + k, grid_amp=2.0)
        P16 = P.view(np.float16).copy()
        P16[2048 + 0 * 64: 2048 + 64] *= np.float16(4.0)     # density row of the second layer: opaque enough to terminate rays
        host_params.append(P16.view(np.uint16)); d_params.append(H.to_dev(host_params[-1], cuda)); d_descs.append(H.to_dev(desc, cuda))
    w, h = 40, 30
    fb = np.zeros((h, w, 4), np.float32); db = np.zeros(w * h, np.float32)
    net_ptrs = (ctypes.c_void_p * 2)(desc.ctypes.data, desc.ctypes.data)
    par_ptrs = (ctypes.c_void_p * 2)(host_params[0].ctypes.data, host_params[1].ctypes.data)
    rgb_act, dens_act = np.array([2, 2], np.int32), np.array([3, 3], np.int32)
    min_t = np.array([0.01, 0.01], np.float32)
    oracle.orc_multi_render.restype = ctypes.c_uint64
    n_ref = oracle.orc_multi_render(2, net_ptrs, par_ptrs, rp.ctypes.data, rgb_act.ctypes.data, dens_act.ctypes.data, min_t.ctypes.data, ds.ctypes.data, cam.ctypes.data, 1,
                                    fb.ctypes.data, db.ctypes.data)
    got, n_hit = _gpu_frame(ngp, cuda, d_descs, d_params, dp, ds, cam, [(2, 3), (2, 3)], 1)
    assert n_ref > 1000 and n_hit > 50
    covered = fb[..., 3] > 0.05
    assert covered.mean() > 0.05
    # fp16 network outputs: a sample whose sigma differs in the last fp16 ulp shifts a pixel by ~1e-3; ray termination flips are rarer still
    diff = np.abs(got - fb)
    assert np.mean(diff) < 2e-3 and np.mean(diff.max(axis=-1) > 2e-2) < 0.01
    if with_masks:
        fb0 = np.zeros((h, w, 4), np.float32)
        _, _, rp0, _, keep0 = _two_nerf_setup(oracle, cuda, 40, 30, False)
        oracle.orc_multi_render(2, net_ptrs, par_ptrs, rp0.ctypes.data, rgb_act.ctypes.data, dens_act.ctypes.data, min_t.ctypes.data, ds.ctypes.data, cam.ctypes.data, 1,
                                fb0.ctypes.data, db.ctypes.data)
        assert np.abs(fb0 - fb).max() > 0.05           # the masks change the picture


def _sparse_cascaded_bitfield(oracle, p_occ, n_cascades, seed):
    """4 x 4 x 4 bricks occupied with probability p_occ in each of the first n_cascades grids (Morton order), coarser mips max-pooled like update_density_grid_mean_and_bitfield"""
    vol = 128 ** 3
    rs = np.random.RandomState(seed)
    bf = np.zeros(vol, np.uint8)   # 8 mips x vol / 8 bytes
    x, y, z = np.meshgrid(np.arange(128), np.arange(128), np.arange(128), indexing="ij")

    def part(v):
        v = v.astype(np.uint32); v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    m = (part(x) | (part(y) << 1) | (part(z) << 2)).ravel()
    for c in range(n_cascades):
        cells = rs.rand(32, 32, 32) < p_occ
        full = np.repeat(np.repeat(np.repeat(cells, 4, 0), 4, 1), 4, 2)
        bits = np.zeros(vol, np.uint8); bits[m] = full.ravel()
        bf[c * vol // 8:(c + 1) * vol // 8] = np.packbits(bits, bitorder="little")
    for lvl in range(1, 8):
        oracle.orc_bitfield_max_pool(vol // 64, bf[(lvl - 1) * vol // 8:].ctypes.data, bf[lvl * vol // 8:].ctypes.data)
    return bf


@pytest.mark.parametrize("p_occ", [1.0, 0.3, 0.05])
def test_sampler_with_cone_stepping_and_cascades_matches_oracle_and_the_stock_tracer(ngp, oracle, cuda, p_occ):
    """every real capture has aabb_scale > 1: three or more cascades and cone stepping (dt = clamp(t / 256, ...), src/nerf_renderer.cu:174, 349).  The fork's march and
    sampler on such a field — march_active_rays, then up to 900 steps of march_proxy_rays_and_generate_next_network_inputs — bit for bit against the oracle; and, because the
    two samplers walk the same candidate sequence through the same occupancy, ray by ray the SAME NUMBER of samples as the stock tracer's generate_next_nerf_network_inputs
    (src/testbed_nerf.cu:705-765): the fork's sampler emits before it tests and steps once past every landing point, which moves samples, not their count (+- 2 per gap)."""
    vol = 128 ** 3
    bf = _sparse_cascaded_bitfield(oracle, p_occ, 3, int(p_occ * 100))
    d_bf = H.to_dev(bf, cuda)
    n, n_steps = 256, 900
    rs = np.random.RandomState(7)
    dirs = rs.randn(n, 3).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    origin = np.float32([0.45, 0.5, 0.55])

    def props(ptr):
        p = _props(np.eye(4, dtype=np.float32), ptr, 0, 0, aabb_scale=4)
        return p
    rp, dp = props(bf.ctypes.data), props(d_bf.data_ptr())
    assert float(rp["cone_angle"][0]) == 1.0 / 256.0
    g = np.zeros(n, capi.GLOBAL_RAY); g["origin"] = origin; g["dir"] = dirs; g["alive"] = 1; g["idx"] = np.arange(n)
    px = np.zeros(n, capi.PROXY_RAY); px["origin"] = origin; px["dir"] = dirs; px["t"] = 1e-5; px["alive"] = 1; px["active"] = 1; px["idx"] = np.arange(n)
    d_g, d_px, d_props = H.to_dev(g, cuda), H.to_dev(px, cuda), H.to_dev(dp, cuda)
    oracle.orc_multi_march_active_rays(n, 1, g.ctypes.data, px.ctypes.data, n, rp.ctypes.data)
    check(ngp.ngp_hip_multi_march_active_rays(None, n, 1, d_g.data_ptr(), d_px.data_ptr(), n, d_props.data_ptr()))
    got = H.to_host(d_px, capi.PROXY_RAY)
    for f in ("alive", "t"):
        np.testing.assert_array_equal(got[f], px[f], err_msg=f)
    net_in = np.zeros(n * n_steps, capi.COORD); d_in = H.dev_zeros(net_in.nbytes, cuda)
    oracle.orc_multi_generate_next_inputs(n, g.ctypes.data, px.ctypes.data, net_in.ctypes.data, n_steps, rp.ctypes.data)
    check(ngp.ngp_hip_multi_generate_next_inputs(None, n, d_g.data_ptr(), d_px.data_ptr(), d_in.data_ptr(), n_steps, d_props.data_ptr()))
    got = H.to_host(d_px, capi.PROXY_RAY)
    for f in ("n_steps", "t", "alive"):
        np.testing.assert_array_equal(got[f], px[f], err_msg=f)
    got_in = H.to_host(d_in, capi.COORD)
    steps = px["n_steps"].astype(np.int64)
    for j in (0, 1, 5, 50, 300):
        sel = np.nonzero((px["alive"] == 1) & (steps > j))[0]
        for f in ("pos", "dt", "dir"):
            np.testing.assert_array_equal(got_in[f][j * n + sel], net_in[f][j * n + sel], err_msg="%s step %d" % (f, j))
    assert (steps > 5).sum() > n // 4 and len(np.unique(net_in["dt"][:n][steps > 0])) >= 1
    # the stock tracer over the same rays and the same grids
    aabb = H.unit_aabb(4)
    pay = np.zeros(n, capi.PAYLOAD); pay["origin"] = origin; pay["dir"] = dirs; pay["t"] = 1e-5; pay["alive"] = 1; pay["idx"] = np.arange(n)
    stock_in = np.zeros(n * n_steps, capi.COORD)
    oracle.orc_generate_next_inputs(n, aabb.ctypes.data, aabb.ctypes.data, pay.ctypes.data, stock_in.ctypes.data, n_steps, bf.ctypes.data, 0, ctypes.c_float(1.0 / 256.0))
    stock_steps = pay["n_steps"].astype(np.int64)
    fin = (steps < n_steps) & (stock_steps < n_steps)          # rays that left the box on both sides
    assert fin.sum() > n // 4
    assert np.abs(steps[fin] - stock_steps[fin]).max() <= 2 + 0.05 * stock_steps[fin].max(), (steps[fin][:16], stock_steps[fin][:16])
    assert abs(int(steps[fin].sum()) - int(stock_steps[fin].sum())) <= 0.03 * stock_steps[fin].sum() + 2 * fin.sum()
