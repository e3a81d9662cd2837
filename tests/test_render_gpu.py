"""GPU parity: single-NeRF renderer kernels (render.hip) vs the CPU oracle, through the C ABI.

Ray set-up and marching (init_rays, advance_pos, generate_next_inputs) are exact; compositing / shading / tonemapping go through
exp/pow and are compared with rtol 2e-3 (v_exp_f32 / v_log_f32 based device functions vs libm).
"""
import numpy as np
import pytest

import capi
import helpers as H
from capi import check

pytestmark = pytest.mark.gpu
W, Hh = 80, 48


def _camera():
    cam = H.look_at_xform([1.6, 1.3, 1.1])
    focal = np.array([70.0, 70.0], np.float32)
    res = np.array([W, Hh], np.int32)
    sc = np.array([0.5, 0.5], np.float32)
    return cam, focal, res, sc


def _init_and_advance(ngp, oracle, cuda, spp, snap, n_cascades=1, cone=0.0, plane_z=1.0, aperture=0.0, exact=True, cam_models=None):
    cam, focal, res, sc = _camera()
    aabb = H.unit_aabb(2 ** (n_cascades - 1))
    grid = H.blob_density_grid(n_cascades)
    bf, _ = H.oracle_bitfield(oracle, grid, n_cascades)
    n = W * Hh
    ident = np.eye(3, dtype=np.float32).reshape(-1)
    zero4, zero3 = np.zeros(4, np.float32), np.zeros(3, np.float32)
    pay = np.zeros(n, H.PAYLOAD)
    depth = np.zeros(n, np.float32)
    oracle.orc_init_rays(spp, pay.ctypes.data, res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, zero4.ctypes.data, sc.ctypes.data, zero3.ctypes.data,
                         snap, aabb.ctypes.data, ident.ctypes.data, H.f32(0.0), 0, None, depth.ctypes.data, H.f32(plane_z), H.f32(aperture), cam_models.ctypes.data if cam_models is not None else None)
    d_pay, d_depth = H.dev_zeros(n * 40, cuda), H.dev_zeros(n * 4, cuda)
    check(ngp.ngp_hip_init_rays(None, spp, d_pay.data_ptr(), res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, zero4.ctypes.data, sc.ctypes.data,
                                zero3.ctypes.data, snap, aabb.ctypes.data, ident.ctypes.data, H.f32(0.0), 0, None, d_depth.data_ptr(), H.f32(plane_z), H.f32(aperture), cam_models.ctypes.data if cam_models is not None else None, None))
    g = H.to_host(d_pay, H.PAYLOAD).copy()
    if not exact:   # depth of field: the lens-disk sample goes through cosf / sinf (device intrinsics vs libm)
        same = g["alive"] == pay["alive"]
        assert same.mean() > 0.995
        al = (pay["alive"] == 1) & same
        np.testing.assert_allclose(g["origin"][al], pay["origin"][al], rtol=0, atol=2e-6)
        np.testing.assert_allclose(g["dir"][al], pay["dir"][al], rtol=0, atol=2e-6)
        np.testing.assert_allclose(g["t"][al], pay["t"][al], rtol=0, atol=1e-5)
        return dict(pay=pay, g=g, al=al, cam=cam)
    np.testing.assert_array_equal(g["alive"], pay["alive"])
    np.testing.assert_array_equal(g["origin"], pay["origin"])
    al = pay["alive"] == 1
    assert plane_z < 0 or (al.any() and (~al).any() or al.all())
    for f in ("dir", "t", "idx", "n_steps"):
        np.testing.assert_array_equal(g[f][al], pay[f][al])
    np.testing.assert_array_equal(H.to_host(d_depth, np.float32), depth)
    if plane_z < 0:
        np.testing.assert_array_equal(g["dir"], pay["dir"])
        return dict(pay=pay, g=g, depth=depth)

    oracle.orc_advance_pos(n, aabb.ctypes.data, ident.ctypes.data, spp, pay.ctypes.data, bf.ctypes.data, 0, H.f32(cone))
    d_bf = H.to_dev(bf, cuda)
    d_pay_s = d_pay.clone()
    check(ngp.ngp_hip_advance_pos(None, n, aabb.ctypes.data, ident.ctypes.data, spp, d_pay.data_ptr(), d_bf.data_ptr(), 0, H.f32(cone)))
    g = H.to_host(d_pay, H.PAYLOAD).copy()
    np.testing.assert_array_equal(g["alive"], pay["alive"])
    al = pay["alive"] == 1
    np.testing.assert_array_equal(g["t"][al], pay["t"][al])
    # ... and with the cascade-0 brick summary staged in LDS: the same march, bit for bit
    d_sum = H.dev_zeros(1024 * 4, cuda)
    check(ngp.ngp_hip_bitfield_brick_summary(None, d_bf.data_ptr(), d_sum.data_ptr()))
    check(ngp.ngp_hip_advance_pos(None, n, aabb.ctypes.data, ident.ctypes.data, spp, d_pay_s.data_ptr(), d_bf.data_ptr(), 0, H.f32(cone), None, d_sum.data_ptr()))
    gs = H.to_host(d_pay_s, H.PAYLOAD)
    np.testing.assert_array_equal(gs["alive"], g["alive"])
    np.testing.assert_array_equal(gs["t"][al], g["t"][al])
    return dict(pay=pay, d_pay=d_pay, bf=bf, d_bf=d_bf, aabb=aabb, cam=cam, focal=focal, res=res, sc=sc, n=n)


@pytest.mark.parametrize("spp,snap,n_cascades,cone", [(0, 1, 1, 0.0), (3, 0, 1, 0.0), (1, 0, 3, 1.0 / 256.0)])
def test_init_rays_and_advance_bit_exact(ngp, oracle, cuda, spp, snap, n_cascades, cone):
    S = _init_and_advance(ngp, oracle, cuda, spp, snap, n_cascades, cone)
    assert (S["pay"]["alive"] == 1).sum() > 50


def test_init_rays_depth_of_field(ngp, oracle, cuda):
    """pixel_to_ray with aperture_size > 0 (common_device.cuh:307-312): origins jittered on the lens disk, the point at focus_z fixed"""
    S = _init_and_advance(ngp, oracle, cuda, 5, 0, plane_z=1.7, aperture=0.03, exact=False)
    P0 = _init_and_advance(ngp, oracle, cuda, 5, 0)["pay"]          # pinhole rays of the same pixels
    al = S["al"] & (P0["alive"] == 1)
    assert al.sum() > 100
    o, d = S["pay"]["origin"][al], S["pay"]["dir"][al]
    o0, d0 = P0["origin"][al], P0["dir"][al]
    off = np.linalg.norm(o - o0, axis=1)
    assert off.max() > 0.01 and off.max() <= 0.03 + 1e-6            # inside the lens disk, and actually spread over it
    # both rays pass through the same point of the focal plane: origin0 + dir0_unnormalised * focus_z
    cam = S["cam"].reshape(4, 3)                                    # column-major 3x4
    fwd = cam[2]
    tz0 = 1.7 / (d0 @ fwd)
    tz = ((o0 + d0 * tz0[:, None] - o) @ fwd) / (d @ fwd)
    np.testing.assert_allclose(o + d * tz[:, None], o0 + d0 * tz0[:, None], atol=2e-5)


def _camera_models(model, W_, H_):
    c = np.zeros(1, dtype=capi.RENDER_CAMERA)
    c["model"] = model
    c["sq_width"], c["sq_height"], c["sq_curvature"] = 0.6, 0.45, 0.35
    # a frustum-like hexahedron in camera space: small back face at z = 0, larger front face at z = 1 (tl, tr, bl, br)
    c["qh_back"][0] = np.array([[-0.1, -0.08, 0.0], [0.1, -0.08, 0.0], [-0.1, 0.08, 0.0], [0.1, 0.08, 0.0]], np.float32).reshape(-1)
    c["qh_front"][0] = np.array([[-0.7, -0.5, 1.0], [0.7, -0.5, 1.0], [-0.7, 0.5, 1.0], [0.7, 0.5, 1.0]], np.float32).reshape(-1)
    return c


@pytest.mark.parametrize("model", [1, 2])
def test_init_rays_extra_camera_models(ngp, oracle, cuda, model):
    """ECameraModel::QuadrilateralHexahedron (1) / SphericalQuadrilateral (2) in the stock tracer (testbed_nerf.cu:1868-1908, camera_models.cuh):
    the same device functions the Blender renderer uses, here against the oracle; sinf / cosf / atan2f differ from libm in the last bits"""
    S = _init_and_advance(ngp, oracle, cuda, 1, 0, exact=False, cam_models=_camera_models(model, W, Hh))
    assert S["al"].sum() > 100
    P0 = _init_and_advance(ngp, oracle, cuda, 1, 0)["pay"]
    both = S["al"] & (P0["alive"] == 1)
    assert both.sum() > 20 and np.abs(S["pay"]["dir"][both] - P0["dir"][both]).max() > 1e-2       # not the pinhole rays
    if model == 2:   # rays start on the curved sensor surface, not at the camera centre
        cam = S["cam"].reshape(4, 3)
        assert np.linalg.norm(S["pay"]["origin"][S["al"]] - cam[3], axis=1).max() > 0.1


def test_init_rays_slice_plane(ngp, oracle, cuda):
    """plane_z < 0 (the Slice render mode, testbed_nerf.cu:1913-1923): no ray is alive, depth = -plane_z, t at the plane; DOF is off"""
    S = _init_and_advance(ngp, oracle, cuda, 2, 0, plane_z=-0.8, aperture=0.05)
    assert (S["g"]["alive"] == 0).all() and (S["depth"] == np.float32(0.8)).all()
    np.testing.assert_array_equal(S["g"]["t"], S["pay"]["t"])
    assert (S["g"]["t"] >= 0.8 - 1e-6).all()


def test_compact_next_inputs_composite(ngp, oracle, cuda):
    S = _init_and_advance(ngp, oracle, cuda, 0, 1)
    n = S["n"]
    # ---- compaction: same SET of alive rays (order is unordered on the GPU as in the reference)
    rgba0 = np.zeros((n, 4), np.float32)
    dep0 = np.zeros(n, np.float32)
    o_pay, o_rgba, o_dep = np.zeros(n, H.PAYLOAD), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    f_pay, f_rgba, f_dep = np.zeros(n, H.PAYLOAD), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    cnt, fcnt = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    oracle.orc_compact_rays(n, rgba0.ctypes.data, dep0.ctypes.data, S["pay"].ctypes.data, o_rgba.ctypes.data, o_dep.ctypes.data, o_pay.ctypes.data,
                            f_rgba.ctypes.data, f_dep.ctypes.data, f_pay.ctypes.data, cnt.ctypes.data, fcnt.ctypes.data)
    d = [H.dev_zeros(n * 16, cuda), H.dev_zeros(n * 4, cuda), H.dev_zeros(n * 16, cuda), H.dev_zeros(n * 4, cuda), H.dev_zeros(n * 40, cuda),
         H.dev_zeros(n * 16, cuda), H.dev_zeros(n * 4, cuda), H.dev_zeros(n * 40, cuda), H.dev_zeros(4, cuda), H.dev_zeros(4, cuda)]
    check(ngp.ngp_hip_compact_rays(None, n, d[0].data_ptr(), d[1].data_ptr(), S["d_pay"].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                   d[5].data_ptr(), d[6].data_ptr(), d[7].data_ptr(), d[8].data_ptr(), d[9].data_ptr()))
    n_alive = int(cnt[0])
    assert int(H.to_host(d[8], np.uint32)[0]) == n_alive and int(H.to_host(d[9], np.uint32)[0]) == int(fcnt[0]) == 0
    g_pay = H.to_host(d[4], H.PAYLOAD)[:n_alive]
    assert sorted(g_pay["idx"].tolist()) == sorted(o_pay["idx"][:n_alive].tolist())

    # ---- next inputs on the ORACLE-ordered compacted rays (upload them so both sides see one order)
    n_steps = 4
    d_pay2 = H.to_dev(o_pay[:n_alive], cuda)
    pay_before_inputs = o_pay[:n_alive].copy()
    o_in = np.zeros(n_alive * n_steps, H.COORD)
    oracle.orc_generate_next_inputs(n_alive, S["aabb"].ctypes.data, S["aabb"].ctypes.data, o_pay.ctypes.data, o_in.ctypes.data, n_steps, S["bf"].ctypes.data, 0, H.f32(0.0))
    d_in = H.dev_zeros(n_alive * n_steps * 28, cuda)
    check(ngp.ngp_hip_generate_next_inputs(None, n_alive, S["aabb"].ctypes.data, S["aabb"].ctypes.data, d_pay2.data_ptr(), d_in.data_ptr(), n_steps, S["d_bf"].data_ptr(), 0, H.f32(0.0), 0))
    g_pay2 = H.to_host(d_pay2, H.PAYLOAD)
    np.testing.assert_array_equal(g_pay2["n_steps"], o_pay["n_steps"][:n_alive])
    np.testing.assert_array_equal(g_pay2["t"], o_pay["t"][:n_alive])
    g_in = H.to_host(d_in, H.COORD)
    for j in range(n_steps):
        m = o_pay["n_steps"][:n_alive] > j
        assert g_in[j * n_alive:(j + 1) * n_alive][m].tobytes() == o_in[j * n_alive:(j + 1) * n_alive][m].tobytes()
    # ... with the brick summary in LDS and the caller's counter zeroed by the kernel: same samples
    d_sum = H.dev_zeros(1024 * 4, cuda)
    check(ngp.ngp_hip_bitfield_brick_summary(None, S["d_bf"].data_ptr(), d_sum.data_ptr()))
    d_pay3 = H.to_dev(pay_before_inputs, cuda)
    d_in3 = H.dev_zeros(n_alive * n_steps * 28, cuda)
    d_word = H.to_dev(np.array([77, 78], np.uint32), cuda)
    check(ngp.ngp_hip_generate_next_inputs(None, n_alive, S["aabb"].ctypes.data, S["aabb"].ctypes.data, d_pay3.data_ptr(), d_in3.data_ptr(), n_steps, S["d_bf"].data_ptr(), 0, H.f32(0.0), 0,
                                           d_sum.data_ptr(), d_word.data_ptr()))
    assert H.to_host(d_pay3, H.PAYLOAD).tobytes() == g_pay2.tobytes()
    g_in3 = H.to_host(d_in3, H.COORD)
    for j in range(n_steps):
        m = o_pay["n_steps"][:n_alive] > j
        assert g_in3[j * n_alive:(j + 1) * n_alive][m].tobytes() == g_in[j * n_alive:(j + 1) * n_alive][m].tobytes()
    assert H.to_host(d_word, np.uint32).tolist() == [0, 78]

    # ---- composite with random network outputs
    pay_before = o_pay[:n_alive].copy()             # state after next-inputs, before composite
    rs = np.random.RandomState(0)
    out = np.zeros((n_alive * n_steps, 4), np.float16)
    out[:, :3] = rs.randn(n_alive * n_steps, 3).astype(np.float16)
    out[:, 3] = (rs.randn(n_alive * n_steps) * 2 + 4).astype(np.float16)
    o_rgba2, o_dep2 = np.zeros((n_alive, 4), np.float32), np.zeros(n_alive, np.float32)
    oracle.orc_composite(n_alive, 1, S["aabb"].ctypes.data, S["cam"].ctypes.data, o_rgba2.ctypes.data, o_dep2.ctypes.data, o_pay.ctypes.data, o_in.ctypes.data,
                         out.ctypes.data, 4, n_steps, 2, 3, H.f32(0.01))
    d_rgba2, d_dep2 = H.dev_zeros(n_alive * 16, cuda), H.dev_zeros(n_alive * 4, cuda)
    d_out = H.to_dev(out, cuda)
    check(ngp.ngp_hip_composite(None, n_alive, 1, S["aabb"].ctypes.data, S["cam"].ctypes.data, d_rgba2.data_ptr(), d_dep2.data_ptr(), d_pay2.data_ptr(), d_in.data_ptr(),
                                d_out.data_ptr(), 4, n_steps, 2, 3, H.f32(0.01), 1, H.f32(1.0), -1, None))
    g_pay3 = H.to_host(d_pay2, H.PAYLOAD)
    same = g_pay3["alive"] == o_pay["alive"][:n_alive]
    assert same.mean() > 0.99  # termination is a float threshold; allow a hair of disagreement
    np.testing.assert_allclose(H.to_host(d_rgba2, np.float32).reshape(n_alive, 4)[same], o_rgba2[same], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(H.to_host(d_dep2, np.float32)[same], o_dep2[same], rtol=2e-3, atol=1e-5)
    assert (o_pay["alive"][:n_alive] == 0).any() and (o_pay["alive"][:n_alive] == 1).any()
    # ---- the visualisation modes that need nothing but the sample (testbed_nerf.cu:938-968): AO 0, Positions 3 (+ show_accel), Depth 4
    for mode, accel in ((0, -1), (3, -1), (3, 0), (3, 2), (4, -1)):
        o_p, o_c, o_d = pay_before.copy(), np.zeros((n_alive, 4), np.float32), np.zeros(n_alive, np.float32)
        oracle.orc_composite_mode(n_alive, 1, S["aabb"].ctypes.data, S["cam"].ctypes.data, o_c.ctypes.data, o_d.ctypes.data, o_p.ctypes.data, o_in.ctypes.data,
                                  out.ctypes.data, 4, n_steps, 2, 3, H.f32(0.01), mode, H.f32(3.0), accel)
        d_p, d_c, d_d = H.to_dev(pay_before, cuda), H.dev_zeros(n_alive * 16, cuda), H.dev_zeros(n_alive * 4, cuda)
        check(ngp.ngp_hip_composite(None, n_alive, 1, S["aabb"].ctypes.data, S["cam"].ctypes.data, d_c.data_ptr(), d_d.data_ptr(), d_p.data_ptr(), d_in.data_ptr(),
                                    d_out.data_ptr(), 4, n_steps, 2, 3, H.f32(0.01), mode, H.f32(3.0), accel, None))
        ok = H.to_host(d_p, H.PAYLOAD)["alive"] == o_p["alive"]
        assert ok.mean() > 0.99
        np.testing.assert_allclose(H.to_host(d_c, np.float32).reshape(n_alive, 4)[ok], o_c[ok], rtol=2e-3, atol=1e-5)
        assert np.abs(o_c[:, :3] - o_rgba2[:, :3]).max() > 1e-2          # not the Shade colours
    assert ngp.ngp_hip_composite(None, n_alive, 1, S["aabb"].ctypes.data, S["cam"].ctypes.data, d_c.data_ptr(), d_d.data_ptr(), d_p.data_ptr(), d_in.data_ptr(),
                                 d_out.data_ptr(), 4, n_steps, 2, 3, H.f32(0.01), 9, H.f32(1.0), -1, None) != 0                   # no such ERenderMode
    assert b"render mode" in ngp.ngp_hip_last_error()


def test_shade_accumulate_tonemap(ngp, oracle, cuda):
    rs = np.random.RandomState(1)
    n_hit, res = 1500, np.array([64, 32], np.int32)
    npx = 64 * 32
    rgba = rs.rand(n_hit, 4).astype(np.float32)
    depth = rs.rand(n_hit).astype(np.float32)
    pay = np.zeros(n_hit, H.PAYLOAD)
    pay["idx"] = rs.permutation(npx)[:n_hit]
    for linear in (0, 1):
        fb, db = rs.rand(npx, 4).astype(np.float32), rs.rand(npx).astype(np.float32)
        d_fb, d_db = H.to_dev(fb, cuda), H.to_dev(db, cuda)
        oracle.orc_shade(n_hit, rgba.ctypes.data, depth.ctypes.data, pay.ctypes.data, linear, fb.ctypes.data, db.ctypes.data)
        d_rgba, d_depth, d_pay = H.to_dev(rgba, cuda), H.to_dev(depth, cuda), H.to_dev(pay, cuda)
        check(ngp.ngp_hip_shade(None, n_hit, d_rgba.data_ptr(), d_depth.data_ptr(), d_pay.data_ptr(), linear, d_fb.data_ptr(), d_db.data_ptr(), 1))
        np.testing.assert_allclose(H.to_host(d_fb, np.float32).reshape(npx, 4), fb, rtol=2e-4, atol=1e-6)
        np.testing.assert_array_equal(H.to_host(d_db, np.float32), db)
    pay["n_steps"] = rs.randint(0, 300, n_hit)
    for mode in (0, 4, 6):                          # AO / Depth: no sRGB decode; Cost: n_steps / 128, opaque
        fb, db = rs.rand(npx, 4).astype(np.float32), rs.rand(npx).astype(np.float32)
        d_fb, d_db, d_pay = H.to_dev(fb, cuda), H.to_dev(db, cuda), H.to_dev(pay, cuda)
        oracle.orc_shade_mode(n_hit, rgba.ctypes.data, depth.ctypes.data, pay.ctypes.data, 0, fb.ctypes.data, db.ctypes.data, mode)
        check(ngp.ngp_hip_shade(None, n_hit, d_rgba.data_ptr(), d_depth.data_ptr(), d_pay.data_ptr(), 0, d_fb.data_ptr(), d_db.data_ptr(), mode))
        np.testing.assert_allclose(H.to_host(d_fb, np.float32).reshape(npx, 4), fb, rtol=2e-6, atol=1e-7)
        np.testing.assert_array_equal(H.to_host(d_db, np.float32), db)
    for cs in (0, 1):
        acc = rs.rand(npx, 4).astype(np.float32)
        d_acc = H.to_dev(acc, cuda)
        for spp in (0.0, 3.0):
            oracle.orc_accumulate(res.ctypes.data, fb.ctypes.data, acc.ctypes.data, H.f32(spp), cs)
            d_fb2 = H.to_dev(fb, cuda)
            check(ngp.ngp_hip_accumulate(None, res.ctypes.data, d_fb2.data_ptr(), d_acc.data_ptr(), H.f32(spp), cs))
        np.testing.assert_allclose(H.to_host(d_acc, np.float32).reshape(npx, 4), acc, rtol=2e-4, atol=1e-6)
        bg = np.array([0.1, 0.3, 0.6, 0.8], np.float32)
        for curve in (0, 1, 2, 3):
            for out_cs in (0, 1):
                surf = np.zeros((npx, 4), np.float32)
                oracle.orc_tonemap(res.ctypes.data, H.f32(0.5), bg.ctypes.data, acc.ctypes.data, cs, out_cs, curve, 1, surf.ctypes.data)
                d_s = H.dev_zeros(npx * 16, cuda)
                check(ngp.ngp_hip_tonemap(None, res.ctypes.data, H.f32(0.5), bg.ctypes.data, d_acc.data_ptr(), cs, out_cs, curve, 1, d_s.data_ptr()))
                np.testing.assert_allclose(H.to_host(d_s, np.float32).reshape(npx, 4), surf, rtol=3e-4, atol=2e-6)


def test_full_frame_matches_oracle(ngp, oracle, cuda):
    """render_nerf (testbed_nerf.cu:2354-2500) driven from Python over the C ABI vs orc_render_nerf on a small frame."""
    desc = H.make_desc(ngp, log2_hashmap_size=14)
    params = H.random_params(desc, seed=3, grid_amp=2.0, mlp_gain=2.0)
    params[2048:2048 + 64] *= 6.0  # density logit row of W2: make the blobs opaque enough to terminate rays
    cam, focal, res, sc = _camera()
    aabb = H.unit_aabb()
    grid = H.blob_density_grid(1)
    bf, _ = H.oracle_bitfield(oracle, grid, 1)
    n = W * Hh
    ident = np.eye(3, dtype=np.float32).reshape(-1)
    fb_ref, db_ref = np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    oracle.orc_render_nerf(desc.ctypes.data, params.ctypes.data, 0, res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, sc.ctypes.data, 1, aabb.ctypes.data,
                           ident.ctypes.data, aabb.ctypes.data, H.f32(0.0), bf.ctypes.data, H.f32(0.0), 2, 3, H.f32(0.01), 0, fb_ref.ctypes.data, db_ref.ctypes.data)

    import torch
    d_desc, d_P, d_bf = H.to_dev(desc, cuda), H.to_dev(params, cuda), H.to_dev(bf, cuda)
    zero4, zero3 = np.zeros(4, np.float32), np.zeros(3, np.float32)
    pay = [H.dev_zeros(n * 40, cuda), H.dev_zeros(n * 40, cuda)]
    rgba = [H.dev_zeros(n * 16, cuda), H.dev_zeros(n * 16, cuda)]
    dep = [H.dev_zeros(n * 4, cuda), H.dev_zeros(n * 4, cuda)]
    hp, hr, hd = H.dev_zeros(n * 40, cuda), H.dev_zeros(n * 16, cuda), H.dev_zeros(n * 4, cuda)
    net_in, net_out = H.dev_zeros(n * 8 * 28, cuda), H.dev_zeros(n * 8 * 8, cuda)
    fb, db = H.dev_zeros(n * 16, cuda), H.dev_zeros(n * 4, cuda)
    cnt, hcnt = H.dev_zeros(4, cuda), H.dev_zeros(4, cuda)
    check(ngp.ngp_hip_init_rays(None, 0, pay[0].data_ptr(), res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, zero4.ctypes.data, sc.ctypes.data, zero3.ctypes.data,
                                1, aabb.ctypes.data, ident.ctypes.data, H.f32(0.0), 0, None, db.data_ptr(), H.f32(1.0), H.f32(0.0), None, None))
    check(ngp.ngp_hip_advance_pos(None, n, aabb.ctypes.data, ident.ctypes.data, 0, pay[0].data_ptr(), d_bf.data_ptr(), 0, H.f32(0.0)))
    n_alive, i, dbi = n, 1, 0
    while i < 10000:
        cur, tmp = (dbi + 1) % 2, dbi % 2
        dbi += 1
        cnt.zero_()
        check(ngp.ngp_hip_compact_rays(None, n_alive, rgba[tmp].data_ptr(), dep[tmp].data_ptr(), pay[tmp].data_ptr(), rgba[cur].data_ptr(), dep[cur].data_ptr(), pay[cur].data_ptr(),
                                       hr.data_ptr(), hd.data_ptr(), hp.data_ptr(), cnt.data_ptr(), hcnt.data_ptr()))
        n_alive = int(H.to_host(cnt, np.uint32)[0])
        if n_alive == 0:
            break
        n_steps = min(max(n // n_alive, 1), 8)
        check(ngp.ngp_hip_generate_next_inputs(None, n_alive, aabb.ctypes.data, aabb.ctypes.data, pay[cur].data_ptr(), net_in.data_ptr(), n_steps, d_bf.data_ptr(), 0, H.f32(0.0), 0))
        check(ngp.ngp_hip_nerf_inference(None, d_desc.data_ptr(), d_P.data_ptr(), net_in.data_ptr(), 7, n_alive * n_steps, net_out.data_ptr(), 4))
        check(ngp.ngp_hip_composite(None, n_alive, i, aabb.ctypes.data, cam.ctypes.data, rgba[cur].data_ptr(), dep[cur].data_ptr(), pay[cur].data_ptr(), net_in.data_ptr(),
                                    net_out.data_ptr(), 4, n_steps, 2, 3, H.f32(0.01), 1, H.f32(1.0), -1, None))
        i += n_steps
    n_hit = int(H.to_host(hcnt, np.uint32)[0])
    check(ngp.ngp_hip_shade(None, n_hit, hr.data_ptr(), hd.data_ptr(), hp.data_ptr(), 0, fb.data_ptr(), db.data_ptr(), 1))
    torch.cuda.synchronize()
    got = H.to_host(fb, np.float32).reshape(n, 4)
    assert (fb_ref[:, 3] > 0.5).sum() > 100  # there is an object in view
    diff = np.abs(got - fb_ref)
    # fp16 MLP + early-termination thresholds: demand a tight mean and a small tail
    assert diff.mean() < 2e-3, diff.mean()
    assert (diff.max(axis=1) > 0.05).mean() < 0.01
    np.testing.assert_array_equal(got[:, 3] > 0, fb_ref[:, 3] > 0)


@pytest.mark.parametrize("lens_mode,params", [
    (1, [0.05, -0.01, 0.001, -0.002, 0, 0, 0]),                       # OpenCV k1 k2 p1 p2
    (2, [0.0, 1.1e-3 * 48, 0.0, 2e-9, 0.0, 48.0, 36.0]),              # FTheta p0..p4, w, h (alpha = p1 * r: ~equidistant fisheye)
    (3, [0, 0, 0, 0, 0, 0, 0]),                                       # LatLong
])
def test_init_rays_lens_models(ngp, oracle, cuda, lens_mode, params):
    """render lenses of pixel_to_ray (common_device.cuh:260-317) beyond Perspective; sinf / cosf differ by ulps between libm and the device"""
    cam, focal, res, sc = _camera()
    aabb = H.unit_aabb(2)
    n = W * Hh
    ident = np.eye(3, dtype=np.float32).reshape(-1)
    zero4, zero3 = np.zeros(4, np.float32), np.zeros(3, np.float32)
    lp = np.array(params, np.float32)
    if lens_mode == 2:
        lp[5], lp[6] = W, Hh
    pay, depth = np.zeros(n, H.PAYLOAD), np.zeros(n, np.float32)
    oracle.orc_init_rays(2, pay.ctypes.data, res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, zero4.ctypes.data, sc.ctypes.data, zero3.ctypes.data,
                         0, aabb.ctypes.data, ident.ctypes.data, H.f32(0.05), lens_mode, lp.ctypes.data, depth.ctypes.data, H.f32(1.0), H.f32(0.0), None)
    d_pay, d_depth = H.dev_zeros(n * 40, cuda), H.dev_zeros(n * 4, cuda)
    check(ngp.ngp_hip_init_rays(None, 2, d_pay.data_ptr(), res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, zero4.ctypes.data, sc.ctypes.data,
                                zero3.ctypes.data, 0, aabb.ctypes.data, ident.ctypes.data, H.f32(0.05), lens_mode, lp.ctypes.data, d_depth.data_ptr(), H.f32(1.0), H.f32(0.0), None, None))
    g = H.to_host(d_pay, H.PAYLOAD)
    same = g["alive"] == pay["alive"]
    assert same.mean() > 0.995                      # a ray grazing the box may flip with a 1-ulp different direction
    al = (pay["alive"] == 1) & same
    assert al.sum() > 100
    tol = 0 if lens_mode == 1 else 2e-6
    np.testing.assert_allclose(g["dir"][al], pay["dir"][al], rtol=0, atol=tol)
    np.testing.assert_allclose(g["origin"][al], pay["origin"][al], rtol=0, atol=tol)
    np.testing.assert_allclose(g["t"][al], pay["t"][al], rtol=0, atol=1e-5 if lens_mode != 1 else 0)
    # the lens actually bends the rays: directions differ from the pinhole ones
    pin = np.zeros(n, H.PAYLOAD)
    oracle.orc_init_rays(2, pin.ctypes.data, res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, zero4.ctypes.data, sc.ctypes.data, zero3.ctypes.data,
                         0, aabb.ctypes.data, ident.ctypes.data, H.f32(0.05), 0, None, depth.ctypes.data, H.f32(1.0), H.f32(0.0), None)
    both = al & (pin["alive"] == 1)
    assert both.sum() > 50 and np.abs(pin["dir"][both] - pay["dir"][both]).max() > 1e-3


def test_pyngp_depth_of_field_and_autofocus(cuda):
    """python_api.cu:662-666: `aperture_size` / `dof`, `slice_plane_z`, `autofocus`, `autofocus_target` drive the stock renderer"""
    import scene
    ds = scene.make_dataset(n_train=8, n_test=1, res=48, device=cuda)
    t = scene.build_testbed(ds)
    scene.train(t, 80)
    t.shall_train = False
    t.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
    sharp = t.render(48, 48, 4, True)
    t.autofocus_target = [0.5, 0.5, 0.5]
    t.autofocus = True
    t.aperture_size = 0.06
    assert t.dof == np.float32(0.06)
    blurred = t.render(48, 48, 4, True)
    cam = np.asarray(t.camera_matrix, np.float32)
    want = max(float(cam[:, 2] @ (np.array([0.5, 0.5, 0.5], np.float32) - cam[:, 3])), 0.1) - t.scale
    assert abs(t.slice_plane_z - want) < 1e-5
    assert np.isfinite(blurred).all() and blurred[..., 3].max() > 0.5
    assert np.abs(blurred - sharp).mean() > 1e-3          # the lens blur is visible (the scene is only trained for 80 steps)
    # gradients across pixels are weaker in the blurred image away from the focal plane
    def sharpness(img):
        return float(np.abs(np.diff(img[..., :3], axis=0)).mean() + np.abs(np.diff(img[..., :3], axis=1)).mean())
    assert sharpness(blurred) < sharpness(sharp)
    t.aperture_size = 0.0
    again = t.render(48, 48, 4, True)
    np.testing.assert_allclose(again, sharp, atol=1e-6)


def test_pyngp_stock_renderer_camera_models(cuda):
    """python_api.cu:691-693: render_camera_model, camera_spherical_quadrilateral, camera_quadrilateral_hexahedron"""
    import pyngp
    import scene
    ds = scene.make_dataset(n_train=8, n_test=1, res=48, device=cuda)
    t = scene.build_testbed(ds)
    scene.train(t, 60)
    t.shall_train = False
    t.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
    pin = t.render(40, 30, 1, True)
    assert t.render_camera_model == pyngp.CameraModel.Perspective
    sq = pyngp.SphericalQuadrilateralConfig.Zero()
    sq.width, sq.height, sq.curvature = 0.8, 0.6, 0.2
    t.camera_spherical_quadrilateral = sq
    got = t.camera_spherical_quadrilateral
    assert (got.width, got.height, got.curvature) == (np.float32(0.8), np.float32(0.6), np.float32(0.2))
    t.render_camera_model = pyngp.CameraModel.SphericalQuadrilateral
    img = t.render(40, 30, 1, True)
    assert np.isfinite(img).all() and np.abs(img - pin).mean() > 1e-3
    t.render_camera_model = pyngp.CameraModel.Perspective
    np.testing.assert_allclose(t.render(40, 30, 1, True), pin, atol=1e-6)


def test_pyngp_render_modes(cuda):
    """python_api.cu:660 `render_mode`: AO, Positions, Depth and Cost through Testbed.render"""
    import pyngp
    import scene
    ds = scene.make_dataset(n_train=8, n_test=1, res=48, device=cuda)
    t = scene.build_testbed(ds)
    scene.train(t, 80)
    t.shall_train = False
    t.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
    t.background_color = [0.0, 0.0, 0.0, 0.0]
    shade = t.render(40, 30, 1, True)
    t.render_mode = pyngp.RenderMode.Depth
    depth = t.render(40, 30, 1, True)
    hit = depth[..., 3] > 0.9                                                                                   # opaque pixels
    assert hit.sum() > 30 and np.isfinite(depth).all()
    d = depth[hit]
    assert (d[:, 0] > 0).all() and np.allclose(d[:, 0], d[:, 1]) and np.allclose(d[:, 0], d[:, 2])              # grey = distance along the view axis / dataset scale
    t.render_mode = pyngp.RenderMode.Positions
    pos = t.render(40, 30, 1, True)
    assert (pos[hit][:, :3] > 0.2).all() and (pos[hit][:, :3] < 0.8).all()                                       # (p - 0.5) / 2 + 0.5 of points inside the unit cube
    t.render_mode = pyngp.RenderMode.Cost
    cost = t.render(40, 30, 1, True)
    assert (cost[hit][:, 3] == 1.0).all() and cost[hit][:, 0].max() > 0
    t.render_mode = pyngp.RenderMode.AO
    assert np.isfinite(t.render(40, 30, 1, True)).all()
    t.render_mode = pyngp.RenderMode.Shade
    np.testing.assert_allclose(t.render(40, 30, 1, True), shade, atol=1e-6)
