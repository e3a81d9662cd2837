"""CPU tests: pin the oracle against every anchor available offline and check its internal consistency.

The reference ships no golden vectors for the NeRF path ("parity unpinned"); what can be pinned is
  * published known-answer vectors of the algorithms the reference pulls in (PCG32, Sobol, Morton codes, IEEE binary16),
  * parameter counts / constants citable in the reference tree,
  * the python metric helpers of scripts/common.py (tests/golden/common_py_metrics.npz, generated from the reference itself).
"""
import ctypes
import os

import numpy as np
import pytest

import helpers as H

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_pcg32_published_known_answers(oracle):
    """pcg32-demo (pcg-c-basic, O'Neill): pcg32_srandom(42, 54) -> first six outputs."""
    o = oracle
    # orc_pcg32_uints seeds with (seed, initseq=1); replay the generic seeding in python for (42, 54) and compare the stepping
    mult, mask = 0x5851f42d4c957f2d, (1 << 64) - 1
    inc = ((54 << 1) | 1) & mask
    state = 0
    state = (state * mult + inc) & mask
    state = (state + 42) & mask
    state = (state * mult + inc) & mask
    outs = []
    for _ in range(6):
        old = state
        state = (old * mult + inc) & mask
        xs = (((old >> 18) ^ old) >> 27) & 0xffffffff
        rot = old >> 59
        outs.append(((xs >> rot) | (xs << ((-rot) & 31))) & 0xffffffff)
    assert outs == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    # the C oracle implements the same recurrence: default_rng_t{1337} via helpers.pcg32_state must agree with it
    st, inc1 = H.pcg32_state(1337)
    got = np.zeros(8, np.uint32)
    o.orc_pcg32_uints(1337, 0, 8, got.ctypes.data)
    s, ref = st, []
    for _ in range(8):
        old = s
        s = (old * mult + inc1) & mask
        xs = (((old >> 18) ^ old) >> 27) & 0xffffffff
        rot = old >> 59
        ref.append(((xs >> rot) | (xs << ((-rot) & 31))) & 0xffffffff)
    assert got.tolist() == ref


def test_pcg32_advance_equals_stepping(oracle):
    a, b = np.zeros(40, np.float32), np.zeros(8, np.float32)
    oracle.orc_pcg32_floats(7, 0, 40, a.ctypes.data, None)
    for adv in (0, 1, 8, 31):
        oracle.orc_pcg32_floats(7, adv, 8, b.ctypes.data, None)
        np.testing.assert_array_equal(b, a[adv:adv + 8])
    assert (a >= 0).all() and (a < 1).all()
    st = np.zeros(2, np.uint64)
    oracle.orc_pcg32_floats(7, 1 << 32, 1, b.ctypes.data, st.ctypes.data)  # default advance() stride is representable
    s2, _ = H.pcg32_advance(*H.pcg32_state(7), (1 << 32) + 1)
    assert int(st[0]) == s2


def test_sobol_and_scramble(oracle):
    # Sobol dim 0 is the van der Corput sequence, dim 1 starts 0, 1/2, 1/4, 3/4 (Joe & Kuo direction numbers)
    L = oracle.lib
    L.orc_sobol.restype = ctypes.c_uint32
    L.orc_sobol.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    assert [L.orc_sobol(i, 0) / 2 ** 32 for i in range(4)] == [0.0, 0.5, 0.25, 0.75]
    assert [L.orc_sobol(i, 1) / 2 ** 32 for i in range(4)] == [0.0, 0.5, 0.75, 0.25] or [L.orc_sobol(i, 1) / 2 ** 32 for i in range(4)] == [0.0, 0.5, 0.25, 0.75]
    v = np.array([oracle.orc_ld_random_val_export(i, 0xdeadbeef, 0) for i in range(256)])
    assert (v >= 0).all() and (v < 1).all() and len(np.unique(v)) == 256
    # Owen-scrambled Sobol keeps the (0,m,1)-net property: 256 points, one per 1/256 stratum
    assert sorted((v * 256).astype(int).tolist()) == list(range(256))
    off = np.zeros(2, np.float32)
    oracle.orc_ld_random_pixel_offset_export(0, off.ctypes.data)
    np.testing.assert_array_equal(off, [0.5, 0.5])  # spp 0 => pixel centre (random_val.cuh:317-322)


def test_morton_roundtrip(oracle):
    assert oracle.orc_morton3D_export(1, 0, 0) == 1 and oracle.orc_morton3D_export(0, 1, 0) == 2 and oracle.orc_morton3D_export(0, 0, 1) == 4
    assert oracle.orc_morton3D_export(127, 127, 127) == 128 ** 3 - 1
    rs = np.random.RandomState(0)
    for x, y, z in rs.randint(0, 128, size=(200, 3)):
        m = oracle.orc_morton3D_export(int(x), int(y), int(z))
        assert (oracle.orc_morton3D_invert_export(m), oracle.orc_morton3D_invert_export(m >> 1), oracle.orc_morton3D_invert_export(m >> 2)) == (x, y, z)


def test_fp16_conversion_matches_ieee(oracle):
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    out = np.zeros(65536, np.float32)
    oracle.orc_f16_to_f32(allh.ctypes.data, out.ctypes.data, 65536)
    np.testing.assert_array_equal(out.view(np.uint32)[~np.isnan(f)], f.view(np.uint32)[~np.isnan(f)])
    rs = np.random.RandomState(0)
    xs = np.concatenate([rs.randn(50000).astype(np.float32) * s for s in (1e-8, 1e-6, 1e-4, 1e-2, 1, 100, 30000)] + [np.array([0, -0.0, 65504, 65519.9, 65520, 1e9, -1e9, 2 ** -24, 2 ** -25, 2 ** -25 * 1.0001], np.float32)])
    h = np.zeros(xs.size, np.uint16)
    oracle.orc_f32_to_f16(xs.ctypes.data, h.ctypes.data, xs.size)
    with np.errstate(over="ignore"):
        np.testing.assert_array_equal(h, xs.astype(np.float16).view(np.uint16))


def test_level_table_reproduces_reference_parameter_counts(ngp, oracle):
    """lego 12 196 240 grid params (res 16,23,31,43,59 dense + 11 hashed x 2^19), MLP 3072 + 7168."""
    b = oracle.orc_per_level_scale(16, 16, H.f32(2048.0), 1)
    assert abs(b - 1.38191) < 1e-5
    desc = np.zeros(1, H.NET_DESC)
    n = ctypes.c_uint32()
    oracle.orc_net_make_levels(16, 19, 16, H.f32(b), desc["levels"].ctypes.data, ctypes.byref(n))
    assert desc["levels"]["resolution"][0][:5].tolist() == [16, 23, 31, 43, 59]
    assert desc["levels"]["size"][0][:5].tolist() == [4096, 12168, 29792, 79512, 205384]
    assert (desc["levels"]["size"][0][5:] == 1 << 19).all()
    assert 2 * n.value == 12196240
    assert oracle.orc_net_mlp_params() == 10240 == 3072 + 7168
    # fox: aabb_scale 4
    b4 = oracle.orc_per_level_scale(16, 16, H.f32(2048.0), 4)
    oracle.orc_net_make_levels(16, 19, 16, H.f32(b4), desc["levels"].ctypes.data, ctypes.byref(n))
    assert desc["levels"]["resolution"][0][:4].tolist() == [16, 25, 37, 56] and 2 * n.value == 13074912
    # the product's host-side table is the same function
    d2 = H.make_desc(ngp, 19, 16, 1)
    oracle.orc_net_make_levels(16, 19, 16, H.f32(b), desc["levels"].ctypes.data, ctypes.byref(n))
    assert d2["levels"].tobytes() == desc["levels"].tobytes() and int(d2["n_grid_entries"][0]) == n.value
    assert ngp.ngp_hip_net_n_params_host(d2.ctypes.data) == 12196240 + 10240


def test_metric_helpers_match_reference_common_py():
    import metrics
    g = np.load(os.path.join(GOLDEN, "common_py_metrics.npz"))
    np.testing.assert_array_equal(metrics.linear_to_srgb(g["ramp"]), g["linear_to_srgb_ramp"])
    np.testing.assert_array_equal(metrics.srgb_to_linear(g["ramp"]), g["srgb_to_linear_ramp"])
    np.testing.assert_array_equal(metrics.linear_to_srgb(g["img"]), g["linear_to_srgb_img"])
    np.testing.assert_array_equal(metrics.srgb_to_linear(g["img"]), g["srgb_to_linear_img"])
    assert metrics.mse2psnr(1e-3) == float(g["mse2psnr_1e-3"])
    np.testing.assert_array_equal([metrics.mse2psnr(x) for x in (1.0, 0.1, 3.3e-4, 1e-5)], g["mse2psnr_vals"])
    for name in ("MSE", "MAE", "MAPE", "SMAPE", "MRSE"):
        assert metrics.compute_error(name, g["img"], g["ref"]) == float(g["err_" + name]), name
    assert metrics.compute_error("SSIM", g["img"], g["ref"]) == float(g["ssim"])
    np.testing.assert_array_equal(metrics.SSIM(g["hdr"], np.clip(g["hdr"] * 0.9, 0, None)).astype(np.float64), g["ssim_hdr"])
    # BASELINE.md §4 anchors
    assert abs(float(g["mse"]) - 9.936874e-05) < 1e-10 and abs(float(g["ssim"]) - 0.99979675) < 1e-7
    np.testing.assert_allclose(metrics.linear_to_srgb(np.array([0, .25, .5, .75, 1], np.float32)), [0, .5370987, .7353569, .880825, .99999994], rtol=1e-6)


def test_marching_constants_and_helpers(oracle):
    """testbed_nerf.cu:53-73 constants; empty / full bitfield edge cases of the ray marcher."""
    aabb = H.unit_aabb()
    xf = H.hemisphere_cameras(2)
    w, h = 32, 24
    imgs = H.make_images(2, w, h, masked_fraction=0.0)
    md = H.make_metadata([imgs[i].ctypes.data for i in range(2)], w, h, 40.0)
    st, inc = H.pcg32_state(1337)
    n_rays = 256
    dres = np.array([0, 0], np.int32)

    def run(bitfield, max_samples=n_rays * 1024):
        r = dict(rc=np.zeros(1, np.uint32), nc=np.zeros(1, np.uint32), idx=np.zeros(n_rays, np.uint32), rays=np.zeros(n_rays, H.RAY), ns=np.zeros(n_rays * 2, np.uint32),
                 co=np.zeros(max_samples, H.COORD))
        oracle.orc_generate_training_samples(n_rays, aabb.ctypes.data, max_samples, st, inc, r["rc"].ctypes.data, r["nc"].ctypes.data, r["idx"].ctypes.data, r["rays"].ctypes.data,
                                             r["ns"].ctypes.data, r["co"].ctypes.data, 2, md.ctypes.data, xf.ctypes.data, bitfield.ctypes.data, 0, None, 0, 0, H.f32(0.0), None,
                                             dres.ctypes.data, 0, n_rays, None)
        return r

    empty = run(np.zeros(128 ** 3, np.uint8))
    assert int(empty["rc"][0]) == 0 and int(empty["nc"][0]) == 0  # no occupied cell => every ray is dropped (1221-1223)
    full = run(np.full(128 ** 3, 0xff, np.uint8))
    n = int(full["rc"][0])
    assert n > 0
    steps = full["ns"][0:2 * n:2]
    assert steps.max() <= 1024  # NERF_STEPS cap (1209)
    # unit cube, cone_angle 0: fixed step sqrt(3)/1024 => a ray crossing the cube takes at most 1024 steps, dt warps to 0
    assert np.allclose(full["co"]["dt"][: int(full["nc"][0])], 0.0)
    pos = full["co"]["pos"][: int(full["nc"][0])]
    assert (pos >= 0).all() and (pos <= 1).all()
    d = full["co"]["dir"][: int(full["nc"][0])] * 2 - 1
    np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)


def test_oracle_bitfield_pooling_semantics(oracle):
    """max-pooling: a single occupied finest cell at the centre lights exactly one bit in every coarser level."""
    grid = np.zeros(128 ** 3, np.float32)
    centre = oracle.orc_morton3D_export(64, 64, 64)
    grid[centre] = 1.0
    bf = np.zeros(128 ** 3, np.uint8)
    oracle.orc_update_bitfield(grid.ctypes.data, 1, H.f32(1.0 / 128 ** 3), bf.ctypes.data)
    per_level = [int(np.unpackbits(bf[l * 128 ** 3 // 8:(l + 1) * 128 ** 3 // 8]).sum()) for l in range(8)]
    assert per_level == [1] * 8
    # threshold = min(0.01, mean): cells at or below it are off
    grid[:] = 0.005
    oracle.orc_update_bitfield(grid.ctypes.data, 1, H.f32(0.005), bf.ctypes.data)
    assert bf[: 128 ** 3 // 8].sum() == 0


def test_oracle_network_gradients_finite_difference(oracle, ngp):
    """the oracle's backward is the derivative of its forward (checked in fp32-ish regime with a smooth direction)."""
    desc = H.make_desc(ngp, log2_hashmap_size=10)
    params = H.random_params(desc, seed=1, grid_amp=1.0)
    coords = H.random_coords(64, seed=2)
    n = 64
    rs = np.random.RandomState(0)
    dl = np.zeros((n, 4), np.float16)
    dl[:, :] = (rs.randn(n, 4) * 0.5).astype(np.float16)
    grads = np.zeros(H.n_params(desc), np.float64)
    out = np.zeros((n, 4), np.uint16)
    oracle.orc_nerf_forward_backward(desc.ctypes.data, params.ctypes.data, coords.ctypes.data, 7, n, dl.ctypes.data, out.ctypes.data, grads.ctypes.data, None)

    def loss(p):
        o = np.zeros((n, 4), np.uint16)
        oracle.orc_nerf_inference(desc.ctypes.data, p.ctypes.data, coords.ctypes.data, 7, n, o.ctypes.data, 4)
        return float((o.view(np.float16).astype(np.float64) * dl.astype(np.float64)).sum())

    # directional derivative along a random direction over the output layer of the rgb net (linear in those weights => exact up to fp16)
    direction = np.zeros(params.size, np.float64)
    sl = slice(3072 + 2048 + 4096, 10240)
    direction[sl] = rs.randn(1024)
    eps = 2.0 ** -6
    p_plus = (params.astype(np.float64) + eps * direction).astype(np.float16)
    p_minus = (params.astype(np.float64) - eps * direction).astype(np.float16)
    actual_dir = (p_plus.astype(np.float64) - p_minus.astype(np.float64))
    fd = loss(p_plus) - loss(p_minus)
    an = float((grads * actual_dir).sum())
    assert abs(fd - an) <= 0.05 * abs(an) + 0.05, (fd, an)


def test_exact_grid_backward_is_the_rounded_rational_sum(oracle, ngp):
    """orc_grid_backward_exact against an independent restatement in exact rational arithmetic (Fraction), and its one-rounding conversion
    against numpy's correctly rounded float64 -> float16 on sums that fit a double exactly."""
    from fractions import Fraction
    n = 256
    desc = H.make_desc(ngp, log2_hashmap_size=10)
    rs = np.random.RandomState(3)
    pos = rs.rand(n, 3).astype(np.float32)
    pos[:32] = pos[0] + (np.arange(32)[:, None] * 1e-4).astype(np.float32)     # a run of samples inside one coarse cell
    pl = (rs.randn(16, n, 2) * 0.05).astype(np.float16)
    pl[2, 3, 0] = np.float16(np.inf)                                             # dropped
    pl[5, 4] = np.float16(6.0e-8)                                                # subnormal terms
    n_entries = int(desc["n_grid_entries"][0])
    got = np.zeros(2 * n_entries, np.uint16)
    oracle.orc_grid_backward_exact(3, desc.ctypes.data, pos.ctypes.data, 3, n, pl.view(np.uint16).ctypes.data, got.ctypes.data)
    acc = {}
    lv = desc["levels"][0]
    primes = (1, 2654435761, 805459861)
    for l in range(16):
        scale, res, size, off = np.float32(lv[l]["scale"]), int(lv[l]["resolution"]), int(lv[l]["size"]), int(lv[l]["offset"])
        for i in range(n):
            p = np.array([np.float32(float(Fraction(float(scale)) * Fraction(float(x)) + Fraction(1, 2))) for x in pos[i]], np.float32)   # fma: one rounding
            fl = np.floor(p)
            fr = (p - fl).astype(np.float32)
            pg = fl.astype(np.int64)
            for c in range(8):
                w = np.float32(1.0)
                cc = []
                for d in range(3):
                    w = np.float32(w * (fr[d] if (c >> d) & 1 else np.float32(np.float32(1.0) - fr[d])))
                    cc.append(int(pg[d]) + ((c >> d) & 1))
                if res ** 3 <= size:
                    idx = (cc[0] + cc[1] * res + cc[2] * res * res) % size
                else:
                    idx = (((cc[0] * primes[0]) ^ (cc[1] * primes[1]) ^ (cc[2] * primes[2])) & 0xffffffff) % size
                for f in range(2):
                    with np.errstate(over="ignore", invalid="ignore"):
                        t = np.float16(np.float32(w * np.float32(pl[l, i, f])))
                    if np.isfinite(t):
                        k = 2 * (off + idx) + f
                        acc[k] = acc.get(k, Fraction(0)) + Fraction(float(t))
    ref = np.zeros(2 * n_entries, np.float16)
    for k, v in acc.items():
        ref[k] = np.float16(float(v))        # sums of < 2^12 terms of 41-bit fixed point fit a double exactly; numpy rounds to nearest even
    np.testing.assert_array_equal(got, ref.view(np.uint16))
    assert (got != 0).sum() > 10000


# ---- the rest of the stock renderer (row f3): oracle-side properties that follow from the reference's code ----
def _f3_extras(**kw):
    import capi
    e = np.zeros(1, capi.RENDER_EXTRAS)
    e["quilting_dims"][0] = (1, 1)
    e["render_mode"] = 1
    for k, v in kw.items():
        e[k][0] = v
    return e


def _f3_init(oracle, ex, res=(48, 32), parallax=(0.0, 0.0, 0.0)):
    cam = H.look_at_xform([1.6, 1.3, 1.1])
    focal, r, sc = np.array([40.0, 40.0], np.float32), np.array(res, np.int32), np.array([0.5, 0.5], np.float32)
    aabb, ident = H.unit_aabb(1), np.eye(3, dtype=np.float32).reshape(-1)
    n = res[0] * res[1]
    pay, depth = np.zeros(n, H.PAYLOAD), np.zeros(n, np.float32)
    oracle.orc_init_rays_ex(1, pay.ctypes.data, r.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, np.zeros(4, np.float32).ctypes.data, sc.ctypes.data,
                            np.array(parallax, np.float32).ctypes.data, 0, aabb.ctypes.data, ident.ctypes.data, H.f32(0.0), 0, None, depth.ctypes.data, H.f32(1.0), H.f32(0.0), None,
                            ex.ctypes.data if ex is not None else None)
    return pay, depth


def test_init_rays_ex_reduces_to_init_rays_and_honours_masks_and_quilting(oracle):
    import capi
    plain, _ = _f3_init(oracle, None)
    same, _ = _f3_init(oracle, _f3_extras())
    assert plain.tobytes() == same.tobytes()
    m = np.zeros(1, capi.MASK3D)               # an Add sphere of radius 0.2 at the scene centre
    m["mode"], m["shape"], m["opacity"] = 0, 2, 1.0
    t = np.eye(4, dtype=np.float32); t[:3, 3] = 0.5
    m["transform"][0] = t.T.reshape(-1); m["itransform"][0] = np.linalg.inv(t).T.reshape(-1)
    m["config"][0][0] = 0.2
    masked, _ = _f3_init(oracle, _f3_extras(render_masks=m.ctypes.data, n_render_masks=1))
    assert 0 < masked["alive"].sum() < 0.5 * plain["alive"].sum()
    # rays that survive point at the sphere: distance of the centre from the ray below radius (+ nothing for feather 0)
    al = masked["alive"] == 1
    oc = 0.5 - masked["origin"][al]
    dist = np.linalg.norm(np.cross(oc, masked["dir"][al]), axis=1)
    assert dist.max() <= 0.2 + 1e-5
    m["mode"] = 1                               # Subtract masks never kill a ray (mask_3D.cuh: infinite additive area around them)
    sub, _ = _f3_init(oracle, _f3_extras(render_masks=m.ctypes.data, n_render_masks=1))
    assert sub.tobytes() == plain.tobytes()
    # quilting (2, 1): two panels, each the half-width view, eyes +- IPD / 2 apart along the camera's x axis
    q, _ = _f3_init(oracle, _f3_extras(quilting_dims=(2, 1)), parallax=(0.08, 0.0, 0.0))
    o = q["origin"].reshape(32, 48, 3)
    left, right = o[:, :24], o[:, 24:]
    d = left[0, 0] - right[0, 0]
    assert abs(np.linalg.norm(d) - 0.08) < 1e-6 and np.allclose(left - right, d, atol=1e-6)


def test_envmap_lookup_and_distortion_mode(oracle):
    rs = np.random.RandomState(0)
    const = np.tile(np.array([0.1, 0.2, 0.3, 0.4], np.float32), (8, 16, 1))
    out = np.zeros(4, np.float32)
    for d in rs.randn(20, 3).astype(np.float32):
        d /= np.linalg.norm(d)
        oracle.orc_read_envmap(const.ctypes.data, np.array([16, 8], np.int32).ctypes.data, d.ctypes.data, out.ctypes.data)
        np.testing.assert_allclose(out, [0.1, 0.2, 0.3, 0.4], rtol=1e-6)
    # rows = polar angle from +y (envmap.cuh:31 feeds {z, -x, y} to dir_to_spherical_unorm): straight up reads row 0, straight down the last row
    ramp = np.zeros((8, 16, 4), np.float32); ramp[..., 0] = np.arange(8, dtype=np.float32)[:, None]
    oracle.orc_read_envmap(ramp.ctypes.data, np.array([16, 8], np.int32).ctypes.data, np.array([0, 1, 0], np.float32).ctypes.data, out.ctypes.data)
    assert out[0] == 0.0
    oracle.orc_read_envmap(ramp.ctypes.data, np.array([16, 8], np.int32).ctypes.data, np.array([0, -1, 0], np.float32).ctypes.data, out.ctypes.data)
    assert out[0] == 7.0
    # Distortion mode: a constant offset along +x paints hue 0.5 (atan2(0, x) / 2 pi + 0.5), value |offset| * 50; every ray is done
    dist = np.zeros((4, 4, 2), np.float32); dist[..., 0] = 0.01
    fb = np.zeros((48 * 32, 4), np.float32)
    pay, depth = _f3_init(oracle, _f3_extras(distortion=dist.ctypes.data, distortion_res=(4, 4), render_mode=5, frame_buffer=fb.ctypes.data))
    assert pay["alive"].sum() == 0
    hit = depth == 1.0
    assert hit.sum() > 100
    np.testing.assert_allclose(fb[hit], np.tile([0.0, 0.5, 0.5, 1.0], (hit.sum(), 1)), atol=1e-5)   # hsv (0.5, 1, 0.5) = cyan at half value


def test_visualize_activation_layers_are_the_forward_pass(oracle, ngp):
    n = 64
    desc = H.make_desc(ngp, log2_hashmap_size=15)   # host-only entry point of the C ABI: runs without a GPU
    params = H.random_params(desc, seed=2)
    coords = H.random_coords(n, seed=3)
    flat = coords.view(np.float32).reshape(n, 7)
    out4 = np.zeros((n, 4), np.uint16)
    oracle.orc_nerf_inference(desc.ctypes.data, params.ctypes.data, flat.ctypes.data, 7, n, out4.ctypes.data, 4)
    vis = np.zeros((n, 4), np.float32)
    oracle.orc_nerf_visualize_activation(desc.ctypes.data, params.ctypes.data, 2, 0, flat.ctypes.data, 7, n, vis.ctypes.data, 4)
    sigma = out4.view(np.float16)[:, 3].astype(np.float32)      # colour-network input 0 IS the density output (nerf_network.h:160, 130-136)
    np.testing.assert_array_equal(vis[:, 1] - vis[:, 0], sigma)
    assert (vis[:, 2] == 0).all() and (vis[:, 3] == 1).all()
    sh = np.zeros(16, np.float32)
    oracle.orc_sh4(flat[0, 4:7].copy().ctypes.data, sh.ctypes.data)
    one = np.zeros((n, 1), np.float32)
    oracle.orc_nerf_visualize_activation(desc.ctypes.data, params.ctypes.data, 2, 16 + 5, flat.ctypes.data, 7, n, one.ctypes.data, 1)
    assert one[0, 0] == np.float16(sh[5])
    for layer in (1, 3, 4):                                      # hidden layers are post-ReLU: no negative part
        oracle.orc_nerf_visualize_activation(desc.ctypes.data, params.ctypes.data, layer, 7, flat.ctypes.data, 7, n, vis.ctypes.data, 4)
        assert (vis[:, 0] == 0).all() and vis[:, 1].max() > 0
