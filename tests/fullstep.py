"""Replay of ONE captured product-path training step (pyngp.Testbed.debug_capture_next_step / debug_captured / debug_scene) through the
CPU oracle, stage by stage and teacher-forced: every oracle stage consumes the DEVICE inputs of that stage, so a deviation is attributed
to the stage that caused it.  Test infrastructure (used by tests/test_baseline_configs_gpu.py at the BASELINE.json sizes and by
bench.py's cpu_baseline leg); the product path never imports this.

Reference stages: generate_training_samples_nerf (src/testbed_nerf.cu:1085-1260), inference (3256), compute_loss_kernel_train_nerf
(1280-1597), fill_rollover* (3314-3322), forward / backward (3330-3331).
"""
import time

import numpy as np

import helpers as H

MIN_STEP = np.float64(1.73205080757 / 1024)


def host_scene(tb, host_images):
    """the scene as the kernels see it, with the pixel pointers redirected to host copies of the same images"""
    sc = tb.debug_scene()
    md = np.frombuffer(sc["metadata"].tobytes(), dtype=H.IMAGE_META).copy()
    assert len(md) == len(host_images)
    for i, img in enumerate(host_images):
        assert img.flags["C_CONTIGUOUS"] and img.dtype == np.uint8 and img.shape == (int(md["res"][i][1]), int(md["res"][i][0]), 4)
        md["pixels"][i] = img.ctypes.data
    md["depth"] = 0
    md["rays"] = 0
    xf = np.frombuffer(sc["xforms"].tobytes(), dtype=H.XFORM).copy()
    aabb = np.zeros(1, dtype=H.AABB)
    aabb["min"][0], aabb["max"][0] = sc["aabb"][0], sc["aabb"][1]
    desc = np.frombuffer(sc["desc"].tobytes(), dtype=H.NET_DESC).copy()
    return dict(md=md, xf=xf, aabb=aabb, desc=desc, bitfield=np.ascontiguousarray(sc["bitfield"]), sc=sc, keep=host_images)


def oracle_march(orc, S, cap, n_rays=None):
    """the oracle's march of the captured step's rays [0, n_rays) (default: all R): same rng, same budget"""
    R = int(cap["R"]) if n_rays is None else int(n_rays)
    max_samples = int(cap["max_inference"])
    r = dict(rc=np.zeros(1, np.uint32), nc=np.zeros(1, np.uint32), idx=np.zeros(R, np.uint32), rays=np.zeros(R, H.RAY), ns=np.zeros(R * 2, np.uint32),
             co=np.zeros(max_samples, H.COORD))
    dist = np.zeros((32, 32, 2), np.float32)
    dres = np.array([32, 32], np.int32)
    orc.orc_generate_training_samples(R, S["aabb"].ctypes.data, max_samples, int(cap["rng_state"]), int(cap["rng_inc"]), r["rc"].ctypes.data, r["nc"].ctypes.data,
                                      r["idx"].ctypes.data, r["rays"].ctypes.data, r["ns"].ctypes.data, r["co"].ctypes.data, len(S["xf"]), S["md"].ctypes.data,
                                      S["xf"].ctypes.data, S["bitfield"].ctypes.data, 0, None, 0, 0, H.f32(S["sc"]["cone_angle_constant"]), dist.ctypes.data,
                                      dres.ctypes.data, int(cap["ray_offset"]), int(cap["n_rays_global"]), None)
    return r


def device_march(cap):
    return dict(rc=cap["gen_counters"][0:1], nc=cap["gen_counters"][1:2], idx=cap["ray_indices"], rays=cap["rays"].view(H.RAY), ns=cap["numsteps"],
                co=cap["coords"].view(H.COORD))


def compare_march(r, g, max_samples=None, note=""):
    """bit-exact: the sample counter, and per ray the kept-ray record and every sample record (slot order is scheduling-dependent).
    When the step's demand exceeds its budget (numsteps counter > max_samples) the rays whose run would not fit are dropped AFTER the counter was
    bumped (testbed_nerf.cu:1225-1228) — which ones depends on the order the atomics were served in, in the reference as much as here — so the kept
    sets may then differ by a few rays; every ray kept by both must still agree bit for bit and every kept run must fit the budget."""
    n_ref, n_got = int(r["rc"][0]), int(g["rc"][0])
    if int(g["nc"][0]) != int(r["nc"][0]):                                                         # bit-exact sample count; say which rays disagree
        ref = {int(r["idx"][k]): int(r["ns"][2 * k]) for k in range(n_ref)}
        got = {int(g["idx"][k]): int(g["ns"][2 * k]) for k in range(n_got)}
        only_r, only_g = sorted(set(ref) - set(got)), sorted(set(got) - set(ref))
        diff = [(k, ref[k], got[k]) for k in ref if k in got and ref[k] != got[k]]
        raise AssertionError("sample counter: device %d, oracle %d (budget %s); rays kept: device %d, oracle %d; only oracle %d rays / %d samples, only device %d rays / %d samples; "
                             "kept by both with different counts: %d %s %s" % (int(g["nc"][0]), int(r["nc"][0]), max_samples, n_got, n_ref, len(only_r), sum(ref[k] for k in only_r),
                                                                             len(only_g), sum(got[k] for k in only_g), len(diff), diff[:8], note))
    overflow = max_samples is not None and int(r["nc"][0]) > max_samples
    if not overflow:
        assert n_got == n_ref, (n_got, n_ref)                                                      # bit-exact ray count
    ref_slot = {int(r["idx"][k]): k for k in range(n_ref)}
    got_slot = {int(g["idx"][k]): k for k in range(n_got)}
    assert len(ref_slot) == n_ref and len(got_slot) == n_got
    if not overflow:
        assert ref_slot.keys() == got_slot.keys()
    else:
        excess = (int(r["nc"][0]) - max_samples) / float(int(r["nc"][0]))      # each side drops about this fraction of its rays, not the same ones
        assert len(ref_slot.keys() ^ got_slot.keys()) <= 4.0 * excess * n_ref + 16, (n_ref, n_got, len(ref_slot.keys() ^ got_slot.keys()), excess)
    gns = g["ns"].reshape(-1, 2)[:n_got].astype(np.int64)
    order = np.argsort(gns[:, 1], kind="stable")
    ends = gns[order, 1] + gns[order, 0]
    assert (gns[order[1:], 1] >= ends[:-1]).all()                                                  # the kept rays' runs do not overlap ...
    if not overflow:
        assert gns[order[0], 1] == 0 and (gns[order[1:], 1] == ends[:-1]).all()                    # ... and tile [0, total)
    else:
        assert ends.max() <= max_samples
    rco, gco = r["co"].view(np.uint8).reshape(-1, 28), g["co"].view(np.uint8).reshape(-1, 28)
    rrays, grays = r["rays"].view(np.uint8).reshape(-1, 24), g["rays"].view(np.uint8).reshape(-1, 24)
    n_samples = 0
    for ray, kr in ref_slot.items():
        if ray not in got_slot:
            continue
        kg = got_slot[ray]
        nr, br = int(r["ns"][2 * kr]), int(r["ns"][2 * kr + 1])
        ng, bg = int(g["ns"][2 * kg]), int(g["ns"][2 * kg + 1])
        assert nr == ng, ray
        assert (rrays[kr] == grays[kg]).all(), ray
        assert (rco[br:br + nr] == gco[bg:bg + ng]).all(), ray
        n_samples += nr
    return n_got, int(gns[:, 0].sum())


def covered_rows(cap):
    """rows of the pre-compaction sample array that belong to a kept ray (dropped rays leave stale rows behind)"""
    n_rays = int(cap["gen_counters"][0])
    ns = cap["numsteps"].reshape(-1, 2)[:n_rays].astype(np.int64)
    mask = np.zeros(int(cap["max_inference"]), bool)
    starts, counts = ns[:, 1], ns[:, 0]
    idx = np.repeat(starts - np.cumsum(counts) + counts, counts) + np.arange(int(counts.sum()))
    mask[idx] = True
    return mask


def oracle_inference(orc, S, cap, rows=None):
    co = cap["coords"].reshape(-1, 7)
    if rows is not None:
        co = np.ascontiguousarray(co[rows])
    n = len(co)
    co = np.ascontiguousarray(np.nan_to_num(co, nan=0.0, posinf=0.0, neginf=0.0))
    out = np.zeros((n, 4), np.uint16)
    orc.orc_nerf_inference(S["desc"].ctypes.data, cap["params"].ctypes.data, co.ctypes.data, 7, n, out.ctypes.data, 4)
    return out.view(np.float16).astype(np.float32)


def oracle_loss(orc, S, cap, background_color=(0.0, 0.0, 0.0), color_space=0, random_bg=1, linear_colors=0, snap=0):
    """compute_loss on the DEVICE's network outputs and march records"""
    R, B = int(cap["n_rays_global"]), int(cap["target_batch_size"])
    n_alive = int(cap["gen_counters"][0])
    sc = S["sc"]
    n_img = len(S["xf"])
    em_res = np.array(sc["error_map_res"], np.int32)
    bg = np.array(background_color, np.float32)
    o = dict(cnt=np.zeros(1, np.uint32), ns=cap["numsteps"].copy(), co=np.zeros(B, H.COORD), dl=np.zeros((B, 4), np.float16), loss=np.zeros(R, np.float32),
             em=np.zeros(max(1, n_img * int(em_res[0]) * int(em_res[1])), np.float32))
    exposure = np.zeros((n_img, 3), np.float32)
    mlp = np.ascontiguousarray(cap["mlp_out"])
    orc.orc_compute_loss(R, S["aabb"].ctypes.data, int(cap["rng_state"]), int(cap["rng_inc"]), B, n_alive, H.f32(128.0), 4, bg.ctypes.data, color_space, random_bg, linear_colors,
                         n_img, S["md"].ctypes.data, mlp.ctypes.data, o["cnt"].ctypes.data, cap["ray_indices"].ctypes.data, cap["rays"].ctypes.data, o["ns"].ctypes.data,
                         cap["coords"].ctypes.data, o["co"].ctypes.data, o["dl"].ctypes.data, int(sc["loss_type"]), o["loss"].ctypes.data, 0, None, int(sc["rgb_activation"]),
                         int(sc["density_activation"]), snap, o["em"].ctypes.data, em_res.ctypes.data, H.f32(float(cap["density_grid_mean"][0])), exposure.ctypes.data,
                         H.f32(sc["near_distance"]), None, None, None, H.f32(0.0), 1, None)
    return o


def borderline_rays(cap):
    """ray slots whose running transmittance comes within 2e-3 (relative) of the T < 1e-4 early-out at any step (float64 replay): the oracle's
    libm expf / sequential product and the kernel's v_exp_f32 / wave prefix product may cut those one sample apart"""
    n_alive = int(cap["gen_counters"][0])
    ns = cap["numsteps"]
    sig = cap["mlp_out"].view(np.float16).reshape(-1, 4)[:, 3].astype(np.float64)
    dt_w = cap["coords"].reshape(-1, 7)[:, 3].astype(np.float64)
    out = set()
    for i in range(n_alive):
        n, b = int(ns[2 * i]), int(ns[2 * i + 1])
        dt = dt_w[b:b + n] * (MIN_STEP * 128 - MIN_STEP) + MIN_STEP     # unwarp_dt (testbed_nerf.cu:313-316): max_stepsize = MIN_CONE_STEPSIZE * 2^(NERF_CASCADES - 1) — NOT MAX_CONE_STEPSIZE
        T = np.cumprod(np.exp(-np.exp(np.minimum(sig[b:b + n], 15.0)) * dt))
        if np.any(np.abs(T - 1e-4) < 2e-7):
            out.add(i)
    return out


def describe_compaction_mismatch(cap, o_ns, slot):
    """why did ray slot `slot` keep different sample counts?  (float64 replay of its transmittance around the oracle's and the device's cut)"""
    ns = cap["numsteps"]
    n, b = int(ns[2 * slot]), int(ns[2 * slot + 1])
    sig = cap["mlp_out"].view(np.float16).reshape(-1, 4)[b:b + n, 3].astype(np.float64)
    dt = cap["coords"].reshape(-1, 7)[b:b + n, 3].astype(np.float64) * (MIN_STEP * 128 - MIN_STEP) + MIN_STEP
    T = np.cumprod(np.exp(-np.exp(np.minimum(sig, 15.0)) * dt))
    ko, kd = int(o_ns[2 * slot]), int(cap["numsteps_compacted"][2 * slot])
    lo, hi = max(0, min(ko, kd) - 3), min(n, max(ko, kd) + 2)
    return "slot %d: %d marched samples, oracle keeps %d, device keeps %d; T[%d:%d] = %s; sigma = %s" % (slot, n, ko, kd, lo, hi, ["%.4g" % t for t in T[lo:hi]], ["%.3f" % x for x in sig[lo:hi]])


def oracle_backward(orc, S, cap, n=None):
    """forward + backward of the (rolled-over) compacted batch as the device ran it; returns float64 gradients [n_params]"""
    B = int(cap["target_batch_size"]) if n is None else int(n)
    co = np.ascontiguousarray(cap["coords_compacted_rolled"].reshape(-1, 7)[:B])
    dl = np.ascontiguousarray(cap["dloss_rolled"].reshape(-1, 4)[:B])
    g = np.zeros(len(cap["params"]), np.float64)
    orc.orc_nerf_forward_backward(S["desc"].ctypes.data, cap["params"].ctypes.data, co.ctypes.data, 7, B, dl.ctypes.data, None, g.ctypes.data, None)
    return g


def timed_cpu_step(orc, S, cap, n_rays, budget_s=10.0):
    """bench.py's cpu_baseline: the oracle runs the captured step's first n_rays rays end to end — march, inference of their samples, loss +
    compaction, forward / backward of the compacted samples — repeated until budget_s seconds have passed.  Returns (compacted samples / s, details)."""
    sub = dict(cap)
    t0 = time.time()
    reps, kept, pre = 0, 0, 0
    stages = {"march": 0.0, "inference": 0.0, "loss": 0.0, "forward_backward": 0.0}
    while reps == 0 or time.time() - t0 < budget_s:
        t = time.time()
        r = oracle_march(orc, S, cap, n_rays)
        stages["march"] += time.time() - t
        n_s = min(int(r["nc"][0]), int(cap["max_inference"]))
        t = time.time()
        out = np.zeros((max(n_s, 1), 4), np.uint16)
        orc.orc_nerf_inference(S["desc"].ctypes.data, cap["params"].ctypes.data, r["co"].ctypes.data, 7, n_s, out.ctypes.data, 4)
        stages["inference"] += time.time() - t
        t = time.time()
        sub.update(R=n_rays, n_rays_global=int(cap["n_rays_global"]), gen_counters=np.array([r["rc"][0], r["nc"][0]], np.uint32), ray_indices=r["idx"], rays=r["rays"].view(np.float32),
                   numsteps=r["ns"], coords=r["co"].view(np.float32), mlp_out=out.reshape(-1))
        o = oracle_loss(orc, S, sub)
        stages["loss"] += time.time() - t
        n_c = min(int(o["cnt"][0]), int(cap["target_batch_size"]))
        t = time.time()
        g = np.zeros(len(cap["params"]), np.float64)
        if n_c:
            orc.orc_nerf_forward_backward(S["desc"].ctypes.data, cap["params"].ctypes.data, o["co"].ctypes.data, 7, n_c, o["dl"].ctypes.data, None, g.ctypes.data, None)
        stages["forward_backward"] += time.time() - t
        reps += 1
        kept += n_c
        pre += n_s
    dt = time.time() - t0
    return kept / dt, dict(reps=reps, rays=n_rays, samples=pre // reps, compacted=kept // reps, seconds=round(dt, 2), stage_seconds={k: round(v, 2) for k, v in stages.items()})
