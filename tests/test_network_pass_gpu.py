"""GPU tests of the measured choice between the two organisations of the network pass (Testbed.network_pass): the reference dispatches ONE
inference_mixed_precision for every scene (src/testbed_nerf.cu:3256, nerf_network.h:103-137); this build owns two organisations of that pass with identical
results — fused (gathers inside the MLP kernel) and two_kernel (XCD-affine encode into level planes + MLP kernel) — and runs whichever it measured faster on the
workload at hand.  What must hold: (1) whatever the host chose, a training step's network pass produces the bits of the kernel library's entry points on the
step's own inputs — both organisations, replayed through the C ABI; (2) 'auto' calibrates by itself, reports what it measured, and a reset starts over;
(3) the plumbing modes (grid -> MLP: image, SDF) go through the same policy."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]

import capi  # noqa: E402
import helpers as H  # noqa: E402

pytestmark = pytest.mark.gpu
check = capi.check


def _testbed(cuda, n_train=8, res=64):
    import scene
    ds = scene.make_dataset(n_train=n_train, n_test=1, res=res, device=cuda)
    return ds, scene.build_testbed(ds)


@pytest.mark.parametrize("forced", ["fused", "two_kernel"])
def test_a_step_under_either_organisation_carries_the_library_bits(ngp, cuda, forced):
    ds, tb = _testbed(cuda)
    tb.network_pass = forced
    assert tb.network_pass == forced
    for _ in range(20):                       # past the first occupancy-grid updates: a few thousand samples per step
        tb.frame()
    tb.debug_capture_next_step()
    tb.frame()
    cap = tb.debug_captured()
    rep = tb.network_pass_report
    assert rep["policy"] == "forced" and rep["running"] == forced and rep["calibrations"] == 0
    n = int(cap["max_inference"])
    n_valid = min(int(cap["gen_counters"][1]), n)
    assert n_valid > 1000
    desc = np.frombuffer(tb.debug_scene()["desc"].tobytes(), dtype=H.NET_DESC).copy()
    d_desc, d_p, d_c = H.to_dev(desc, cuda), H.to_dev(np.ascontiguousarray(cap["params"]), cuda), H.to_dev(np.ascontiguousarray(cap["coords"]), cuda)
    got_step = np.asarray(cap["mlp_out"]).reshape(n, 4)
    # replay through BOTH entry points of the C ABI: the step's outputs are theirs, whichever the host launched
    out_f, xs_f = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    check(ngp.ngp_hip_nerf_forward(None, d_desc.data_ptr(), d_p.data_ptr(), d_c.data_ptr(), 7, n, out_f.data_ptr(), 4, xs_f.data_ptr(), None))
    ws_bytes = int(ngp.ngp_hip_nerf_encode_workspace_bytes(n))
    ws = H.dev_zeros(ws_bytes, cuda)
    out_w, xs_w = H.dev_zeros(n * 8, cuda), H.dev_zeros(n * 64, cuda)
    check(ngp.ngp_hip_nerf_forward_ws(None, d_desc.data_ptr(), d_p.data_ptr(), d_c.data_ptr(), 7, n, out_w.data_ptr(), 4, xs_w.data_ptr(), ws.data_ptr(), ws_bytes, None))
    f, w = H.to_host(out_f, np.uint16).reshape(n, 4), H.to_host(out_w, np.uint16).reshape(n, 4)
    np.testing.assert_array_equal(f[:n_valid], w[:n_valid])
    np.testing.assert_array_equal(H.to_host(xs_f, np.uint16).reshape(n, 32)[:n_valid], H.to_host(xs_w, np.uint16).reshape(n, 32)[:n_valid])
    np.testing.assert_array_equal(got_step[:n_valid], f[:n_valid])
    assert (got_step[:n_valid].view(np.float16).astype(np.float32) != 0).any()


def test_auto_calibrates_reports_and_starts_over_after_a_reset(cuda):
    ds, tb = _testbed(cuda)
    assert tb.network_pass == "auto"
    rep = tb.network_pass_report
    assert rep["policy"] == "auto" and rep["calibrations"] == 0 and rep["running"] == "fused"
    tb.async_training_steps = True
    for _ in range(48 + 12 + 4):              # first calibration: steps 48 .. 59 (6 launches per organisation), decided when their events have finished
        tb.frame()
    tb.sync()
    tb.frame()
    rep = tb.network_pass_report
    assert rep["calibrations"] == 1 and rep["last_calibration_step"] == 48, rep
    assert rep["fused_us"] > 0 and rep["two_kernel_us"] > 0 and rep["running"] in ("fused", "two_kernel")
    slower, faster = max(rep["fused_us"], rep["two_kernel_us"]), min(rep["fused_us"], rep["two_kernel_us"])
    if faster < 0.97 * slower:                # the decision follows the measurement (3 % hysteresis in favour of what was running: fused)
        assert rep["running"] == ("fused" if rep["fused_us"] < rep["two_kernel_us"] else "two_kernel")
    else:
        assert rep["running"] == "fused"
    assert np.isfinite(tb.loss)
    with pytest.raises(RuntimeError):
        tb.network_pass = "sometimes"
    tb.reset(True)                            # a new network: what was measured on the old one is forgotten
    rep = tb.network_pass_report
    assert rep["calibrations"] == 0 and rep["running"] == "fused"
    for _ in range(4):
        tb.frame()
    assert tb.training_step == 4


@pytest.mark.parametrize("mode", ["image", "sdf"])
def test_plumbing_modes_go_through_the_same_policy(cuda, mode):
    import pyngp
    cfg = os.path.join(ROOT, "blender-ngp_amd", "configs", mode, "base.json")
    rs = np.random.RandomState(3)

    def make():
        tb = pyngp.Testbed(pyngp.TestbedMode.Image if mode == "image" else pyngp.TestbedMode.Sdf)
        if mode == "image":
            yy, xx = np.mgrid[0:96, 0:128].astype(np.float32)
            img = np.stack([0.5 + 0.5 * np.sin(xx / 9.0), 0.5 + 0.5 * np.cos(yy / 7.0), (xx + yy) / 224.0, np.ones_like(xx)], -1).astype(np.float32)
            tb.set_image_data(np.ascontiguousarray(img))
        else:
            pts = np.random.RandomState(0).rand(1 << 16, 3).astype(np.float32)
            tb.override_sdf_training_data(pts, (np.linalg.norm(pts - 0.5, axis=1) - 0.3).astype(np.float32))
        tb.reload_network_from_file(cfg)
        tb.shall_train = True
        return tb

    losses = {}
    for org in ("fused", "two_kernel", "auto"):
        tb = make()
        tb.network_pass = org
        for _ in range(33):
            tb.train(1 << 12)
        losses[org] = float(tb.loss)
        rep = tb.network_pass_report
        if org == "auto":
            assert rep["calibrations"] >= 1 and rep["last_calibration_step"] == 8 and rep["fused_us"] > 0 and rep["two_kernel_us"] > 0, rep
        else:
            assert rep["running"] == org and rep["calibrations"] == 0
        del tb
    # the same model whichever organisation ran (identical forward bits; the backward's partial sums are the same launches): the loss of step 32 agrees closely
    assert np.isfinite(list(losses.values())).all()
    assert max(losses.values()) - min(losses.values()) <= 0.02 * max(abs(v) for v in losses.values()) + 1e-7, losses


def test_backward_over_the_live_samples_inside_a_training_step(ngp, cuda):
    """`compact_backward` (opt-in): the step lists the samples of its compacted batch whose loss gradient is not zero in all four channels and runs the backward pass over
    those (ngp_hip_compact_live_samples + ngp_hip_nerf_backward_live).  A captured step's gradient against the whole-batch entry point on the step's own buffers:
    hash-grid part bit for bit, MLP part within 2e-3 of the largest weight gradient; a good part of the batch is dead (ray tails), and training goes on."""
    ds, tb = _testbed(cuda, n_train=12, res=96)
    tb.compact_backward = True
    for _ in range(60):
        tb.frame()
    tb.debug_capture_next_step()
    tb.frame()
    cap = tb.debug_captured()
    frac = float(tb.backward_live_fraction)
    n = int(cap["target_batch_size"])
    dl = np.ascontiguousarray(cap["dloss_rolled"]).reshape(n, 4)
    live = int(((dl & 0x7fff) != 0).any(axis=1).sum())
    assert 0 < live < n and abs(frac - live / n) < 1e-6, (frac, live, n)
    desc = np.frombuffer(tb.debug_scene()["desc"].tobytes(), dtype=H.NET_DESC).copy()
    d_desc, d_p = H.to_dev(desc, cuda), H.to_dev(np.ascontiguousarray(cap["params"]), cuda)
    d_c, d_x, d_dl = H.to_dev(np.ascontiguousarray(cap["coords_compacted_rolled"]), cuda), H.to_dev(np.ascontiguousarray(cap["x_saved"]), cuda), H.to_dev(dl, cuda)
    sb = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    scratch, grads = H.dev_zeros(sb, cuda), H.dev_zeros(H.n_params(desc) * 2, cuda)
    check(ngp.ngp_hip_nerf_backward(None, d_desc.data_ptr(), None, d_p.data_ptr(), d_c.data_ptr(), 7, n, d_x.data_ptr(), d_dl.data_ptr(), 4, grads.data_ptr(), scratch.data_ptr(), sb, None, None, None, None))
    want, got = H.to_host(grads, np.uint16), np.asarray(cap["grads"])
    np.testing.assert_array_equal(got[10240:], want[10240:])
    a, b = got[:10240].view(np.float16).astype(np.float32), want[:10240].view(np.float16).astype(np.float32)
    assert np.abs(a - b).max() <= 2e-3 * np.abs(b).max() + 1e-7 and np.abs(b).max() > 0
    for _ in range(10):
        tb.frame()
    tb.sync()
    assert np.isfinite(tb.loss) and tb.training_step == 71
    print("live samples of the captured step: %.1f %% of %d" % (100.0 * frac, n))
