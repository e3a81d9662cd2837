"""nerf.training.optimize_extrinsics (python_api.cu:811; testbed_nerf.cu:1600-1712, 2598-2633, 3056-3093) through the pyngp boundary:
the camera gradients of a product-path training step against the oracle's replay of that step, the update applied to the transforms, and
a registration run that pulls perturbed cameras back onto a trained scene."""
import os
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import fullstep as F
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]
pytestmark = pytest.mark.gpu


def _host_images(ds):
    return [np.ascontiguousarray(x.cpu().numpy() if hasattr(x, "cpu") else x) for x in ds["train_images"]]


def test_camera_gradients_of_a_training_step_match_the_oracle_and_move_the_transforms(cuda, oracle):
    import scene
    ds = scene.make_dataset(n_train=8, n_test=1, res=48, device=cuda)
    t = scene.build_testbed(ds)
    t.training_batch_size = 1 << 14
    tr = t.nerf.training
    scene.train(t, 40)
    before = np.array([tr.get_camera_extrinsics(i) for i in range(8)])
    pos0, rot0, it0 = tr._cam_offsets()
    assert (pos0 == 0).all() and (rot0 == 0).all() and (it0 == 0).all()
    tr.optimize_extrinsics = True
    tr.n_steps_between_cam_updates = 1
    xf0 = np.array([np.asarray(a) for a, _ in tr.transforms])                 # [n][3][4] start matrices (ngp frame) the captured step trains with
    t.debug_capture_next_step()
    t.frame()
    cap = t.debug_captured()
    S = F.host_scene(t, _host_images(ds))
    B = int(cap["target_batch_size"])
    n_alive, n_kept = int(cap["gen_counters"][0]), min(int(cap["measured_batch_size"]), B)
    assert B == 1 << 14 and n_alive > 100 and n_kept > 1000

    # ---- the network's input gradient over the compacted batch (teacher-forced on the device's rolled-over batch)
    co = np.ascontiguousarray(cap["coords_compacted_rolled"].reshape(-1, 7)[:n_kept])
    dl = np.ascontiguousarray(cap["dloss_rolled"].reshape(-1, 4)[:n_kept])
    cg_ref = np.zeros((n_kept, 6), np.float32)
    oracle.orc_nerf_input_gradient(S["desc"].ctypes.data, cap["params"].ctypes.data, co.ctypes.data, 7, n_kept, dl.ctypes.data, cg_ref.ctypes.data)
    cg_dev = np.ascontiguousarray(cap["coords_gradient"].reshape(-1, 6)[:n_kept])
    for name, sl in (("pos", slice(0, 3)), ("dir", slice(3, 6))):
        rel = np.linalg.norm(cg_dev[:, sl].astype(np.float64) - cg_ref[:, sl]) / np.linalg.norm(cg_ref[:, sl].astype(np.float64))
        assert rel < 3e-2, (name, rel)

    # ---- compute_cam_gradient_train_nerf on the DEVICE's input gradient (stage parity) and on the oracle's (end to end)
    got_pos, got_rot = tr._cam_gradients()
    assert got_pos.shape == (8, 3) and np.abs(got_pos).max() > 0 and np.abs(got_rot).max() > 0
    ns = np.ascontiguousarray(cap["numsteps_compacted"])
    ns_fit = ns.copy()
    for i in range(n_alive):                                                    # a ray whose samples got no room was given numsteps 0 by the loss kernel already
        assert int(ns[2 * i]) == 0 or int(ns[2 * i + 1]) + int(ns[2 * i]) <= B
    md = S["md"]
    for cg, tol in ((cg_dev, 2e-4), (cg_ref, 3e-2)):
        ref_pos, ref_rot = np.zeros((8, 3), np.float32), np.zeros((8, 3), np.float32)
        full = np.zeros((B, 6), np.float32); full[:n_kept] = cg
        oracle.orc_compute_cam_gradient(int(cap["n_rays_global"]), S["aabb"].ctypes.data, int(cap["rng_state"]), int(cap["rng_inc"]), n_alive, 0, ref_pos.ctypes.data, ref_rot.ctypes.data, 8,
                                        md.ctypes.data, cap["ray_indices"].ctypes.data, np.ascontiguousarray(cap["rays"]).ctypes.data, ns_fit.ctypes.data,
                                        np.ascontiguousarray(cap["coords_compacted_rolled"]).ctypes.data, full.ctypes.data, None)
        for got, ref in ((got_pos, ref_pos), (got_rot, ref_rot)):
            np.testing.assert_allclose(got, ref, rtol=0, atol=tol * np.abs(ref).max())

    # ---- the update (3063-3093): first Adam step = -lr * sign(g) per component (bias-corrected m / sqrt(v) = +-1), applied by update_transforms
    pos1, rot1, it1 = tr._cam_offsets()
    assert (it1 == 1).all()
    lr = max(tr.extrinsic_learning_rate, t.learning_rate / 1000.0)            # 3076: 0.33^(0 / 128) = 1
    scale = 8 / 128.0 / 1.0                                                    # per_camera_loss_scale = n_images / LOSS_SCALE / n_steps_between_cam_updates (3061)
    g = got_pos * np.float32(scale)
    strong = np.abs(g) > 1e-5                                                    # eps = 1e-8 in the denominator: tiny gradients move less than lr (1e-6 was seen 2.6 % short, one run in twenty)
    np.testing.assert_allclose(pos1[strong], (-lr * np.sign(g))[strong], rtol=2e-2)
    after = np.array([tr.get_camera_extrinsics(i) for i in range(8)])
    assert np.abs(after - before).max() > 1e-4 and np.abs(after - before).max() < 0.02
    # get_camera_extrinsics returns NeRF-convention matrices: compare in the ngp frame through the transforms list
    xf = np.array([np.asarray(a) for a, _ in tr.transforms])
    for i in range(8):
        m0, m1 = xf0[i], xf[i]
        np.testing.assert_allclose(m1[:, 3] - m0[:, 3], pos1[i], atol=2e-6)
        np.testing.assert_allclose(m1[:, :3], Rotation.from_rotvec(rot1[i].astype(np.float64)).as_matrix() @ m0[:, :3], atol=3e-6)
    # training goes on with the moved cameras; switching the optimisation off keeps them where they are
    tr.n_steps_between_cam_updates = 16
    scene.train(t, 80)
    pos2, rot2, it2 = tr._cam_offsets()
    assert (it2 >= 2).all() and np.isfinite(t.loss) and np.isfinite(pos2).all() and np.isfinite(rot2).all()
    tr.optimize_extrinsics = False
    scene.train(t, 100)
    pos3, _, it3 = tr._cam_offsets()
    assert (it3 == it2).all() and (pos3 == pos2).all()
    # reset_camera_extrinsics + update through set_camera_extrinsics: back to the dataset pose
    tr.set_camera_extrinsics(0, ds["train_poses"][0][:3, :], True)
    np.testing.assert_allclose(tr.get_camera_extrinsics(0), before[0], atol=1e-6)


def test_focal_length_and_extra_dims_switches_train_nothing_on_a_plain_dataset(cuda):
    import scene
    ds = scene.make_dataset(n_train=6, n_test=1, res=32, device=cuda)
    a, b = scene.build_testbed(ds), scene.build_testbed(ds)
    b.nerf.training.optimize_focal_length = True                                 # the reference's kernel never writes that gradient: same training
    scene.train(a, 20); scene.train(b, 20)
    # (two runs of the same configuration drift apart themselves: the hash-grid gradients are summed with fp16 atomics and Adam's first steps are sign-like)
    assert abs(a.loss - b.loss) < 0.1 * a.loss and a.training_step == b.training_step == 20
    # optimize_extra_dims without latent codes in the dataset trains nothing either (testbed_nerf.cu:2925: n_extra_learnable_dims > 0 && optimize_extra_dims);
    # with them: tests/test_netx_e2e_gpu.py
    c = scene.build_testbed(ds)
    c.nerf.training.optimize_extra_dims = True
    scene.train(c, 20)
    assert c.training_step == 20 and abs(a.loss - c.loss) < 0.1 * a.loss


def test_registration_pulls_perturbed_cameras_back(cuda):
    """Train on the true poses, then displace a third of the cameras and let ONLY the extrinsics train (weights frozen through shall_train_network /
    shall_train_encoding): the displaced cameras move back towards where the scene says they are, the others stay put."""
    import scene
    n = 24
    ds = scene.make_dataset(n_train=n, n_test=1, res=64, device=cuda)
    t = scene.build_testbed(ds)
    tr = t.nerf.training
    scene.train(t, 1200)
    true_pos = np.array([np.asarray(a)[:, 3] for a, _ in tr.transforms])
    true_rot = np.array([np.asarray(a)[:, :3] for a, _ in tr.transforms])
    rs = np.random.RandomState(4)
    moved = np.arange(0, n, 3)
    for i in moved:
        m = np.array(ds["train_poses"][i], np.float64)
        m[:3, 3] += rs.randn(3) / np.sqrt(3) * 0.06                             # NeRF units; x 0.33 dataset scale = ~0.02 of the unit cube
        m[:3, :3] = Rotation.from_rotvec(rs.randn(3) / np.sqrt(3) * np.deg2rad(1.0)).as_matrix() @ m[:3, :3]
        tr.set_camera_extrinsics(int(i), m[:3, :].astype(np.float32), True)

    def errors():
        p = np.array([np.asarray(a)[:, 3] for a, _ in tr.transforms])
        r = np.array([np.asarray(a)[:, :3] for a, _ in tr.transforms])
        ang = np.array([np.linalg.norm(Rotation.from_matrix(r[i] @ true_rot[i].T).as_rotvec()) for i in range(n)])
        return np.linalg.norm(p - true_pos, axis=1), np.rad2deg(ang)

    e0_pos, e0_rot = errors()
    assert e0_pos[moved].min() > 0.005 and np.delete(e0_pos, moved).max() < 1e-6
    t.shall_train_network = False
    t.shall_train_encoding = False
    tr.optimize_extrinsics = True
    hist = []
    for stop in (1400, 1800, 2200, 2800):
        scene.train(t, stop)
        ep, er = errors()
        hist.append((stop, float(ep[moved].mean()), float(er[moved].mean()), float(np.delete(ep, moved).max())))
    print("registration: (step, mean position error of the displaced cameras, mean rotation error [deg], max drift of the others)", [(0, float(e0_pos[moved].mean()), float(e0_rot[moved].mean()), 0.0)] + hist)
    e1_pos, e1_rot = errors()
    assert e1_pos[moved].mean() < 0.7 * e0_pos[moved].mean() and hist[-1][1] < hist[0][1], hist   # measured: 0.0170 -> 0.0139 (200 steps) -> 0.0101 (800)
    assert e1_rot[moved].mean() < 0.75 * e0_rot[moved].mean(), hist                              # measured: 0.78 deg -> 0.48 deg (800 steps)
    assert np.delete(e1_pos, moved).max() < 0.4 * e0_pos[moved].mean(), hist                          # measured: 0.003-0.004 (the frozen scene is only so sharp at 64 x 64)
