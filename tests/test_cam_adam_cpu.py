"""Host-side camera optimizers (blender-ngp_amd/host/cam_adam.h) against an independent float64 restatement of
include/neural-graphics-primitives/adam_optimizer.h:23-162 written with scipy's rotations (the reference composes with Eigen, an absent submodule).

fp32 host arithmetic vs float64: the position iterates agree to 1e-6 relative; the rotation iterates go through matrix -> quaternion -> angle-axis
round trips whose fp32 error is ~1e-7 of a radian per step on offsets of 1e-3..1e-1 rad."""
import os
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]



@pytest.fixture(scope="module")
def pyngp():
    import torch  # noqa: F401  (one HIP runtime per process: torch's first, see capi.load_ngp_hip)
    return pytest.importorskip("pyngp")


B1, B2, EPS = np.float64(np.float32(0.9)), np.float64(np.float32(0.99)), np.float64(np.float32(1e-8))


def _adam_f64(grads, lrs, rotation):
    m = np.zeros(3); v = np.zeros(3); x = np.zeros(3)
    out = []
    for it, (g, lr) in enumerate(zip(grads.astype(np.float64), lrs.astype(np.float64)), start=1):
        alr = lr * np.sqrt(1 - B2 ** it) / (1 - B1 ** it)
        m = B1 * m + (1 - B1) * g
        v = B2 * v + (1 - B2) * g * g
        upd = alr * m / (np.sqrt(v) + EPS)
        if rotation:   # variable <- log( exp(-upd) * exp(variable) )
            x = (Rotation.from_rotvec(-upd) * Rotation.from_rotvec(x)).as_rotvec()
        else:
            x = x - upd
        out.append(x.copy())
    return np.array(out)


@pytest.mark.parametrize("rotation", [False, True])
def test_adam_iterates(pyngp, rotation):
    rs = np.random.RandomState(3)
    n = 400
    grads = (rs.randn(n, 3) * np.array([1.0, 0.1, 10.0]) + np.array([0.5, -0.02, 0.0])).astype(np.float32)
    grads[7] = 0.0                                                         # a zero gradient in the middle
    lrs = np.maximum(1e-3 * 0.33 ** (np.arange(n) // 128), 1e-5).astype(np.float32)   # the schedule of testbed_nerf.cu:3076-3077
    got = (pyngp._rotation_adam_steps if rotation else pyngp._vec3_adam_steps)(grads, lrs)
    ref = _adam_f64(grads, lrs, rotation)
    assert np.abs(ref).max() > 1e-2
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6)


def test_adam_zero_gradient_from_zero_state_stays_zero(pyngp):
    """optimize_focal_length in the reference: zero gradient on a zero variable -> 0 / (0 + eps) = 0, the variable never moves (testbed_nerf.cu:3095-3101)."""
    z = np.zeros((20, 3), np.float32)
    lrs = np.full(20, 1e-3, np.float32)
    assert (pyngp._vec3_adam_steps(z, lrs) == 0).all()
    assert (pyngp._rotation_adam_steps(z, lrs) == 0).all()


@pytest.mark.parametrize("aa", [(0.3, -0.2, 0.1), (0.0, 0.0, 1e-4), (2.0, 2.0, 0.5), (0.0, 3.1, 0.0), (0, 0, 0)])
def test_angle_axis_matrix_round_trip(pyngp, aa):
    aa = np.array(aa, np.float32)
    mat, back = pyngp._angle_axis_round_trip(aa)
    np.testing.assert_allclose(mat, Rotation.from_rotvec(aa.astype(np.float64)).as_matrix(), atol=3e-7)
    np.testing.assert_allclose(back, aa, atol=5e-4 if np.linalg.norm(aa) > 3 else 2e-6)   # near pi the quaternion's w ~ 0: fp32 conditioning
