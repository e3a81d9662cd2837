"""Data-parallel step inside the product, as far as ONE GPU can exercise it: the RCCL entry points of the C ABI on a
one-rank communicator, and Testbed.init_data_parallel with a world of one rank — the same code path as N ranks (shared-memory counter exchange,
stream-ordered RCCL all-reduce of the gradient vector, error-map / exposure-gradient reductions), weak and strong scaling.  The partitioning
contract for N > 1 (ray slices, global normalisation, identical counters) is covered on CPU by tests/test_dp_cpu.py (gloo, world size 2)."""
import os
import sys

import numpy as np
import pytest

import helpers as H
from capi import check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]
pytestmark = pytest.mark.gpu


def test_rccl_entry_points_on_a_one_rank_communicator(ngp, cuda):
    import torch
    assert ngp.ngp_rccl_available() == 1
    uid = np.zeros(128, np.uint8)
    check(ngp.ngp_rccl_get_unique_id(uid.ctypes.data))
    assert uid.any()
    comm = ngp.ngp_rccl_init(0, 1, uid.ctypes.data)
    assert comm
    try:
        g = torch.arange(4096, device=cuda, dtype=torch.float16) * 0.25
        f = torch.linspace(-1, 1, 1000, device=cuda, dtype=torch.float32)
        d = torch.tensor([1.0, 2.0, 3.5], device=cuda, dtype=torch.float64)
        want = (g.clone(), f.clone(), d.clone())
        st = torch.cuda.current_stream().cuda_stream
        check(ngp.ngp_rccl_allreduce_grads(comm, st, g.data_ptr(), g.numel()))
        check(ngp.ngp_rccl_allreduce_f32(comm, st, f.data_ptr(), f.numel()))
        check(ngp.ngp_rccl_allreduce_counters(comm, st, d.data_ptr(), d.numel()))
        torch.cuda.synchronize()
        assert torch.equal(g, want[0]) and torch.equal(f, want[1]) and torch.equal(d, want[2])   # sum over one rank
        assert ngp.ngp_rccl_init(2, 2, uid.ctypes.data) is None                                    # rank out of range
    finally:
        check(ngp.ngp_rccl_finalize(comm))


@pytest.mark.parametrize("strong", [False, True])
def test_testbed_data_parallel_step_with_one_rank(cuda, strong):
    import scene
    ds = scene.make_dataset(n_train=8, n_test=1, res=64, device=cuda)
    a = scene.build_testbed(ds)
    b = scene.build_testbed(ds)
    b.init_data_parallel(0, 1, "t_%d_%d" % (os.getpid(), int(strong)), strong)
    assert b.world_size == 1 and b.rank == 0 and b.strong_scaling == strong
    b.nerf.training.optimize_exposure = True          # its gradients go through ngp_rccl_allreduce_f32 in the data-parallel step
    b.nerf.training.n_steps_between_error_map_updates = 16   # ... and so does the error map at every CDF rebuild
    b.nerf.training.optimize_extrinsics = True        # ... and the per-image camera gradients at every camera update
    scene.train(a, 80)
    scene.train(b, 80)
    pos, rot, it = b.nerf.training._cam_offsets()
    assert (it == 5).all() and np.isfinite(pos).all() and np.abs(pos).max() > 0 and np.abs(rot).max() > 0
    assert a.training_step == b.training_step == 80
    assert b.nerf.training.is_cdf_valid
    assert np.isfinite(b.loss) and 0.4 < a.loss / b.loss < 2.5
    assert 0.6 < a.nerf.training.rays_per_batch / b.nerf.training.rays_per_batch < 1.67
    assert b.nerf.training.measured_batch_size > 0
    b.shutdown_data_parallel()
    scene.train(b, 90)                                 # back on the single-GPU step
    assert b.training_step == 90
