"""One rank of a several-ranks-on-ONE-GPU run of the product's data-parallel path (tests/test_dp_loopback_gpu.py starts `world` of these).

The nccl* names behind blender-ngp_amd/csrc/comm.hip are served by tests/loopback/librccl_loopback.so (NGP_RCCL_LIBRARY, set by the test); everything else is the
product binary: Testbed.init_data_parallel, frame(), optimizer_step_sharded, dp_gather_*, the row-sharded render.  The worker writes what it saw to
<out>/rank<r>.npz (+ a snapshot) and the test compares the ranks with each other and with a single-rank run.

usage: dp_worker.py <rank> <world> <key> <out_dir> <options json>
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]


def _wait_for(path, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise RuntimeError("timed out waiting for %s" % path)
        time.sleep(0.01)


def abi_collectives(ngp, capi, torch, rank, world, out_dir):
    """every ngp_rccl_* entry point the step uses, on rank-dependent data with closed-form results"""
    dev = torch.device("cuda:0")
    uid = np.zeros(128, np.uint8)
    uid_path = os.path.join(out_dir, "abi_uid.bin")
    if rank == 0:
        capi.check(ngp.ngp_rccl_get_unique_id(uid.ctypes.data))
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid.tobytes())
        os.replace(uid_path + ".tmp", uid_path)
    else:
        _wait_for(uid_path)
        uid = np.frombuffer(open(uid_path, "rb").read(), np.uint8).copy()
    comm = ngp.ngp_rccl_init(rank, world, uid.ctypes.data)
    assert comm, ngp.ngp_hip_last_error()
    st = torch.cuda.current_stream().cuda_stream
    try:
        assert ngp.ngp_rccl_comm_size(comm) == world and ngp.ngp_rccl_comm_rank(comm) == rank
        n = 40_000
        base = torch.arange(n, device=dev, dtype=torch.float32)
        tri = world * (world - 1) / 2
        # all-reduce, three dtypes: rank r contributes (i % 64) * 0.25 + r  ->  world * (i % 64) * 0.25 + tri  (exact in fp16)
        g = ((base % 64) * 0.25 + rank).to(torch.float16)
        capi.check(ngp.ngp_rccl_allreduce_grads(comm, st, g.data_ptr(), n))
        f = base * 0.5 + rank
        capi.check(ngp.ngp_rccl_allreduce_f32(comm, st, f.data_ptr(), n))
        d = torch.tensor([1.0 + rank, 2.0, 3.5 * (rank + 1)], device=dev, dtype=torch.float64)
        capi.check(ngp.ngp_rccl_allreduce_counters(comm, st, d.data_ptr(), 3))
        torch.cuda.synchronize()
        assert torch.equal(g.float(), world * (base % 64) * 0.25 + tri)
        assert torch.equal(f, world * base * 0.5 + tri)
        assert d.tolist() == [world + tri, 2.0 * world, 3.5 * (tri + world)]
        # in-place all-gather: rank r's chunk = r * 1000 + i
        per = 5_000
        buf32 = torch.full((world * per,), -1.0, device=dev)
        buf32[rank * per:(rank + 1) * per] = rank * 1000.0 + base[:per] * 0.125
        capi.check(ngp.ngp_rccl_allgather_f32(comm, st, buf32.data_ptr(), per))
        buf16 = torch.full((world * per,), -1.0, device=dev, dtype=torch.float16)
        buf16[rank * per:(rank + 1) * per] = (rank * 8.0 + (base[:per] % 32)).to(torch.float16)
        capi.check(ngp.ngp_rccl_allgather_f16(comm, st, buf16.data_ptr(), per))
        torch.cuda.synchronize()
        for q in range(world):
            assert torch.equal(buf32[q * per:(q + 1) * per], q * 1000.0 + base[:per] * 0.125)
            assert torch.equal(buf16[q * per:(q + 1) * per].float(), q * 8.0 + (base[:per] % 32))
        # reduce-scatter: rank r contributes in[j] = j + r  ->  rank q receives world * j + tri for j in its slice
        src = torch.arange(world * per, device=dev, dtype=torch.float32) + rank
        dst = torch.zeros(per, device=dev)
        capi.check(ngp.ngp_rccl_reduce_scatter_f32(comm, st, src.data_ptr(), dst.data_ptr(), per))
        torch.cuda.synchronize()
        assert torch.equal(dst, world * torch.arange(rank * per, (rank + 1) * per, device=dev, dtype=torch.float32) + tri)
        # all-to-all of fp16 slices: rank r sends slice q = (r * 16 + q) + (i % 8) * 0.5 to rank q; slice q of recv then holds rank q's slice r
        send = torch.empty(world * per, device=dev, dtype=torch.float16)
        for q in range(world):
            send[q * per:(q + 1) * per] = (rank * 16.0 + q + (base[:per] % 8) * 0.5).to(torch.float16)
        recv = torch.full((world * per,), -1.0, device=dev, dtype=torch.float16)
        capi.check(ngp.ngp_rccl_alltoall_f16(comm, st, send.data_ptr(), recv.data_ptr(), per))
        torch.cuda.synchronize()
        for q in range(world):
            assert torch.equal(recv[q * per:(q + 1) * per].float(), q * 16.0 + rank + (base[:per] % 8) * 0.5)
    finally:
        capi.check(ngp.ngp_rccl_finalize(comm))


def main():
    rank, world, key, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    opt = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
    steps = int(opt.get("steps", 50))
    strong = bool(opt.get("strong", True))
    import torch
    import capi
    import scene
    assert os.environ.get("NGP_RCCL_LIBRARY", "").endswith("librccl_loopback.so"), "the loopback library must be chosen by the test (NGP_RCCL_LIBRARY)"
    ngp = capi.load_ngp_hip()
    dev = torch.device("cuda:0")
    if opt.get("abi", False):
        abi_collectives(ngp, capi, torch, rank, world, out_dir)
    ds = scene.make_dataset(n_train=8, n_test=1, res=64, device=dev)
    tb = scene.build_testbed(ds)
    tb.network_pass = "fused"            # (the auto tuner chooses by timing: pinned, so that the ranks and the single-rank run execute the same kernels)
    for k, v in opt.get("testbed", {}).items():
        setattr(tb, k, v)
    first = 1
    if opt.get("snapshot"):               # every rank (and the single-rank run of the test) starts from the same file
        tb.load_snapshot(opt["snapshot"])
        tb.shall_train = True
        first = tb.training_step + 1
    if strong and (1 << 18) % (256 * world):
        tb.training_batch_size = 256 * world * ((1 << 18) // (256 * world))   # strong scaling splits the batch evenly: 2^18 rounded down to a multiple of 256 x world
    tb.init_data_parallel(rank, world, key, strong)
    assert tb.dp_comm_size == world and tb.rank == rank and tb.world_size == world
    res = {}
    scene.train(tb, first)
    res["params_step1"] = tb.debug_params("training")
    res["measured_step1"] = np.array([tb.nerf.training.measured_batch_size, tb.nerf.training.measured_batch_size_before_compaction, tb.nerf.training.rays_per_batch], np.int64)
    res["loss_step1"] = np.float64(tb.loss)
    scene.train(tb, steps)
    res["params"] = tb.debug_params("training")
    res["loss"] = np.float64(tb.loss)
    res["rays_per_batch"] = np.int64(tb.nerf.training.rays_per_batch)
    res["stale"] = np.array([tb.dp_state_stale, tb.dp_inference_stale])
    # a rank-local read of the inference weights must refuse a stale copy (sharded Ema), not return old numbers
    refused = False
    try:
        tb.debug_params("inference")
    except RuntimeError:
        refused = True
    res["refused_stale_inference"] = np.bool_(refused)
    # the collective render: rows per rank + all-gather, gathers the inference weights by itself
    tb.shall_train = False
    tb.background_color = [0.0, 0.0, 0.0, 0.0]
    tb.snap_to_pixel_centers = True
    tb.fov_axis = 0
    tb.fov = ds["camera_angle_x"] * 180 / np.pi
    tb.set_nerf_camera_matrix(ds["test_poses"][0][:3, :])
    tb.render_sharded = True
    res["frame_sharded"] = tb.render(56, 45, 1, True)        # 45 rows: uneven over 2 and 3 ranks
    assert not tb.dp_inference_stale
    tb.render_sharded = False
    res["frame_local"] = tb.render(56, 45, 1, True)          # the same frame traced by this rank alone
    res["inference"] = tb.debug_params("inference")
    tb.dp_gather_optimizer_state()                            # collective
    assert not tb.dp_state_stale
    tb.save_snapshot(os.path.join(out_dir, "rank%d.msgpack" % rank), True)
    tb.shutdown_data_parallel()
    tb.set_distributed(0, 1)
    tb.shall_train = True
    scene.train(tb, steps + 2)                                # whole state again: the single-GPU step accepts it
    res["loss_after"] = np.float64(tb.loss)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    print("rank %d of %d done: loss %.5f -> %.5f" % (rank, world, float(res["loss"]), float(res["loss_after"])), flush=True)


if __name__ == "__main__":
    main()
