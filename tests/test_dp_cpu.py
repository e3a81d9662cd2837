"""CPU tests (gloo, world_size 2) of the data-parallel path.

No GPU here, so the per-rank compute is done by the oracle; what is under test is the PARTITIONING CONTRACT that bench.py /
Testbed.train_nerf_dp_* rely on:
  * rank r marches global rays [r*R, (r+1)*R) of a step of W*R rays (ray_offset / n_rays_global arguments),
  * the loss is normalised by the GLOBAL ray count, so all_reduce(SUM) of the rank gradients == the full-batch gradient,
  * counters and the loss scalar are summed with the same collective, every rank ends up with identical values,
and the control flow of bench.dp_step (prep schedule, begin -> all-reduce -> end) with a recording fake Testbed.
"""
import os
import socket
import sys

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(orc, ngp):
    n_img, w, h = 4, 48, 32
    imgs = H.make_images(n_img, w, h, masked_fraction=0.0)
    md = H.make_metadata([imgs[i].ctypes.data for i in range(n_img)], w, h, 55.0)
    xf = H.hemisphere_cameras(n_img)
    grid = H.blob_density_grid(1)
    bf, mean = H.oracle_bitfield(orc, grid, 1)
    desc = H.make_desc(ngp, log2_hashmap_size=10)
    params = H.random_params(desc, seed=1, grid_amp=0.5)
    return dict(imgs=imgs, md=md, xf=xf, bf=bf, mean=mean, desc=desc, params=params, n_img=n_img)


def _rank_gradient(orc, S, n_rays_local, ray_offset, n_rays_global):
    """one rank's share of a training step, oracle only: march -> inference -> loss (global normalisation) -> backward"""
    aabb = H.unit_aabb()
    st, inc = H.pcg32_state(1337)
    max_samples = n_rays_local * 256
    r = dict(rc=np.zeros(1, np.uint32), nc=np.zeros(1, np.uint32), idx=np.zeros(n_rays_local, np.uint32), rays=np.zeros(n_rays_local, H.RAY),
             ns=np.zeros(n_rays_local * 2, np.uint32), co=np.zeros(max_samples, H.COORD))
    dres = np.array([0, 0], np.int32)
    orc.orc_generate_training_samples(n_rays_local, aabb.ctypes.data, max_samples, st, inc, r["rc"].ctypes.data, r["nc"].ctypes.data, r["idx"].ctypes.data, r["rays"].ctypes.data,
                                      r["ns"].ctypes.data, r["co"].ctypes.data, S["n_img"], S["md"].ctypes.data, S["xf"].ctypes.data, S["bf"].ctypes.data, 0, None, 0, 0,
                                      H.f32(0.0), None, dres.ctypes.data, ray_offset, n_rays_global, None)
    n_alive, n_samples = int(r["rc"][0]), int(r["nc"][0])
    out = np.zeros((max(n_samples, 1), 4), np.uint16)
    orc.orc_nerf_inference(S["desc"].ctypes.data, S["params"].ctypes.data, r["co"].ctypes.data, 7, n_samples, out.ctypes.data, 4)
    B = n_samples + 8
    cnt = np.zeros(1, np.uint32)
    co_c, dl, loss = np.zeros(B, H.COORD), np.zeros((B, 4), np.float16), np.zeros(n_rays_global, np.float32)
    bg = np.zeros(3, np.float32)
    em_res = np.array([8, 8], np.int32)
    em = np.zeros(S["n_img"] * 64, np.float32)
    exposure = np.zeros((S["n_img"], 3), np.float32)
    orc.orc_compute_loss(n_rays_global, aabb.ctypes.data, st, inc, B, n_alive, H.f32(128.0), 4, bg.ctypes.data, 0, 1, 0, S["n_img"], S["md"].ctypes.data, out.ctypes.data,
                         cnt.ctypes.data, r["idx"].ctypes.data, r["rays"].ctypes.data, r["ns"].ctypes.data, r["co"].ctypes.data, co_c.ctypes.data, dl.ctypes.data, 4,
                         loss.ctypes.data, 0, None, 2, 3, 0, em.ctypes.data, em_res.ctypes.data, H.f32(S["mean"]), exposure.ctypes.data, H.f32(0.2), None, None, None, H.f32(0.0), 2, None)
    n_c = int(cnt[0])
    grads = np.zeros(H.n_params(S["desc"]), np.float64)
    if n_c:
        orc.orc_nerf_forward_backward(S["desc"].ctypes.data, S["params"].ctypes.data, co_c.ctypes.data, 7, n_c, dl.ctypes.data, None, grads.ctypes.data, None)
    return grads, np.array([n_samples, n_c, float(loss.sum())])


def _worker(rank, world, port, result_path):
    import torch
    import torch.distributed as dist
    sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
    import capi
    import helpers as Hh
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    orc, ngp = Hh.load_oracle(), capi.load_ngp_hip()
    S = _scene(orc, ngp)
    R = 256  # rays per rank
    g, c = _rank_gradient(orc, S, R, rank * R, world * R)
    tg, tc = torch.from_numpy(g), torch.from_numpy(c)
    dist.all_reduce(tg)
    dist.all_reduce(tc)
    if rank == 0:
        full_g, full_c = _rank_gradient(orc, S, world * R, 0, world * R)
        np.savez(result_path, summed=tg.numpy(), full=full_g, counters=tc.numpy(), full_counters=full_c)
    # every rank holds the same reduced counters
    gathered = [torch.zeros_like(tc) for _ in range(world)]
    dist.all_gather(gathered, tc)
    assert all(torch.equal(gathered[0], x) for x in gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gradients_sum_to_full_batch(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    out = str(tmp_path / "dp.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    # the two shards together see exactly the rays of the full batch: identical sample / compaction counts and loss
    np.testing.assert_array_equal(r["counters"][:2], r["full_counters"][:2])
    assert r["counters"][0] > 0 and r["counters"][1] > 0
    np.testing.assert_allclose(r["counters"][2], r["full_counters"][2], rtol=1e-5)
    scale = np.abs(r["full"]).max()
    assert scale > 0
    np.testing.assert_allclose(r["summed"], r["full"], rtol=1e-9, atol=1e-12 * scale)


class _FakeTestbed:
    """records the call sequence of bench.dp_step; mimics the pyngp extension API used by it"""
    def __init__(self, step, counters):
        self.training_step = step
        self.calls = []
        self._c = counters

    def training_prep_nerf(self, B):
        self.calls.append("prep")

    def train_nerf_dp_begin(self, B, get_loss=False):
        self.calls.append("begin")
        return self._c

    def local_loss_sum(self):
        self.calls.append("loss")
        return 0.5

    def train_nerf_dp_backward(self, B, before, after, get_loss, loss_sum):
        self.calls.append(("backward", before, after, get_loss, round(loss_sum, 6)))

    def train_nerf_dp_end(self):
        self.calls.append("end")
        self.training_step += 1


def _dp_step_worker(rank, world, port, result_path):
    import torch
    import torch.distributed as dist
    sys.path[:0] = [ROOT]
    import bench
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    log = []
    for step in (0, 16, 17, 300, 512):
        tb = _FakeTestbed(step, (1000 + rank, 400 + 10 * rank))
        grads = torch.full((8,), float(rank + 1), dtype=torch.float16)
        bench.dp_step(tb, torch, dist, 1 << 18, bench.DpState(torch, grads, "cpu"))
        log.append((step, tb.calls, grads.tolist()))
    if rank == 0:
        import pickle
        pickle.dump(log, open(result_path, "wb"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_step_control_flow(tmp_path):
    import pickle
    import torch.multiprocessing as mp
    port = _free_port()
    out = str(tmp_path / "flow.pkl")
    mp.spawn(_dp_step_worker, args=(2, port, out), nprocs=2, join=True)
    log = pickle.load(open(out, "rb"))
    by_step = {s: (calls, g) for s, calls, g in log}
    # prep schedule of Testbed::train (testbed.cu:2538): every step until 32, then every step//16-th (capped at 16)
    assert by_step[0][0][0] == "prep" and by_step[16][0][0] == "prep"
    assert by_step[17][0][0] == "prep"            # n_prep_to_skip = clamp(17 // 16, 1, 16) = 1
    assert by_step[300][0][0] == "begin"          # n_prep_to_skip = 16 and 300 % 16 != 0
    assert by_step[512][0][0] == "prep"           # 512 % 16 == 0
    for s, (calls, g) in by_step.items():
        assert calls[-1] == "end"
        bwd = calls[-2]
        assert bwd[0] == "backward" and bwd[1] == 2001 and bwd[2] == 810   # counters summed over the two ranks
        assert bwd[3] == (s % 16 == 0)
        assert bwd[4] == (1.0 if s % 16 == 0 else 0.0)
        assert g == [3.0] * 8                                          # gradient buffer was all-reduced in place
        assert calls.index("begin") < len(calls) - 2


def _shm_worker(rank, world, port):
    import torch.distributed as dist
    sys.path[:0] = [ROOT]
    import bench
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    s = bench.ShmCounters(rank, world, "t%d" % port)
    dist.barrier()
    s.attach()
    dist.barrier()
    for step in range(300):
        got = s.all_sum(step, rank + step, 10 * rank, 0.5)
        assert got == (sum(k + step for k in range(world)), 10.0 * sum(range(world)), 0.5 * world), (step, got)
    dist.barrier()
    s.close()
    dist.destroy_process_group()


def test_shared_memory_counter_exchange():
    """bench.ShmCounters: the single-node counter all-reduce of the data-parallel step (two ranks, 300 back-to-back steps)"""
    import torch.multiprocessing as mp
    mp.spawn(_shm_worker, args=(2, _free_port()), nprocs=2, join=True)


def _cpp_shm_worker(rank, world, key, q):
    sys.path.insert(0, os.path.join(ROOT, "blender-ngp_amd"))
    import torch  # noqa: F401
    import pyngp
    x = pyngp.ShmCounterExchange(rank, world, key, 30.0)
    if rank == 0:
        x.publish_blob(bytes(range(128)))
    blob = x.fetch_blob()
    x.barrier()
    out = []
    for step in range(200):
        out.append(x.all_sum(step, float(rank + 1) * (step + 1), float(step), 0.25 * (rank + 1)))
    x.barrier()
    q.put((rank, blob == bytes(range(128)), out[0], out[-1]))


def test_cpp_shared_memory_counter_exchange_and_rccl_binding():
    """the product's own exchange (host/dp.cpp, what Testbed.init_data_parallel uses): three ranks, 200 steps, blob hand-over (the RCCL unique id),
    barrier; and the RCCL library binds (dlopen) on this machine"""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "blender-ngp_amd"))
    import torch  # noqa: F401
    import pyngp
    assert pyngp.rccl_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = "test_%d" % os.getpid()
    world = 3
    ps = [ctx.Process(target=_cpp_shm_worker, args=(r, world, key, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    for rank, blob_ok, first, last in res:
        assert blob_ok
        assert first == (6.0, 0.0, 1.5)                  # (1 + 2 + 3) * 1, 3 * 0, 0.25 * 6
        assert last == (6.0 * 200, 3 * 199.0, 1.5)
    assert not os.path.exists("/dev/shm/ngp_dp_" + key)   # rank 0 unlinked the name


def _late_rank0_worker(rank, world, key, q, delay):
    import time
    sys.path.insert(0, os.path.join(ROOT, "blender-ngp_amd"))
    import torch  # noqa: F401
    import pyngp
    time.sleep(delay)
    x = pyngp.ShmCounterExchange(rank, world, key, 30.0)
    if rank == 0:
        x.publish_blob(bytes([7] * 128))
    blob = x.fetch_blob()
    x.barrier()
    q.put((rank, blob == bytes([7] * 128), x.all_sum(0, 1.0, 2.0, 3.0)))
    x.barrier()


def test_cpp_exchange_ignores_the_leftover_segment_of_a_crashed_job():
    """a segment a crashed job left in /dev/shm carries a valid magic, the right world size and a published (dead) communicator
    id.  Rank 1 arrives BEFORE this job's rank 0 has replaced it; it must not keep that mapping — nobody answers its nonce there — and must end up
    on the new segment with the new id."""
    import multiprocessing as mp
    import struct
    key = "stale_%d" % os.getpid()
    world = 2
    n_bytes = 256 + 2 * world * 64 + world * 64
    stale = bytearray(n_bytes)
    struct.pack_into("<QIIQ", stale, 0, 0x6e67705f64703032, world, 0, 1)    # magic "ngp_dp02", world, pad, blob_ready = 1
    stale[24:24 + 128] = bytes([0xEE] * 128)                                 # the dead job's unique id
    struct.pack_into("<QQQ", stale, 256 + 2 * world * 64 + 64, 5, 5, 5)      # ... whose rank 1 had completed its handshake
    path = "/dev/shm/ngp_dp_" + key
    with open(path, "wb") as f:
        f.write(stale)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_late_rank0_worker, args=(r, world, key, q, 1.5 if r == 0 else 0.0)) for r in range(world)]
        for p in ps:
            p.start()
        res = [q.get(timeout=120) for _ in ps]
        for p in ps:
            p.join(30)
            assert p.exitcode == 0
        for rank, blob_ok, sums in res:
            assert blob_ok, "rank %d read the crashed job's communicator id" % rank
            assert sums == (2.0, 4.0, 6.0)
    finally:
        if os.path.exists(path):
            os.unlink(path)
    assert not os.path.exists(path)


# ---------------------------------------------------------------------------------------------------------------- the sharded optimizer step
def _sharded_optimizer_worker(rank, world, port, q):
    """reduce-scatter (fp32 sums) -> Adam on the rank's shard -> all-gather of the fp16 weights -> Ema over everything, with the oracle as each rank's compute, against the
    replicated step (all-reduce, whole Adam / Ema on every rank).  What Testbed::optimizer_step_sharded does on the GPUs with ngp_rccl_* and the NGP_OPT_* stages."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    orc = H.load_oracle()
    n, nm = 10240 + 30011, 10240
    shard = ((n + world - 1) // world + 7) // 8 * 8
    rs0 = np.random.RandomState(5)                       # the replicated state: identical on every rank
    master = (rs0.randn(n) * 0.1).astype(np.float32)
    p16, m1, m2 = master.astype(np.float16), (rs0.randn(n) * 1e-3).astype(np.float32), (rs0.rand(n) * 1e-5).astype(np.float32)
    ema, inf = master.copy(), master.astype(np.float16)
    hp = (H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-15), H.f32(1e-6), H.f32(128.0), H.f32(0.95))
    rs = np.random.RandomState(100 + rank)               # this rank's gradients
    grads = (rs.randn(n) * 0.3).astype(np.float16)
    grads[rs.rand(n) < 0.5] = 0
    # ---- the wire: fp32 sums (reduce-scatter == a slice of the all-reduce)
    g32 = torch.zeros(shard * world, dtype=torch.float32)
    g32[:n] = torch.from_numpy(grads.astype(np.float32))
    dist.all_reduce(g32)
    gsum16 = g32.numpy()[:n].astype(np.float16)
    # ---- replicated reference: the whole step on every rank
    ref = [a.copy() for a in (master, p16, m1, m2, ema, inf)]
    orc.orc_adam_ema_step(n, nm, 9, *hp, gsum16.ctypes.data, *[a.ctypes.data for a in ref])
    # ---- sharded: Adam on my shard only (the Ema outputs of this call are discarded) ...
    off = shard * rank
    mine = max(0, min(shard, n - off))
    my = [np.ascontiguousarray(a[off:off + mine]) for a in (master, p16, m1, m2)]
    junk_ema, junk_inf = np.zeros(mine, np.float32), np.zeros(mine, np.float16)
    if mine:
        gs = np.ascontiguousarray(gsum16[off:off + mine])
        orc.orc_adam_ema_step(mine, max(nm - off, 0), 9, *hp, gs.ctypes.data, my[0].ctypes.data, my[1].ctypes.data, my[2].ctypes.data, my[3].ctypes.data, junk_ema.ctypes.data, junk_inf.ctypes.data)
    # ... all-gather of the fp16 weights (equal shards, the last one padded) ...
    chunk = torch.zeros(shard * 2, dtype=torch.uint8)                 # (gloo moves bytes; RCCL moves ncclFloat16 — an all-gather does no arithmetic)
    chunk[:mine * 2] = torch.from_numpy(my[1].view(np.uint8))
    parts = [torch.zeros_like(chunk) for _ in range(world)]
    dist.all_gather(parts, chunk)
    new_p16 = torch.cat(parts).numpy()[:n * 2].view(np.float16).copy()
    # ... Ema over all parameters from the gathered weights: the oracle's step with nothing to optimise (no matrix parameters, zero gradients) is its Ema stage
    zeros = np.zeros(n, np.float16)
    m_, a_, b_ = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    new_ema, new_inf = ema.copy(), inf.copy()
    orc.orc_adam_ema_step(n, 0, 9, *hp, zeros.ctypes.data, m_.ctypes.data, new_p16.ctypes.data, a_.ctypes.data, b_.ctypes.data, new_ema.ctypes.data, new_inf.ctypes.data)
    ok = dict(params=np.array_equal(new_p16.view(np.uint16), ref[1].view(np.uint16)), ema=np.array_equal(new_ema, ref[4]), inference=np.array_equal(new_inf.view(np.uint16), ref[5].view(np.uint16)),
              master=np.array_equal(my[0], ref[0][off:off + mine]), m1=np.array_equal(my[2], ref[2][off:off + mine]), m2=np.array_equal(my[3], ref[3][off:off + mine]))
    q.put((rank, ok, int((ref[1] != p16).sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_equals_replicated_bit_for_bit(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_sharded_optimizer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, n_changed in res:
        assert all(ok.values()), (rank, ok)
        assert n_changed > 10000            # the step did move the weights


# ---------------------------------------------------------------------------------------------------------------- round 5: fp16 on the wire, Ema sharded
def _fp16_wire_worker(rank, world, port, q):
    """Testbed::optimizer_step_sharded as of round 5, stage by stage with gloo standing in for RCCL and the oracle for the kernels:
      all-to-all of the ranks' fp16 gradient slices (no arithmetic on the wire) -> the owner adds the world slices IN RANK ORDER in fp32, one fp16 rounding
      -> Adam AND Ema on the rank's shard (one call) -> all-gather of the fp16 training weights; inference weights / Ema gathered on demand.
    Against (a) the replicated step fed with the rank-ordered fp32 sum (the definition), (b) the round-3 wire: fp32 all-reduce of the widened vectors."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    orc = H.load_oracle()
    n, nm = 10240 + 30011, 10240
    shard = ((n + world - 1) // world + 7) // 8 * 8
    rs0 = np.random.RandomState(5)
    master = (rs0.randn(n) * 0.1).astype(np.float32)
    p16, m1, m2 = master.astype(np.float16), (rs0.randn(n) * 1e-3).astype(np.float32), (rs0.rand(n) * 1e-5).astype(np.float32)
    ema, inf = master.copy(), master.astype(np.float16)
    hp = (H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-15), H.f32(1e-6), H.f32(128.0), H.f32(0.95))
    rs = np.random.RandomState(100 + rank)
    grads = np.zeros(shard * world, np.float16)
    grads[:n] = (rs.randn(n) * np.exp(rs.randn(n) * 3)).astype(np.float16)      # magnitudes over many binades: fp32 sums of three are NOT all exact
    grads[:n][rs.rand(n) < 0.5] = 0
    # ---- the wire: slice q of my vector to rank q (bytes; gloo has no fp16 arithmetic and needs none)
    send = [torch.from_numpy(grads[q * shard:(q + 1) * shard].view(np.uint8).copy()) for q in range(world)]
    recv = [torch.zeros(shard * 2, dtype=torch.uint8) for _ in range(world)]
    recv[rank].copy_(send[rank])                                                # the own slice: a device copy in ngp_rccl_alltoall_f16
    ops = []
    for qq in range(world):                                                     # grouped point-to-point sends / receives, as ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd
        if qq != rank:
            ops += [dist.P2POp(dist.isend, send[qq], qq), dist.P2POp(dist.irecv, recv[qq], qq)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    slices = np.stack([r.numpy().view(np.float16) for r in recv])               # [world][shard]: slice q = rank q's contribution to MY shard
    acc = slices[0].astype(np.float32)
    for qq in range(1, world):
        acc = (acc + slices[qq].astype(np.float32)).astype(np.float32)          # rank order, fp32
    my_gsum16 = acc.astype(np.float16)                                          # one rounding (ngp_hip_sum_slices_f16)
    # ---- (a) the definition, computed whole on every rank from all vectors (an all-gather of the raw gradients, test only)
    allg = [torch.zeros(shard * world * 2, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(allg, torch.from_numpy(grads.view(np.uint8).copy()))
    whole = allg[0].numpy().view(np.float16).astype(np.float32)
    for qq in range(1, world):
        whole = (whole + allg[qq].numpy().view(np.float16).astype(np.float32)).astype(np.float32)
    gsum16 = whole.astype(np.float16)
    off = shard * rank
    mine = max(0, min(shard, n - off))
    wire_ok = np.array_equal(my_gsum16.view(np.uint16), gsum16[off:off + shard].view(np.uint16))
    # ---- (b) the round-3 wire: fp32 all-reduce (gloo's own order) of the widened vectors, then one rounding
    g32 = torch.from_numpy(grads.astype(np.float32))
    dist.all_reduce(g32)
    old16 = g32.numpy().astype(np.float16)
    n_differ_from_round3 = int((old16[:n].view(np.uint16) != gsum16[:n].view(np.uint16)).sum())
    # ---- replicated reference step with the definition's gradient
    ref = [a.copy() for a in (master, p16, m1, m2, ema, inf)]
    orc.orc_adam_ema_step(n, nm, 9, *hp, np.ascontiguousarray(gsum16[:n]).ctypes.data, *[a.ctypes.data for a in ref])
    # ---- sharded: Adam + Ema on my shard in ONE call
    my = [np.ascontiguousarray(a[off:off + mine]) for a in (master, p16, m1, m2, ema, inf)]
    if mine:
        gs = np.ascontiguousarray(my_gsum16[:mine])
        orc.orc_adam_ema_step(mine, max(nm - off, 0), 9, *hp, gs.ctypes.data, *[a.ctypes.data for a in my])
    ok = dict(wire=wire_ok)
    for name, idx in (("master", 0), ("params", 1), ("m1", 2), ("m2", 3), ("ema", 4), ("inference", 5)):
        a, b = my[idx], ref[idx][off:off + mine]
        ok[name] = np.array_equal(a.view(np.uint8), b.view(np.uint8))
    # ---- the all-gathers (training weights every step; inference weights on demand) reassemble the replicated vectors
    for name, idx in (("params_gathered", 1), ("inference_gathered", 5)):
        chunk = torch.zeros(shard * 2, dtype=torch.uint8)
        chunk[:mine * 2] = torch.from_numpy(my[idx].view(np.uint8))
        parts = [torch.zeros_like(chunk) for _ in range(world)]
        dist.all_gather(parts, chunk)
        ok[name] = np.array_equal(torch.cat(parts).numpy()[:n * 2], ref[idx].view(np.uint8))
    q.put((rank, ok, n_differ_from_round3, int((ref[1] != p16).sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_fp16_wire_and_sharded_ema_equal_the_replicated_step_bit_for_bit(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_fp16_wire_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    n = 10240 + 30011
    for rank, ok, n_differ_from_round3, n_changed in res:
        assert all(ok.values()), (rank, ok)
        assert n_changed > 10000
        if world == 2:
            assert n_differ_from_round3 == 0      # a + b = b + a: the fp16 wire gives the bits of the fp32 reduce-scatter it replaces
        else:
            # three addends: the fp32 sum depends on the order where it is inexact; after the one fp16 rounding the library-ordered sum and the rank-ordered
            # sum differ in a handful of elements at most — the rank-ordered one is the DEFINED result from this round on
            assert n_differ_from_round3 <= n // 500, n_differ_from_round3


def test_loopback_librccl_is_what_the_binding_resolves(tmp_path):
    """tests/loopback/librccl_loopback.so (the several-ranks-on-one-GPU stand-in of tests/test_dp_loopback_gpu.py) exports every nccl* name csrc/comm.hip looks up, and
    NGP_RCCL_LIBRARY makes the product library bind it: checked without a GPU (unique id, a one-rank communicator, the refusal of an id it did not make)."""
    import ctypes
    import subprocess
    lb_dir = os.path.join(ROOT, "tests", "loopback")
    so = os.path.join(lb_dir, "librccl_loopback.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(lb_dir, "rccl_loopback.hip")):
        subprocess.check_call(["make", "-C", lb_dir], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclAllGather", "ncclReduceScatter", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd",
                 "ncclCommCount", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(lib, name), name
    code = """
import sys, numpy as np
sys.path[:0] = [%r]
import capi
ngp = capi.load_ngp_hip()
assert ngp.ngp_rccl_available() == 1
uid = np.zeros(128, np.uint8)
capi.check(ngp.ngp_rccl_get_unique_id(uid.ctypes.data))
assert bytes(uid[:14]) == b"/ngp_loopback_", bytes(uid[:20])
comm = ngp.ngp_rccl_init(0, 1, uid.ctypes.data)
assert comm and ngp.ngp_rccl_comm_size(comm) == 1 and ngp.ngp_rccl_comm_rank(comm) == 0
capi.check(ngp.ngp_rccl_finalize(comm))
bad = np.zeros(128, np.uint8); bad[:4] = [1, 2, 3, 4]
assert ngp.ngp_rccl_init(0, 1, bad.ctypes.data) is None          # an id of another library is refused, not dereferenced
print("ok")
""" % os.path.join(ROOT, "blender-ngp_amd")
    env = dict(os.environ, NGP_RCCL_LIBRARY=so)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:]
    # a path that does not exist is an error of the binding, not a silent fall-back to another RCCL
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path[:0] = [%r]; import capi; print(capi.load_ngp_hip().ngp_rccl_available())" % os.path.join(ROOT, "blender-ngp_amd")],
                       env=dict(os.environ, NGP_RCCL_LIBRARY=str(tmp_path / "nope.so")), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("0"), r.stdout[-500:]
