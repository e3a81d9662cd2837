"""A frame rendered in row shards (SURVEY.md §8e "Render: image tiles/rows per rank + gather"; VERDICT r02 row e-r), CPU side: the PARTITION CONTRACT on the oracle's
renderer (orc_render_nerf_rows) — rank r renders rows [r * ceil(H / P), (r + 1) * ceil(H / P)), payload.idx and every per-pixel random number keyed by the pixel's
index in the whole frame — with two gloo ranks gathering their rows, against the frame rendered at once.  Bit for bit: rays do not interact."""
import os
import socket
import sys

import numpy as np

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    import capi
    ngp = capi.load_ngp_hip()                      # host-side helper (the level table); no device call
    orc = H.load_oracle()
    grid = H.blob_density_grid(1)
    bf, _ = H.oracle_bitfield(orc, grid, 1)
    desc = H.make_desc(ngp, log2_hashmap_size=12)
    params = H.random_params(desc, seed=3, grid_amp=2.0)
    return orc, desc, params, bf


def _render_rows(orc, desc, params, bf, W, Hh, row_begin, row_end, spp_index=1, snap=0):
    res = np.array([W, Hh], np.int32)
    focal = np.array([0.9 * W, 0.9 * W], np.float32)
    cam = np.ascontiguousarray(H.hemisphere_cameras(3)[1]["start"].reshape(-1).astype(np.float32))
    sc = np.array([0.5, 0.5], np.float32)
    aabb = H.unit_aabb()
    ident = np.eye(3, dtype=np.float32).reshape(-1)
    frame, depth = np.zeros((Hh, W, 4), np.float32), np.zeros((Hh, W), np.float32)
    n = orc.orc_render_nerf_rows(desc.ctypes.data, params.ctypes.data, spp_index, res.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, sc.ctypes.data, snap, aabb.ctypes.data,
                                 ident.ctypes.data, aabb.ctypes.data, H.f32(0.05), bf.ctypes.data, H.f32(0.0), 2, 3, H.f32(0.01), 0, frame.ctypes.data, depth.ctypes.data, row_begin, row_end)
    return frame, depth, int(n)


def _rows_of(rank, world, height):
    per = (height + world - 1) // world
    a = min(height, rank * per)
    return a, min(height, a + per)


def test_row_shards_equal_the_whole_frame_for_every_partition():
    orc, desc, params, bf = _scene()
    W, Hh = 40, 27                                   # 27 rows: uneven over 2, 4 and 5 ranks
    whole, whole_depth, n_whole = _render_rows(orc, desc, params, bf, W, Hh, 0, Hh)
    assert n_whole > 500 and (whole[..., 3] > 0.01).any() and (whole[..., 3] == 0).any()   # an object in front of empty pixels
    for world in (2, 4, 5, 27, 32):
        frame, depth, n = np.zeros_like(whole), np.zeros_like(whole_depth), 0
        for r in range(world):
            a, b = _rows_of(r, world, Hh)
            f, d, k = _render_rows(orc, desc, params, bf, W, Hh, a, b)
            assert not f[:a].any() and not f[b:].any()           # a shard writes its own rows only
            frame[a:b], depth[a:b] = f[a:b], d[a:b]
            n += k
        np.testing.assert_array_equal(frame, whole)
        np.testing.assert_array_equal(depth, whole_depth)
        assert n == n_whole                                      # the same network samples, only cut differently into passes


def _worker(rank, world, port, W, Hh, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    orc, desc, params, bf = _scene()
    a, b = _rows_of(rank, world, Hh)
    f, d, n = _render_rows(orc, desc, params, bf, W, Hh, a, b)
    per = (Hh + world - 1) // world
    chunk = torch.zeros(per * W * 4)
    chunk[:(b - a) * W * 4] = torch.from_numpy(f[a:b].reshape(-1))
    parts = [torch.zeros_like(chunk) for _ in range(world)]
    dist.all_gather(parts, chunk)                                # what ngp_rccl_allgather_f32 does on the GPUs: equal chunks, rank order
    frame = torch.cat(parts)[:Hh * W * 4].reshape(Hh, W, 4).numpy()
    q.put((rank, frame, n))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_gather_the_frame_rendered_at_once():
    import torch.multiprocessing as mp
    W, Hh, world = 32, 21, 2
    orc, desc, params, bf = _scene()
    whole, _, n_whole = _render_rows(orc, desc, params, bf, W, Hh, 0, Hh)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, W, Hh, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert sum(n for _, _, n in res) == n_whole
    for rank, frame, _ in res:
        np.testing.assert_array_equal(frame, whole)              # every rank holds the whole frame
