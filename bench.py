#!/usr/bin/env python
"""bench.py — train-samples/s (+ render MP/s, PSNR) of the NeRF hot path on the procedural-lego workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one Testbed::train() call (src/testbed.cu:2527-2587): occupancy-grid prep on its schedule + ray marching +
inference + loss/compaction + forward/backward + optimizer at the reference's batch of 2^18 compacted samples.
value = sum over the K timed steps and all ranks of measured_batch_size (src/testbed_nerf.cu:2883) / wall time (max over ranks).
N > 1: weak scaling — every rank trains its own 2^18-sample batch on a disjoint ray slice; one fp16 RCCL all-reduce of the
gradient vector per step (hash table + MLP), replicated optimizer step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "blender-ngp_amd")
for p in (PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FWD_BYTES_PER_SAMPLE = 588     # 512 B gathered + 12 B position + 64 B encoded features
BWD_BYTES_PER_SAMPLE = 1100    # 64 B dL/dy + 12 B pos + 512 B read + 512 B write (atomic RMW)
OPT_BYTES_PER_PARAM = 36       # grads 2 + master 4+4 + m 4+4 + v 4+4 + fp16 2 + ema 4+4 ... see DESIGN.md (28 B + 8 B EMA)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


class CudaArray:
    """zero-copy torch view of a device buffer owned by the C++ Testbed (the gradient vector for the RCCL all-reduce)."""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class DpState:
    """per-rank buffers of the data-parallel step: the gradient view, a 3-word scratch tensor and the stream the collectives are
    ordered on (the Testbed's own HIP stream wrapped as a torch ExternalStream; None on CPU / gloo)"""
    def __init__(self, torch, grads, device, comm_stream=None, ctl_group=None, ctl_stream=None):
        self.grads = grads
        self.ctl_group = ctl_group     # process group of the 24-byte counter exchange (high-priority RCCL stream on GPU)
        self.ctl_stream = ctl_stream   # high-priority torch stream for its H2D / D2H copies
        self.scratch = torch.zeros(3, dtype=torch.float64, device=device)
        self.host = torch.zeros(3, dtype=torch.float64)
        if torch.device(device).type == "cuda":
            self.host = self.host.pin_memory()
        self.comm_stream = comm_stream
        self._torch = torch

    def on_comm_stream(self):
        import contextlib
        return self._torch.cuda.stream(self.comm_stream) if self.comm_stream is not None else contextlib.nullcontext()

    def on_ctl_stream(self):
        import contextlib
        return self._torch.cuda.stream(self.ctl_stream) if self.ctl_stream is not None else contextlib.nullcontext()


def make_dp_state(torch, dist, tb, grads, dev):
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    ctl_group = dist.new_group(backend="nccl", pg_options=opts)
    ctl_stream = torch.cuda.Stream(device=dev, priority=-1)
    return DpState(torch, grads, dev, torch.cuda.ExternalStream(tb.stream_ptr(), device=dev), ctl_group, ctl_stream)


def dp_step(tb, torch, dist, B, st):
    """Testbed::train (testbed.cu:2527-2587) with the two exchanges of the data-parallel step:
    counters (+ loss) right after the loss kernel, gradients between backward and optimizer."""
    step = tb.training_step
    n_prep_to_skip = min(max(step // 16, 1), 16)
    if step % n_prep_to_skip == 0:
        tb.training_prep_nerf(B)  # replicated: same params + same rng on every rank => bit-identical grids, no collective
    get_loss = step % 16 == 0
    c0, c1 = tb.train_nerf_dp_begin(B, get_loss)   # returns once the loss kernel ran; forward / backward are already queued
    st.host[0], st.host[1], st.host[2] = c0, c1, (tb.local_loss_sum() if get_loss else 0.0)
    with st.on_ctl_stream():
        # 24 bytes on high-priority streams so that it is not queued behind backward: every rank derives the same rays_per_batch
        st.scratch.copy_(st.host, non_blocking=True)
        dist.all_reduce(st.scratch, group=st.ctl_group)
        st.host.copy_(st.scratch)
    n0, n1, loss_sum = st.host.tolist()
    tb.train_nerf_dp_backward(B, int(n0), int(n1), get_loss, float(loss_sum))   # feedback + next step's march on stream B
    with st.on_comm_stream():
        # RCCL over xGMI, fp16 sum of the loss-scaled gradients (24.4 MB for lego), stream-ordered: after backward, before Adam
        dist.all_reduce(st.grads)
    tb.train_nerf_dp_end()
    return c1


def cpu_baseline(n=65536, k=2):
    """The oracle (CPU port of the same step) on a bounded sample: inference of k*n + forward/backward of n samples through the
    lego-sized network (T = 2^19) + one Adam/EMA pass over all 12.2 M parameters.  Reported as compacted samples / s."""
    import helpers as H
    import capi
    ngp = capi.load_ngp_hip()
    orc = H.load_oracle()
    desc = H.make_desc(ngp, 19)
    P = H.random_params(desc, 0, 0.01)
    c = H.random_coords(k * n, 1)
    out = np.zeros((k * n, 4), np.uint16)
    dl = (np.random.RandomState(0).randn(n, 4) * 0.01).astype(np.float16)
    npar = H.n_params(desc)
    g = np.zeros(npar, np.float64)
    t0 = time.time()
    orc.orc_nerf_inference(desc.ctypes.data, P.ctypes.data, c.ctypes.data, 7, k * n, out.ctypes.data, 4)
    orc.orc_nerf_forward_backward(desc.ctypes.data, P.ctypes.data, c.ctypes.data, 7, n, dl.ctypes.data, None, g.ctypes.data, None)
    g16 = g.astype(np.float16)
    master = P.astype(np.float32)
    m1, m2, ema, inf = np.zeros(npar, np.float32), np.zeros(npar, np.float32), np.zeros(npar, np.float32), P.copy()
    orc.orc_adam_ema_step(npar, 10240, 1, H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-15), H.f32(1e-6), H.f32(128.0), H.f32(0.95), g16.ctypes.data, master.ctypes.data,
                          P.ctypes.data, m1.ctypes.data, m2.ctypes.data, ema.ctypes.data, inf.ctypes.data)
    dt = time.time() - t0
    return {"value": n / dt, "unit": "samples/s", "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)), "kind": "port",
            "sample": "oracle: %d pre-compaction inference + %d compacted fwd/bwd samples (1/%d of a 2^18 step) + full 12.2M-param Adam/EMA, %.1f s" % (k * n, n, (1 << 18) // n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--n_train", type=int, default=100)
    ap.add_argument("--n_test", type=int, default=3)
    ap.add_argument("--eval_spp", type=int, default=1)
    ap.add_argument("--no_render", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--force_dp", action="store_true", help="run the data-parallel step path (RCCL all-reduce) even with one rank")
    a = ap.parse_args()

    import torch  # first: one HIP runtime per process
    import scene

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dp = world > 1 or a.force_dp
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29500 + os.getpid() % 1000), rank=0, world_size=1, device_id=dev)

    B = 1 << 18
    ds = scene.make_dataset(a.n_train, a.n_test, a.res, dev)
    tb = scene.build_testbed(ds)
    tb.set_distributed(rank, world)
    dp = None
    if use_dp:
        grads = torch.as_tensor(CudaArray(tb.gradients_ptr(), tb.n_params(), "<f2"), device=dev)
        assert grads.data_ptr() == tb.gradients_ptr()
        dp = make_dp_state(torch, dist, tb, grads, dev)

    def one_step():
        if not use_dp:
            tb.frame()
            return tb.nerf.training.measured_batch_size
        dp_step(tb, torch, dist, B, dp)
        return tb.nerf.training.measured_batch_size

    for _ in range(a.warmup):
        one_step()
    tb.set_profiling(True)
    tb.reset_profile()
    if use_dp:
        dist.barrier()
    torch.cuda.synchronize()
    tb.sync()
    t0 = time.perf_counter()
    samples = 0
    rays = 0
    for _ in range(a.steps):
        rays += tb.nerf.training.rays_per_batch
        samples += min(one_step(), B)
    tb.sync()
    torch.cuda.synchronize()
    if use_dp:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = tb.profile()
    tb.set_profiling(False)
    pre_compaction = prof["nerf_inference"]["units"]

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([samples, rays, pre_compaction], dtype=torch.float64, device=dev)
        dist.all_reduce(s)
        samples, rays, pre_compaction = (float(x) for x in s.tolist())

    if rank != 0:
        if use_dp:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel, from HIP-event timings taken on the launch stream during the timed region
    bytes_per_unit = {"nerf_inference": FWD_BYTES_PER_SAMPLE, "nerf_forward": FWD_BYTES_PER_SAMPLE, "nerf_backward": BWD_BYTES_PER_SAMPLE, "optimizer_step": OPT_BYTES_PER_PARAM}
    kernels = {}
    for name, e in prof.items():
        if e["launches"]:
            k = {"ms_total": round(e["ms"], 3), "launches": int(e["launches"]), "avg_us": round(1000.0 * e["ms"] / e["launches"], 2), "units_per_launch": round(e["units"] / e["launches"], 1)}
            if name in bytes_per_unit:
                k["algorithmic_GBps"] = round(bytes_per_unit[name] * e["units"] / (e["ms"] * 1e-3) / 1e9, 1)
            kernels[name] = k
    dom = max((n for n in kernels if n in bytes_per_unit), key=lambda n: kernels[n]["ms_total"])
    achieved = kernels[dom]["algorithmic_GBps"]
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom)
        except Exception:
            traffic = None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_unit": bytes_per_unit[dom], "units_per_launch": kernels[dom]["units_per_launch"], "avg_launch_us": kernels[dom]["avg_us"]}

    # ---- render MP/s + PSNR on the trained model (rank 0), outside the timed region
    extra = {}
    if not a.no_render:
        psnr, ssim, per = scene.eval_test_views(tb, ds, spp=a.eval_spp, max_views=a.n_test)
        t1 = time.perf_counter()
        n_frames = 3
        for i in range(n_frames):
            tb.set_nerf_camera_matrix(ds["test_poses"][i % len(ds["test_poses"])][:3, :])
            tb.render(a.res, a.res, 1, True)
        rdt = (time.perf_counter() - t1) / n_frames
        extra = {"render_MP_per_s": round(a.res * a.res / rdt / 1e6, 2), "render_ms_per_frame": round(rdt * 1e3, 2), "render_network_samples_per_frame": int(tb.render_samples_evaluated),
                 "psnr_db": round(psnr, 2), "ssim": round(ssim, 4), "psnr_at_step": int(tb.training_step), "eval_views": len(per), "eval_spp": a.eval_spp}

    line = {
        "metric": "train samples/s (compacted samples back-propagated per second), nerf-synthetic/lego stand-in",
        "value": round(samples / dt, 1), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000.0 * dt / a.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "procedural-lego %dx%d x%d RGBA8 views (scale 0.33, offset 0.5, aabb_scale 1), configs/nerf/base.json (L=16 F=2 T=2^19, 64-wide MLPs), batch 2^18 compacted samples per GPU" % (a.res, a.res, a.n_train),
                   "global_batch": B * world, "parallelism": "dp%d" % world if use_dp else "single"},
        "rays_per_s": round(rays / dt, 1), "pre_compaction_samples_per_s": round(pre_compaction / dt, 1),
        "roofline": roofline, "kernels": kernels,
    }
    line.update(extra)
    if world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
    print(json.dumps(line), flush=True)
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
