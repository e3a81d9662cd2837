#!/usr/bin/env python
"""bench.py — train-samples/s (+ render MP/s, PSNR) of the NeRF hot path on the procedural-lego workload.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one Testbed::train() call (src/testbed.cu:2527-2587): occupancy-grid prep on its schedule + ray marching +
inference + loss/compaction + forward/backward + optimizer at the reference's batch of 2^18 compacted samples.
value = sum over the K timed steps and all ranks of measured_batch_size (src/testbed_nerf.cu:2883) / wall time (max over ranks).
N > 1: strong scaling is the headline (the batch of 2^18 compacted samples per step is split over the ranks — the reference's convergence per
step); the weak-scaled figure (2^18 per rank) is measured in the same run and printed beside it (`weak_scaling`).  Per step: fp32 reduce-scatter of the gradients,
Adam on each rank's shard, fp16 all-gather of the weights (RCCL over xGMI).
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
BYTES_PER_UNIT = {"nerf_inference": FWD_BYTES_PER_SAMPLE, "nerf_forward": FWD_BYTES_PER_SAMPLE, "nerf_backward": BWD_BYTES_PER_SAMPLE, "optimizer_step": OPT_BYTES_PER_PARAM}
MARCH_BYTES_PER_SAMPLE = 28    # one NerfCoordinate written per sample (nerf.h:62-107)
MARCH_BYTES_PER_RAY = 40       # ray index 4 + Ray 24 + numsteps 8 written, one RGBA8 pixel read (testbed_nerf.cu:1232-1258)
KERNEL_SET = "r06a"            # bumped whenever a kernel of a timed launch group changes: PMC numbers of another set are not quoted
GROUP_KERNELS = {"grad_exchange": "data-parallel step: fp16 -> fp32 copy, RCCL reduce-scatter (fp32 sums), fp32 -> fp16 of this rank's shard", "param_gather": "data-parallel step: RCCL all-gather of the fp16 weights",
                 "nerf_backward": "one ngp_hip_nerf_backward call: MLP dgrad+wgrad kernel, hash-grid backward (bin count, scan, bin scatter, owners, combine)",
                 "nerf_inference": "one ngp_hip_nerf_forward call: fused hash-grid encode + both MLPs (single kernel)", "optimizer_step": "adam_ema_vec4_kernel (single kernel)"}
SURVEY_STEPS = 48              # untimed steps with every launch group bracketed by events (picks the dominant group, fills "kernels")
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


class CudaArray:
    """zero-copy torch view of a device buffer owned by the C++ Testbed (the gradient vector for the RCCL all-reduce)."""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class ShmCounters:
    """all-reduce (sum) of three numbers per step between the ranks of ONE node through POSIX shared memory: every rank owns a 64-byte
    slot per step parity, writes {values, step tag} and spins until all slots carry the tag.  A rank can be at most one step ahead of
    the slowest one (it needs everybody's values to proceed), so two parities are enough."""
    def __init__(self, rank, world, key):
        from multiprocessing import shared_memory
        import numpy as np
        self.rank, self.world, self.np = rank, world, np
        name = "ngp_dp_%s" % key
        size = 2 * world * 64
        if rank == 0:
            try:
                old = shared_memory.SharedMemory(name=name)
                old.close(); old.unlink()
            except FileNotFoundError:
                pass
            self.mem = shared_memory.SharedMemory(name=name, create=True, size=size)
            self.mem.buf[:size] = bytes(size)
        self.name, self.size = name, size

    def attach(self):   # after a barrier that follows rank 0's constructor
        from multiprocessing import shared_memory
        if self.rank != 0:
            self.mem = shared_memory.SharedMemory(name=self.name)
        self.a = self.np.ndarray((2, self.world, 8), dtype=self.np.float64, buffer=self.mem.buf)

    def all_sum(self, step, v0, v1, v2):
        a = self.a[step & 1]
        mine = a[self.rank]
        mine[0], mine[1], mine[2] = v0, v1, v2
        mine[3] = float(step + 1)           # tag last (x86 stores are not reordered with older stores)
        tags = a[:, 3]
        want = float(step + 1)
        t0 = time.perf_counter()
        while not (tags == want).all():
            if time.perf_counter() - t0 > 60.0:
                raise RuntimeError("shared-memory counter exchange timed out at step %d (tags %s)" % (step, tags.tolist()))
        return float(a[:, 0].sum()), float(a[:, 1].sum()), float(a[:, 2].sum())

    def close(self):
        self.a = None
        self.mem.close()
        if self.rank == 0:
            self.mem.unlink()


class DpState:
    """per-rank buffers of the data-parallel step: the gradient view (split into its MLP and hash-grid parts), a 3-double scratch
    tensor for the counter exchange and the streams the collectives are ordered on.  On CPU / gloo (tests) everything is None and the
    exchanges are plain blocking all-reduces."""
    def __init__(self, torch, grads, device, train_stream=None, ctl_group=None, ctl_stream=None, grid_stream=None, n_mlp=0):
        self.grads = grads
        self.n_mlp = n_mlp
        self.ctl_group = ctl_group     # process group of the 24-byte counter exchange (high-priority RCCL stream on GPU)
        self.ctl_stream = ctl_stream   # high-priority torch stream the exchange is ordered on
        self.grid_stream = grid_stream # side stream of the hash-grid gradient all-reduce (starts while the MLP weight gradients are computed)
        self.scratch = torch.zeros(3, dtype=torch.float64, device=device)
        self.host = torch.zeros(3, dtype=torch.float64)
        self.on_gpu = torch.device(device).type == "cuda"
        # counter exchange: "shm" (single node: host shared memory, microseconds), "device" (RCCL, stream-ordered), "host" (RCCL after a host hop)
        self.ctl_mode = os.environ.get("BENCH_DP_CTL", "shm")
        # "1": all-reduce the hash-grid gradients on a side stream while the MLP weight gradients are computed.  Measured on one rank the
        # cross-stream event hops cost more (~90 us) than the overlap can win back (~85 us), so it is off by default
        self.bucketed = os.environ.get("BENCH_DP_BUCKET", "0") == "1"
        self.shm = None
        if self.on_gpu:
            self.host = self.host.pin_memory()
            self.ctl_done = torch.cuda.Event()
            self.grid_done = torch.cuda.Event()
        self.train_stream = train_stream   # the Testbed's own HIP stream as a torch ExternalStream
        self._torch = torch


def make_dp_state(torch, dist, tb, grads, dev):
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    ctl_group = dist.new_group(backend="nccl", pg_options=opts)
    ctl_stream = torch.cuda.Stream(device=dev, priority=-1)
    st = DpState(torch, grads, dev, torch.cuda.ExternalStream(tb.stream_ptr(), device=dev), ctl_group, ctl_stream, torch.cuda.Stream(device=dev), int(tb.n_mlp_params))
    tb.set_dp_counter_buffer(st.scratch.data_ptr())   # every step's post kernel leaves {samples, compacted samples, loss sum} there
    world, rank = dist.get_world_size(), dist.get_rank()
    single_node = int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world
    if st.ctl_mode == "shm" and single_node:
        st.shm = ShmCounters(rank, world, os.environ.get("MASTER_PORT", "0"))
        dist.barrier()
        st.shm.attach()
        dist.barrier()
    return st


def dp_step(tb, torch, dist, B, st):
    """Testbed::train (testbed.cu:2527-2587) with the two exchanges of the data-parallel step:
    counters (+ loss) right after the loss kernel, gradients between backward and optimizer."""
    step = tb.training_step
    n_prep_to_skip = min(max(step // 16, 1), 16)
    if step % n_prep_to_skip == 0:
        tb.training_prep_nerf(B)  # replicated: same params + same rng on every rank => bit-identical grids, no collective
    get_loss = step % 16 == 0
    if st.on_gpu and st.ctl_mode == "shm" and st.shm is not None:
        # single node: the host polls its own counters (host-mapped memory) and the ranks exchange them through shared memory
        c0, c1 = tb.train_nerf_dp_begin(B, get_loss)
        n0, n1, loss_sum = st.shm.all_sum(step, c0, c1, tb.local_loss_sum() if get_loss else 0.0)
        st.host[0], st.host[1], st.host[2] = n0, n1, loss_sum
    elif st.on_gpu and st.ctl_mode == "device":
        # the whole step is queued; the counters land in st.scratch (device) in stream order.  24 bytes are all-reduced on
        # high-priority streams right behind the loss kernel (not behind backward), one host wait for the summed values
        tb.train_nerf_dp_begin(B, get_loss, False)
        tb.stream_wait_counters(st.ctl_stream.cuda_stream)
        with torch.cuda.stream(st.ctl_stream):
            dist.all_reduce(st.scratch, group=st.ctl_group)
            st.host.copy_(st.scratch, non_blocking=True)
            st.ctl_done.record()
        st.ctl_done.synchronize()
        c1 = 0
    elif st.on_gpu:
        c0, c1 = tb.train_nerf_dp_begin(B, get_loss)   # returns once the loss kernel ran; backward is already queued
        st.host[0], st.host[1], st.host[2] = c0, c1, (tb.local_loss_sum() if get_loss else 0.0)
        with torch.cuda.stream(st.ctl_stream):
            st.scratch.copy_(st.host, non_blocking=True)
            dist.all_reduce(st.scratch, group=st.ctl_group)
            st.host.copy_(st.scratch)
    else:
        c0, c1 = tb.train_nerf_dp_begin(B, get_loss)
        st.host[0], st.host[1], st.host[2] = c0, c1, (tb.local_loss_sum() if get_loss else 0.0)
        dist.all_reduce(st.host, group=st.ctl_group)
    n0, n1, loss_sum = st.host.tolist()
    tb.train_nerf_dp_backward(B, int(n0), int(n1), get_loss, float(loss_sum))   # feedback + next step's march on stream B
    if st.on_gpu and st.bucketed:
        # RCCL over xGMI, fp16 sums of the loss-scaled gradients.  The hash-grid part (24 MB for lego) is final before the MLP weight
        # gradients are computed: its all-reduce starts there on a side stream; the 20 KB MLP part follows on the training stream,
        # which then waits for the side stream — both are complete before Adam
        tb.stream_wait_grid_gradients(st.grid_stream.cuda_stream)
        with torch.cuda.stream(st.grid_stream):
            dist.all_reduce(st.grads[st.n_mlp:])
            st.grid_done.record()
        with torch.cuda.stream(st.train_stream):
            dist.all_reduce(st.grads[:st.n_mlp])
            st.train_stream.wait_event(st.grid_done)
    elif st.on_gpu:
        with torch.cuda.stream(st.train_stream):
            dist.all_reduce(st.grads)
    else:
        dist.all_reduce(st.grads)
    tb.train_nerf_dp_end()
    return c1


def request_rate_roofs(torch, dev):
    """The box's random-gather ceilings, measured in this run (~20 ms): ngp_hip_probe_gather_rate (csrc/probe.hip: independent random 4-byte loads, 8 in flight per lane) on
    a table inside the vector L1's reach (16 KiB), on one 2 MiB slice per XCD (its L2 holds it: the XCD-affine encoder's situation), and on 24 MiB shared by all XCDs (the
    hash table of base.json: Infinity-Cache resident, every L2 miss crosses the fabric).  G gathers / s, chip-wide, best of 3 after a warm-up launch."""
    import ctypes
    import capi
    ngp = capi.load_ngp_hip()
    table = torch.ones(8 << 20, dtype=torch.int32, device=dev)     # 32 MiB
    sink = torch.zeros(16, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for name, n_entries, per_xcd in (("L1", 4096, 0), ("L2", 8 * (1 << 19), 1), ("fabric", 6 << 20, 0)):
        n = ctypes.c_uint64(0)
        best = None
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            capi.check(ngp.ngp_hip_probe_gather_rate(st, table.data_ptr(), n_entries, per_xcd, 4096, 64, rep, sink.data_ptr(), ctypes.addressof(n)))
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            if rep and (best is None or ms < best):
                best = ms
        out[name] = round(n.value / (best * 1e-3) / 1e9, 1)
    return out


def cpu_baseline(tb, ds, res, budget_s=12.0):
    """The oracle (CPU port, OpenMP over the host cores) timed on a BOUNDED sample of the same workload: one real training step of the benchmarked
    Testbed is captured (tests/fullstep.py) and the oracle runs a slice of its rays end to end — DDA march, inference of every sample, loss +
    compaction, forward / backward of the compacted samples — plus one Adam / Ema pass over all parameters; the step rate is the slice's time
    scaled to the whole ray batch.  Second leg: the oracle's renderer on a 96 x 96 crop-resolution frame of a test view (MP/s)."""
    import helpers as H
    import fullstep as F
    orc, orc_build = H.load_oracle_native()   # -O3 -march=native for the host this runs on; contraction off: the results of the portable build
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    tb.shall_train = True
    tb.debug_capture_next_step()
    tb.frame()
    cap = tb.debug_captured()
    S = F.host_scene(tb, ds["train_images"])
    R = int(cap["R"])
    n_rays = max(256, min(R, 64 * cores))          # ~8 s of work on 8 cores, < 1 s on 256 (repeated until the budget is used)
    rate_slice, det = F.timed_cpu_step(orc, S, cap, n_rays, budget_s)
    t_slice = det["seconds"] / det["reps"]
    npar = len(cap["params"])
    g16 = cap["grads"].copy()
    master = cap["params"].view(np.float16).astype(np.float32)
    p16 = cap["params"].copy()
    m1, m2, ema, inf = np.zeros(npar, np.float32), np.zeros(npar, np.float32), np.zeros(npar, np.float32), cap["params"].copy()
    t0 = time.time()
    orc.orc_adam_ema_step(npar, 10240, 1, H.f32(1e-2), H.f32(0.9), H.f32(0.99), H.f32(1e-15), H.f32(1e-6), H.f32(128.0), H.f32(0.95), g16.ctypes.data, master.ctypes.data,
                          p16.ctypes.data, m1.ctypes.data, m2.ctypes.data, ema.ctypes.data, inf.ctypes.data)
    t_adam = time.time() - t0
    kept_full = min(int(cap["measured_batch_size"]), int(cap["target_batch_size"]))
    t_step = t_slice * (R / n_rays) + t_adam
    out = {"value": kept_full / t_step, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": "oracle on %d of the %d rays of one captured 2^18-sample step (march %d samples -> inference -> loss/compaction %d samples -> forward/backward), %d x in %.1f s, "
                     "scaled to the whole batch + one Adam/Ema pass over %.1f M parameters (%.2f s)" % (n_rays, R, det["samples"], det["compacted"], det["reps"], det["seconds"], npar / 1e6, t_adam),
           "stage_seconds": det["stage_seconds"], "cpu_ms_per_step": round(1000.0 * t_step, 1),
           "build": "gcc -O3 -march=native -fopenmp -ffp-contract=off (oracle/Makefile `native`, built on this host)" if orc_build == "native" else "gcc -O2 -msse2 -fopenmp (%s)" % orc_build}
    # ---- render leg: orc_render_nerf (the NerfTracer loop of src/testbed_nerf.cu:2140-2267 on the CPU), EMA weights
    try:
        rr = 96
        sc = S["sc"]
        cam = np.ascontiguousarray(tb.camera_matrix.T.reshape(-1).astype(np.float32))   # 3x4 -> column-major 12 floats
        fl = 0.5 * rr / np.tan(0.5 * ds["camera_angle_x"])
        focal = np.array([fl, fl], np.float32)
        resv = np.array([rr, rr], np.int32)
        scv = np.array([0.5, 0.5], np.float32)
        ident = np.eye(3, dtype=np.float32).reshape(-1)
        frame, depth = np.zeros((rr, rr, 4), np.float32), np.zeros((rr, rr), np.float32)
        inf_params = tb.debug_params("inference")
        t0 = time.time()
        n_net = orc.orc_render_nerf(S["desc"].ctypes.data, inf_params.ctypes.data, 0, resv.ctypes.data, focal.ctypes.data, cam.ctypes.data, cam.ctypes.data, scv.ctypes.data, 1,
                                    S["aabb"].ctypes.data, ident.ctypes.data, S["aabb"].ctypes.data, H.f32(0.0), S["bitfield"].ctypes.data, H.f32(sc["cone_angle_constant"]),
                                    int(sc["rgb_activation"]), int(sc["density_activation"]), H.f32(1e-4), 0, frame.ctypes.data, depth.ctypes.data)
        dt = time.time() - t0
        out["render_MP_per_s"] = round(rr * rr / dt / 1e6, 5)
        out["render_sample"] = "oracle renderer, one %dx%d frame of the last test view (%d network samples), %.2f s" % (rr, rr, int(n_net), dt)
    except Exception as e:   # the training leg is the contract; say why the render leg is missing
        out["render_MP_per_s"] = None
        out["render_sample"] = "render leg failed: %r" % (e,)
    return out


def self_launch(n_gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>` — one process per GPU, LOCAL_RANK -> device, rank 0 prints the
    JSON line.  exec, not spawn: the launcher keeps this PID, so whoever started bench.py still owns (and can time out) the whole job."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    os.environ.setdefault("OMP_NUM_THREADS", "1")              # what torchrun would set (with a warning) anyway
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: exec %s" % (n_gpus, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


def preflight(rank, local_rank, world, dev):
    """`python bench.py --gpus N --preflight`: everything the multi-GPU run needs BEFORE any training — the kernel library, the RCCL it binds, the shared-memory
    rendezvous of Testbed.init_data_parallel, a communicator over all ranks, and one 1 KiB collective of each kind the step uses (all-reduce, fp16 all-to-all, all-gather)
    with its result checked — each stage under a 30 s watchdog that names the stage and the rank and exits, so that a first run on an 8-GPU node fails on plumbing with
    a sentence, not with a hang.  Rank 0 prints one JSON line {"preflight": "ok", ...}; nothing is trained."""
    import ctypes
    import threading
    import numpy as np
    import torch
    for p in (PKG,):
        if p not in sys.path:
            sys.path.insert(0, p)
    stage = {"name": "start", "t0": time.perf_counter()}
    times = {}

    def watchdog():
        while True:
            time.sleep(1.0)
            if stage["name"] == "done":
                return
            if time.perf_counter() - stage["t0"] > 30.0:
                print(json.dumps({"preflight": "failed", "rank": rank, "world": world, "stuck_in": stage["name"], "after_s": 30,
                                  "hint": "a stage that never returns on SOME ranks usually means the others never reached it: read their lines"}), flush=True)
                os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()

    def enter(name):
        now = time.perf_counter()
        if stage["name"] not in ("start",):
            times[stage["name"]] = round((now - stage["t0"]) * 1e3, 2)
        stage["name"], stage["t0"] = name, now

    try:
        enter("load libngp_hip.so")
        import capi
        ngp = capi.load_ngp_hip()
        enter("bind RCCL")
        if not ngp.ngp_rccl_available():
            raise RuntimeError("no librccl could be loaded next to the HIP runtime in use (%s)" % ngp.ngp_hip_last_error().decode())
        enter("Testbed.init_data_parallel (shared-memory rendezvous + ncclCommInitRank)")
        import pyngp
        tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
        key = "preflight_%s" % os.environ.get("MASTER_PORT", "0")
        tb.init_data_parallel(rank, world, key, True)
        if int(tb.dp_comm_size) != world:
            raise RuntimeError("the communicator reports %d ranks, WORLD_SIZE is %d" % (int(tb.dp_comm_size), world))
        enter("Testbed.shutdown_data_parallel")
        tb.shutdown_data_parallel()
        del tb
        # a communicator of this script's own for the 1 KiB collectives: the unique id travels through a file in /dev/shm (no torch.distributed involved)
        enter("unique id exchange (/dev/shm)")
        id_path = "/dev/shm/ngp_preflight_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
        uid = np.zeros(128, np.uint8)
        if rank == 0:
            capi.check(ngp.ngp_rccl_get_unique_id(uid.ctypes.data))
            with open(id_path + ".tmp", "wb") as f:
                f.write(uid.tobytes())
            os.replace(id_path + ".tmp", id_path)
        else:
            while not os.path.exists(id_path):
                time.sleep(0.01)
            uid = np.frombuffer(open(id_path, "rb").read(), np.uint8).copy()
        enter("ngp_rccl_init")
        comm = ngp.ngp_rccl_init(rank, world, uid.ctypes.data)
        if not comm:
            raise RuntimeError("ngp_rccl_init: %s" % ngp.ngp_hip_last_error().decode())
        st = torch.cuda.current_stream().cuda_stream
        enter("all-reduce, 1 KiB fp32")
        x = torch.full((256,), float(rank + 1), device=dev, dtype=torch.float32)
        capi.check(ngp.ngp_rccl_allreduce_f32(comm, st, x.data_ptr(), x.numel()))
        torch.cuda.synchronize()
        if not bool((x == world * (world + 1) / 2).all()):
            raise RuntimeError("all-reduce gave %r, expected %r" % (float(x[0]), world * (world + 1) / 2))
        enter("all-to-all, fp16 slices (grouped ncclSend / ncclRecv)")
        per = 512 // world // 8 * 8 or 8
        send = torch.empty(world * per, device=dev, dtype=torch.float16)
        for q in range(world):
            send[q * per:(q + 1) * per] = float(rank * 16 + q)
        recv = torch.zeros_like(send)
        capi.check(ngp.ngp_rccl_alltoall_f16(comm, st, send.data_ptr(), recv.data_ptr(), per))
        torch.cuda.synchronize()
        want = torch.cat([torch.full((per,), float(q * 16 + rank), device=dev, dtype=torch.float16) for q in range(world)])
        if not torch.equal(recv, want):
            raise RuntimeError("all-to-all: slice from rank %d is wrong" % int((recv != want).nonzero()[0] // per))
        out = torch.zeros(per, device=dev, dtype=torch.float16)
        capi.check(ngp.ngp_hip_sum_slices_f16(st, world, per, recv.data_ptr(), out.data_ptr()))
        enter("all-gather, fp16")
        g = torch.zeros(world * per, device=dev, dtype=torch.float16)
        g[rank * per:(rank + 1) * per] = float(rank + 1)
        capi.check(ngp.ngp_rccl_allgather_f16(comm, st, g.data_ptr(), per))
        torch.cuda.synchronize()
        if not torch.equal(g, torch.cat([torch.full((per,), float(q + 1), device=dev, dtype=torch.float16) for q in range(world)])):
            raise RuntimeError("all-gather: wrong contents")
        enter("ngp_rccl_finalize")
        capi.check(ngp.ngp_rccl_finalize(comm))
        enter("done")
        if rank == 0:
            try:
                os.unlink(id_path)
            except OSError:
                pass
            print(json.dumps({"preflight": "ok", "n_gpus": world, "device": torch.cuda.get_device_name(local_rank), "stages_ms_rank0": times}), flush=True)
    except Exception as e:
        failed = stage["name"]
        stage["name"] = "done"
        print(json.dumps({"preflight": "failed", "rank": rank, "world": world, "stage": failed, "error": str(e)}), flush=True)
        raise SystemExit(3)


def main():
    if os.environ.get("BENCH_HANG_DUMP"):   # dev: dump every thread's stack and exit if the run is still going after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_HANG_DUMP"]), exit=True)
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
    ap.add_argument("--dp_impl", choices=["product", "torch"], default="product", help="product: Testbed.init_data_parallel — the step's two exchanges inside the C++ Testbed (shared-memory "
                    "counters, its own RCCL communicator), frame() as on one GPU; torch: the same exchanges driven from this script through torch.distributed (dp_step). product falls back to torch if it cannot initialise")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong", help="strong (default, SURVEY 8e): 2^18 compacted samples per step over ALL GPUs — the reference's convergence per step; with N > 1 the line also carries a "
                    "`weak_scaling` object measured in the same run (2^18 per GPU and step, N x the global batch).  weak: the weak-scaled run as the headline")
    ap.add_argument("--min_train_step", type=int, default=1000, help="BASELINE.md M1 quotes the metric on steps [1000, 2000): the timed region never starts before this training step, whatever --warmup says")
    ap.add_argument("--legs", default="fox,bl_render,plumbing", help="comma list of the extra legs of a one-GPU run (bench_legs.py): fox = BASELINE config #2 on the fox photographs, bl_render = the Blender "
                    "multi-NeRF renderer next to the stock tracer, plumbing = configs #1 / #5 at 2^18; 'none' switches them off")
    ap.add_argument("--preflight", action="store_true", help="multi-GPU plumbing check only (no training): library, RCCL binding, init_data_parallel's rendezvous, a communicator over all ranks and one "
                    "1 KiB collective of each kind the step uses, every stage under a 30 s watchdog that names the stage and the rank; prints {\"preflight\": \"ok\"} on rank 0")
    ap.add_argument("--scene", default="", help="transforms_train.json of a dataset ON DISK in the nerf-synthetic layout (e.g. .../nerf_synthetic/lego/transforms_train.json; pass the file, not the "
                    "directory: a directory loads every *.json in it, testbed_nerf.cu:2738-2743).  Training data then goes through the product loader (Testbed.load_training_data: mini_json, PNG / JPEG / EXR "
                    "readers, nerf_matrix_to_ngp) instead of the in-memory procedural scene; a file without \"scale\" / \"offset\" / \"aabb\" keys gets \"scale\": 0.33, \"offset\": [0.5, 0.5, 0.5] injected in a "
                    "patched copy (this fork defaults to 1.0 / 0: SURVEY.md fact 5).  `data` becomes \"real\" and config.workload names the path")
    ap.add_argument("--test_scene", default="", help="transforms_test.json for the PSNR gate and the evaluation renders (run.py --test_transforms: run.py:216-303); without it the gate is off and the "
                    "render leg traces training poses")
    ap.add_argument("--psnr_gate", type=float, default=35.0, help="BASELINE config #3 'train to 35 PSNR then render': keep pre-training (untimed) until the held-out PSNR reaches this")
    a = ap.parse_args()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)   # does not return

    import torch  # first: one HIP runtime per process
    import scene

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("rank %d of %d: bench.py needs a GPU (no CPU fallback in the product path)" % (rank, world))
    # NGP_BENCH_LOOPBACK=1 (tests only): all ranks on device 0, the nccl* names served by tests/loopback/librccl_loopback.so (NGP_RCCL_LIBRARY) and this script's own
    # control traffic (barriers, max-over-ranks) on gloo — the N > 1 code path of this file and of the Testbed on a one-GPU box.  Its timings say nothing about scaling.
    loopback = os.environ.get("NGP_BENCH_LOOPBACK") == "1"
    if loopback:
        if not os.environ.get("NGP_RCCL_LIBRARY"):
            raise SystemExit("NGP_BENCH_LOOPBACK=1 needs NGP_RCCL_LIBRARY=<tests/loopback/librccl_loopback.so>: real RCCL refuses two ranks on one device")
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d of %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, world, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if a.preflight:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        preflight(rank, local_rank, world, dev)
        return
    dist = None
    use_dp = world > 1 or a.force_dp
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if loopback:
            dist.init_process_group("gloo")
            if a.dp_impl != "product":
                raise SystemExit("NGP_BENCH_LOOPBACK: only --dp_impl product (the torch driver all-reduces the gradients with torch's own RCCL)")
        elif world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # launched by torch.distributed.run: its env:// rendezvous
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29500 + os.getpid() % 1000), rank=0, world_size=1, device_id=dev)

    ctl_dev = torch.device("cpu") if loopback else dev   # where this script's own control reductions live (gloo reduces host tensors)
    B = 1 << 18
    if a.scene:
        ds = scene.load_disk_dataset(a.scene, a.test_scene or None, max_test=a.n_test, decode_train=(world == 1 and not a.no_cpu_baseline))
        a.n_train = ds["n_train"]
        if not ds["test_images"]:
            a.psnr_gate = 0.0
    else:
        ds = scene.make_dataset(a.n_train, a.n_test, a.res, dev)
    W, Hh = int(ds.get("w", ds["res"])), int(ds.get("h", ds["res"]))   # frame size of the render leg = the dataset's image size (800 x 800 for lego and its stand-in)
    views = ds["test_poses"] if len(ds["test_poses"]) else ds["train_poses"]
    tb = scene.build_testbed(ds)
    if os.environ.get("NGP_BENCH_SIDE_EMA") is not None:   # dev A / B (also read by bench_legs.dev_overrides): 0 = the Ema stage back on the training chain
        tb.ema_on_side_stream = bool(int(os.environ["NGP_BENCH_SIDE_EMA"]))
    if os.environ.get("NGP_BENCH_COMPACT_BWD") is not None:   # dev A / B: 0 = the backward pass over all B slots
        tb.compact_backward = bool(int(os.environ["NGP_BENCH_COMPACT_BWD"]))
    if os.environ.get("BENCH_NO_PREFETCH"):   # dev: the march in stream order, i.e. uncontended (its in-situ stand-alone time)
        tb.prefetch_samples = False
    tb.async_training_steps = True   # frame() without the reference's per-step stream drain (pyngp property; the timed region is still bracketed by syncs)
    tb.set_distributed(rank, world)
    dp = None
    dp_impl = None
    if use_dp and a.dp_impl == "product" and (world > 1 or a.force_dp):
        try:
            tb.init_data_parallel(rank, world, "bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getpid() if world == 1 else 0), a.scaling == "strong")
            ok = 1.0
        except Exception as e:   # e.g. no RCCL to bind, a rendezvous that times out
            print("rank %d: product data-parallel path failed to initialise (%s); using the torch.distributed driver" % (rank, e), file=sys.stderr, flush=True)
            ok = 0.0
            if loopback:
                raise
        t = torch.tensor([ok], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)   # all ranks or none
        if float(t.item()) == 1.0:
            dp_impl = "product"
            tb.render_sharded = True   # every render() of this script is issued by all ranks: rows per rank + RCCL all-gather (opt-in since round 4)
        else:
            tb.shutdown_data_parallel()
            tb.set_distributed(rank, world)
    if use_dp and dp_impl is None:
        if a.scaling == "strong":
            print("rank %d: the torch.distributed driver only knows the weak split; reporting weak scaling" % rank, file=sys.stderr, flush=True)
            a.scaling = "weak"
        dp_impl = "torch"
        grads = torch.as_tensor(CudaArray(tb.gradients_ptr(), tb.n_params(), "<f2"), device=dev)
        assert grads.data_ptr() == tb.gradients_ptr()
        dp = make_dp_state(torch, dist, tb, grads, dev)
    if a.scaling == "strong":
        tb.training_batch_size = B   # per step over all ranks; the Testbed back-propagates B / world per rank

    def one_step():
        if dp is None:
            tb.frame()
            return tb.nerf.training.measured_batch_size
        dp_step(tb, torch, dist, B, dp)
        return tb.nerf.training.measured_batch_size

    for _ in range(a.warmup):
        one_step()
    # ---- untimed pre-training up to the metric's definition: steady state (training_step >= 1000) on a model that passed the 35 dB gate
    pretrain_steps = 0
    while tb.training_step < a.min_train_step:
        one_step()
        pretrain_steps += 1
    psnr_at_bench = None
    if not a.no_render and a.psnr_gate > 0:
        def gate_psnr():
            tb.sync()
            v = scene.eval_test_views(tb, ds, spp=1, max_views=min(a.n_test, 2))[0]
            tb.shall_train = True   # eval_test_views switches training off like run.py does
            if use_dp:              # every rank must take the same decision
                t = torch.tensor([v], dtype=torch.float64, device=ctl_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                v = float(t.item())
            return v
        psnr_at_bench = gate_psnr()
        while psnr_at_bench < a.psnr_gate and tb.training_step < 4000:
            for _ in range(250):
                one_step()
            pretrain_steps += 250
            psnr_at_bench = gate_psnr()
    # Bracketing a launch group with HIP events costs dispatch gaps on the step's critical chain (~5 us per bracket), so only the
    # dominant kernel group — the one the roofline is quoted for — is timed live inside the timed region.  Which one that is, and the
    # per-group table in "kernels", comes from SURVEY_STEPS further untimed steps with every group bracketed.
    tb.set_profiling(True)
    tb.reset_profile()
    for _ in range(SURVEY_STEPS):
        one_step()
    survey = tb.profile()
    dom = max((n for n in survey if n in BYTES_PER_UNIT and survey[n]["launches"]), key=lambda n: survey[n]["ms"])
    # ... and only on every 4th step of the timed region (a bracket is two event records = two ~5 us dispatch gaps on the step's critical chain): the
    # average launch duration the roofline uses is still measured live, over the timed region, on a quarter of its launches
    profile_every = 4 if a.steps >= 16 else 1
    tb.set_profiling(True, [dom], profile_every)
    tb.reset_profile()
    if use_dp:
        dist.barrier()
    torch.cuda.synchronize()
    tb.sync()
    timed_from = tb.training_step
    t0 = time.perf_counter()
    samples = 0
    rays = 0
    pre_compaction = 0
    for _ in range(a.steps):
        rays += tb.nerf.training.rays_per_batch
        samples += min(one_step(), B)
        pre_compaction += tb.nerf.training.measured_batch_size_before_compaction
    tb.sync()
    torch.cuda.synchronize()
    if use_dp:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = tb.profile()
    tb.set_profiling(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([samples, rays, pre_compaction], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(s)
        samples, rays, pre_compaction = (float(x) for x in s.tolist())

    # ---- N > 1, strong-scaled headline: the weak-scaled figure from the same run, beside it (2^18 per rank and step; rays_per_batch re-adapts through the counter feedback)
    weak = None
    if world > 1 and dp_impl == "product" and a.scaling == "strong":
        tb.strong_scaling = False
        for _ in range(96):
            one_step()
        tb.sync(); torch.cuda.synchronize(); dist.barrier()
        t1 = time.perf_counter()
        w_samples = 0
        for _ in range(a.steps):
            w_samples += min(one_step(), B)
        tb.sync(); torch.cuda.synchronize(); dist.barrier()
        w_dt = time.perf_counter() - t1
        t = torch.tensor([w_dt], dtype=torch.float64, device=ctl_dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); w_dt = float(t.item())
        t = torch.tensor([w_samples], dtype=torch.float64, device=ctl_dev); dist.all_reduce(t); w_samples = float(t.item())
        weak = {"value": round(w_samples / w_dt, 1), "unit": "samples/s", "ms_per_step": round(1000.0 * w_dt / a.steps, 4), "steps": a.steps, "global_batch": B * world,
                "note": "same run, after the strong-scaled timed region: 2^18 compacted samples per GPU and step, 96 untimed steps for rays_per_batch to re-adapt"}
        tb.strong_scaling = True

    # ---- render MP/s + PSNR on the trained model, outside the timed region.  With the product's data-parallel communicator every render() is a collective: rank r traces
    # the rows [r * ceil(H / N), ...) of the frame and the rows are all-gathered over RCCL (Testbed::fetch_render_surface), so ALL ranks run this leg
    extra = {}
    render_sharded = dp_impl == "product" and use_dp
    if not a.no_render and (render_sharded or rank == 0):
        if len(ds["test_images"]):
            psnr, ssim, per = scene.eval_test_views(tb, ds, spp=a.eval_spp, max_views=a.n_test)
        else:   # a dataset on disk without test transforms: frames are timed, nothing is scored
            psnr, ssim, per = float("nan"), float("nan"), []
            tb.shall_train = False; tb.background_color = [0.0, 0.0, 0.0, 1.0]; tb.snap_to_pixel_centers = True; tb.nerf.render_min_transmittance = 1e-4
            tb.fov_axis = 0; tb.fov = ds["camera_angle_x"] * 180 / np.pi
        n_frames = 9
        tb.set_nerf_camera_matrix(views[0][:3, :])
        # untimed: the PSNR / SSIM arithmetic above ran on the host for seconds and the GPU's clocks have dropped — the first frames after it take 3-5x as long
        # (r03_d: 28.2 ms, then 5.3-6.0 ms), so the leg warms up with a fixed number of frames before the timed ones, like the training leg's warm-up steps
        n_warm = 12   # (r04_k: with 4 warm-up frames the first timed frame of the driver's short run still took 14 ms — the clocks had not come back up after the host-side PSNR / SSIM seconds)
        for i in range(n_warm):
            tb.set_nerf_camera_matrix(views[i % len(views)][:3, :])
            tb.render(W, Hh, 1, True)
        if render_sharded:
            dist.barrier()
        frame_ms = []
        for i in range(n_frames):
            t1 = time.perf_counter()
            tb.set_nerf_camera_matrix(views[i % len(views)][:3, :])
            tb.render(W, Hh, 1, True)
            frame_ms.append((time.perf_counter() - t1) * 1e3)
        rdt = sum(frame_ms) / n_frames / 1e3
        n_render_samples = float(tb.render_samples_evaluated)
        if render_sharded and world > 1:
            t = torch.tensor([rdt], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rdt = float(t.item())
            t = torch.tensor([n_render_samples], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(t)
            n_render_samples = float(t.item())
        extra = {"render_MP_per_s": round(W * Hh / rdt / 1e6, 2), "render_resolution": [W, Hh], "render_ms_per_frame": round(rdt * 1e3, 2), "render_frames": n_frames, "render_warmup_frames": n_warm, "render_ms_min_max": [round(min(frame_ms), 2), round(max(frame_ms), 2)], "render_ms_frames": [round(x, 2) for x in frame_ms], "render_network_samples_per_frame": int(n_render_samples),
                 "render_ranks": world if render_sharded else 1, "render_rows_per_rank": (Hh + world - 1) // world if render_sharded else Hh,
                 "psnr_db": round(psnr, 2) if per else None, "ssim": round(ssim, 4) if per else None, "psnr_at_step": int(tb.training_step), "eval_views": len(per), "eval_spp": a.eval_spp}

    if rank != 0:
        if use_dp:
            dist.barrier()
            if dp is not None and dp.shm is not None:
                dp.shm.close()
            tb.shutdown_data_parallel()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel, from HIP-event timings taken on the launch stream during the timed region
    bytes_per_unit = BYTES_PER_UNIT

    def table(p):
        out = {}
        for name, e in p.items():
            if e["launches"]:
                k = {"ms_total": round(e["ms"], 3), "launches": int(e["launches"]), "avg_us": round(1000.0 * e["ms"] / e["launches"], 2), "units_per_launch": round(e["units"] / e["launches"], 1)}
                if name in bytes_per_unit:
                    k["algorithmic_GBps"] = round(bytes_per_unit[name] * e["units"] / (e["ms"] * 1e-3) / 1e9, 1)
                out[name] = k
        return out

    kernels = table(survey)            # all groups, from the untimed survey steps
    kernels.update(table(prof))        # the dominant group, from the timed region
    if "optimizer_step" in kernels:
        # 36 B / parameter is the DENSE model (every parameter has a gradient); hash-grid entries nobody touched skip their 24 B of moments and master weight, so the
        # model over-counts and can exceed what the chip copies at.  The figure to read is the counters' (L2 <-> fabric bytes of a separate rocprofv3 --pmc run)
        o = kernels["optimizer_step"]
        o["dense_model_GBps_upper_bound"] = o.pop("algorithmic_GBps")
        try:
            tj0 = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if tj0.get("_meta", {}).get("kernel_set") == KERNEL_SET and tj0.get("optimizer_step"):
                o["counter_bytes_per_launch"] = tj0["optimizer_step"]
                o["counter_GBps"] = round(float(tj0["optimizer_step"]) / (o["avg_us"] * 1e-6) / 1e9, 1)
        except Exception:
            pass
    achieved = kernels[dom]["algorithmic_GBps"]
    # traffic: HBM bytes per launch from the PMC passes of tools/gpu_profile_round.sh — a SEPARATE rocprofv3 run (counters cannot be read inside
    # this process); the file names the run it came from and the kernel set it was taken on.  Stale (other kernel set) => null.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            meta = tj.get("_meta", {})
            if meta.get("kernel_set") == KERNEL_SET:
                traffic = tj.get(dom)
                traffic_source = "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes '%s' on kernel set '%s' (another run of this command on another box); these count L2->fabric bytes (Infinity-Cache hits included), an upper bound on HBM bytes" % (meta.get("tag"), meta.get("kernel_set"))
            else:
                traffic_source = "profiles/pmc_traffic.json is from kernel set '%s', this build is '%s': not quoted" % (meta.get("kernel_set"), KERNEL_SET)
        except Exception:
            traffic = None
    # `traffic` (and `mfma` below) are NOT measured by this run: gpurun refuses counter collection inside a traced run and a --pmc re-exec would triple the run's length.  They are
    # static text from the profile named in traffic_source, quoted only while that profile's kernel_set tag equals this file's KERNEL_SET; `traffic_static` says so in the line.
    roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_static": traffic is not None, "traffic_source": traffic_source, "launch_group": GROUP_KERNELS.get(dom),
                "algorithmic_bytes_per_unit": bytes_per_unit[dom], "units_per_launch": kernels[dom]["units_per_launch"], "avg_launch_us": kernels[dom]["avg_us"]}
    # the longest SINGLE kernel on the step's critical chain that has a byte model: the network pass (one C-ABI call = one kernel, nerf_forward_kernel<2, 0>: fused hash
    # encode + both MLPs over every marched sample).  `roofline` above is quoted for the largest launch GROUP (seven kernels behind one call) as in rounds 1-2; this is the
    # per-kernel view next to it, HIP events on the launch stream over the untimed survey steps (inside the timed region only the dominant group is bracketed).
    if "nerf_inference" in kernels and "algorithmic_GBps" in kernels["nerf_inference"]:
        fk = kernels["nerf_inference"]
        f_traffic = None
        try:
            tj2 = json.load(open(tpath))
            if tj2.get("_meta", {}).get("kernel_set") == KERNEL_SET:
                f_traffic = tj2.get("nerf_inference")
        except Exception:
            pass
        line_single = {"kernel": "nerf_forward_kernel<2,0>", "bound": "hbm", "achieved": fk["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fk["algorithmic_GBps"] / HBM_PEAK_GBS, 4),
                       "traffic": f_traffic, "algorithmic_bytes_per_unit": bytes_per_unit["nerf_inference"], "units_per_launch": fk["units_per_launch"], "avg_launch_us": fk["avg_us"],
                       "measured": "HIP events on the launch stream, %d untimed survey steps" % SURVEY_STEPS}
    else:
        line_single = None
    # ---- the request-rate roof beside the byte roof: a gather is one look-up in the vector L1 wherever it ends up hitting, and
    # profiles/r04_forward_counters.md has the pass at 0.55 look-ups / clock / CU with the L1 90 % busy — that, not a byte figure, is what binds it.  Peaks: measured here.
    if line_single is not None:
        try:
            roofs = request_rate_roofs(torch, dev)
            two_kernel = str(dict(tb.network_pass_report).get("running", "")) == "two_kernel"
            per_sample = 64 if two_kernel else 128   # 16 levels x 8 corners as 4-byte gathers (fused kernel) or x 4 x-pairs as 8-byte gathers (encode_planes_kernel)
            fk = kernels["nerf_inference"]
            ach = per_sample * fk["units_per_launch"] / (fk["avg_us"] * 1e-6) / 1e9
            rr = {"achieved_G_per_s": round(ach, 1), "peak_G_per_s": roofs["L1"], "level": "L1", "frac": round(ach / roofs["L1"], 4), "gathers_per_sample": per_sample,
                  "peaks_G_per_s": roofs, "measured": "ngp_hip_probe_gather_rate in this run (4096 x 256 threads x 64 random 4-byte loads; tables of 16 KiB / 8 x 2 MiB per XCD / 24 MiB shared)",
                  "note": "every gather passes the vector L1's tag look-up (the L1 level's peak); the coarse levels and ray-coherent samples hit there, the fine hashed levels go on to the L2 (peak 'L2') "
                          "and, in the fused kernel where every XCD walks all 16 tables, across the fabric to the Infinity Cache (peak 'fabric')"}
            line_single["request_rate"] = rr
            byte_headroom, req_headroom = 1.0 / max(line_single["frac"], 1e-9), 1.0 / max(rr["frac"], 1e-9)
            line_single["bound"] = "l1_request_rate" if req_headroom < byte_headroom else "hbm"
            line_single["headroom"] = {"bytes_x": round(byte_headroom, 2), "requests_x": round(req_headroom, 2), "bound_is": "the roof with the smaller headroom"}
            if dom == "nerf_inference":
                roofline["request_rate"], roofline["bound"] = rr, line_single["bound"]
            else:   # the backward group: five kernels in a row, none of them a gather stream — its byte fraction is reported, a look-up roof would not describe it
                roofline["bound_note"] = ("launch group of five latency-bound kernels (MFMA chain of the MLP backward, LDS-atomic ranks, scattered 12-byte record stores, ds_add_u64 owners, combine: "
                                          "DESIGN.md 5.2-5.3); `bound` is the byte roof SURVEY 8(d) assigns the hash pass, the request-rate roof measured in this run is quoted for the network pass in "
                                          "roofline_longest_single_kernel, whose bound it is")
        except Exception as e:   # the probe is an extra: say why it is missing and keep the line
            line_single["request_rate"] = {"failed": repr(e)}
    if line_single is not None and line_single.get("traffic"):
        line_single["traffic_over_algorithmic"] = round(line_single["traffic"] / (line_single["algorithmic_bytes_per_unit"] * line_single["units_per_launch"]), 2)
        line_single["traffic_factor"] = "2 x FETCH_SIZE + WRITE_SIZE; the factor 2 holds for this pass's 4-byte and 8-byte gathers too: profiles/r06_fetch_calibration.json (one 128-byte line per touch, tallied at 64)"
    mpath = os.path.join(ROOT, "profiles", "mfma_util.json")
    if os.path.exists(mpath):
        try:
            mj = json.load(open(mpath))
            if mj.get("_meta", {}).get("kernel_set") == KERNEL_SET:
                roofline["mfma"] = {k: v for k, v in mj.items() if not k.startswith("_")}
                roofline["mfma_source"] = "profiles/mfma_util.json (%s)" % mj["_meta"].get("tag")
                roofline["mfma_static"] = True
        except Exception:
            pass
    # the march (stream B, hidden behind the backward): bytes it moves and how many waves it keeps resident
    if "generate_training_samples" in kernels and a.steps:
        g = kernels["generate_training_samples"]
        rays_l, samples_l = rays / a.steps / world, pre_compaction / a.steps / world
        g["algorithmic_GBps"] = round((MARCH_BYTES_PER_SAMPLE * samples_l + MARCH_BYTES_PER_RAY * rays_l) / (g["avg_us"] * 1e-6) / 1e9, 1)
        g["rays_per_launch"] = round(rays_l, 1)
        g["resident_waves"] = int(min((rays_l + 15) // 16, 640) * 4)
        g["note"] = "wave-per-ray march, %d persistent waves (640 4-wave workgroups: 2.5 per CU) on 1024 SIMDs; runs one step ahead on a second stream beside the backward pass" % g["resident_waves"]

    line = {
        "metric": "train samples/s (compacted samples back-propagated per second), " + ("dataset on disk" if a.scene else "nerf-synthetic/lego stand-in"),
        "value": round(samples / dt, 1), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000.0 * dt / a.steps, 4),
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f16", "data": "real" if a.scene else "synthetic",
        "config": {"workload": ("%s: %d frames %dx%d through Testbed.load_training_data (scale %.4g, offset %s%s, aabb_scale %d)" % (ds["source"], a.n_train, W, Hh, ds["scale"] if not ds["scale_offset_injected"] else 0.33,
                                ds["offset"] if not ds["scale_offset_injected"] else [0.5, 0.5, 0.5], " injected: the file has no such keys" if ds["scale_offset_injected"] else "", ds["aabb_scale"])
                                if a.scene else "procedural-lego %dx%d x%d RGBA8 views (scale 0.33, offset 0.5, aabb_scale 1)" % (a.res, a.res, a.n_train))
                               + ", configs/nerf/base.json (L=16 F=2 T=2^19, 64-wide MLPs), batch 2^18 compacted samples per GPU",
                   "scene": ds.get("source"), "test_scene": os.path.abspath(a.test_scene) if a.test_scene else None,
                   "global_batch": B if a.scaling == "strong" else B * world, "parallelism": "dp%d" % world if use_dp else "single", "dp_impl": dp_impl},
        "rays_per_s": round(rays / dt, 1), "pre_compaction_samples_per_s": round(pre_compaction / dt, 1),
        "pretrain_steps": pretrain_steps, "timed_from_training_step": int(timed_from), "psnr_at_bench": None if psnr_at_bench is None else round(psnr_at_bench, 2),
        "roofline": roofline, "roofline_longest_single_kernel": line_single, "kernels": kernels, "kernels_note": "%s: timed region, HIP events around the launches of every %s step; other groups: %d untimed survey steps" % (dom, "4th" if profile_every == 4 else "single", SURVEY_STEPS),
    }
    line.update(extra)
    try:   # what this command gave on other boxes of the pool while it was developed (a static record kept by the builder): the driver's 20-step window is one draw from this
        sp = json.load(open(os.path.join(ROOT, "profiles", "box_spread.json")))
        if sp.get("kernel_set") == KERNEL_SET and sp.get("ms_per_step"):
            v = sorted(sp["ms_per_step"])
            line["ms_per_step_other_boxes"] = {"min": v[0], "max": v[-1], "runs": len(v), "static": True, "source": "profiles/box_spread.json (%s)" % sp.get("note", "")}
    except Exception:
        pass
    line["network_pass"] = dict(tb.network_pass_report)   # which organisation of the network pass the Testbed measured faster on this workload and runs (Testbed.network_pass = 'auto')
    if weak is not None:
        line["weak_scaling"] = weak
    if use_dp:   # what the communicator itself says, and what the step's exchanges cost (HIP events on the training stream, survey steps)
        dp_info = {"impl": dp_impl, "world_size_env": world, "loopback_on_one_device": loopback, "rccl_comm_ranks": int(tb.dp_comm_size) if dp_impl == "product" else int(dist.get_world_size()),
                   "sharded_optimizer": bool(tb.dp_sharded_optimizer) if dp_impl == "product" else False}
        for k_name, label in (("grad_exchange", "grad_exchange_us_per_step"), ("param_gather", "param_gather_us_per_step")):
            if k_name in kernels:
                dp_info[label] = kernels[k_name]["avg_us"]
        dp_info["exchange_us_per_step"] = round(sum(dp_info.get(k, 0.0) for k in ("grad_exchange_us_per_step", "param_gather_us_per_step")), 2)
        line["data_parallel"] = dp_info
    if world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(tb, ds, W)
    # ---- the other configs BASELINE names and the fork's own renderer, measured inside this run (bench_legs.py); a leg that fails says why and does not take the line down
    legs = [x for x in a.legs.split(",") if x and x != "none"] if world == 1 else []
    if legs:
        import bench_legs
        import gc
        tb.set_profiling(False)
        # The legs that build their own Testbed run after this one is gone: HIP maps a process's streams onto 4 hardware queues, and with this Testbed's two streams alive
        # the next Testbed's training stream and its run-ahead stream land on ONE queue — its march then runs inside the chain instead of beside it (fox 0.72 -> 0.79 ms per
        # step, tools/fox_ab_probe.py + tools/fox_timeline.sh).  bl_render needs this Testbed's model and goes first.
        legs.sort(key=lambda x: x != "bl_render")
        for leg in legs:
            t_leg = time.perf_counter()
            if leg != "bl_render" and tb is not None and not use_dp:
                tb.sync()
                tb = None
                gc.collect()
            try:
                if leg == "fox":
                    line["fox"] = bench_legs.fox_leg(max(a.steps, 100), BYTES_PER_UNIT, a.min_train_step)
                elif leg == "bl_render" and not a.no_render and len(ds["test_poses"]):
                    line["bl_render"] = bench_legs.bl_render_leg(tb, ds, W)
                elif leg == "plumbing":
                    line["plumbing"] = bench_legs.plumbing_leg()
                else:
                    continue
            except Exception as e:
                line[leg] = {"failed": repr(e)}
            line[leg]["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    print(json.dumps(line), flush=True)
    if use_dp:
        dist.barrier()
        if dp is not None and dp.shm is not None:
            dp.shm.close()
        tb.shutdown_data_parallel()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
