"""dev (GPU box): what does the run-ahead march cost the kernel it runs beside?  On a trained lego stand-in: the march (wave-per-ray, throttled and unthrottled), the network pass
of a training step (fused forward over the step's own samples) and the step's backward pass, each alone and in pairs on two streams.  The step today runs march || backward; the
question is what march || forward would cost (the forward is bound by L1 look-ups, the march by VALU issue)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
import capi, helpers as H, scene
from capi import check
dev = torch.device("cuda:0")
ds = scene.make_dataset(100, 1, 800, dev)
tb = scene.build_testbed(ds)
tb.async_training_steps = True
scene.train(tb, 1100)
tb.sync()
tb.debug_capture_next_step()
tb.frame()
tb.sync()
cap = tb.debug_captured()
P = tb.debug_pointers()
ngp = capi.load_ngp_hip()
R = int(P["rays_per_batch"])
n_fwd = min(int(cap["gen_counters"][1]), int(cap["max_inference"])) // 256 * 256
B = int(cap["target_batch_size"])
print("rays %d, samples of the step %d, compacted batch %d" % (R, n_fwd, B), flush=True)
aabb = H.unit_aabb()
max_samples = 16 << 18
bufs = dict(rc=H.dev_zeros(4, dev), nc=H.dev_zeros(4, dev), idx=H.dev_zeros(R * 4, dev), rays=H.dev_zeros(R * 24, dev), ns=H.dev_zeros(R * 8, dev), co=H.dev_zeros(max_samples * 28, dev))
dist = H.dev_zeros(32 * 32 * 2 * 4, dev)
dres = np.array([32, 32], np.int32)
d_coords = H.to_dev(np.ascontiguousarray(cap["coords"]), dev)
d_cc, d_x, d_dl = H.to_dev(np.ascontiguousarray(cap["coords_compacted_rolled"]), dev), H.to_dev(np.ascontiguousarray(cap["x_saved"]), dev), H.to_dev(np.ascontiguousarray(cap["dloss_rolled"]), dev)
out, xs = H.dev_zeros(n_fwd * 8, dev), H.dev_zeros(n_fwd * 64, dev)
desc_host = np.frombuffer(tb.debug_scene()["desc"].tobytes(), dtype=H.NET_DESC).copy()
sb = ngp.ngp_hip_nerf_backward_scratch_bytes_for(desc_host.ctypes.data, B)
scratch, grads = H.dev_zeros(sb, dev), H.dev_zeros(H.n_params(desc_host) * 2, dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def march(stream, mode):
    bufs["rc"].zero_(); bufs["nc"].zero_()
    check(ngp.ngp_hip_generate_training_samples(stream.cuda_stream, R, aabb.ctypes.data, max_samples, int(P["rng_state"]), int(P["rng_inc"]), bufs["rc"].data_ptr(), bufs["nc"].data_ptr(),
                                                bufs["idx"].data_ptr(), bufs["rays"].data_ptr(), bufs["ns"].data_ptr(), bufs["co"].data_ptr(), int(P["n_images"]), int(P["metadata"]),
                                                int(P["xforms"]), int(P["bitfield"]), 0, None, 0, 0, H.f32(P["cone_angle_constant"]), dist.data_ptr(), dres.ctypes.data, 0, R, None, None, mode))
def forward(stream):
    check(ngp.ngp_hip_nerf_forward(stream.cuda_stream, int(P["desc"]), int(P["params"]), d_coords.data_ptr(), 7, n_fwd, out.data_ptr(), 4, xs.data_ptr(), None))
def backward(stream):
    check(ngp.ngp_hip_nerf_backward(stream.cuda_stream, int(P["desc"]), desc_host.ctypes.data, int(P["params"]), d_cc.data_ptr(), 7, B, d_x.data_ptr(), d_dl.data_ptr(), 4, grads.data_ptr(), scratch.data_ptr(), sb, None, None, None, None))

def timed(jobs, iters=20):
    """jobs: list of (stream, fn).  Every stream runs its job `iters` times back to back; returns the per-iteration time of each stream (us)."""
    for st, fn in jobs:
        with torch.cuda.stream(st):
            for _ in range(3):
                fn(st)
    torch.cuda.synchronize()
    ev = []
    for st, fn in jobs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev.append((e0, e1))
    for (st, fn), (e0, e1) in zip(jobs, ev):
        with torch.cuda.stream(st):
            e0.record(st)
    for k in range(iters):                 # interleave the launches so that both queues stay fed
        for st, fn in jobs:
            with torch.cuda.stream(st):
                fn(st)
    for (st, fn), (e0, e1) in zip(jobs, ev):
        with torch.cuda.stream(st):
            e1.record(st)
    torch.cuda.synchronize()
    return [round(1000 * e0.elapsed_time(e1) / iters, 1) for e0, e1 in ev]

with torch.cuda.stream(sB):
    pass
print("alone: forward %s  backward %s  march (all workgroups) %s  march (throttled, 640 workgroups) %s" % (timed([(sA, forward)]), timed([(sA, backward)]), timed([(sB, lambda s: march(s, 2))]), timed([(sB, lambda s: march(s, 3))])), flush=True)
for name, fn in (("forward", forward), ("backward", backward)):
    for mode, mname in ((3, "throttled"), (2, "all workgroups")):
        a, b = timed([(sA, fn), (sB, lambda s: march(s, mode))])
        print("%s || march (%s): %s %.1f us, march %.1f us per iteration (both streams back to back: the slower stream runs partly alone)" % (name, mname, name, a, b), flush=True)
