"""dev tool: host-side phase timing of bench.dp_step on one rank (world 1, RCCL) — where does the host block?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd")]
import torch
import torch.distributed as dist
import bench, scene

dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29613", rank=0, world_size=1, device_id=dev)
B = 1 << 18
ds = scene.make_dataset(100, 1, 800, dev)
tb = scene.build_testbed(ds)
tb.set_distributed(0, 1)
grads = torch.as_tensor(bench.CudaArray(tb.gradients_ptr(), tb.n_params(), "<f2"), device=dev)
st = bench.make_dp_state(torch, dist, tb, grads, dev)
for _ in range(500):
    bench.dp_step(tb, torch, dist, B, st)
names = ["prep", "begin", "h2d+allreduce(scratch)", "d2h", "backward(feedback+prefetch)", "allreduce(grads)", "end"]
acc = [0.0] * len(names)
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
tb.sync(); torch.cuda.synchronize()
T0 = time.perf_counter()
N = 300
for _ in range(N):
    t = [time.perf_counter()]
    step = tb.training_step
    if step % min(max(step // 16, 1), 16) == 0:
        tb.training_prep_nerf(B)
    t.append(time.perf_counter())
    c0, c1 = tb.train_nerf_dp_begin(B, False)
    t.append(time.perf_counter())
    if mode != "noscratch":
        st.host[0], st.host[1], st.host[2] = c0, c1, 0.0
        with st.on_ctl_stream():
            st.scratch.copy_(st.host, non_blocking=True)
            dist.all_reduce(st.scratch, group=st.ctl_group)
    t.append(time.perf_counter())
    if mode != "noscratch":
        with st.on_ctl_stream():
            st.host.copy_(st.scratch)
        n0, n1, _ = st.host.tolist()
    else:
        n0, n1 = c0, c1
    t.append(time.perf_counter())
    tb.train_nerf_dp_backward(B, int(n0), int(n1), False, 0.0)
    t.append(time.perf_counter())
    if mode != "nograds":
        with st.on_comm_stream():
            dist.all_reduce(st.grads)
    t.append(time.perf_counter())
    tb.train_nerf_dp_end()
    t.append(time.perf_counter())
    for i in range(len(names)):
        acc[i] += t[i + 1] - t[i]
tb.sync(); torch.cuda.synchronize()
total = time.perf_counter() - T0
print("mode", mode, "ms/step %.4f" % (total / N * 1e3))
for n, a in zip(names, acc):
    print("  %-30s %8.1f us" % (n, a / N * 1e6))
dist.destroy_process_group()
