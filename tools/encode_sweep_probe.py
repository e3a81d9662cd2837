"""dev (GPU box): the two-kernel network pass (XCD-affine encode + MLP kernel) timed on the samples of a REAL fox training step and on 2^18 random points (the SDF batch), for build
variants of the encoder's work-item constants (NGP_ENC_CHUNK / NGP_ENC_CLAIM / NGP_ENC_BLOCKS: tools/gpu_r05_enc.sh).  argv[1] = capture | time; the capture is written once by the product build."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import capi, helpers as H
CAP = "/tmp/fox_step_capture.npz"
dev = torch.device("cuda:0")
if sys.argv[1] == "capture":
    import pyngp, bench_legs
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
    tb.load_training_data(bench_legs.FOX); tb.reload_network_from_file(os.path.join(bench_legs.CFG, "nerf", "base.json"))
    tb.async_training_steps = True; tb.shall_train = True
    while tb.training_step < 1300: tb.frame()
    tb.debug_capture_next_step(); tb.frame()
    cap = tb.debug_captured()
    desc = np.frombuffer(tb.debug_scene()["desc"].tobytes(), dtype=H.NET_DESC).copy()
    np.savez(CAP, coords=np.asarray(cap["coords"]), params=np.asarray(cap["params"]), desc=desc.view(np.uint8), n=int(cap["max_inference"]))
    print("captured", int(cap["max_inference"]), "samples")
    sys.exit(0)
z = np.load(CAP)
ngp = capi.load_ngp_hip()
n = int(z["n"])
desc = z["desc"].view(H.NET_DESC)
d_desc, d_p, d_c = H.to_dev(desc, dev), H.to_dev(np.ascontiguousarray(z["params"]), dev), H.to_dev(np.ascontiguousarray(z["coords"]), dev)
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / iters
out, xs = H.dev_zeros(n * 8, dev), H.dev_zeros(n * 64, dev)
wsb = int(ngp.ngp_hip_nerf_encode_workspace_bytes(max(n, 1 << 24))); ws = H.dev_zeros(wsb, dev)
res = {"fox_forward_ws_us": round(timeit(lambda: capi.check(ngp.ngp_hip_nerf_forward_ws(st, d_desc.data_ptr(), d_p.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr(), ws.data_ptr(), wsb, None))), 1), "n": n}
res["fox_forward_fused_us"] = round(timeit(lambda: capi.check(ngp.ngp_hip_nerf_forward(st, d_desc.data_ptr(), d_p.data_ptr(), d_c.data_ptr(), 7, n, out.data_ptr(), 4, xs.data_ptr(), None))), 1)
rs = np.random.RandomState(0)
for label, m in (("random_2^18", 1 << 18), ("random_1M", 1 << 20), ("random_3.1M", 3145728), ("random_8.3M", 8294400)):
    pts = rs.rand(m, 3).astype(np.float32); d_pts = H.to_dev(pts, dev); o0 = H.dev_zeros(m * 2, dev)
    res["density_ws_%s_us" % label] = round(timeit(lambda: capi.check(ngp.ngp_hip_nerf_density_ws(st, d_desc.data_ptr(), d_p.data_ptr(), d_pts.data_ptr(), 3, m, o0.data_ptr(), ws.data_ptr(), wsb, None)), 20), 1)
print(json.dumps(res))
