"""dev tool: train a little, then render N frames; wall time per frame (for rocprofv3 --kernel-trace --stats runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd")]
import torch
import scene
dev = torch.device("cuda", 0)
ds = scene.make_dataset(100, 2, 800, dev)
tb = scene.build_testbed(ds)
scene.train(tb, int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
psnr, ssim, per = scene.eval_test_views(tb, ds, spp=1, max_views=1)
res = ds["res"]
if os.environ.get("NGP_RENDER_STREAMS"):
    tb.nerf.render_n_streams = int(os.environ["NGP_RENDER_STREAMS"])
if os.environ.get("NGP_RENDER_CAP"):
    tb.nerf.render_max_steps_per_pass = int(os.environ["NGP_RENDER_CAP"])
tb.render(res, res, 1, True)
t0 = time.perf_counter()
for _ in range(n):
    tb.render(res, res, 1, True)
dt = (time.perf_counter() - t0) / n
print("psnr %.2f  eval wall per frame %.2f ms" % (psnr, dt * 1e3), per)
for tile, fused in ((False, False), (True, False), (False, True), (True, True), (False, False), (True, True)):
    tb.nerf.render_fused_compaction = fused
    tb.nerf.render_tile_order = tile
    tb.render(res, res, 1, True)
    t0 = time.perf_counter()
    for _ in range(n):
        tb.render(res, res, 1, True)
    dt = (time.perf_counter() - t0) / n
    print("tile_order %s fused_compaction %s: %.2f ms / frame, %d network samples" % (tile, fused, dt * 1e3, tb.render_samples_evaluated))
