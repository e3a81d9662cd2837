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
if os.environ.get("NGP_RENDER_FACTOR"):
    tb.nerf.render_pass_samples_factor = float(os.environ["NGP_RENDER_FACTOR"])
if os.environ.get("NGP_RENDER_SKIPS"):
    tb.nerf.render_max_skips_per_pass = int(os.environ["NGP_RENDER_SKIPS"])
tb.render(res, res, 1, True)
t0 = time.perf_counter()
for _ in range(n):
    tb.render(res, res, 1, True)
dt = (time.perf_counter() - t0) / n
print("psnr %.2f  eval wall per frame %.2f ms" % (psnr, dt * 1e3), per)
if os.environ.get("NGP_PROBE_ONLY"):   # tools/render_kstats.sh: exactly 1 evaluation frame + 1 warm frame + n timed frames were rendered
    print("frames_rendered %d  wall per frame %.2f ms  network samples %d" % (n + 2, dt * 1e3, tb.render_samples_evaluated))
    sys.exit(0)
variants = ((True, True),) if os.environ.get("NGP_PROBE_PLAIN") else ((False, False), (True, False), (False, True), (True, True), (False, False), (True, True))
for tile, fused in variants:
    tb.nerf.render_fused_compaction = fused
    tb.nerf.render_tile_order = tile
    tb.render(res, res, 1, True)
    t0 = time.perf_counter()
    for _ in range(n):
        tb.render(res, res, 1, True)
    dt = (time.perf_counter() - t0) / n
    print("tile_order %s fused_compaction %s: %.2f ms / frame, %d network samples" % (tile, fused, dt * 1e3, tb.render_samples_evaluated))

if os.environ.get("NGP_PROBE_FUSED"):   # the network pass: XCD-affine encode + MLP kernel vs the fused kernel, alternating on one model
    tb.nerf.render_fused_compaction = True
    tb.nerf.render_tile_order = True
    for fused in (False, True, False, True):
        tb.nerf.render_fused_network = fused
        tb.render(res, res, 1, True)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            tb.render(res, res, 1, True)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("fused network pass %s: median %.2f ms  min %.2f ms / frame, %d network samples" % (fused, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, tb.render_samples_evaluated))
    tb.nerf.render_fused_network = False

if os.environ.get("NGP_PROBE_SWEEP"):   # one model, many tracer settings: "factor:skips:cap,..."
    tb.nerf.render_fused_compaction = True
    tb.nerf.render_tile_order = True
    for item in os.environ["NGP_PROBE_SWEEP"].split(","):
        f, k, c = item.split(":")
        tb.nerf.render_pass_samples_factor = float(f)
        tb.nerf.render_max_skips_per_pass = int(k)
        tb.nerf.render_max_steps_per_pass = int(c)
        tb.render(res, res, 1, True)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            tb.render(res, res, 1, True)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("factor %s skips %s cap %s: median %.2f ms  min %.2f ms / frame, %d network samples" % (f, k, c, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, tb.render_samples_evaluated))
