"""dev tool: PSNR / SSIM on held-out views of the procedural lego stand-in as training proceeds (run.py's evaluation protocol, 4 views, 4 spp).

    python tools/quality_curve.py [steps ...]      default 250 500 1000 2000 5000 10000 20000 35000
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd")]
import torch
import scene

steps = [int(x) for x in sys.argv[1:]] or [250, 500, 1000, 2000, 5000, 10000, 20000, 35000]
dev = torch.device("cuda", 0)
ds = scene.make_dataset(100, 4, 800, dev)
tb = scene.build_testbed(ds)
rows = []
t_train = 0.0
for stop in steps:
    tb.shall_train = True
    t0 = time.perf_counter()
    scene.train(tb, stop)
    tb.sync()
    t_train += time.perf_counter() - t0
    psnr, ssim, per = scene.eval_test_views(tb, ds, spp=4, max_views=4)
    rows.append({"step": stop, "train_seconds": round(t_train, 2), "psnr_db": round(psnr, 2), "ssim": round(ssim, 4), "loss": float(tb.loss)})
    print(rows[-1], flush=True)
    # eval changed render settings only; training state is untouched
print(json.dumps(rows))
