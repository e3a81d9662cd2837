"""dev tool: single-GPU training at the per-rank batch sizes strong scaling produces (2^18 / 8 = 2^15 ...).  usage: python tools/small_batch_probe.py B [steps]"""
import sys, os, faulthandler
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
sys.path[:0] = [os.path.join(sys.path[0], "blender-ngp_amd"), os.path.join(sys.path[0], "tests")]
faulthandler.dump_traceback_later(50, exit=True)
import numpy as np, torch, scene
dev = torch.device("cuda", 0)
B = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ds = scene.make_dataset(20, 1, 200, dev)
tb = scene.build_testbed(ds)
tb.training_batch_size = B
tb.async_training_steps = True
for k in range(steps):
    tb.frame()
    if k % 50 == 0:
        print("B", B, "step", tb.training_step, "loss", tb.loss, "rays", tb.nerf.training.rays_per_batch, "measured", tb.nerf.training.measured_batch_size, tb.nerf.training.measured_batch_size_before_compaction, flush=True)
tb.sync()
psnr, ssim, _ = scene.eval_test_views(tb, ds, spp=1)
print("B", B, "done: loss", tb.loss, "psnr", round(psnr, 2), flush=True)
