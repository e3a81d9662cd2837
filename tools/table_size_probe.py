"""dev tool: held-out PSNR after N steps on a small scene as a function of log2_hashmap_size (same seed, same data)."""
import json, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import torch
import scene
import pyngp
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 10
res = int(sys.argv[3]) if len(sys.argv) > 3 else 96
ds = scene.make_dataset(n_train=n_train, n_test=2, res=res, device=dev)
base = json.load(open(os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json")))
for log2 in (14, 15, 16, 17, 18, 19, 20, 21):
    cfg = json.loads(json.dumps(base)); cfg["encoding"]["log2_hashmap_size"] = log2
    t = scene.build_testbed(ds, network_config=cfg) if "network_config" in scene.build_testbed.__code__.co_varnames else None
    if t is None:
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False, dir=os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf")) as f:
            json.dump(cfg, f); path = f.name
        t = scene.build_testbed(ds); t.reload_network_from_file(path); os.unlink(path)
    scene.train(t, steps); t.sync()
    psnr, ssim, _ = scene.eval_test_views(t, ds, spp=1)
    print("log2_hashmap_size %2d: %9d params  loss %.5f  held-out %.2f dB" % (log2, t.n_params(), t.loss, psnr), flush=True)
