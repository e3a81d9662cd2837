"""dev tool: hunts a rare per-ray disagreement of the compacted sample count (loss kernel vs oracle) OFF the T < 1e-4 knife edge on the fox photographs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa
import pyngp, scene
import helpers as H, fullstep as F
FOX = os.path.join(ROOT, "tests", "golden", "_generated", "fox", "transforms.json")
orc = H.load_oracle()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
tb.load_training_data(FOX)
tb.reload_network_from_file(os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json"))
tb.shall_train = True
tr = tb.nerf.training
scene.train(tb, 250)
imgs = [np.ascontiguousarray(tr.get_image_rgba8(i)) for i in range(len(list(tr.paths)))]
B = 1 << 18
bad = 0
for it in range(iters):
    tb.debug_capture_next_step(); tb.frame()
    cap = tb.debug_captured()
    S = F.host_scene(tb, imgs)
    n_rays = int(cap["gen_counters"][0])
    o = F.oracle_loss(orc, S, cap)
    border = F.borderline_rays(cap)
    gns, ons = cap["numsteps_compacted"], o["ns"]
    mism = [i for i in range(n_rays) if int(ons[2 * i]) != int(gns[2 * i]) and i not in border and int(ons[2 * i + 1]) + int(ons[2 * i]) < B and int(gns[2 * i + 1]) + int(gns[2 * i]) < B]
    print("step %d: %d rays, %d borderline, %d mismatches off the edge" % (int(cap["step"]), n_rays, len(border), len(mism)), flush=True)
    for i in mism[:2]:
        bad += 1
        ns = cap["numsteps"]
        n, b = int(ns[2 * i]), int(ns[2 * i + 1])
        out = cap["mlp_out"].view(np.float16).reshape(-1, 4)[b:b + n].astype(np.float64)
        co = cap["coords"].reshape(-1, 7)[b:b + n]
        print("  slot %d: marched %d base %d, oracle keeps %d (base %d), device keeps %d (base %d); density_activation %s rgb_activation %s" % (i, n, b, int(ons[2 * i]), int(ons[2 * i + 1]), int(gns[2 * i]), int(gns[2 * i + 1]), S["sc"]["density_activation"], S["sc"]["rgb_activation"]))
        print("  sigma", np.round(out[:, 3], 3).tolist())
        print("  dt_w ", np.round(co[:, 3].astype(np.float64), 5).tolist())
        print("  pos0 ", co[0, :3].tolist(), "dir", co[0, 4:7].tolist())
        print("  " + F.describe_compaction_mismatch(cap, ons, i))
print("mismatching rays: %d over %d steps" % (bad, iters))
