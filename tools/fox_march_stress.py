"""dev tool: hunts a rare device / oracle disagreement of the cone-stepping march's sample counter on the fox photographs (seen once in ~12 runs of
tests/test_baseline_configs_gpu.py::test_fox_photographs_...).  Trains N steps, captures one step, replays the march through the oracle; on a mismatch prints which rays differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa
import pyngp, scene
import helpers as H, fullstep as F
FOX = os.path.join(ROOT, "tests", "golden", "_generated", "fox", "transforms.json")
orc = H.load_oracle()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tb = pyngp.Testbed(pyngp.TestbedMode.Nerf)
tb.load_training_data(FOX)
tr = tb.nerf.training
imgs = None
bad = 0
for it in range(iters):
    tb.reload_network_from_file(os.path.join(ROOT, "blender-ngp_amd", "configs", "nerf", "base.json"))
    tb.shall_train = True
    scene.train(tb, (300 if iters == 1 else 290 + it))
    if imgs is None:
        imgs = [np.ascontiguousarray(tr.get_image_rgba8(i)) for i in range(len(list(tr.paths)))]
    tb.debug_capture_next_step(); tb.frame()
    cap = tb.debug_captured()
    S = F.host_scene(tb, imgs)
    r = F.oracle_march(orc, S, cap)
    g = F.device_march(cap)
    nr, ng = int(r["nc"][0]), int(g["nc"][0])
    ok = nr == ng
    print("iter %d step %d: R %d max_inference %d  samples oracle %d device %d  rays oracle %d device %d  %s" % (it, int(cap["step"]), int(cap["R"]), int(cap["max_inference"]), nr, ng, int(r["rc"][0]), int(g["rc"][0]), "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        bad += 1
        n_ref, n_got = int(r["rc"][0]), int(g["rc"][0])
        ref = {int(r["idx"][k]): int(r["ns"][2 * k]) for k in range(n_ref)}
        got = {int(g["idx"][k]): int(g["ns"][2 * k]) for k in range(n_got)}
        only_r, only_g = sorted(set(ref) - set(got)), sorted(set(got) - set(ref))
        diff = [(k, ref[k], got[k]) for k in ref if k in got and ref[k] != got[k]]
        print("  rays only in oracle %d (samples %d), only on device %d (samples %d), kept by both with different counts %d: %s" % (len(only_r), sum(ref[k] for k in only_r), len(only_g), sum(got[k] for k in only_g), len(diff), diff[:12]))
        print("  sum over oracle-kept %d, over device-kept %d" % (sum(ref.values()), sum(got.values())))
print("mismatches: %d of %d" % (bad, iters))
