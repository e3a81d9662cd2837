"""dev tool: distribution of (marched samples, kept samples) per ray in one steady-state training step of the bench scene — how much of the
pre-compaction network pass is spent on samples behind a ray's termination (T < 1e-4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import scene
dev = torch.device("cuda", 0)
ds = scene.make_dataset(100, 2, 800, dev)
tb = scene.build_testbed(ds)
scene.train(tb, int(sys.argv[1]) if len(sys.argv) > 1 else 1200)
tb.debug_capture_next_step()
tb.frame()
cap = tb.debug_captured()
n_rays = int(cap["gen_counters"][0])
ns = cap["numsteps"].reshape(-1, 2)[:n_rays].astype(np.int64)
sig = cap["mlp_out"].view(np.float16).reshape(-1, 4)[:, 3].astype(np.float64)
dtw = cap["coords"].reshape(-1, 7)[:, 3].astype(np.float64)
MIN_STEP = 1.73205080757 / 1024
kept = np.zeros(n_rays, np.int64)
for i in range(n_rays):
    n, b = ns[i]
    dt = dtw[b:b + n] * (MIN_STEP * 128 - MIN_STEP) + MIN_STEP
    T_before = np.concatenate([[1.0], np.cumprod(np.exp(-np.exp(sig[b:b + n]) * dt))[:-1]])
    bad = np.nonzero(T_before < 1e-4)[0]
    kept[i] = bad[0] if len(bad) else n
tot, k = ns[:, 0].sum(), kept.sum()
print("rays %d  marched %d (%.1f / ray)  kept %d (%.1f / ray)  ratio %.3f" % (n_rays, tot, tot / n_rays, k, k / n_rays, tot / k))
for g in (8, 16, 32, 64):
    ev = np.minimum(ns[:, 0], (kept + g - 1) // g * g + (0 if g else 0))
    ev = np.minimum(ns[:, 0], np.maximum((kept + g) // g * g, g))   # evaluate in groups of g until a group ends terminated (needs one sample past the last kept to know)
    print("granularity %3d: evaluate %d samples = %.3f of marched" % (g, ev.sum(), ev.sum() / tot))
print("histogram of marched per ray:", np.histogram(ns[:, 0], bins=[0, 1, 8, 16, 32, 64, 128, 256, 1025])[0].tolist())
print("histogram of kept per ray:   ", np.histogram(kept, bins=[0, 1, 8, 16, 32, 64, 128, 256, 1025])[0].tolist())
full = kept == ns[:, 0]
print("rays that keep every sample: %d (%.1f %%), their samples: %d" % (full.sum(), 100.0 * full.mean(), ns[full, 0].sum()))
# two-pass schemes: pass 1 evaluates the first K samples of every ray, pass 2 the rest of the rays whose optical depth after K samples has not passed -ln(1e-4) + 0.5
tau_stop = -np.log(1e-4) + 0.5
for K in (32, 48, 64, 96):
    ev = 0
    for i in range(n_rays):
        n, b = ns[i]
        dt = dtw[b:b + n] * (MIN_STEP * 128 - MIN_STEP) + MIN_STEP
        tau = np.cumsum(np.exp(sig[b:b + n]) * dt)
        first = min(n, K)
        ev += first
        if n > K and tau[K - 1] < tau_stop:
            ev += n - K
    print("two passes, K = %3d: evaluate %d samples = %.3f of marched" % (K, ev, ev / tot))
