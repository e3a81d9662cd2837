"""Procedural stand-in for nerf-synthetic/lego (SURVEY.md §8d input S1) + the run.py-style driver around `pyngp`.

The real lego dataset is not in the reference tree and there is no network, so the benchmark scene is generated here:
100 train + 20 test cameras on the upper hemisphere (radius 4.0311 NeRF units, camera_angle_x = 0.6911112, 800x800, RGBA8),
dataset keys "scale": 0.33, "offset": [0.5, 0.5, 0.5], "aabb_scale": 1, and an analytic scene of 8 coloured boxes / spheres
inside |x| < 1.2 with density 64 inside.  Ground truth images are produced by an exact analytic volume integrator (piecewise
constant density along each ray) written with torch tensor ops — input synthesis only, never part of the measured hot path.

The driver mirrors scripts/run.py: `while testbed.frame()` until training_step >= n_steps (run.py:187-210), then the
test-view PSNR / SSIM loop (run.py:216-303) via metrics.eval_psnr_ssim.
"""
import math
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

CAMERA_ANGLE_X = 0.6911112
RADIUS = 4.0311
SIGMA = 64.0

# (kind, centre, half-size or radius, linear base colour)
PRIMITIVES = [
    ("box", (0.0, 0.0, -0.35), (0.95, 0.6, 0.12), (0.75, 0.55, 0.10)),
    ("box", (-0.45, 0.0, 0.10), (0.22, 0.45, 0.30), (0.80, 0.10, 0.08)),
    ("box", (0.45, 0.15, 0.05), (0.25, 0.25, 0.25), (0.10, 0.35, 0.80)),
    ("sphere", (0.0, -0.25, 0.25), 0.28, (0.10, 0.70, 0.20)),
    ("sphere", (0.55, -0.35, -0.05), 0.16, (0.85, 0.80, 0.15)),
    ("box", (0.0, 0.42, 0.45), (0.55, 0.08, 0.08), (0.60, 0.60, 0.65)),
    ("sphere", (-0.55, 0.40, 0.55), 0.14, (0.70, 0.15, 0.70)),
    ("box", (0.75, 0.0, 0.45), (0.07, 0.45, 0.07), (0.15, 0.65, 0.65)),
]


def camera_poses(n, seed, elev_range=(0.15, 1.35)):
    """c2w 4x4 matrices in the NeRF / Blender convention (x right, y up, camera looks along -z), looking at the origin."""
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        az = 2.0 * math.pi * (i + rs.rand() * 0.5) / n
        el = rs.uniform(*elev_range)
        pos = RADIUS * np.array([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)])
        fwd = -pos / np.linalg.norm(pos)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, -fwd, pos
        out.append(m.astype(np.float32))
    return out


def render_ground_truth(c2w, w, h, device):
    """Analytic RGBA8 (straight alpha, sRGB) image of the primitive scene, plus the premultiplied-linear float image."""
    import torch
    f = 0.5 * w / math.tan(0.5 * CAMERA_ANGLE_X)
    ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32), torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
    d_cam = torch.stack([(xs + 0.5 - 0.5 * w) / f, -(ys + 0.5 - 0.5 * h) / f, -torch.ones_like(xs)], dim=-1).reshape(-1, 3)
    R = torch.tensor(c2w[:3, :3], device=device)
    o = torch.tensor(c2w[:3, 3], device=device).expand_as(d_cam)
    d = d_cam @ R.T
    d = d / d.norm(dim=-1, keepdim=True)
    t0s, t1s, cols = [], [], []
    for kind, c, s, col in PRIMITIVES:
        c_t = torch.tensor(c, device=device, dtype=torch.float32)
        if kind == "box":
            hs = torch.tensor(s, device=device, dtype=torch.float32)
            inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
            ta, tb = (c_t - hs - o) * inv, (c_t + hs - o) * inv
            t0 = torch.minimum(ta, tb).amax(dim=-1)
            t1 = torch.maximum(ta, tb).amin(dim=-1)
        else:
            oc = o - c_t
            b = (oc * d).sum(-1)
            disc = b * b - ((oc * oc).sum(-1) - s * s)
            sq = torch.sqrt(disc.clamp(min=0))
            t0 = torch.where(disc > 0, -b - sq, torch.full_like(b, 1e9))
            t1 = torch.where(disc > 0, -b + sq, torch.full_like(b, -1e9))
        t0 = t0.clamp(min=0)
        hit = t1 > t0
        t0s.append(torch.where(hit, t0, torch.full_like(t0, 1e9)))
        t1s.append(torch.where(hit, t1, torch.full_like(t1, 1e9)))
        mid = o + d * (0.5 * (t0 + t1)).unsqueeze(-1)
        base = torch.tensor(col, device=device, dtype=torch.float32)
        shade = 0.75 + 0.25 * torch.sin(7.0 * mid[:, 0:1]) * torch.cos(5.0 * mid[:, 1:2] + 3.0 * mid[:, 2:3])
        cols.append((base * shade).clamp(0, 1))
    T0, T1, C = torch.stack(t0s, 1), torch.stack(t1s, 1), torch.stack(cols, 1)
    order = T0.argsort(dim=1)
    T0, T1 = T0.gather(1, order), T1.gather(1, order)
    C = C.gather(1, order.unsqueeze(-1).expand(-1, -1, 3))
    alpha = 1.0 - torch.exp(-SIGMA * (T1 - T0).clamp(min=0, max=100.0))
    alpha = torch.where(T0 < 1e8, alpha, torch.zeros_like(alpha))
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha[:, :-1]], dim=1), dim=1)
    wgt = alpha * trans
    rgb = (wgt.unsqueeze(-1) * C).sum(1)
    a = wgt.sum(1, keepdim=True)
    straight = torch.where(a > 1e-6, rgb / a.clamp(min=1e-6), torch.zeros_like(rgb))
    srgb = torch.where(straight > 0.0031308, 1.055 * straight.clamp(min=1e-8) ** (1.0 / 2.4) - 0.055, 12.92 * straight)
    rgba8 = torch.cat([srgb, a], dim=-1).clamp(0, 1).mul(255.0).round().to(torch.uint8).reshape(h, w, 4)
    return rgba8.cpu().numpy()


def make_dataset(n_train=100, n_test=20, res=800, device=None, seed=0, aabb_scale=1, height=None):
    """res = image width (and height unless `height` is given: the fox-shaped config is 1920 x 1080 with aabb_scale 4, i.e. 3 cascades)."""
    import torch
    device = device or torch.device("cuda:0")
    train = camera_poses(n_train, seed)
    test = camera_poses(n_test, seed + 1)
    h = height or res
    fl = 0.5 * res / math.tan(0.5 * CAMERA_ANGLE_X)
    return dict(res=res, w=res, h=h, focal=fl, camera_angle_x=CAMERA_ANGLE_X, scale=0.33, offset=[0.5, 0.5, 0.5], aabb_scale=aabb_scale,
                train_poses=train, test_poses=test,
                train_images=[render_ground_truth(m, res, h, device) for m in train],
                test_images=[render_ground_truth(m, res, h, device) for m in test])


# ---- a scene on disk: transforms_train.json [+ transforms_test.json] in the nerf-synthetic layout (run.py --scene / --test_transforms) -------------------------
def _frame_image_path(base, file_path):
    """the file a frame names: nerf_loader.cu:562-570 appends .png (then .exr) to an extension-less path; run.py:240-252 also tries .jpg / .jpeg"""
    p = file_path if os.path.isabs(file_path) else os.path.join(base, file_path)
    if os.path.splitext(os.path.basename(p))[1] and os.path.isfile(p):
        return p
    for ext in (".png", ".exr", ".jpg", ".jpeg"):
        if os.path.isfile(p + ext):
            return p + ext
    return p


def prepare_training_json(train_json, workdir):
    """this fork loads with NERF_SCALE = 1 and offset 0 (nerf_loader.h:28, nerf_loader.cu:406-407), so a stock nerf-synthetic file — no "scale",
    "offset" or "aabb" key — would train the object around the origin of a cube centred on (0.5, 0.5, 0.5).  To benchmark it as the paper did the file must carry
    "scale": 0.33, "offset": [0.5, 0.5, 0.5] (honoured at nerf_loader.cu:472-474, 499-504).  A file that has the keys is used as it is; otherwise a patched copy is
    written to `workdir` (frame paths made absolute, so that it may live anywhere) and that path is returned."""
    if os.path.isdir(train_json):   # a directory would make the loader take EVERY *.json in it, test and validation frames included (testbed_nerf.cu:2738-2743)
        raise ValueError("%s is a directory: pass the transforms_train.json itself (README.md:117 of the reference does the same)" % train_json)
    meta = json.load(open(train_json))
    if "frames" not in meta or not meta["frames"]:
        raise ValueError("%s: no \"frames\"" % train_json)
    if "scale" in meta or "offset" in meta or "aabb" in meta:
        return os.path.abspath(train_json), meta, False
    base = os.path.dirname(os.path.abspath(train_json))
    meta["scale"], meta["offset"] = 0.33, [0.5, 0.5, 0.5]
    for f in meta["frames"]:
        if not os.path.isabs(f["file_path"]):
            f["file_path"] = os.path.join(base, f["file_path"])
    os.makedirs(workdir, exist_ok=True)
    out = os.path.join(workdir, os.path.basename(train_json))
    with open(out, "w") as fh:
        json.dump(meta, fh)
    return out, meta, True


def load_disk_dataset(train_json, test_json=None, max_test=None, decode_train=False, workdir=None):
    """The dict build_testbed / eval_test_views / bench.py work with, for a dataset on disk: training goes through the PRODUCT loader (Testbed.load_training_data on
    `train_path`), evaluation reads the test transforms as run.py:216-303 does (reference images decoded by the product's own PNG / JPEG reader, pyngp.decode_image).
    decode_train: also keep host copies of the training images in loader order (bench.py's cpu_baseline hands them to the oracle)."""
    import tempfile
    import pyngp
    workdir = workdir or tempfile.mkdtemp(prefix="ngp_scene_")
    train_path, meta, patched = prepare_training_json(train_json, workdir)
    base = os.path.dirname(os.path.abspath(train_json))
    frames = [f for f in meta["frames"] if os.path.isfile(_frame_image_path(base, f["file_path"]))]   # frames whose file is missing are dropped (nerf_loader.cu:383)
    first = pyngp.decode_image(_frame_image_path(base, frames[0]["file_path"]))
    h, w = int(meta.get("h", first.shape[0])), int(meta.get("w", first.shape[1]))
    if "camera_angle_x" in meta:
        angle_x = float(meta["camera_angle_x"])
    elif "fl_x" in meta:
        angle_x = 2.0 * math.atan(0.5 * w / float(meta["fl_x"]))
    else:
        raise ValueError("%s: neither camera_angle_x nor fl_x" % train_json)
    ds = dict(train_path=train_path, source=os.path.abspath(train_json), scale_offset_injected=patched, res=w, w=w, h=h, camera_angle_x=angle_x,
              focal=float(meta.get("fl_x", 0.5 * w / math.tan(0.5 * angle_x))), scale=float(meta.get("scale", 1.0)), offset=meta.get("offset", [0.0, 0.0, 0.0]),
              aabb_scale=int(meta.get("aabb_scale", 1)), n_train=len(frames), train_poses=[np.asarray(f["transform_matrix"], np.float32) for f in frames],
              test_poses=[], test_images=[])
    if decode_train:
        ds["train_images"] = [np.ascontiguousarray(pyngp.decode_image(_frame_image_path(base, f["file_path"]))) for f in frames]
    if test_json:
        tmeta = json.load(open(test_json))
        tbase = os.path.dirname(os.path.abspath(test_json))
        tframes = tmeta["frames"][:max_test] if max_test else tmeta["frames"]
        ds["test_poses"] = [np.asarray(f["transform_matrix"], np.float32) for f in tframes]
        ds["test_images"] = [_frame_image_path(tbase, f["file_path"]) for f in tframes]                  # paths: decoded one at a time by eval_test_views
        if "camera_angle_x" in tmeta:
            ds["test_camera_angle_x"] = float(tmeta["camera_angle_x"])
    return ds


def build_testbed(ds, config_path=None, seed=1337):
    """Equivalent of `testbed = ngp.Testbed(mode); testbed.load_training_data(scene); testbed.reload_network_from_file(cfg)` (run.py:84-125)
    for a dataset held in memory (create_empty_nerf_dataset + set_image / set_camera_*: python_api.cu:619, 820-830)."""
    import torch  # noqa: F401  (one HIP runtime per process: load torch's first, see capi.load_ngp_hip)
    import pyngp as ngp
    t = ngp.Testbed(ngp.TestbedMode.Nerf)
    t.seed = seed
    if ds.get("train_path"):          # a dataset on disk: the product's transforms.json loader + image decoders (run.py:111-115)
        t.load_training_data(ds["train_path"])
    else:
        n = len(ds["train_images"])
        t.create_empty_nerf_dataset(n, ds["aabb_scale"], False)
        t.nerf.training.set_dataset_transform(ds["scale"], ds["offset"])
        w, h = ds.get("w", ds["res"]), ds.get("h", ds["res"])
        for i in range(n):
            t.nerf.training.set_image_rgba8(i, ds["train_images"][i])
            t.nerf.training.set_camera_intrinsics(i, ds["focal"], ds["focal"], 0.5 * w, 0.5 * h)
            t.nerf.training.set_camera_extrinsics(i, ds["train_poses"][i][:3, :], True)
        t.nerf.training.n_images_for_training = n
    t.reload_network_from_file(config_path or os.path.join(HERE, "configs", "nerf", "base.json"))
    # run.py:131-150 defaults for NeRF
    t.nerf.render_with_lens_distortion = True
    t.exposure = 0.0
    t.shall_train = True
    # soak runs of the host-side options (INTEGRATION.md "Host-side options"): NGP_SCENE_TESTBED_OPTIONS='{"compact_backward": true}' python -m pytest tests -m gpu
    # sets them on every Testbed built here (the tests and the bench build theirs through this function); unset = the product defaults
    for key, value in json.loads(os.environ.get("NGP_SCENE_TESTBED_OPTIONS", "{}")).items():
        setattr(t, key, value)
    return t


def train(testbed, n_steps, log_every=0):
    """run.py:187-210"""
    t0 = time.time()
    if not testbed.shall_train:
        raise RuntimeError("scene.train: testbed.shall_train is off (eval_test_views switches it off): frame() would not advance training_step")
    while testbed.frame():
        if testbed.training_step >= n_steps:
            break
        if not testbed.shall_train:   # the Testbed switched training off itself ("Nerf training generated 0 samples. Aborting training.", testbed_nerf.cu:2966-2970): run.py's loop would spin here for ever
            raise RuntimeError("scene.train: training stopped at step %d (the step generated no samples)" % testbed.training_step)
        if log_every and testbed.training_step % log_every == 0:
            print("step %d loss %.5f rays %d (%.1fs)" % (testbed.training_step, testbed.loss, testbed.nerf.training.rays_per_batch, time.time() - t0), flush=True)


def eval_test_views(testbed, ds, spp=8, max_views=None):
    """run.py:216-303: black background, pixel-centre sampling, min transmittance 1e-4, PSNR on sRGB-clipped images, mean over views."""
    import metrics
    testbed.background_color = [0.0, 0.0, 0.0, 1.0]
    testbed.snap_to_pixel_centers = True
    testbed.nerf.render_min_transmittance = 1e-4
    testbed.fov_axis = 0
    testbed.fov = ds.get("test_camera_angle_x", ds["camera_angle_x"]) * 180 / np.pi
    testbed.shall_train = False
    w, h = ds.get("w", ds["res"]), ds.get("h", ds["res"])
    psnrs, ssims = [], []
    views = list(zip(ds["test_poses"], ds["test_images"]))[:max_views]
    for pose, img8 in views:
        if isinstance(img8, str):     # a dataset on disk: the frame's reference image, read as run.py:253 reads it
            import pyngp
            img8 = pyngp.decode_image(img8)
            h, w = img8.shape[0], img8.shape[1]
        ref = metrics.read_image_rgba8(img8)
        testbed.set_nerf_camera_matrix(pose[:3, :])
        image = testbed.render(w, h, spp, True)
        p, s = metrics.eval_psnr_ssim(image, ref)
        psnrs.append(p)
        ssims.append(s)
    return float(np.mean(psnrs)), float(np.mean(ssims)), psnrs


def smoke_train(steps=3, res=64, n_train=8):
    import torch
    ds = make_dataset(n_train=n_train, n_test=1, res=res, device=torch.device("cuda:0"))
    t = build_testbed(ds)
    train(t, steps)
    assert t.training_step == steps and np.isfinite(t.loss)
    img = t.render(res, res, 1, True)
    assert img.shape == (res, res, 4) and np.isfinite(img).all()
    print("smoke: %d training steps on a %dx%d x%d procedural scene, loss %.4f, rays/batch %d" % (steps, res, res, n_train, t.loss, t.nerf.training.rays_per_batch))


if __name__ == "__main__":
    import argparse
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--n_train", type=int, default=50)
    ap.add_argument("--n_test", type=int, default=4)
    ap.add_argument("--spp", type=int, default=2)
    a = ap.parse_args()
    t0 = time.time()
    ds = make_dataset(a.n_train, a.n_test, a.res, torch.device("cuda:0"))
    print("dataset %.1fs" % (time.time() - t0), flush=True)
    tb = build_testbed(ds)
    t0 = time.time()
    train(tb, a.steps, log_every=max(1, a.steps // 10))
    dt = time.time() - t0
    print("trained %d steps in %.1fs (%.2f ms/step)" % (a.steps, dt, 1000 * dt / a.steps), flush=True)
    psnr, ssim, per = eval_test_views(tb, ds, spp=a.spp)
    print("PSNR=%.2f SSIM=%.4f per-view=%s render_ms=%.1f" % (psnr, ssim, ["%.1f" % p for p in per], tb.render_ms))


def write_dataset(ds, directory, with_test=False, stock_keys=False):
    """Write a dataset of make_dataset() in the nerf-synthetic on-disk layout (transforms_train.json + train/r_%03d.png, RGBA8): what
    `Testbed.load_training_data` ingests (src/nerf_loader.cu).  Returns the path of transforms_train.json."""
    import json
    from PIL import Image
    os.makedirs(os.path.join(directory, "train"), exist_ok=True)
    frames = []
    for i, (img, pose) in enumerate(zip(ds["train_images"], ds["train_poses"])):
        name = "train/r_%03d" % i
        Image.fromarray(np.ascontiguousarray(img), "RGBA").save(os.path.join(directory, name + ".png"))
        m = np.eye(4); m[:3, :4] = np.asarray(pose)[:3, :4]
        frames.append({"file_path": "./" + name, "transform_matrix": m.tolist()})
    meta = {"fl_x": float(np.float32(ds["focal"])), "fl_y": float(np.float32(ds["focal"])), "cx": 0.5 * ds.get("w", ds["res"]), "cy": 0.5 * ds.get("h", ds["res"]), "w": ds.get("w", ds["res"]), "h": ds.get("h", ds["res"]),
            "camera_angle_x": ds["camera_angle_x"], "aabb_scale": ds["aabb_scale"], "scale": ds["scale"], "offset": list(ds["offset"]), "frames": frames}
    if stock_keys:   # what a stock nerf-synthetic transforms_train.json carries: camera_angle_x and frames, nothing else (scene.prepare_training_json injects scale / offset)
        meta = {"camera_angle_x": ds["camera_angle_x"], "frames": frames}
    path = os.path.join(directory, "transforms_train.json")
    with open(path, "w") as f:
        json.dump(meta, f)
    if with_test and ds.get("test_images"):   # transforms_test.json + test/r_%d.png, as nerf-synthetic ships them (no scale / offset keys needed: only poses and images are read)
        os.makedirs(os.path.join(directory, "test"), exist_ok=True)
        tframes = []
        for i, (img, pose) in enumerate(zip(ds["test_images"], ds["test_poses"])):
            name = "test/r_%d" % i
            Image.fromarray(np.ascontiguousarray(img), "RGBA").save(os.path.join(directory, name + ".png"))
            m = np.eye(4); m[:3, :4] = np.asarray(pose)[:3, :4]
            tframes.append({"file_path": "./" + name, "transform_matrix": m.tolist()})
        with open(os.path.join(directory, "transforms_test.json"), "w") as f:
            json.dump({"camera_angle_x": ds["camera_angle_x"], "frames": tframes}, f)
    return path
