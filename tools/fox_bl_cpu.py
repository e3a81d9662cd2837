"""dev (this container, CPU): both whole-frame oracles on the fox snapshot tools/fox_bl_dump.py wrote — orc_render_nerf (the stock tracer, src/testbed_nerf.cu) and
orc_multi_render (the fork's Blender renderer, src/nerf_renderer.cu) — sample counts and frames against what the GPU produced, and the step sizes along one ray."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
import msgpack
import numpy as np
import capi
import helpers as H
import test_multi_render_gpu as M

out = os.path.join(ROOT, "gpurun_out")
view = json.load(open(os.path.join(out, "fox_view.json")))
snap = msgpack.unpackb(open(os.path.join(out, "fox_snap.msgpack"), "rb").read(), raw=False)["snapshot"]
orc = H.load_oracle()
ngp = capi.load_ngp_hip()
desc = H.make_desc(ngp, log2_hashmap_size=19, base_resolution=16, aabb_scale=4)
params = np.frombuffer(snap["params_binary"], np.float16).copy()
assert params.size >= H.n_params(desc), (params.size, H.n_params(desc))
grid = np.frombuffer(snap["density_grid_binary"], np.float16).astype(np.float32)
vol = 128 ** 3
nc = grid.size // vol
orc.orc_density_grid_mean.restype = ctypes.c_float
bf, mean = H.oracle_bitfield(orc, grid, nc)
print("cascades", nc, "mean", mean, "occupied", [int(np.unpackbits(bf[c * vol // 8:(c + 1) * vol // 8]).sum()) for c in range(nc)], flush=True)
w, h = view["w"], view["h"]
cam34 = np.asarray(view["camera_matrix"], np.float32)              # 3 x 4, row lists
cam_cm = cam34.T.reshape(-1).copy()                                 # column-major 3x4
res = np.array([w, h], np.int32)
focal = np.array([view["focal_px"]] * 2, np.float32)
sc = np.array([0.5, 0.5], np.float32)
aabb = H.unit_aabb(4)
eye3 = np.eye(3, dtype=np.float32).reshape(-1).copy()
fb = np.zeros((h, w, 4), np.float32); db = np.zeros((h, w), np.float32)
orc.orc_render_nerf.restype = ctypes.c_uint64
if "skip_stock" not in sys.argv:
    n_stock = orc.orc_render_nerf(desc.ctypes.data, params.ctypes.data, 0, res.ctypes.data, focal.ctypes.data, cam_cm.ctypes.data, cam_cm.ctypes.data, sc.ctypes.data, 1, aabb.ctypes.data,
                                  eye3.ctypes.data, aabb.ctypes.data, ctypes.c_float(0.0), bf.ctypes.data, ctypes.c_float(1.0 / 256.0), 2, 3, ctypes.c_float(0.01), 1, fb.ctypes.data, db.ctypes.data)
    gpu = np.load(os.path.join(out, "fox_stock.npy")).astype(np.float32)
    print("stock: oracle samples", n_stock, "GPU", view["stock_samples"], "(GPU counts padded launches); max |frame diff|", float(np.abs(fb - gpu).max()), "mean", float(np.abs(fb - gpu).mean()), flush=True)
ds = M._ds(orc, w, h, 0)
cam = np.zeros(1, capi.RENDER_CAMERA)
cam["transform"][0] = cam_cm; cam["model"] = 0; cam["focal_length"] = view["focal_px"]; cam["focus_z"] = 1.0
rp = M._props(np.eye(4, dtype=np.float32), bf.ctypes.data, 0, 0, aabb_scale=4)
fb2 = np.zeros((h, w, 4), np.float32); db2 = np.zeros((h, w), np.float32)
orc.orc_multi_render.restype = ctypes.c_uint64
nets = (ctypes.c_void_p * 1)(desc.ctypes.data); pars = (ctypes.c_void_p * 1)(params.ctypes.data)
act_rgb = np.array([2], np.int32); act_d = np.array([3], np.int32); min_t = np.array([0.01], np.float32)
n_bl = orc.orc_multi_render(1, nets, pars, rp.ctypes.data, act_rgb.ctypes.data, act_d.ctypes.data, min_t.ctypes.data, ds.ctypes.data, cam.ctypes.data, 0, fb2.ctypes.data, db2.ctypes.data)
gpu_bl = np.load(os.path.join(out, "fox_bl_0.npy")).astype(np.float32)
print("bl: oracle samples", n_bl, "GPU (reference sequence)", view["bl"]["reference_sequence"], "max |frame diff|", float(np.abs(fb2 - gpu_bl).max()), "mean", float(np.abs(fb2 - gpu_bl).mean()),
      "flipped", float(np.abs(fb2[::-1] - gpu_bl).mean()), flush=True)
# the same frame with a render box that contains the camera: the proxy rays' t then starts at the camera like the stock tracer's, and so does the cone's step size
if "big_box" in sys.argv:
    big = np.zeros(1, H.AABB); big["min"][0] = -12.0; big["max"][0] = 12.0
    rp2 = M._props(np.eye(4, dtype=np.float32), bf.ctypes.data, 0, 0, aabb_scale=4, render_aabb=big)
    fb3 = np.zeros((h, w, 4), np.float32); db3 = np.zeros((h, w), np.float32)
    n_big = orc.orc_multi_render(1, nets, pars, rp2.ctypes.data, act_rgb.ctypes.data, act_d.ctypes.data, min_t.ctypes.data, ds.ctypes.data, cam.ctypes.data, 0, fb3.ctypes.data, db3.ctypes.data)
    print("bl, render box [-12, 12]^3 (camera inside): oracle samples", n_big, " mean |frame - box-entry frame|", float(np.abs(fb3 - fb2).mean()), flush=True)
