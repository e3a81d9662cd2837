"""ctypes view of a C header: parses `int name(args);` prototypes and attaches argtypes/restype to a loaded library.

Used for libngp_hip.so (include/ngp_hip.h) by the tests and bench.py, and for the CPU oracle by the tests.  This module is
plumbing only: it contains no algorithm and never falls back to a CPU implementation — if libngp_hip.so is missing,
`load_ngp_hip()` raises.
"""
import ctypes
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "uint32_t": ctypes.c_uint32, "uint64_t": ctypes.c_uint64, "int64_t": ctypes.c_int64,
    "uint16_t": ctypes.c_uint16, "uint8_t": ctypes.c_uint8, "float": ctypes.c_float, "double": ctypes.c_double, "void": None,
}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return text


def parse_prototypes(header_path):
    """Returns {name: (restype, [argtypes], [argnames])} for every function prototype in the header."""
    text = _strip_comments(open(header_path).read())
    text = re.sub(r"#[^\n]*", " ", text)
    # drop typedef struct bodies and enums
    text = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"enum\s*\{.*?\}\s*;", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(\w+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("static") or "typedef" in ret or not ret:
            continue
        ret = ret.replace("extern", "").replace('"C"', "").strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _SCALARS.get(ret.replace("const", "").strip(), ctypes.c_int)
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"\[[^\]]*\]", "*", a)  # array params decay to pointers
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split("*")[-1].strip())
                else:
                    toks = a.replace("const", "").split()
                    argtypes.append(_SCALARS[toks[0]])
                    argnames.append(toks[-1])
        protos[name] = (restype, argtypes, argnames)
    return protos


class CLib:
    def __init__(self, so_path, header_path, prefix):
        if not os.path.exists(so_path):
            raise FileNotFoundError("%s not built (run `python blender-ngp_amd/build.py` / `make -C oracle`)" % so_path)
        self.lib = ctypes.CDLL(so_path)
        self.so_path = so_path
        self.protos = {k: v for k, v in parse_prototypes(header_path).items() if k.startswith(prefix)}
        for name, (restype, argtypes, _) in self.protos.items():
            fn = getattr(self.lib, name)  # raises AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes

    def __getattr__(self, name):
        fn = getattr(self.lib, name)
        proto = self.protos.get(name)
        if proto is None:
            return fn
        argtypes = proto[1]
        n_required = len(argtypes)
        while n_required and argtypes[n_required - 1] is ctypes.c_void_p:
            n_required -= 1

        def call(*args):   # the optional trailing pointers of an entry point (events, dL_dinput, NgpNetVariant*, NgpRenderExtras* ...) default to NULL
            if n_required <= len(args) < len(argtypes):
                args = args + (None,) * (len(argtypes) - len(args))
            return fn(*args)
        call.__name__ = name
        self.__dict__[name] = call
        return call


def ptr(x):
    """Device/host pointer of a torch tensor, numpy array, ctypes object, int or None as c_void_p-compatible int."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return ctypes.addressof(x)


_ngp = None


def load_ngp_hip():
    """libngp_hip.so with typed entry points; every call returns 0 or raises with ngp_hip_last_error()."""
    global _ngp
    if _ngp is None:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  If torch is going to be used for device
        # memory in this process it must be loaded FIRST so that libngp_hip.so (NEEDED libamdhip64.so.7) binds to the same
        # runtime instance; two instances in one process do not both see the GPU (hipErrorNoDevice on the second).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # NGP_HIP_LIBRARY_DIR: a tool that wants the development build (build.py --dev -> lib_dev/, csrc/ngp_dev_knobs.h) points here; tests and bench.py never set it
        libdir = os.environ.get("NGP_HIP_LIBRARY_DIR") or os.path.join(HERE, "lib")
        _ngp = CLib(os.path.join(libdir, "libngp_hip.so"), os.path.join(ROOT, "include", "ngp_hip.h"), ("ngp_hip_", "ngp_rccl_"))
    return _ngp


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_ngp_hip()
        raise RuntimeError("ngp_hip call failed (%d): %s" % (rc, lib.ngp_hip_last_error().decode()))


# numpy views of the PODs in include/ngp_hip.h (identical layouts are used by the oracle)
AABB = np.dtype([("min", "<f4", 3), ("max", "<f4", 3)])
RAY = np.dtype([("o", "<f4", 3), ("d", "<f4", 3)])
XFORM = np.dtype([("start", "<f4", 12), ("end", "<f4", 12)])
COORD = np.dtype([("pos", "<f4", 3), ("dt", "<f4"), ("dir", "<f4", 3)])
PAYLOAD = np.dtype([("origin", "<f4", 3), ("dir", "<f4", 3), ("t", "<f4"), ("max_weight", "<f4"), ("idx", "<u4"), ("n_steps", "<u2"), ("alive", "u1"), ("pad", "u1")])
IMAGE_META = np.dtype([
    ("pixels", "<u8"), ("image_data_type", "<i4"), ("res", "<i4", 2), ("focal_length", "<f4", 2), ("principal_point", "<f4", 2),
    ("rolling_shutter", "<f4", 4), ("lens_mode", "<i4"), ("lens_params", "<f4", 7), ("depth", "<u8"), ("rays", "<u8")], align=True)
GRID_LEVEL = np.dtype([("scale", "<f4"), ("resolution", "<u4"), ("offset", "<u4"), ("size", "<u4")])
NET_DESC = np.dtype([("n_levels", "<u4"), ("n_grid_entries", "<u4"), ("levels", GRID_LEVEL, 16)])

# Blender multi-NeRF renderer records (include/ngp_hip.h)
GLOBAL_RAY = np.dtype([("origin", "<f4", 3), ("dir", "<f4", 3), ("rgba", "<f4", 4), ("idx", "<u4"), ("depth", "<f4"), ("alive", "u1"), ("pad", "u1", 3)])
PROXY_RAY = np.dtype([("origin", "<f4", 3), ("dir", "<f4", 3), ("t", "<f4"), ("idx", "<u4"), ("n_steps", "<u2"), ("alive", "u1"), ("active", "u1"), ("mask_alpha", "<f4")])
MASK3D = np.dtype([("mode", "<i4"), ("shape", "<i4"), ("transform", "<f4", 16), ("itransform", "<f4", 16), ("config", "<f4", 6), ("feather", "<f4"), ("opacity", "<f4")])
NERF_PROPS = np.dtype([
    ("transform", "<f4", 16), ("itransform", "<f4", 16), ("density_grid_bitfield", "<u8"), ("grid_size", "<u4"), ("grid_volume", "<u4"),
    ("render_aabb", AABB), ("train_aabb", AABB), ("masks", "<u8"), ("n_masks", "<u4"), ("cone_angle", "<f4"), ("min_cone_stepsize", "<f4"),
    ("max_cone_stepsize", "<f4"), ("nerf_cascades", "<u4"), ("opacity", "<f4")], align=True)
ERROR_MAP_CDF = np.dtype([("cdf_x_cond_y", np.uint64), ("cdf_y", np.uint64), ("cdf_img", np.uint64), ("res", np.int32, 2)], align=True)   # NgpErrorMapCdf
DOWNSAMPLE_INFO = np.dtype([("max_res", "<i4", 2), ("scaled_res", "<i4", 2), ("skip", "<i4", 2), ("max_pixels", "<u4"), ("scaled_pixels", "<u4")])
RENDER_CAMERA = np.dtype([("transform", "<f4", 12), ("model", "<i4"), ("focal_length", "<f4"), ("sq_width", "<f4"), ("sq_height", "<f4"), ("sq_curvature", "<f4"),
                          ("qh_front", "<f4", 12), ("qh_back", "<f4", 12), ("near_distance", "<f4"), ("aperture_size", "<f4"), ("focus_z", "<f4")])
assert GLOBAL_RAY.itemsize == 52 and PROXY_RAY.itemsize == 40 and MASK3D.itemsize == 168 and NERF_PROPS.itemsize == 224
assert DOWNSAMPLE_INFO.itemsize == 32 and RENDER_CAMERA.itemsize == 176
NET_VARIANT = np.dtype([("n_extra_dims", "<u4"), ("n_rgb_hidden_layers", "<u4"), ("extra_dims", "<u8"), ("sample_slot", "<u8"), ("dL_dextra", "<u8"), ("flags", "<u4")], align=True)   # NgpNetVariant
NETX_SCALAR = 1   # NgpNetVariant.flags: the scalar checker kernels instead of the MFMA kernels
LOSS_EXTRAS = np.dtype([("envmap_data", "<u8"), ("envmap_gradient", "<u8"), ("envmap_res", "<i4", 2), ("envmap_loss_type", "<i4"),
                        ("sharpness_data", "<u8"), ("sharpness_res", "<i4", 2), ("sharpness_grid", "<u8"), ("x_row_index_out", "<u8")], align=True)   # NgpLossExtras
RENDER_EXTRAS = np.dtype([("render_masks", "<u8"), ("n_render_masks", "<u4"), ("glow_mode", "<i4"), ("glow_y_cutoff", "<f4"), ("envmap", "<u8"), ("envmap_res", "<i4", 2),
                          ("distortion", "<u8"), ("distortion_res", "<i4", 2), ("quilting_dims", "<i4", 2), ("render_mode", "<i4"), ("frame_buffer", "<u8"),
                          ("row_begin", "<i4"), ("row_end", "<i4"), ("tile_order", "<i4")], align=True)   # NgpRenderExtras

assert AABB.itemsize == 24 and RAY.itemsize == 24 and XFORM.itemsize == 96 and COORD.itemsize == 28 and PAYLOAD.itemsize == 40
assert NET_DESC.itemsize == 8 + 16 * 16
