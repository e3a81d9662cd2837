"""Build the gfx950 kernel library (libngp_hip.so) and the pybind11 host module (pyngp) in-tree.

hipcc cross-compiles gfx950 without a GPU.  Outputs stay next to the sources (git-ignored, shipped by gpurun):
    blender-ngp_amd/lib/libngp_hip.so      kernels + C ABI (include/ngp_hip.h)
    blender-ngp_amd/pyngp*.so              C++ Testbed + pybind11 bindings (the reference's `pyngp` module name)
"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

KERNEL_SOURCES = ["density_grid.hip", "train_samples.hip", "network.hip", "loss.hip", "render.hip", "multi_render.hip", "comm.hip", "probe.hip"]
HOST_SOURCES = ["testbed.cpp", "python_api.cpp", "mini_json.cpp", "snapshot.cpp", "nerf_renderer.cpp", "nerf_loader.cpp", "png_reader.cpp", "exr_reader.cpp", "jpeg_reader.cpp", "hdr_reader.cpp", "image_io.cpp", "plumbing.cpp", "dp.cpp"]

# -ffp-contract=off: the index / count paths must round exactly like the CPU oracle; network.hip re-enables contraction locally.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"] + os.environ.get("NGP_EXTRA_HIP_FLAGS", "").split()


def _newer(src_list, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in src_list)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_kernels(force=False, verbose=False, dev=False):
    """dev=True: the same sources with -DNGP_DEV_KNOBS (csrc/ngp_dev_knobs.h: the sweep / ablation environment knobs of tools/) into lib_dev/ — never what pyngp links or the tests load"""
    libdir, objdir = (LIBDIR + "_dev", OBJDIR + "_dev") if dev else (LIBDIR, OBJDIR)
    os.makedirs(libdir, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".cuh") or h.endswith(".h")] + [os.path.join(ROOT, "include", "ngp_hip.h")]   # (network_generic.cuh / network_netx_mfma.cuh are included by network.hip)
    objs, jobs = [], []
    for s in KERNEL_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            jobs.append([HIPCC] + HIP_FLAGS + (["-DNGP_DEV_KNOBS"] if dev else []) + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out.strip():
                print(out)
    lib = os.path.join(libdir, "libngp_hip.so")
    if force or jobs or not os.path.exists(lib):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib, "-ldl"])
    return lib


def build_host(force=False, verbose=False):
    import pybind11
    lib = build_kernels(force=force, verbose=verbose)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(HERE, "pyngp" + ext)
    srcs = [os.path.join(HOST, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST, h) for h in os.listdir(HOST) if h.endswith(".h")] + [os.path.join(ROOT, "include", "ngp_hip.h"), lib]
    if force or _newer(deps, out):
        cmd = ["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__",
               "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROCM, "include")] + srcs + ["-o", out, "-L" + LIBDIR, "-lngp_hip", "-L" + os.path.join(ROCM, "lib"),
               "-lamdhip64", "-lz", "-lrt", "-Wl,-rpath,$ORIGIN/lib", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread"]
        o = _run(cmd)
        if verbose and o.strip():
            print(o)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    only_kernels = "--kernels" in sys.argv
    if "--dev" in sys.argv:
        print(build_kernels(force=force, verbose=True, dev=True))
    elif only_kernels:
        print(build_kernels(force=force, verbose=True))
    else:
        print(build_host(force=force, verbose=True))
