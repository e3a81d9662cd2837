"""CPU tests: the C-ABI library loads without a GPU and exports every symbol include/ngp_hip.h declares (no compute calls)."""
import os
import re
import subprocess

import numpy as np

import capi
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(ngp):
    header = open(os.path.join(ROOT, "include", "ngp_hip.h")).read()
    declared = set(re.findall(r"\b(ngp_(?:hip|rccl)_\w+)\s*\(", header))
    assert len(declared) >= 80 and "ngp_rccl_allreduce_grads" in declared
    nm = subprocess.run(["nm", "-D", "--defined-only", ngp.so_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ngp_(?:hip|rccl)_\w+)", nm))
    assert declared <= exported, sorted(declared - exported)
    assert set(ngp.protos) == declared
    assert ngp.ngp_hip_abi_version() == 1


def test_no_torch_or_oracle_in_the_abi(ngp):
    deps = subprocess.run(["ldd", ngp.so_path], capture_output=True, text=True).stdout
    assert "torch" not in deps and "oracle" not in deps and "libamdhip64" in deps and "rccl" not in deps   # RCCL is bound at run time (csrc/comm.hip)
    header = open(os.path.join(ROOT, "include", "ngp_hip.h")).read()
    assert "torch" not in header and "at::" not in header


def test_pod_layouts_match_the_header():
    src = '#include "%s"\n#include <stdio.h>\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(NgpAabb), sizeof(NgpRay), sizeof(NgpXForm), sizeof(NgpCoord), sizeof(NgpPayload), sizeof(NgpImageMeta), sizeof(NgpNetDesc), sizeof(NgpErrorMapCdf), sizeof(NgpGlobalRay), sizeof(NgpProxyRay), sizeof(NgpMask3D), sizeof(NgpNerfProps), sizeof(NgpDownsampleInfo), sizeof(NgpRenderCamera), sizeof(NgpRenderExtras), sizeof(NgpLossExtras));}' % os.path.join(ROOT, "include", "ngp_hip.h")
    exe = "/tmp/ngp_sizes_%d" % os.getpid()
    subprocess.run(["gcc", "-x", "c", "-", "-o", exe], input=src, text=True, check=True)
    sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    os.remove(exe)
    assert sizes == [capi.AABB.itemsize, capi.RAY.itemsize, capi.XFORM.itemsize, capi.COORD.itemsize, capi.PAYLOAD.itemsize, capi.IMAGE_META.itemsize, capi.NET_DESC.itemsize,
                     capi.ERROR_MAP_CDF.itemsize, capi.GLOBAL_RAY.itemsize, capi.PROXY_RAY.itemsize, capi.MASK3D.itemsize, capi.NERF_PROPS.itemsize, capi.DOWNSAMPLE_INFO.itemsize,
                     capi.RENDER_CAMERA.itemsize, capi.RENDER_EXTRAS.itemsize, capi.LOSS_EXTRAS.itemsize]


def test_host_only_entry_points(ngp):
    d = H.make_desc(ngp, 15, 16, 1)
    assert int(d["n_levels"][0]) == 16 and ngp.ngp_hip_net_n_params_host(d.ctypes.data) == 10240 + 2 * int(d["n_grid_entries"][0])
    assert 16 * (1 << 18) * 4 < ngp.ngp_hip_nerf_backward_scratch_bytes(1 << 18) < (1 << 30)   # dL/dx planes + binning lists; no activation planes (fused backward)
    # sized for a level table: the 16 levels' sort records packed by kind (hashed 48 bytes per sample, dense 160), private copies only where a level needs them
    n = 1 << 18
    any_table = ngp.ngp_hip_nerf_backward_scratch_bytes(n)
    assert ngp.ngp_hip_nerf_backward_scratch_bytes_for(None, n) == any_table
    for aabb_scale, n_dense in ((1, 5), (4, 4)):
        big = H.make_desc(ngp, 19, 16, aabb_scale)
        dense = [int(l["resolution"]) ** 3 <= int(l["size"]) for l in big["levels"][0]]
        assert sum(dense) == n_dense
        packed = ngp.ngp_hip_nerf_backward_scratch_bytes_for(big.ctypes.data, n)
        fine = sum(int(l["resolution"]) >= 4096 for l in big["levels"][0])                    # a pair's x corners can straddle a 4096-entry slice there: two records per pair
        assert fine == (2 if aabb_scale == 4 else 0)
        # hashed levels at 48 (96 where fine) instead of 160 bytes per sample; 16 instead of 64 MiB of private copies
        assert any_table - packed == ((16 - n_dense - fine) * (160 - 48) + fine * (160 - 96)) * n + 48 * (1 << 20)
        assert packed < 410e6 < 770e6 < any_table
    bad = np.zeros(1, H.NET_DESC)
    assert ngp.ngp_hip_net_make_desc_host(8, 19, 16, H.f32(1.5), bad.ctypes.data) != 0  # only L = 16 is built
    assert b"n_levels" in ngp.ngp_hip_last_error()
