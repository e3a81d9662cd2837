#!/usr/bin/env python3
"""GPU box: does the ORDER of the occupancy-grid update's sample positions matter to its density pass?  2^20 positions, one random point in each of 2^20 pseudo-randomly
chosen cells of the 128^3 grid, (a) in generation order, (b) sorted by the cells' Morton index, (c) sorted by 4x4x4 brick only.  Times ngp_hip_nerf_density_ws."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]


def morton_decode(idx):
    import numpy as np
    def compact(v):
        v = v & 0x09249249
        v = (v ^ (v >> 2)) & 0x030c30c3
        v = (v ^ (v >> 4)) & 0x0300f00f
        v = (v ^ (v >> 8)) & 0xff0000ff
        v = (v ^ (v >> 16)) & 0x000003ff
        return v
    idx = idx.astype(np.uint32)
    return compact(idx), compact(idx >> 1), compact(idx >> 2)


def main():
    import numpy as np
    import torch
    import capi
    import helpers as H
    from capi import check
    dev = torch.device("cuda:0")
    ngp = capi.load_ngp_hip()
    n = 1 << 20
    desc = H.make_desc(ngp, 19)
    P = H.random_params(desc, 0, grid_amp=0.1)
    rs = np.random.RandomState(0)
    cells = ((np.arange(n, dtype=np.uint64) * 56924617) % (1 << 21)).astype(np.uint32)
    x, y, z = morton_decode(cells)
    pos = ((np.stack([x, y, z], 1) + rs.rand(n, 3)) / 128.0).astype(np.float32)
    d_desc, d_P = H.to_dev(desc, dev), H.to_dev(P, dev)
    ws_bytes = ngp.ngp_hip_nerf_encode_workspace_bytes(n)
    ws, out = H.dev_zeros(ws_bytes, dev), H.dev_zeros(n * 2, dev)
    orders = {"generation order": np.arange(n), "sorted by cell (Morton)": np.argsort(cells, kind="stable"), "sorted by 4x4x4 brick": np.argsort(cells >> 6, kind="stable"),
              "sorted by 8x8x8 block": np.argsort(cells >> 9, kind="stable")}
    for name, perm in orders.items():
        d_pos = H.to_dev(np.ascontiguousarray(pos[perm]), dev)
        for _ in range(3):
            check(ngp.ngp_hip_nerf_density_ws(None, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), 3, n, out.data_ptr(), ws.data_ptr(), ws_bytes, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            check(ngp.ngp_hip_nerf_density_ws(None, d_desc.data_ptr(), d_P.data_ptr(), d_pos.data_ptr(), 3, n, out.data_ptr(), ws.data_ptr(), ws_bytes, None))
        e1.record()
        torch.cuda.synchronize()
        print("%-26s %7.1f us per density pass" % (name, 100.0 * e0.elapsed_time(e1)))


if __name__ == "__main__":
    main()
