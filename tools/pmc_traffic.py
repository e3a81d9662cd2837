#!/usr/bin/env python
"""Fold two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, collected separately with --kernel-trace only, CSV output) into
profiles/pmc_traffic.json: HBM bytes per C-ABI call of each step stage, per the gfx950 recipe in MI355X_MICROARCH.md §HBM
(unit = KiB; FETCH_SIZE doubled because this rocprofv3 tallies 128-B read requests at 64 B).

usage: pmc_traffic.py <dir with *counter_collection.csv of the FETCH pass> <dir of the WRITE pass> <out.json> [tag]
"""
import csv
import glob
import json
import os
import re
import sys

# kernel-name substring -> stage (one stage = one ngp_hip_* entry point as bench.py times it)
STAGES = [("nerf_forward_kernelILi2ELi0E", "nerf_inference"),   # the training step's one network pass (MODE 2 = forward with saved encodings, PRE 0 = gathers inside); bench.py's group name
          ("nerf_forward_kernelILi2ELi1E", "nerf_inference"),   # ... or its two-kernel organisation: the MLP-only kernel over encode_planes_kernel's level planes (the encode launches of the
          ("nerf_forward_kernelILi1E", "density_grid_prep"),    # training pass and of the occupancy update share one kernel name: see _kernels for the split by dispatch count)
          ("encode_planes_kernel", "density_grid_prep"),
          ("nerf_backward_fused_kernel", "nerf_backward"), ("grid_backward_kernel", "nerf_backward"), ("grid_combine_kernel", "nerf_backward"),
          ("gb_fx_bin_kernel", "nerf_backward"), ("gb_fx_scan_kernel", "nerf_backward"),
          ("nerf_wgrad_kernel", "nerf_backward"), ("wgrad_reduce_kernel", "nerf_backward"), ("adam_ema", "optimizer_step"),
          ("generate_training_samples_kernel", "generate_training_samples"), ("generate_training_samples_wave_kernel", "generate_training_samples"), ("generate_training_samples_cone_wave_kernel", "generate_training_samples"),
          ("expand_training_samples_kernel", "generate_training_samples"),
          ("compute_loss_kernel", "compute_loss")]


def per_kernel(d, counter):
    tot, cnt = {}, {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + d
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
            cnt[k] = cnt.get(k, 0) + 1
    return tot, cnt


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
    out, detail = {}, {}
    launches = {}
    for k in sorted(set(fetch) | set(write)):
        stage = next((s for sub, s in STAGES if sub in k), None)
        f_kib = fetch.get(k, 0.0) / max(nf.get(k, 1), 1)
        w_kib = write.get(k, 0.0) / max(nw.get(k, 1), 1)
        detail[k[:90]] = {"dispatches": nf.get(k, nw.get(k, 0)), "FETCH_SIZE_KiB_avg": round(f_kib, 1), "WRITE_SIZE_KiB_avg": round(w_kib, 1), "stage": stage}
        if stage is None:
            continue
        # kernels launched once per entry-point call add up; (grid_backward may be launched per level under a dev switch — not in bench runs)
        out[stage] = out.get(stage, 0.0) + (2.0 * f_kib + w_kib) * 1024.0
        launches[stage] = launches.get(stage, 0) + 1
    res = {s: round(v) for s, v in out.items()}
    ks = re.search(r'KERNEL_SET = "([^"]+)"', open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read())
    res["_meta"] = {"tag": sys.argv[4] if len(sys.argv) > 4 else os.path.basename(sys.argv[3]).replace("_pmc_traffic.json", ""), "kernel_set": ks.group(1) if ks else None}
    res["_method"] = ("per call: sum over the stage's kernels of (2*FETCH_SIZE + WRITE_SIZE)*1024 B, dispatch averages; separate --pmc passes.  The factor 2 is calibrated on this "
                      "code's access patterns (tools/fetch_calibration.hip -> profiles/r06_fetch_calibration.json): a wide stream, one 4-byte touch per 128-byte line and random 4- / 8-byte "
                      "gathers from a 2 GiB table all show ONE read request and FETCH_SIZE = 64 B per 128-byte line moved (the touch-per-line kernel takes the stream's time), so 2 x "
                      "FETCH_SIZE is the bytes that crossed the L2's memory side for gathers as for streams; Infinity-Cache hits are included (an upper bound on HBM bytes)")
    res["_kernels"] = detail
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
