#!/usr/bin/env python
"""Matrix-core utilisation per kernel from rocprofv3 counter passes (one dir per pass; each holds *counter_collection.csv and *kernel_trace.csv).

usage: mfma_summary.py out.json dir [dir ...]

MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles (32 per v_mfma_f32_32x32x16_f16), summed over the chip's 1024 SIMDs;
GRBM_GUI_ACTIVE = busy cycles of the dispatch summed over the 8 XCDs (measured: 8 x 2.65 GHz x duration).  mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024).  The cross-check
column does the same from the instruction count (SQ_INSTS_MFMA * 32) and from the traced duration at the 2.4 GHz clock; the flop rate is
SQ_INSTS_MFMA * 32*32*16*2 / duration against the 2.5 PFLOP/s dense f16 peak.  ROCm 7.2 has no gfx950 section for the derived metrics
(MfmaUtil, VALUBusy): when present they are the gfx94x formulas and are kept as reported.
"""
import csv, glob, json, os, sys
N_SIMD = 1024
CLOCK_GHZ = 2.4
res = {}
for d in sys.argv[2:]:
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            a = dur.setdefault(k, [0.0, 0])
            a[0] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"]); a[1] += 1
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "ngp" not in k:
                continue
            a = acc.setdefault((k, row["Counter_Name"]), [0.0, 0])
            a[0] += float(row["Counter_Value"]); a[1] += 1
        for (k, c), (s, n) in acc.items():
            short = k.split("(")[0].replace("void ", "").replace("ngp::", "")[:60]
            e = res.setdefault(short, {})
            e[c] = round(s / n, 2)
            e["_dispatches"] = n
            if k in dur and dur[k][1]:
                e["_avg_us_serialised"] = round(dur[k][0] / dur[k][1] / 1000.0, 2)
for k, e in res.items():
    busy, insts, gui, us = e.get("SQ_VALU_MFMA_BUSY_CYCLES"), e.get("SQ_INSTS_MFMA"), e.get("GRBM_GUI_ACTIVE"), e.get("_avg_us_serialised")
    if insts and us:
        e["mfma_tflops"] = round(insts * 32 * 32 * 16 * 2 / (us * 1e-6) / 1e12, 2)
        e["mfma_frac_of_2500_tflops"] = round(e["mfma_tflops"] / 2500.0, 4)
        e["mfma_util_from_insts_and_duration"] = round(insts * 32 / (us * 1e3 * CLOCK_GHZ * N_SIMD), 4)
    if busy and gui:
        e["mfma_util_busy_over_gui_active"] = round(busy / (gui / 8.0 * N_SIMD), 4)
json.dump(res, open(sys.argv[1], "w"), indent=1, sort_keys=True)
# compact form bench.py quotes in its roofline object (profiles/mfma_util.json after a copy): MFMA kernels only
import re
_ks = re.search(r'KERNEL_SET = "([^"]+)"', open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read())
util = {"_meta": {"tag": os.path.basename(sys.argv[1]).replace(".json", ""), "kernel_set": _ks.group(1) if _ks else None, "peak_tflops": 2500.0,
                  "method": "rocprofv3 --pmc SQ_INSTS_MFMA (x 32*32*16*2 flop) over the traced kernel duration; kernels serialised by the counter pass"}}
for k, e in res.items():
    if e.get("SQ_INSTS_MFMA"):
        util[k.replace("_ZN3ngp", "")[:40]] = {"avg_us": e.get("_avg_us_serialised"), "tflops": e.get("mfma_tflops"), "frac_of_peak": e.get("mfma_frac_of_2500_tflops"), "MfmaUtil_pct": e.get("MfmaUtil")}
json.dump(util, open(sys.argv[1].replace(".json", "_util.json"), "w"), indent=1, sort_keys=True)
for k in sorted(res):
    e = res[k]
    if e.get("SQ_INSTS_MFMA"):
        print(k[:50].ljust(50), {x: e[x] for x in e if x.startswith("mfma") or x in ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "_avg_us_serialised", "MfmaUtil", "VALUBusy")})
