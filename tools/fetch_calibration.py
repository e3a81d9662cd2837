#!/usr/bin/env python
"""Fold the plain run and the counter passes of tools/fetch_calibration.hip into one JSON (profiles/r06_fetch_calibration.json).
usage: fetch_calibration.py <plain-run json line file> <out.json> <counter dir> [<counter dir> ...]"""
import csv, glob, json, os, sys

plain = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
kernels = {"stream16": "stream16", "line_touch_4B<128>": "line128_4B", "line_touch_4B<64>": "line64_4B", "random_gather<1>": "random_4B", "random_gather<2>": "random_8B"}
counters = {}
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = next((v for k, v in kernels.items() if k in row["Kernel_Name"].replace("Li", "<").replace("E>", ">") or k.replace("<", "ILi").replace(">", "E") in row["Kernel_Name"]), None)
            if name is None:
                continue
            c = counters.setdefault(name, {}).setdefault(row["Counter_Name"], [])
            c.append(float(row["Counter_Value"]))
out = {"table_bytes": plain["table_bytes"], "kernels": {}}
for name in ("stream16", "line128_4B", "line64_4B", "random_4B", "random_8B"):
    e = dict(plain[name])
    n = e["accesses"]
    for cname, vals in sorted(counters.get(name, {}).items()):
        v = sum(vals) / len(vals)     # per dispatch
        e[cname] = round(v, 1)
        e[cname + "_per_access"] = round(v * (1024.0 if cname in ("FETCH_SIZE", "WRITE_SIZE") else 1.0) / n, 3)
    out["kernels"][name] = e
k = out["kernels"]
s, l128, l64 = k["stream16"], k["line128_4B"], k["line64_4B"]
out["conclusions"] = {
    "stream_GBps": round(plain["table_bytes"] / s["best_ms"] * 1e-6, 1),
    "line128_4B_time_over_stream_time": round(l128["best_ms"] / s["best_ms"], 3),
    "line64_4B_time_over_stream_time": round(l64["best_ms"] / s["best_ms"], 3),
    "reading": "a 4-byte touch per 128-byte line that takes the stream's time moves 128 B per touch; half the time, 64 B.  FETCH_SIZE_per_access is in bytes (counter KiB x 1024 / accesses)",
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["conclusions"]))
for name, e in k.items():
    print(name, {kk: vv for kk, vv in e.items() if kk.endswith("_per_access") or kk in ("best_ms", "G_accesses_per_s")})
