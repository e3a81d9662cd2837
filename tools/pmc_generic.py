#!/usr/bin/env python
"""Average every collected counter per kernel (ngp kernels only) from rocprofv3 *counter_collection.csv files under the given dirs.
usage: pmc_generic.py out.json dir [dir ...]"""
import csv, glob, json, os, sys
res = {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "ngp" not in k:
                continue
            short = k.split("(")[0].replace("_ZN3ngp", "").replace("void ", "").replace("ngp::", "")[:40]
            key = (short, row["Counter_Name"])
            a = acc.setdefault(key, [0.0, 0])
            a[0] += float(row["Counter_Value"]); a[1] += 1
        for (k, c), (s, n) in acc.items():
            res.setdefault(k, {})[c] = round(s / n, 1)
            res[k]["_dispatches"] = n
json.dump(res, open(sys.argv[1], "w"), indent=1, sort_keys=True)
