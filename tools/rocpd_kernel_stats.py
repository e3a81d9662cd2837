#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite database into the per-kernel table `--stats` would print.
usage: rocpd_kernel_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(accum_vgpr_count), max(scratch_size), max(lds_size) "
                       "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,AGPR,ScratchBytes,LDSBytes"]
for r in rows:
    lines.append('"%s",%d,%d,%.1f,%d,%d,%.3f,%s,%s,%s,%s' % (r[0].replace('"', "'"), r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9]))
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
else:
    sys.stdout.write(out)
