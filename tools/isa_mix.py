#!/usr/bin/env python3
"""Static instruction mix of one kernel in a `hipcc --cuda-device-only -S` listing: tools/isa_mix.py file.s <substring of the mangled name> [...]"""
import collections
import re
import sys


def mix(path, key):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(";")[0].strip().endswith(":"))
    ops = collections.Counter()
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if not l.startswith("\t") or t.startswith((".", ";")):
            continue
        ops[t.split()[0]] += 1
    return ops


if __name__ == "__main__":
    path = sys.argv[1]
    for key in sys.argv[2:]:
        ops = mix(path, key)
        print(key, "total", sum(ops.values()))
        print("  ", ", ".join(f"{k} {v}" for k, v in ops.most_common(16)))
