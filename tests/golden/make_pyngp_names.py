#!/usr/bin/env python
"""Writes tests/golden/pyngp_names.json: every name the reference binds in its `pyngp` module (src/python_api.cu:306-888) — module functions,
classes, enum values, methods, properties.  Data only (the API surface a driver script can touch); run in the build container, where
/root/reference exists."""
import json, os, re
src = open("/root/reference/src/python_api.cu").read()
src = src[src.index("PYBIND11_MODULE(pyngp, m)"):]
pat = re.compile(r'\.(?:def|def_readwrite|def_readonly|def_property|def_property_readonly|def_static|value)\(\s*"([A-Za-z_0-9]+)"')
names = set(pat.findall(src))
names |= set(re.findall(r'm\.def\(\s*"([a-z_A-Z0-9]+)"', src))
names |= set(re.findall(r'py::(?:class_|enum_)<[^>]+>\s*\w*\s*\(\s*\w+\s*,\s*"([A-Za-z0-9_]+)"', src))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyngp_names.json")
json.dump(sorted(names), open(out, "w"), indent=0)
print(len(names), "names ->", out)
