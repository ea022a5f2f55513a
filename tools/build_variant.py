#!/usr/bin/env python
"""Builds a tuning variant of libpienerf_hip.so with extra -D flags into pienerf_amd/lib/variants/<name>.so (select it with PN_LIB_PATH).

    python tools/build_variant.py w5 -DPN_MARCH_WAVES=5
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(b.HERE, "lib", "variants")
obj_dir = os.path.join(out_dir, "obj_" + name)
os.makedirs(obj_dir, exist_ok=True)
objs = []
only = os.environ.get("PN_VARIANT_UNITS", "pn_render_ops.hip").split(",")  # the other units are taken from the base build
b.build()
for src, extra in b.UNITS.items():
    if src not in only:
        objs.append(os.path.join(b.OBJ, src.replace(".hip", ".o")))
        continue
    o = os.path.join(obj_dir, src.replace(".hip", ".o"))
    objs.append(o)
    subprocess.run([b.hipcc()] + b.COMMON + extra + flags + ["-c", os.path.join(b.CSRC, src), "-o", o], check=True)
lib = os.path.join(out_dir, name + ".so")
subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
