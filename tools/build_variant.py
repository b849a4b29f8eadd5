#!/usr/bin/env python
"""Builds a VARIANT of libsgr_hip.so with extra compiler flags (A/B of -D switches) into
street_gaussians_amd/variants/libsgr_hip_<name>.so; select it with SGR_LIB=<path> SGR_BINDING=ctypes.

    python tools/build_variant.py <name> <flags...>        e.g.  python tools/build_variant.py nopf -DSGR_BWD_PREFETCH=0
    python tools/build_variant.py <name> --only=sgr_blend_bwd.hip[,more.hip] <flags...>
        recompiles only the named sources with the flags and links them with the SHIPPED build's other objects
        (a switch that lives in one file: seconds instead of minutes; run street_gaussians_amd.build first)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from street_gaussians_amd import build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
only = None
for f in list(flags):
    if f.startswith("--only="):
        only = f.split("=", 1)[1].split(",")
        flags.remove(f)
out_dir = os.path.join(b.HERE, "variants")
obj_dir = os.path.join(b.OBJ, "variant_" + name)
os.makedirs(out_dir, exist_ok=True)
os.makedirs(obj_dir, exist_ok=True)


def one(src):
    if only is not None and src not in only:
        o = os.path.join(b.OBJ, src.replace(".hip", ".o"))  # the shipped build's object
        assert os.path.exists(o), f"{o}: build the shipped library first"
        return o
    o = os.path.join(obj_dir, src.replace(".hip", ".o"))
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.flags_for(src) + flags + ["-c", os.path.join(b.CSRC, src), "-o", o])
    return o


sources = list(b.SOURCES)
if any(f.replace(" ", "") == "-DSGR_WITH_VARIANTS=1" for f in flags):  # the rejected A/B designs ride along
    sources += b.VARIANT_SOURCES
    os.makedirs(os.path.join(obj_dir, "variants"), exist_ok=True)
with ThreadPoolExecutor(max_workers=8) as ex:
    objs = list(ex.map(one, sources))
lib = os.path.join(out_dir, f"libsgr_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
