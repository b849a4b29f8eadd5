#!/usr/bin/env python
"""Compiles every kernel source to ISA (no GPU needed) and lists VGPRs, LDS, scratch and the waves / SIMD they allow on
gfx950 (512 VGPRs per lane and SIMD, allocation granule 8, at most 8 waves), flagging kernels that sit a few registers
past an occupancy step.  How the 65-VGPR blend forward (7 waves instead of 8) was found.

    python tools/occupancy_audit.py [substring of the kernel name]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from street_gaussians_amd import build as b  # noqa: E402

LIMIT = {8: 64, 7: 72, 6: 80, 5: 96, 4: 128, 3: 168, 2: 256, 1: 512}


def waves(v):
    return min(8, 512 // ((v + 7) // 8 * 8))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


want = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for src in b.SOURCES:
    if not src.endswith(".hip"):
        continue
    out = os.path.join("/tmp", "audit_" + src.replace(".hip", ".s"))
    subprocess.run(["/opt/rocm/bin/hipcc"] + b.flags_for(src) + ["--offload-device-only", "-S", os.path.join(b.CSRC, src), "-o", out],
                   capture_output=True, text=True, check=True)
    s = open(out).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        g = lambda k: int(re.search(r"\.amdhsa_" + k + r" (\d+)", m.group(2)).group(1))  # noqa: E731
        rows.append((src, m.group(1), g("next_free_vgpr"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
names = demangle([r[1] for r in rows])
for src, name, v, lds, scratch in rows:
    d = re.sub(r"\(.*", "", names[name])[:64]
    if want not in d:
        continue
    w = waves(v)
    nxt = LIMIT.get(w + 1, 0)
    flag = f"  <-- {v - nxt} past {nxt} ({w + 1} waves)" if w < 8 and v - nxt <= 6 else ""
    print(f"{src[:20]:20s} {d:64s} vgpr {v:4d} waves {w} lds {lds:6d} scratch {scratch}{flag}")
