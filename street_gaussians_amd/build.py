"""Builds libsgr_hip.so (the HIP kernels + C ABI of include/sgr.h) for gfx950 with hipcc, in-tree.

    python -m street_gaussians_amd.build [-f]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsgr_hip.so")
SOURCES = ["sgr_preprocess.hip", "sgr_scan_sort.hip", "sgr_blend_fwd.hip", "sgr_blend_bwd.hip", "sgr_gauss_bwd.hip",
           "sgr_knn.hip", "sgr_multiview.hip", "sgr_scene.hip", "sgr_loss.hip", "sgr_densify.hip", "sgr_api.hip"]
HEADERS = ["sgr_common.h", "sgr_math.h", os.path.join("..", "..", "include", "sgr.h"),
           os.path.join("..", "..", "include", "sgr_scene.h"), os.path.join("..", "..", "include", "sgr_loss.h"), os.path.join("..", "..", "include", "sgr_densify.h")]
# -fno-slp-vectorize: hipcc's SLP pass packs neighbouring scalar f32 ops into v_pk_* and pays for it with v_mov
# shuffles; measured on MI355X it costs 6 % in the blend backward and 7 % in the per-Gaussian backward.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function"]


def source_sha16() -> str:
    """First 16 hex digits of the SHA-256 over the kernel sources and headers: identifies the build a profile was
    taken on (profiles/pmc_blend_bwd.json, bench.py's roofline.traffic)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + [x for x in HEADERS if not x.startswith("..")]):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = _newest([os.path.join(CSRC, h) for h in HEADERS])

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(s), hdr_time):
            return o, False
        cmd = [hipcc] + FLAGS + os.environ.get("SGR_EXTRA_FLAGS", "").split() + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return o, True

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
