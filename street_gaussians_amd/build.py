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
SOURCES = ["sgr_preprocess.hip", "sgr_scan_sort.hip", "sgr_tile_sort.hip", "sgr_blend_fwd.hip", "sgr_blend_bwd.hip", "sgr_gauss_bwd.hip", "sgr_gauss_bwd_strict.hip",
           "sgr_knn.hip", "sgr_multiview.hip", "sgr_scene.hip", "sgr_loss.hip", "sgr_densify.hip", "sgr_api.hip"]
# Designs that were built, measured slower on MI355X and kept as A/B records: the scalar-walk blend
# backward (its own file), and -- behind `#if SGR_WITH_VARIANTS` inside the files above -- the transposed-accumulation
# backward, the one-sweep radix sorts and the wave-cooperative row sum.  NOT part of the shipped library:
# `python tools/build_variant.py <name> -DSGR_WITH_VARIANTS=1` builds a library that contains them (sgr_has_variants() = 1),
# and the tests of those paths run only against such a build.
VARIANT_SOURCES = [os.path.join("variants", "sgr_blend_bwd_sw.hip")]
HEADERS = ["sgr_common.h", "sgr_math.h", "sgr_reduce.h", os.path.join("..", "..", "include", "sgr.h"),
           os.path.join("..", "..", "include", "sgr_scene.h"), os.path.join("..", "..", "include", "sgr_loss.h"), os.path.join("..", "..", "include", "sgr_densify.h")]
# -fno-slp-vectorize: hipcc's SLP pass packs neighbouring scalar f32 ops into v_pk_* and pays for it with v_mov
# shuffles; measured on MI355X it costs 6 % in the blend backward and 7 % in the per-Gaussian backward.
# -mllvm -enable-post-misched=0: without the post-RA machine scheduler the blend kernels keep the order they were
# written in (interleaved DPP groups, paired visits); measured -1.2 % on the blend backward, -0.7 % on the step.
# -mllvm -amdgpu-use-amdgpu-trackers=1 (the AMDGPU register-pressure trackers in the scheduler): -0.3 % on both blend
# kernels, same box.  Per file: the radix-sort kernels like the max-ILP scheduling strategy (-3.5 us on the depth sort +
# scan and on the tile sort at 1 M Gaussians) which costs the blend kernels 2 %.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize", "-mllvm",
         "-enable-post-misched=0", "-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-Wall", "-Wno-unused-function"]
PER_FILE_FLAGS = {"sgr_scan_sort.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
# sources that #include another source (a second instantiation under other names / other FP settings)
INCLUDES = {"sgr_gauss_bwd_strict.hip": ["sgr_gauss_bwd.hip"]}


def flags_for(src: str) -> list:
    """Compiler flags of one kernel source (FLAGS + its per-file additions + SGR_EXTRA_FLAGS from the environment)."""
    return FLAGS + PER_FILE_FLAGS.get(os.path.basename(src), []) + os.environ.get("SGR_EXTRA_FLAGS", "").split()



def source_sha16() -> str:
    """First 16 hex digits of the SHA-256 over the compiler flags, the kernel sources and headers: identifies the build a profile was
    taken on (profiles/pmc_blend_bwd.json, bench.py's roofline.traffic)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(PER_FILE_FLAGS.items())] +
                                os.environ.get("SGR_EXTRA_FLAGS", "").split()).encode())
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
    # objects are only as fresh as the flags they were compiled with
    stamp = os.path.join(OBJ, "flags.txt")
    want = "\n".join(f"{s}: {' '.join(flags_for(s))}" for s in SOURCES)
    if not os.path.exists(stamp) or open(stamp).read() != want:
        force = True

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        dep_time = max([os.path.getmtime(s), hdr_time] + [os.path.getmtime(os.path.join(CSRC, d)) for d in INCLUDES.get(src, [])])
        if not force and os.path.exists(o) and os.path.getmtime(o) > dep_time:
            return o, False
        cmd = [hipcc] + flags_for(src) + ["-c", s, "-o", o]
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
    with open(stamp, "w") as f:
        f.write(want)
    if force or any(c for _, c in results) or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


EXT_NAME = "_C_pybind"


def build_pybind(force: bool = False, verbose: bool = False) -> str:
    """Builds street_gaussians_amd/_C_pybind*.so, the pybind module of csrc/ext.cpp, with torch.utils.cpp_extension
    (the way the reference builds its `_C`, submodules/diff-gaussian-rasterization/setup.py:21-30): host C++ only, linked
    against libsgr_hip.so next to it (rpath $ORIGIN).  In-tree, so it travels with the snapshot like the HIP library."""
    import glob
    import sysconfig
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "ext.cpp")
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    out = os.path.join(HERE, EXT_NAME + suffix)
    deps = [src, os.path.join(HERE, "..", "include", "sgr.h")]
    if not force and os.path.exists(out) and os.path.getmtime(out) > _newest(deps) and os.path.getmtime(out) > os.path.getmtime(LIB):
        return out
    torch_lib = os.path.abspath(os.path.join(os.path.dirname(__import__("torch").__file__), "lib"))
    bdir = os.path.join(OBJ, "pybind")
    os.makedirs(bdir, exist_ok=True)
    for old in glob.glob(os.path.join(HERE, EXT_NAME + "*.so")):
        os.remove(old)
    # torch.utils.cpp_extension drives the compiler (ninja + c++): ext.cpp is host code, the device code is libsgr_hip.so
    built = os.path.join(bdir, EXT_NAME + ".so")
    try:
        ce.load(name=EXT_NAME, sources=[src], build_directory=bdir, is_python_module=False, with_cuda=False,
                verbose=verbose,
                extra_cflags=["-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-Wno-deprecated-declarations"],
                extra_include_paths=["/opt/rocm/include"],
                extra_ldflags=["-L" + HERE, "-lsgr_hip", "-L" + torch_lib, "-lc10_hip", "-ltorch_python",
                               "-Wl,-rpath,'$$ORIGIN'"])
    except OSError:
        # load() also dlopens what it built, from the build directory, where $ORIGIN does not reach libsgr_hip.so:
        # the file is complete, it is loaded from its final place below
        if not os.path.exists(built):
            raise
    if not os.path.exists(built):
        raise RuntimeError(f"torch.utils.cpp_extension did not produce {built}")
    import shutil
    shutil.copyfile(built, out)
    return out


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
    print(build_pybind(force="-f" in sys.argv, verbose=True))
