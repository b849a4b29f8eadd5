"""ctypes front-end of the plain-C oracle (oracle/sgr_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module.  The product package
(``street_gaussians_amd``) never does; it fails loudly when its HIP library is missing.

The functions mirror the reference's native entry points
(/root/reference/submodules/diff-gaussian-rasterization/rasterize_points.h:18-88 and
/root/reference/submodules/simple-knn/spatial.h:14) over numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsgr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "sgr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.sgo_forward.restype = C.c_void_p
        _lib.sgo_free.argtypes = [C.c_void_p]
        _lib.sgo_num_rendered.argtypes = [C.c_void_p]
        _lib.sgo_num_rendered.restype = C.c_int
        for name in ("depths", "clamped", "radii", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched",
                     "point_offsets", "keys_unsorted", "vals_unsorted", "keys", "point_list", "ranges", "n_contrib"):
            f = getattr(_lib, "sgo_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_void_p
        _lib.sgo_get_higher_msb.argtypes = [C.c_uint32]
        _lib.sgo_get_higher_msb.restype = C.c_uint32
    return _lib


def _np(x, dtype=np.float32):
    """numpy view of a numpy array / torch tensor / None (-> None)."""
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    x = np.ascontiguousarray(x, dtype=dtype)
    return x


def _ptr(x):
    if x is None or x.size == 0:
        return None
    return x.ctypes.data_as(C.c_void_p)


def _view(addr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not addr:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


@dataclass
class ForwardResult:
    color: np.ndarray
    radii: np.ndarray
    depth: np.ndarray
    alpha: np.ndarray
    semantic: np.ndarray
    num_rendered: int
    # internals (GeometryState / BinningState / ImageState of the reference)
    depths: np.ndarray = None
    clamped: np.ndarray = None
    means2D: np.ndarray = None
    cov3D: np.ndarray = None
    conic_opacity: np.ndarray = None
    rgb: np.ndarray = None
    tiles_touched: np.ndarray = None
    point_offsets: np.ndarray = None
    keys_unsorted: np.ndarray = None
    vals_unsorted: np.ndarray = None
    keys: np.ndarray = None
    point_list: np.ndarray = None
    ranges: np.ndarray = None
    n_contrib: np.ndarray = None
    _state: int = 0
    _args: dict = field(default_factory=dict)

    def free(self):
        if self._state:
            lib().sgo_free(C.c_void_p(self._state))
            self._state = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def forward(*, means3D, opacities, viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, image_height, image_width,
            sh_degree=0, scale_modifier=1.0, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None, semantics=None, internals=True) -> ForwardResult:
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-124) on the CPU oracle."""
    L = lib()
    means3D = _np(means3D)
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    shs = _np(shs)
    colors_precomp = _np(colors_precomp)
    semantics = _np(semantics)
    S = 0 if semantics is None else (semantics.shape[1] if semantics.ndim == 2 else 0)
    if S > 20:  # the reference's per-pixel arrays hold NUM_CLASSES = 20 entries (cuda_rasterizer/config.h:16)
        raise ValueError("the reference algorithm supports at most 20 semantic channels")
    M = 0 if shs is None or shs.size == 0 else shs.shape[1]
    a = dict(means3D=means3D, opacities=_np(opacities), viewmatrix=_np(viewmatrix), projmatrix=_np(projmatrix),
             campos=_np(campos), bg=_np(bg), shs=shs, colors_precomp=colors_precomp, scales=_np(scales),
             rotations=_np(rotations), cov3D_precomp=_np(cov3D_precomp), semantics=semantics,
             tanfovx=float(tanfovx), tanfovy=float(tanfovy), scale_modifier=float(scale_modifier),
             sh_degree=int(sh_degree), M=M, S=S, H=H, W=W, P=P)
    out_color = np.zeros((3, H, W), np.float32)
    out_depth = np.zeros((1, H, W), np.float32)
    out_alpha = np.zeros((1, H, W), np.float32)
    out_sem = np.zeros((S, H, W), np.float32)
    radii = np.zeros((P,), np.int32)
    if P == 0:
        return ForwardResult(out_color, radii, out_depth, out_alpha, out_sem, 0, _args=a)
    st = L.sgo_forward(
        C.c_int(P), C.c_int(a["sh_degree"]), C.c_int(M), C.c_int(S), _ptr(a["bg"]), C.c_int(W), C.c_int(H),
        _ptr(means3D), _ptr(shs), _ptr(colors_precomp), _ptr(semantics), _ptr(a["opacities"]), _ptr(a["scales"]),
        C.c_float(a["scale_modifier"]), _ptr(a["rotations"]), _ptr(a["cov3D_precomp"]), _ptr(a["viewmatrix"]),
        _ptr(a["projmatrix"]), _ptr(a["campos"]), C.c_float(a["tanfovx"]), C.c_float(a["tanfovy"]),
        _ptr(out_color), _ptr(out_depth), _ptr(out_alpha), _ptr(out_sem), _ptr(radii))
    R = L.sgo_num_rendered(C.c_void_p(st))
    res = ForwardResult(out_color, radii, out_depth, out_alpha, out_sem, R, _state=st, _args=a)
    if internals:
        T = ((W + 15) // 16) * ((H + 15) // 16)
        g = lambda n, shape, dt: _view(getattr(L, "sgo_" + n)(C.c_void_p(st)), shape, dt)
        res.depths = g("depths", (P,), np.float32)
        res.clamped = g("clamped", (P, 3), np.uint8)
        res.means2D = g("means2D", (P, 2), np.float32)
        res.cov3D = g("cov3D", (P, 6), np.float32)
        res.conic_opacity = g("conic_opacity", (P, 4), np.float32)
        res.rgb = g("rgb", (P, 3), np.float32)
        res.tiles_touched = g("tiles_touched", (P,), np.uint32)
        res.point_offsets = g("point_offsets", (P,), np.uint32)
        res.keys_unsorted = g("keys_unsorted", (R,), np.uint64)
        res.vals_unsorted = g("vals_unsorted", (R,), np.uint32)
        res.keys = g("keys", (R,), np.uint64)
        res.point_list = g("point_list", (R,), np.uint32)
        res.ranges = g("ranges", (T, 2), np.uint32)
        res.n_contrib = g("n_contrib", (H, W), np.uint32)
    return res


def backward(fw: ForwardResult, grad_color, grad_depth, grad_alpha, grad_semantic=None, parallel=False) -> dict:
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:126-220) on the CPU oracle.

    parallel=False gives the deterministic accumulation order; parallel=True uses all OpenMP
    threads with unordered atomic float adds (what the reference's atomicAdd does); parallel="exact"
    uses all threads and sums the same float terms in double, rounded to float once -- the order-free
    value the unordered float sums scatter around (used at the BASELINE sizes)."""
    L = lib()
    a = fw._args
    P, M, S, H, W = a["P"], a["M"], a["S"], a["H"], a["W"]
    z = lambda *shape: np.zeros(shape, np.float32)
    g = dict(means2D=z(P, 3), colors=z(P, 3), depths=z(P, 1), conic=z(P, 2, 2), opacity=z(P, 1), means3D=z(P, 3),
             cov3D=z(P, 6), sh=z(P, M, 3), scales=z(P, 3), rotations=z(P, 4), semantics=z(P, S))
    if P == 0:
        return g
    gc, gd, ga = _np(grad_color), _np(grad_depth), _np(grad_alpha)
    gs = _np(grad_semantic) if grad_semantic is not None else z(S, H, W)
    L.sgo_backward(
        C.c_void_p(fw._state), C.c_int(a["sh_degree"]), C.c_int(M), C.c_int(S), _ptr(a["bg"]), _ptr(a["means3D"]),
        _ptr(a["shs"]), _ptr(a["colors_precomp"]), _ptr(a["semantics"]), _ptr(fw.alpha), _ptr(a["scales"]),
        C.c_float(a["scale_modifier"]), _ptr(a["rotations"]), _ptr(a["cov3D_precomp"]), _ptr(a["viewmatrix"]),
        _ptr(a["projmatrix"]), _ptr(a["campos"]), C.c_float(a["tanfovx"]), C.c_float(a["tanfovy"]), _ptr(fw.radii),
        _ptr(gc), _ptr(gd), _ptr(ga), _ptr(gs), _ptr(g["means2D"]), _ptr(g["conic"]), _ptr(g["opacity"]),
        _ptr(g["colors"]), _ptr(g["depths"]), _ptr(g["means3D"]), _ptr(g["cov3D"]), _ptr(g["sh"]), _ptr(g["scales"]),
        _ptr(g["rotations"]), _ptr(g["semantics"]), C.c_int(2 if parallel == "exact" else (1 if parallel else 0)))
    return g


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    """markVisible (rasterize_points.cu:222-241)."""
    means3D = _np(means3D)
    P = means3D.shape[0]
    out = np.zeros((P,), np.uint8)
    if P:
        lib().sgo_mark_visible(C.c_int(P), _ptr(means3D), _ptr(_np(viewmatrix)), _ptr(_np(projmatrix)), _ptr(out))
    return out.astype(bool)


def visible_filter(*, means3D, viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width,
                   scale_modifier=1.0, scales=None, rotations=None, cov3D_precomp=None):
    """RasterizeGaussiansfilterCUDA (rasterize_points.cu:243-307)."""
    means3D = _np(means3D)
    P = means3D.shape[0]
    radii = np.zeros((P,), np.int32)
    means2D = np.zeros((P, 2), np.float32)
    if P:
        lib().sgo_visible_filter(C.c_int(P), C.c_int(int(image_width)), C.c_int(int(image_height)), _ptr(means3D),
                                 _ptr(_np(scales)), C.c_float(scale_modifier), _ptr(_np(rotations)),
                                 _ptr(_np(cov3D_precomp)), _ptr(_np(viewmatrix)), _ptr(_np(projmatrix)),
                                 C.c_float(tanfovx), C.c_float(tanfovy), _ptr(radii), _ptr(means2D))
    return radii, means2D


def dist2(points, return_internals=False):
    """distCUDA2 (KNN/spatial.cu:16-26 -> simple_knn.cu:185-220)."""
    points = _np(points)
    P = points.shape[0]
    out = np.zeros((P,), np.float32)
    morton = np.zeros((P,), np.uint32)
    idx = np.zeros((P,), np.uint32)
    if P:
        lib().sgo_knn(C.c_int(P), _ptr(points), _ptr(out), _ptr(morton), _ptr(idx))
    return (out, morton, idx) if return_internals else out


def get_higher_msb(n: int) -> int:
    return int(lib().sgo_get_higher_msb(C.c_uint32(n)))
