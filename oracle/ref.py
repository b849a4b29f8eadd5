"""ctypes front-end of oracle/_ref/libref_rasterizer.so -- the reference's OWN kernels compiled
unmodified for gfx950 (oracle/ref_build.sh) and run on the MI355X box.

TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as oracle/oracle.py).  Used (a) to pin the C oracle and the
HIP path against outputs of the reference itself, (b) to generate tests/golden/*.npz
(tests/golden/make_golden.py), (c) as the "reference kernels on MI355X" timing in bench.py.
All tensors are torch HIP tensors; the reference launches on the legacy default stream, so every call is
bracketed by torch.cuda.synchronize().
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_rasterizer.so")
_lib = None

INTERNAL = {"depths": (0, torch.float32, 1), "clamped": (1, torch.uint8, 3), "means2D": (2, torch.float32, 2),
            "cov3D": (3, torch.float32, 6), "conic_opacity": (4, torch.float32, 4), "rgb": (5, torch.float32, 3),
            "tiles_touched": (6, torch.int32, 1), "point_offsets": (7, torch.int32, 1)}


FMAD_LIB_PATH = os.path.join(_HERE, "_ref", "libref_rasterizer_fmad.so")
_libs = {}
_variant = "strict"


def available(variant: str = "strict") -> bool:
    return os.path.exists(LIB_PATH if variant == "strict" else FMAD_LIB_PATH)


def use(variant: str) -> None:
    """Selects which build of the reference's kernels the functions below call: "strict" = compiled with
    -ffp-contract=off (bit-comparable with the C oracle), "fmad" = the compiler's default contraction, as the reference's
    own toolchain builds it (nvcc --fmad=true): another valid rounding of the same sources (oracle/ref_build.sh)."""
    global _variant, _lib
    assert variant in ("strict", "fmad")
    _variant = variant
    _lib = _libs.get(variant)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH if _variant == "strict" else FMAD_LIB_PATH)
        _libs[_variant] = L
        L.ref_forward.restype = C.c_void_p
        L.ref_internal.restype = C.c_void_p
        L.ref_internal.argtypes = [C.c_void_p, C.c_int]
        L.ref_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(t):
    if t is None or t.numel() == 0:
        return None
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _c(t, dtype=torch.float32):
    return None if t is None else t.detach().to("cuda", dtype).contiguous()


class RefForward:
    def __init__(self):
        self.handle = None

    def free(self):
        if self.handle:
            lib().ref_free(C.c_void_p(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def internal(self, name):
        """Copy of one array of the reference's opaque buffers (carved with its own fromChunk)."""
        a = self.args
        P, R, H, W = a["P"], self.num_rendered, a["H"], a["W"]
        T = ((W + 15) // 16) * ((H + 15) // 16)
        if name in INTERNAL:
            which, dt, k = INTERNAL[name]
            n, shape = P * k, ((P, k) if k > 1 else (P,))
        elif name == "point_list":
            which, dt, n, shape = 8, torch.int32, R, (R,)
        elif name == "keys":
            which, dt, n, shape = 9, torch.int64, R, (R,)
        elif name == "ranges":
            which, dt, n, shape = 12, torch.int32, 2 * T, (T, 2)
        elif name == "n_contrib":
            which, dt, n, shape = 13, torch.int32, H * W, (H, W)
        else:
            raise KeyError(name)
        out = torch.zeros(n, dtype=dt, device="cuda")
        if n:
            src = lib().ref_internal(C.c_void_p(self.handle), which)
            nbytes = n * out.element_size()
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(src), C.c_size_t(nbytes), 3)  # device to device
            torch.cuda.synchronize()
        return out.reshape(shape)


def forward(*, means3D, opacities, viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, image_height, image_width,
            sh_degree=0, scale_modifier=1.0, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None, semantics=None) -> RefForward:
    L = lib()
    a = dict(means3D=_c(means3D), opacities=_c(opacities), viewmatrix=_c(viewmatrix), projmatrix=_c(projmatrix),
             campos=_c(campos), bg=_c(bg), shs=_c(shs), colors_precomp=_c(colors_precomp), scales=_c(scales),
             rotations=_c(rotations), cov3D_precomp=_c(cov3D_precomp), semantics=_c(semantics))
    P = a["means3D"].shape[0]
    H, W = int(image_height), int(image_width)
    S = 0 if a["semantics"] is None or a["semantics"].dim() != 2 else a["semantics"].shape[1]
    M = 0 if a["shs"] is None or a["shs"].numel() == 0 else a["shs"].shape[1]
    a.update(P=P, H=H, W=W, S=S, M=M, D=int(sh_degree), tanfovx=float(tanfovx), tanfovy=float(tanfovy),
             scale_modifier=float(scale_modifier))
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device="cuda")
    res = RefForward()
    res.args = a
    res.color, res.depth, res.alpha, res.semantic = z(3, H, W), z(1, H, W), z(1, H, W), z(S, H, W)
    res.radii = torch.zeros(P, dtype=torch.int32, device="cuda")
    nr = C.c_int(0)
    torch.cuda.synchronize()
    res.handle = L.ref_forward(
        C.c_int(P), C.c_int(a["D"]), C.c_int(M), C.c_int(S), _p(a["bg"]), C.c_int(W), C.c_int(H), _p(a["means3D"]),
        _p(a["shs"]), _p(a["colors_precomp"]), _p(a["semantics"]), _p(a["opacities"]), _p(a["scales"]),
        C.c_float(a["scale_modifier"]), _p(a["rotations"]), _p(a["cov3D_precomp"]), _p(a["viewmatrix"]),
        _p(a["projmatrix"]), _p(a["campos"]), C.c_float(a["tanfovx"]), C.c_float(a["tanfovy"]), C.c_int(0),
        _p(res.color), _p(res.depth), _p(res.alpha), _p(res.semantic), _p(res.radii), C.c_int(0), C.byref(nr))
    torch.cuda.synchronize()
    res.num_rendered = nr.value
    return res


def backward(res: RefForward, grad_color, grad_depth, grad_alpha, grad_semantic=None) -> dict:
    L = lib()
    a = res.args
    P, M, S, H, W = a["P"], a["M"], a["S"], a["H"], a["W"]
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device="cuda")
    g = dict(means2D=z(P, 3), colors=z(P, 3), depths=z(P, 1), conic=z(P, 2, 2), opacity=z(P, 1), means3D=z(P, 3),
             cov3D=z(P, 6), sh=z(P, M, 3), scales=z(P, 3), rotations=z(P, 4), semantics=z(P, S))
    gc, gd, ga = _c(grad_color), _c(grad_depth), _c(grad_alpha)
    gs = _c(grad_semantic) if grad_semantic is not None else z(S, H, W)
    torch.cuda.synchronize()
    L.ref_backward(
        C.c_void_p(res.handle), C.c_int(a["D"]), C.c_int(M), C.c_int(S), _p(a["bg"]), _p(a["means3D"]), _p(a["shs"]),
        _p(a["colors_precomp"]), _p(a["semantics"]), _p(res.alpha), _p(a["scales"]), C.c_float(a["scale_modifier"]),
        _p(a["rotations"]), _p(a["cov3D_precomp"]), _p(a["viewmatrix"]), _p(a["projmatrix"]), _p(a["campos"]),
        C.c_float(a["tanfovx"]), C.c_float(a["tanfovy"]), _p(res.radii), _p(gc), _p(gd), _p(ga), _p(gs),
        _p(g["means2D"]), _p(g["conic"]), _p(g["opacity"]), _p(g["colors"]), _p(g["depths"]), _p(g["means3D"]),
        _p(g["cov3D"]), _p(g["sh"]), _p(g["scales"]), _p(g["rotations"]), _p(g["semantics"]), C.c_int(0))
    torch.cuda.synchronize()
    return g


def dist2(points):
    pts = _c(points)
    out = torch.zeros(pts.shape[0], dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    lib().ref_knn(C.c_int(pts.shape[0]), _p(pts), _p(out))
    torch.cuda.synchronize()
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    m = _c(means3D)
    out = torch.zeros(m.shape[0], dtype=torch.bool, device="cuda")
    lib().ref_mark_visible(C.c_int(m.shape[0]), _p(m), _p(_c(viewmatrix)), _p(_c(projmatrix)), _p(out))
    torch.cuda.synchronize()
    return out


def visible_filter(*, means3D, viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width,
                   scale_modifier=1.0, scales=None, rotations=None, cov3D_precomp=None):
    m = _c(means3D)
    P = m.shape[0]
    radii = torch.zeros(P, dtype=torch.int32, device="cuda")
    m2d = torch.zeros(P, 2, dtype=torch.float32, device="cuda")
    lib().ref_visible_filter(C.c_int(P), C.c_int(0), C.c_int(int(image_width)), C.c_int(int(image_height)), _p(m),
                             _p(_c(scales)), C.c_float(scale_modifier), _p(_c(rotations)), _p(_c(cov3D_precomp)),
                             _p(_c(viewmatrix)), _p(_c(projmatrix)), C.c_float(tanfovx), C.c_float(tanfovy), C.c_int(0),
                             _p(radii), _p(m2d), C.c_int(0))
    torch.cuda.synchronize()
    return radii, m2d
