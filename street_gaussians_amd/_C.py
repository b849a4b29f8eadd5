"""Native entry points with the signatures of the reference's pybind module
``diff_gaussian_rasterization._C`` (/root/reference/submodules/diff-gaussian-rasterization/ext.cpp:15-20,
rasterize_points.h:18-88) and ``simple_knn._C`` (/root/reference/submodules/simple-knn/ext.cpp:15-17),
implemented over the C ABI of libsgr_hip.so.  PyTorch is used only for device memory and the stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native
from . import _alloc
from ._native import ALLOC_FN, SgrError, SgrLazyError, check

NUM_CHANNELS = 3  # config.h:15

# Two bindings of the same C ABI: the pybind module built from csrc/ext.cpp with torch.utils.cpp_extension (what the
# reference ships, ext.cpp:15-20) and the ctypes one below (no compiler needed).  The pybind module is used for the
# five reference entry points when it has been built (street_gaussians_amd.build.build_pybind) unless
# SGR_BINDING=ctypes; everything beyond the reference's API (statistics sink, introspection, ...) is ctypes.
_ext = None
_ext_tried = False


def _pybind():
    global _ext, _ext_tried
    if not _ext_tried:
        _ext_tried = True
        import os
        if os.environ.get("SGR_BINDING", "") != "ctypes":
            _native.lib()  # libsgr_hip.so first: a missing library must raise its own, clear error
            try:
                from . import _C_pybind as m
                _ext = m
            except ImportError:
                if os.environ.get("SGR_BINDING", "") == "pybind":
                    raise
    return _ext


def binding() -> str:
    """'pybind' or 'ctypes': which binding serves the reference's entry points in this process."""
    return "pybind" if _pybind() is not None else "ctypes"


def set_binding(name: str) -> None:
    """Switch at run time (A/B of the host-side cost): 'pybind' or 'ctypes'."""
    global _ext, _ext_tried
    if name == "ctypes":
        _ext, _ext_tried = None, True
    elif name == "pybind":
        from . import _C_pybind as m
        _ext, _ext_tried = m, True
    else:
        raise ValueError(name)


def _call_ext(fn, *args):
    try:
        return fn(*args)
    except RuntimeError as ex:  # std::runtime_error of the C++ side -> the error type of the ctypes binding
        if isinstance(ex, SgrError):
            raise
        msg = str(ex).split("\n")[0]
        raise (SgrLazyError if "PREVIOUS lazy forward" in msg else SgrError)(msg) from None


class _StatSegment(C.Structure):  # sgr_stat_segment (include/sgr.h)
    _fields_ = [("src_start", C.c_int), ("count", C.c_int), ("dst_offset", C.c_int)]


class _BackwardExtras(C.Structure):  # sgr_backward_extras (include/sgr.h)
    _fields_ = [("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p), ("max_radii2D", C.c_void_p),
                ("segments", C.POINTER(_StatSegment)), ("n_segments", C.c_int), ("color_ready_event", C.c_void_p),
                ("rows", C.c_int), ("masked_color_out", C.c_void_p), ("skip_sh_grad", C.c_int)]


MAX_STAT_SEGMENTS = 128  # SGR_MAX_STAT_SEGMENTS


def _dev_check(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise SgrError(f"{name} must be a HIP (cuda) tensor: street_gaussians_amd has no CPU path")
    if t.dtype != torch.float32 and t.dtype != torch.int32 and t.dtype != torch.uint8 and t.dtype != torch.bool:
        raise SgrError(f"{name} has unsupported dtype {t.dtype}")


def _fptr(t, name="tensor"):
    """device pointer of a float tensor; empty tensor -> NULL ("feature absent")."""
    if t is None or t.numel() == 0:
        return None, None
    _dev_check(t, name)
    if t.dtype != torch.float32:
        raise SgrError(f"{name} must be float32")
    t = t.contiguous()
    return t, C.c_void_p(t.data_ptr())


class _Grow:
    """Growable byte buffer handed to the C side (resizeFunctional, rasterize_points.cu:27-33).

    The callback closes over a one-element holder, NOT over this object: a bound method (`ALLOC_FN(self._alloc)`) made
    `_Grow -> callback -> method -> _Grow` a reference cycle, and the buffer -- hundreds of MB of backward scratch per
    call -- stayed allocated until Python's cyclic collector ran (round 5, tools/densify_gc_trace.py: +1.25 GB per
    iteration at 5 M Gaussians for ~10 iterations in a row; the densify loop's "device allocations in the region")."""

    def __init__(self, device):
        holder = [torch.empty(0, dtype=torch.uint8, device=device)]

        def alloc(nbytes, _user, holder=holder, device=device):
            # (a larger block than asked for is fine -- the native side carves what it needs -- and ladder sizes repeat
            # when the number of Gaussians drifts: _alloc.py)
            holder[0] = torch.empty(_alloc.ladder(int(nbytes)), dtype=torch.uint8, device=device)
            return holder[0].data_ptr()

        self._holder = holder
        self.cb = ALLOC_FN(alloc)

    @property
    def tensor(self):
        return self._holder[0]


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def rasterize_gaussians(background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                        degree, campos, prefiltered, debug):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-124)."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _dev_check(means3D, "means3D")
    ext = _pybind()
    if ext is not None:
        e = torch.Tensor([])
        z = lambda t: e if t is None else t
        return _call_ext(ext.rasterize_gaussians, z(background), means3D, z(colors), z(semantics), z(opacity), z(scales),
                         z(rotations), float(scale_modifier), z(cov3D_precomp), z(viewmatrix), z(projmatrix),
                         float(tan_fovx), float(tan_fovy), int(image_height), int(image_width), z(sh), int(degree),
                         z(campos), bool(prefiltered), bool(debug))
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    S = semantics.size(1) if semantics is not None and semantics.ndimension() == 2 else 0
    M = sh.size(1) if sh is not None and sh.numel() != 0 and sh.size(0) != 0 else 0
    with torch.cuda.device(dev):
        fopt = dict(dtype=torch.float32, device=dev)
        out_color = torch.empty((NUM_CHANNELS, H, W), **fopt)
        out_depth = torch.empty((1, H, W), **fopt)
        out_alpha = torch.empty((1, H, W), **fopt)
        out_semantic = torch.empty((S, H, W), **fopt)
        # the preprocess kernel writes every element (0 for culled Gaussians): no zero fill (rasterize_points.cu:74)
        radii = (torch.empty if P else torch.zeros)((P,), dtype=torch.int32, device=dev)
        geom, binning, img = _Grow(dev), _Grow(dev), _Grow(dev)
        keep = []
        def p(t, n):
            t, ptr = _fptr(t, n)
            keep.append(t)
            return ptr
        rendered = check(_native.lib().sgr_forward(
            geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), M, S, p(background, "bg"), W, H,
            p(means3D, "means3D"), p(sh, "sh"), p(colors, "colors_precomp"), p(semantics, "semantics"),
            p(opacity, "opacities"), p(scales, "scales"), float(scale_modifier), p(rotations, "rotations"),
            p(cov3D_precomp, "cov3D_precomp"), p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"),
            p(campos, "campos"), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
            C.c_void_p(out_color.data_ptr()), C.c_void_p(out_depth.data_ptr()), C.c_void_p(out_alpha.data_ptr()),
            C.c_void_p(out_semantic.data_ptr()) if S else None, C.c_void_p(radii.data_ptr()) if P else None,
            int(bool(debug)), _stream(dev)))
    return rendered, out_color, out_depth, out_alpha, out_semantic, radii, geom.tensor, binning.tensor, img.tensor


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                 dL_dout_alpha, dL_dout_semantic, sh, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, alphas, semantics, debug, stats=None, color_event=None, out=None,
                                 masked_color_out=None, skip_sh_grad=False):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:126-220).  Returns
    (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dsemantic).
    stats (extension): (xyz_gradient_accum [P,2], denom [P,1], max_radii2D [P]) contiguous float32 tensors updated in
    place with this view's densification statistics (sgr_backward_ex); or a 4-tuple whose last element is a list of
    (src_start, count, dst_offset) segments mapping this call's Gaussians to rows of PERSISTENT statistics tensors of
    any length (a frame renders a subset of the sub-models, street_gaussian_model.py:230-250).
    color_event (extension): a torch.cuda.Event recorded on the current stream right after the row-sum stage, i.e. as
    soon as dL_dcolors is final and before the per-Gaussian stage runs (sgr_backward_extras.color_ready_event).
    out (extension): {"means3D" | "means2D" | "colors" | "opacity" | "cov3D" | "sh" | "scales" | "rotations" | "semantics":
    tensor} -- caller-supplied destinations for those gradients (contiguous float32 of the gradient's shape; e.g. views of
    a gradient-exchange bucket, street_gaussians_amd.multiview) instead of fresh allocations; they are what is returned.
    masked_color_out (extension): [P, 3] float32 destination of the clamp-masked colour gradient
    (sgr_backward_extras.masked_color_out).  skip_sh_grad (extension): dL_dsh is not computed, None is returned for it."""
    _dev_check(means3D, "means3D")
    ext = _pybind()
    if ext is not None and stats is None and color_event is None and not out and masked_color_out is None and not skip_sh_grad:
        e = torch.Tensor([])
        z = lambda t: e if t is None else t
        return _call_ext(ext.rasterize_gaussians_backward, z(background), means3D, radii, z(colors), z(scales), z(rotations),
                         float(scale_modifier), z(cov3D_precomp), z(viewmatrix), z(projmatrix), float(tan_fovx),
                         float(tan_fovy), dL_dout_color, dL_dout_depth, dL_dout_alpha, z(dL_dout_semantic), z(sh),
                         int(degree), z(campos), geomBuffer, int(R), binningBuffer, imageBuffer, alphas, z(semantics),
                         bool(debug))
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    S = dL_dout_semantic.size(0) if dL_dout_semantic is not None and dL_dout_semantic.numel() != 0 else 0
    M = sh.size(1) if sh is not None and sh.numel() != 0 and sh.size(0) != 0 else 0
    with torch.cuda.device(dev):
        fopt = dict(dtype=torch.float32, device=dev)
        # every element is written by the kernels (include/sgr.h), so no torch.zeros (rasterize_points.cu:166-176)
        mk0 = (lambda shape, **kw: _alloc.empty(shape, kw["dtype"], kw["device"])) if P else torch.zeros
        out = out or {}

        def mk(shape, name):
            t = out.get(name)
            if t is None:
                return mk0(shape, **fopt)
            if not (t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == tuple(shape)):
                raise SgrError(f"out[{name!r}] must be a contiguous float32 HIP tensor of shape {tuple(shape)}")
            return t
        dL_dmeans3D = mk((P, 3), "means3D")
        dL_dmeans2D = mk((P, 3), "means2D")
        dL_dcolors = mk((P, NUM_CHANNELS), "colors")
        dL_dopacity = mk((P, 1), "opacity")
        dL_dcov3D = mk((P, 6), "cov3D")
        dL_dsh = None if (skip_sh_grad and P) else mk((P, M, 3), "sh")
        dL_dscales = mk((P, 3), "scales")
        dL_drotations = mk((P, 4), "rotations")
        dL_dsemantic = mk((P, S), "semantics")
        if P == 0:
            return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
                    dL_dsemantic)
        scratch = _Grow(dev)
        keep = []
        def p(t, n):
            t, ptr = _fptr(t, n)
            keep.append(t)
            return ptr
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
        extras = None
        if stats is not None:
            acc, den, mr = stats[:3]
            segments = stats[3] if len(stats) > 3 else None
            rows = P if segments is None else den.numel()
            for t, n in ((acc, 2 * rows), (den, rows), (mr, rows)):
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
                    raise SgrError("densification statistics must be contiguous float32 HIP tensors " +
                                   ("covering all P Gaussians" if segments is None else "of one common length"))
            seg_arr, nseg = None, 0
            if segments is not None:
                nseg = len(segments)
                if nseg > MAX_STAT_SEGMENTS:
                    raise SgrError(f"at most {MAX_STAT_SEGMENTS} statistics segments per call")
                seg_arr = (_StatSegment * max(nseg, 1))()
                for k, (s0, cnt, d0) in enumerate(segments):
                    if d0 < 0 or cnt < 0 or d0 + cnt > rows:
                        raise SgrError("statistics segment outside the persistent tensors")
                    seg_arr[k] = _StatSegment(int(s0), int(cnt), int(d0))
                keep.append(seg_arr)
            extras = _BackwardExtras(acc.data_ptr(), den.data_ptr(), mr.data_ptr(), seg_arr, nseg, None, int(rows), None, 0)
        if color_event is not None or masked_color_out is not None or skip_sh_grad:
            if extras is None:
                extras = _BackwardExtras(None, None, None, None, 0, None, 0, None, 0)
            if color_event is not None:
                extras.color_ready_event = C.c_void_p(int(color_event.cuda_event))
            if masked_color_out is not None:
                m = masked_color_out
                if not (m.is_cuda and m.dtype == torch.float32 and m.is_contiguous() and m.numel() == 3 * P):
                    raise SgrError("masked_color_out must be a contiguous float32 HIP tensor of 3 * P elements")
                extras.masked_color_out = C.c_void_p(m.data_ptr())
            extras.skip_sh_grad = 1 if skip_sh_grad else 0
        check(_native.lib().sgr_backward_ex(
            P, int(degree), M, int(R), S, p(background, "bg"), W, H, p(means3D, "means3D"), p(sh, "sh"),
            p(colors, "colors_precomp"), p(semantics, "semantics"), p(alphas, "alpha"), p(scales, "scales"),
            float(scale_modifier), p(rotations, "rotations"), p(cov3D_precomp, "cov3D_precomp"),
            p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), p(campos, "campos"), float(tan_fovx),
            float(tan_fovy), vp(radii.contiguous()), vp(geomBuffer), vp(binningBuffer), vp(imageBuffer),
            p(dL_dout_color, "dL_dout_color"), p(dL_dout_depth, "dL_dout_depth"), p(dL_dout_alpha, "dL_dout_alpha"),
            p(dL_dout_semantic, "dL_dout_semantic"), vp(dL_dmeans2D), vp(dL_dopacity), vp(dL_dcolors), vp(dL_dmeans3D),
            vp(dL_dcov3D), vp(dL_dsh), vp(dL_dscales), vp(dL_drotations), vp(dL_dsemantic), scratch.cb, None,
            int(bool(debug)), _stream(dev), C.byref(extras) if extras is not None else None))
    return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
            dL_dsemantic)


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:222-241)."""
    _dev_check(means3D, "means3D")
    ext = _pybind()
    if ext is not None:
        return _call_ext(ext.mark_visible, means3D, viewmatrix, projmatrix)
    dev = means3D.device
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P:
        with torch.cuda.device(dev):
            m, v, pr = means3D.contiguous(), viewmatrix.contiguous(), projmatrix.contiguous()
            check(_native.lib().sgr_mark_visible(P, C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()),
                                                 C.c_void_p(pr.data_ptr()), C.c_void_p(present.data_ptr()),
                                                 _stream(dev)))
    return present


def rasterize_gaussians_filter(means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                               tan_fovx, tan_fovy, image_height, image_width, prefiltered, debug):
    """RasterizeGaussiansfilterCUDA (rasterize_points.cu:243-307)."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _dev_check(means3D, "means3D")
    ext = _pybind()
    if ext is not None:
        e = torch.Tensor([])
        z = lambda t: e if t is None else t
        return _call_ext(ext.rasterize_gaussians_filter, means3D, z(scales), z(rotations), float(scale_modifier),
                         z(cov3D_precomp), viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy), int(image_height),
                         int(image_width), bool(prefiltered), bool(debug))
    dev = means3D.device
    P = means3D.size(0)
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    means2D = torch.zeros((P, 2), dtype=torch.float32, device=dev)
    if P:
        with torch.cuda.device(dev):
            keep = []
            def p(t, n):
                t, ptr = _fptr(t, n)
                keep.append(t)
                return ptr
            check(_native.lib().sgr_visible_filter(
                P, int(image_width), int(image_height), p(means3D, "means3D"), p(scales, "scales"),
                float(scale_modifier), p(rotations, "rotations"), p(cov3D_precomp, "cov3D_precomp"),
                p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), float(tan_fovx), float(tan_fovy),
                int(bool(prefiltered)), C.c_void_p(radii.data_ptr()), C.c_void_p(means2D.data_ptr()),
                int(bool(debug)), _stream(dev)))
    return radii, means2D


def distCUDA2(points):
    """distCUDA2 (simple-knn/spatial.cu:16-26)."""
    _dev_check(points, "points")
    ext = _pybind()
    if ext is not None:
        return _call_ext(ext.distCUDA2, points)
    dev = points.device
    P = points.size(0)
    means = torch.zeros((P,), dtype=torch.float32, device=dev)
    if P:
        with torch.cuda.device(dev):
            pts = points.contiguous()
            if pts.dtype != torch.float32:
                raise SgrError("points must be float32")
            scratch = _Grow(dev)
            check(_native.lib().sgr_knn(P, C.c_void_p(pts.data_ptr()), C.c_void_p(means.data_ptr()), scratch.cb, None,
                                        _stream(dev)))
            torch.cuda.current_stream(dev).synchronize()  # scratch must outlive the kernels
    return means


_EXPORT = {"depths": (0, torch.float32, lambda P, R, N, T: (P,)), "clamped": (1, torch.uint8, lambda P, R, N, T: (P, 3)),
           "means2D": (2, torch.float32, lambda P, R, N, T: (P, 2)),
           "conic_opacity": (4, torch.float32, lambda P, R, N, T: (P, 4)), "rgb": (5, torch.float32, lambda P, R, N, T: (P, 3)),
           "tiles_touched": (6, torch.int32, lambda P, R, N, T: (P,)), "point_offsets": (7, torch.int32, lambda P, R, N, T: (P,)),
           "point_list": (8, torch.int32, lambda P, R, N, T: (R,)), "keys": (9, torch.int64, lambda P, R, N, T: (R,)),
           "ranges": (12, torch.int32, lambda P, R, N, T: (T, 2)), "n_contrib": (13, torch.int32, lambda P, R, N, T: (N,)),
           "extents": (14, torch.float32, lambda P, R, N, T: (P, 2)), "hits": (15, torch.uint8, lambda P, R, N, T: (R,)),
           "tile_rect": (16, torch.int32, lambda P, R, N, T: (P, 4)),
           "num_rendered_reference": (17, torch.int32, lambda P, R, N, T: (1,)),
           "tile_mask": (18, torch.int64, lambda P, R, N, T: (P,)),
           "hit_list": (19, torch.int32, lambda P, R, N, T: (R,)), "n_contrib_k": (20, torch.int32, lambda P, R, N, T: (N,))}


def masked_color_grad(geomBuffer, grad_colors, P):
    """sgr_masked_color_grad: dL/dcolour with the channels the forward clamped set to zero -> [P, 3]."""
    _dev_check(grad_colors, "grad_colors")
    dev = grad_colors.device
    out = torch.empty((int(P), 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().sgr_masked_color_grad(int(P), C.c_void_p(geomBuffer.data_ptr()),
                                                  C.c_void_p(grad_colors.contiguous().data_ptr()),
                                                  C.c_void_p(out.data_ptr()), _stream(dev)))
    return out


def sh_grad_from_views(means3D, campos, drgb, degree, M):
    """sgr_sh_grad_from_views: sum_v Y(normalize(means3D - campos[v])) (x) drgb[v] -> dL/dSH [P, M, 3].
    campos [V, 3], drgb [V, P, 3]."""
    _dev_check(means3D, "means3D")
    dev = means3D.device
    P, V = means3D.size(0), campos.size(0)
    if drgb.shape != (V, P, 3):
        raise RuntimeError("drgb must have dimensions (num_views, num_points, 3)")
    out = torch.empty((P, int(M), 3), dtype=torch.float32, device=dev)
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    means3D, campos, drgb = f32(means3D), f32(campos), f32(drgb)
    with torch.cuda.device(dev):
        check(_native.lib().sgr_sh_grad_from_views(P, int(degree), int(M), V, C.c_void_p(means3D.data_ptr()),
                                                   C.c_void_p(campos.data_ptr()), C.c_void_p(drgb.data_ptr()),
                                                   C.c_void_p(out.data_ptr()), _stream(dev)))
    return out


def sh_grad_from_rows(P, degree, M, V, means_ptr, means_stride, campos_ptr, campos_stride, drgb_ptr, drgb_stride, device):
    """sgr_sh_grad_from_views_ex on raw device addresses + per-view strides (in floats): the rebuild of
    multiview.FactoredGradReducer reads its inputs straight out of the all-gathered payload rows.  -> [P, M, 3]."""
    out = torch.empty((int(P), int(M), 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        check(_native.lib().sgr_sh_grad_from_views_ex(int(P), int(degree), int(M), int(V), C.c_void_p(means_ptr),
                                                      int(means_stride), C.c_void_p(campos_ptr), int(campos_stride),
                                                      C.c_void_p(drgb_ptr), int(drgb_stride), C.c_void_p(out.data_ptr()),
                                                      _stream(device)))
    return out


NO_CULL, NO_DPP, NO_DET, NO_HITS, USE_V2, USE_ONESWEEP, PRE_STAGE_SH, EXACT, USE_SW, USE_RS_WAVE = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
def has_variants() -> bool:
    """True when the loaded library also holds the rejected A/B designs (USE_V2, USE_ONESWEEP, USE_SW, USE_RS_WAVE): a
    tools/build_variant.py build with -DSGR_WITH_VARIANTS=1.  The shipped library ignores those switches."""
    return bool(_native.lib().sgr_has_variants())


def set_lazy(on: bool) -> bool:
    """sgr_set_lazy: the forward without a host wait (include/sgr.h).  Returns the previous setting."""
    return bool(_native.lib().sgr_set_lazy(1 if on else 0))


def lazy_status():
    """sgr_lazy_status of the calling thread -> (num_rendered, capacity, flags); synchronise the stream first."""
    r, c, f = C.c_int(0), C.c_int(0), C.c_int(0)
    check(_native.lib().sgr_lazy_status(C.byref(r), C.byref(c), C.byref(f)))
    return r.value, c.value, f.value


NO_TILE_MASK = 2048  # bounding-box rects without the per-tile mask (A/B)
TILE_SORT = 4096  # binning chain A/B: index-order emission + per-tile LDS radix sort by depth (csrc/sgr_tile_sort.hip)
LPT = 16384  # blend launches ALWAYS walk the tiles longest list first (default: decided per frame, longest list > 2.5 x the mean)
NO_HLIST = 65536  # the blend backward steps through list POSITIONS instead of the forward's compact list of hit instances (A/B)
KEY32 = 262144  # 32-bit tile keys in the instance list (default: 16-bit whenever the frame has fewer than 65535 tiles; A/B)
HLIST_ALWAYS = 131072  # ... the compact list in every mode (default: with the reference's rects, REF_RECT, only)
NO_LPT = 32768  # blend launches never do: the XCD-aware supertile order without looking at the lists (round-5 behaviour)
REF_RECT_PLAIN = 8192  # with REF_RECT: the reference's rects WITHOUT the dead-instance marks (the round-5 form of the strict mode, A/B)
REF_RECT = 1024  # emit every Gaussian for the reference's whole tile rect (default: cut down to where alpha >= 1/255 is possible)


def test_switches(mask: int = -1) -> int:
    """sgr_test_switches: A/B switches of the blend kernels (tests and tools only); returns the previous mask."""
    return int(_native.lib().sgr_test_switches(int(mask)))


def export_internal(name, P, R, image_height, image_width, geomBuffer, binningBuffer, imageBuffer):
    """Parity-test introspection (sgr_export_internal): dense copy of one internal array."""
    which, dtype, shp = _EXPORT[name]
    H, W = int(image_height), int(image_width)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    dev = geomBuffer.device
    out = torch.zeros(shp(P, R, H * W, T), dtype=dtype, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
    with torch.cuda.device(dev):
        check(_native.lib().sgr_export_internal(which, P, R, W, H, vp(geomBuffer), vp(binningBuffer), vp(imageBuffer),
                                                vp(out), _stream(dev)))
        torch.cuda.current_stream(dev).synchronize()
    return out
