"""Public Python API, signature-for-signature the reference's
``diff_gaussian_rasterization`` package
(/root/reference/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py):
``GaussianRasterizationSettings`` (:167-179), ``GaussianRasterizer`` (:181-260) and the autograd
function ``_RasterizeGaussians`` (:46-165), running on the MI355X-native kernels.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, semantics, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, stats=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, semantics, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, stats)


# extension (not in the reference): the number of tile instances R of this process' most recent forward -- the reference
# keeps it inside the autograd context only (__init__.py:93); bench.py and the densify-loop test report it
_LAST = {"num_rendered": 0}


def last_num_rendered() -> int:
    return int(_LAST["num_rendered"])


# callables invoked at the end of every rasterizer backward with that view's dL/dcolour, geometry buffer and camera
# centre; empty unless a multiview.FactoredGradReducer is alive (extension, not part of the reference API)
BACKWARD_OBSERVERS = []


# callables invoked BEFORE the native backward with the call's input tensors; each may return a dict with
#   "out": {gradient name: destination tensor}   (see _C.rasterize_gaussians_backward: e.g. views of an exchange bucket),
#   "masked_color_out": [P, 3] destination of the clamp-masked colour gradient, "skip_sh_grad": True
# (extension, not part of the reference API; empty unless a multiview reducer is alive).  Entries may be weakref.WeakMethod
# objects (the reducers register themselves that way): a reducer that is dropped without close() does not stay alive -- with
# its exchange buffers -- through this list; dead entries are pruned as they are met.
BACKWARD_SINKS = []


def _live_sinks():
    import weakref
    out = []
    for s in list(BACKWARD_SINKS):
        fn = s() if isinstance(s, weakref.WeakMethod) else s
        if fn is None:
            BACKWARD_SINKS.remove(s)
        else:
            out.append(fn)
    return out


def remove_sink(method) -> None:
    """Removes `method` (or the weak reference to it) from BACKWARD_SINKS."""
    import weakref
    for s in list(BACKWARD_SINKS):
        fn = s() if isinstance(s, weakref.WeakMethod) else s
        if fn is None or fn == method:
            BACKWARD_SINKS.remove(s)


_COLOR_EVENTS = {}


def _color_event(device):
    """One event per device, created (first record) here so that the native side only re-records it."""
    ev = _COLOR_EVENTS.get(device)
    if ev is None:
        ev = torch.cuda.Event()
        with torch.cuda.device(device):
            ev.record()
        _COLOR_EVENTS[device] = ev
    return ev


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, semantics, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, stats=None):
        # argument order of the native entry point (reference __init__.py:62-83)
        args = (raster_settings.bg, means3D, colors_precomp, semantics, opacities, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix,
                raster_settings.tanfovx, raster_settings.tanfovy, raster_settings.image_height,
                raster_settings.image_width, sh, raster_settings.sh_degree, raster_settings.campos,
                raster_settings.prefiltered, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # copy them before they can be corrupted
            try:
                (num_rendered, color, depth, alpha, semantic, radii, geomBuffer, binningBuffer,
                 imgBuffer) = _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            (num_rendered, color, depth, alpha, semantic, radii, geomBuffer, binningBuffer,
             imgBuffer) = _C.rasterize_gaussians(*args)

        _LAST["num_rendered"] = num_rendered
        ctx.raster_settings = raster_settings
        ctx.stats = stats  # extension: densification statistics updated by the backward (GaussianRasterizer.stats_sink)
        ctx.num_rendered = num_rendered
        # (extension) identity of the opacity input, which the reference does not save: a BACKWARD_SINKS provider matches its
        # parameters against the call's inputs by storage address
        ctx.opacities_key = (opacities.data_ptr(), tuple(opacities.shape)) if isinstance(opacities, torch.Tensor) else None
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer, alpha, semantics)
        ctx.mark_non_differentiable(radii)
        # autograd would otherwise hand the backward a zero-filled gradient for EVERY output it has none for, radii
        # included (a P-element fill kernel per step); the backward below fills in only the ones it reads
        ctx.set_materialize_grads(False)
        return color, radii, depth, alpha, semantic

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha, grad_semantic):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer,
         alpha, semantics) = ctx.saved_tensors
        if grad_color is None:
            grad_color = torch.zeros((3,) + alpha.shape[1:], dtype=alpha.dtype, device=alpha.device)
        if grad_depth is None:
            grad_depth = torch.zeros_like(alpha)
        if grad_alpha is None:
            grad_alpha = torch.zeros_like(alpha)
        if grad_semantic is None:
            ns = semantics.shape[1] if semantics is not None and semantics.dim() == 2 else 0
            grad_semantic = torch.zeros((ns,) + alpha.shape[1:], dtype=alpha.dtype, device=alpha.device)
        args = (raster_settings.bg, means3D, radii, colors_precomp, scales, rotations, raster_settings.scale_modifier,
                cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.tanfovx,
                raster_settings.tanfovy, grad_color, grad_depth, grad_alpha, grad_semantic, sh,
                raster_settings.sh_degree, raster_settings.campos, geomBuffer, num_rendered, binningBuffer, imgBuffer,
                alpha, semantics, raster_settings.debug)
        stats = ctx.stats
        # view-sharded training: the observers want dL/dcolour as early as it exists -- an event recorded between the row
        # sum and the per-Gaussian stage (sgr_backward_extras.color_ready_event)
        color_event = None
        if BACKWARD_OBSERVERS and means3D.is_cuda:
            color_event = _color_event(means3D.device)
        kw = {}
        if BACKWARD_SINKS and means3D.is_cuda:  # (weakly held reducers: see BACKWARD_SINKS)
            inputs = {"means3D": means3D, "scales": scales, "rotations": rotations, "sh": sh, "semantics": semantics,
                      "colors": colors_precomp, "cov3D": cov3Ds_precomp}
            for sink in _live_sinks():
                d = sink(inputs=inputs, opacities_key=ctx.opacities_key, num_points=means3D.shape[0])
                if d:
                    if d.get("out"):
                        kw.setdefault("out", {}).update(d["out"])
                    if d.get("masked_color_out") is not None:
                        kw["masked_color_out"] = d["masked_color_out"]
                    if d.get("skip_sh_grad"):
                        kw["skip_sh_grad"] = True
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
                 grad_scales, grad_rotations, grad_semantics) = _C.rasterize_gaussians_backward(*args, stats=stats,
                                                                                                 color_event=color_event, **kw)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations, grad_semantics) = _C.rasterize_gaussians_backward(*args, stats=stats,
                                                                                             color_event=color_event, **kw)

        for observer in list(BACKWARD_OBSERVERS):  # view-sharded training (multiview.FactoredGradReducer)
            observer(grad_colors=grad_colors_precomp, geomBuffer=geomBuffer, campos=raster_settings.campos,
                     sh_degree=raster_settings.sh_degree, num_points=means3D.shape[0], means3D=means3D,
                     color_ready=color_event)

        # same order as the reference (__init__.py:152-163); gradients of inputs that do not take part in
        # autograd (None / empty placeholders) are dropped instead of returned and ignored
        need = ctx.needs_input_grad
        grads = (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_semantics, grad_opacities,
                 grad_scales, grad_rotations, grad_cov3Ds_precomp, None, None)
        grads = tuple(g if (g is not None and need[i]) else None for i, g in enumerate(grads))
        return grads


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings
        # extension (not in the reference): (xyz_gradient_accum [P,2], denom [P,1], max_radii2D [P]) float32 tensors that
        # the backward of the next forward updates in place with the view's densification statistics
        # (street_gaussian_model.py:551-571) -- see street_gaussians_amd.scene.FlatStats
        self.stats_sink = None

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, semantics=None):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        if semantics is None:
            # the reference hard-codes .cuda() (reference __init__.py:218-219); same device as the Gaussians here
            semantics = torch.zeros(means3D.shape[0], 0, dtype=torch.float32, device=means3D.device)

        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, semantics, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings, self.stats_sink)

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        raster_settings = self.raster_settings
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        with torch.no_grad():
            radii, means2D = _C.rasterize_gaussians_filter(
                means3D, scales, rotations, raster_settings.scale_modifier, cov3D_precomp, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy,
                raster_settings.image_height, raster_settings.image_width, raster_settings.prefiltered,
                raster_settings.debug)
        return radii, means2D
