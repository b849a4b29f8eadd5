"""Adaptive density control as plan + gather (include/sgr_densify.h).

``densify_and_prune`` computes what ``GaussianModel.densify_and_prune`` of the reference computes
(/root/reference/lib/models/gaussian_model.py:522-553: densify_and_clone :494-520, densify_and_split :448-492,
prune_points :409-427, with the Adam-state surgery of cat_optimizer / prune_optimizer :363-407) on plain tensors:
it returns the new parameter tensors, the new Adam moments and the reference's ``scalar_dict`` counters.  Wrapping the
results back into ``nn.Parameter`` / ``optimizer.state`` stays with the caller.

The split draws ``samples = normal(0, std)``; here the caller may pass the standard normals (``normals``), otherwise
they are drawn with ``torch.randn`` -- same distribution, not the same random stream.  GPU only.

``variant`` selects the prune rule of the model classes street_gaussians instantiates (both override the base method):
``"bkgd"`` = GaussianModelBkgd.densify_and_prune (gaussian_model_bkgd.py:74-114: big points farther than
2 * sphere_radius from sphere_center are exempt; scalars also carry points_below_min_opacity / points_big_ws) and
``"actor"`` = GaussianModelActor.densify_and_prune (gaussian_model_actor.py:204-261: points whose sampled extent
leaves the tracking box are pruned).  Those rules look at the NEW points' positions, so the candidates (kept originals,
clones, split children) are laid out first and pruned in a second step (sgr_densify_prune_mask / _compact).
``reset_opacity`` is GaussianModel.reset_opacity (gaussian_model.py:410-414)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _alloc, _native
from ._native import SgrError, check

PARAMS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "semantic")  # optimiser group names (:409-412)


class _CParams(C.Structure):
    _fields_ = [("max_grad", C.c_float), ("min_opacity", C.c_float), ("extent", C.c_float), ("percent_dense", C.c_float),
                ("percent_big_ws", C.c_float), ("prune_big", C.c_int32), ("grad_column", C.c_int32), ("n_split", C.c_int32),
                ("defer_prune", C.c_int32)]


_VARIANTS = {None: 0, "base": 0, "bkgd": 1, "actor": 2}


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


# ---- memory for a loop that re-sizes under load ------------------------------------------------------------------------
# Every densify step changes N: parameters, gradients, Adam moments and the rasterizer's three scratch buffers come back in
# new sizes, and with an empty caching allocator every new size is a device allocation (tens of ms for a multi-GB block on
# some hosts -- in the middle of a training step).  The reference lives with that (torch.cat per parameter,
# gaussian_model.py:363-407).  Here the trainer reserves ONE block of about twice the bytes that are live at the largest
# size it expects and hands it straight back to torch's caching allocator, which then serves every re-sized tensor by
# splitting it: no device allocation inside the loop.  The pool belongs to the library (round 4 left a fixed 48 GB
# reservation in bench.py); it grows (by the difference, at least an eighth) when the estimate outgrows it; a request that did not fit is not repeated.
def live_bytes_estimate(n_points: int, sh_coeffs: int = 16, semantic_channels: int = 0, instances_per_point: float = 8.0) -> int:
    """Bytes live at n_points Gaussians in one training iteration: raw parameters, gradients, two Adam moments, the activated
    rasterizer inputs with their gradients, the geometry / binning buffers and the backward's partial rows
    (140 B per Gaussian, 18.5 + 48 B per tile instance)."""
    per_point = 4 * (3 + 3 * sh_coeffs + 1 + 3 + 4 + semantic_channels)  # one copy of the parameters
    inst = int(instances_per_point * n_points)
    return int(n_points * (4 * per_point + 2 * per_point + 140 + 64) + inst * (18.5 + 48 + 4 * semantic_channels))


class Pool:
    """reserve(n_points): makes sure torch's caching allocator holds one free block of `factor` x the live-bytes estimate
    (head-room for the size steps of the ladder, _alloc.py, and for the densify step itself, where the old and the new
    parameters + Adam moments are alive together)."""

    def __init__(self, device, factor: float = 1.5, **estimate_kw):
        self.device, self.factor, self.kw = torch.device(device), float(factor), estimate_kw
        self.reserved = 0  # bytes handed to the allocator so far (the sum of the blocks below)
        self.failed_at = None  # smallest request that did not fit: not asked for again (a failed torch allocation flushes the
                               # caching allocator first -- exactly the stall the pool exists to remove)

    def reserve(self, n_points: int) -> int:
        want = int(self.factor * live_bytes_estimate(n_points, **self.kw))
        if want <= self.reserved:
            return self.reserved
        want = max(want, self.reserved + self.reserved // 8)
        # grow by the DELTA: the block reserved earlier stays with the allocator (possibly split among live tensors), so a
        # fresh block of `want` bytes next to it would make the allocator hold reserved + want
        delta = want - self.reserved
        if self.failed_at is not None and delta >= self.failed_at:
            return self.reserved
        try:
            blk = torch.empty(delta, dtype=torch.uint8, device=self.device)
            del blk  # stays with the caching allocator as a free block
            self.reserved = want
        except RuntimeError:
            self.failed_at = delta  # not enough memory for the head-room: the loop allocates as it goes, and is not asked again
        return self.reserved


def densify_and_prune(params: Dict[str, torch.Tensor], xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, *,
                      max_grad: float, min_opacity: float, extent: float, percent_dense: float, percent_big_ws: float,
                      prune_big: bool, states: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]] = None,
                      grad_column: int = 0, n_split: int = 2, normals: Optional[torch.Tensor] = None,
                      variant: Optional[str] = None, sphere_center=None, sphere_radius: Optional[float] = None,
                      box_min=None, box_max=None, box_normals: Optional[torch.Tensor] = None, normal_source=None):
    """params: {'xyz' [N,3], 'f_dc' [N,C,3], 'f_rest' [N,M-1,3], 'opacity' [N,1], 'scaling' [N,3], 'rotation' [N,4],
    'semantic' [N,S]} raw parameters; states: optional {name: (exp_avg, exp_avg_sq)} shaped like the parameters.
    Returns (new_params, new_states, scalars, index) with scalars = {'points_total', 'points_clone', 'points_split',
    'points_pruned'} and index = {'src', 'kind'} (source row and 0 keep / 1 clone / 2 split child per result row).
    variant "bkgd" needs sphere_center [3] and sphere_radius; variant "actor" needs box_min / box_max [3] and takes
    box_normals [n_candidates, 2, 3] (standard normals; drawn when omitted).
    normal_source: a callable (rows, device[, cols]) -> standard normals, asked for the split's samples (and the actor
    variant's box samples) when the tensors are not given -- how view-sharded training keeps replicated Gaussians
    identical across ranks (multiview.ReplicatedNormals)."""
    if variant not in _VARIANTS:
        raise ValueError(f"unknown variant {variant!r}")
    if _VARIANTS[variant] != 0:
        return _densify_two_step(params, xyz_gradient_accum, denom, max_grad=max_grad, min_opacity=min_opacity,
                                 extent=extent, percent_dense=percent_dense, percent_big_ws=percent_big_ws,
                                 prune_big=prune_big, states=states, grad_column=grad_column, n_split=n_split,
                                 normals=normals, variant=variant, sphere_center=sphere_center,
                                 sphere_radius=sphere_radius, box_min=box_min, box_max=box_max, box_normals=box_normals,
                                 normal_source=normal_source)
    xyz = params["xyz"]
    if not xyz.is_cuda:
        raise SgrError("densify_and_prune needs HIP (cuda) tensors: there is no CPU path")
    dev, N = xyz.device, xyz.shape[0]
    L = _native.lib()
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    cp = _CParams(float(max_grad), float(min_opacity), float(extent), float(percent_dense), float(percent_big_ws),
                  int(bool(prune_big)), int(grad_column), int(n_split), 0)
    counts = (C.c_int64 * 6)()
    work = torch.empty(L.sgr_densify_work_bytes(N), dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    acc, den, sc, op = f32(xyz_gradient_accum), f32(denom), f32(params["scaling"]), f32(params["opacity"])
    with torch.cuda.device(dev):
        check(L.sgr_densify_plan(N, C.byref(cp), _p(acc), _p(den), _p(sc), _p(op), _p(work), counts, stream))
        n_out, n_norm = int(counts[4]), int(counts[5])
        src = torch.empty(n_out, dtype=torch.int32, device=dev)
        kind = torch.empty(n_out, dtype=torch.uint8, device=dev)
        srow = torch.empty(n_out, dtype=torch.int32, device=dev)
        check(L.sgr_densify_map(N, C.byref(cp), _p(work), _p(src), _p(kind), _p(srow), stream))

        def gather(t, zero_new):
            t = f32(t)
            width = t[0].numel() if N else 0
            out = _alloc.empty((n_out,) + tuple(t.shape[1:]), torch.float32, dev)  # ladder-sized backing: see _alloc.py
            check(L.sgr_densify_gather(n_out, width, _p(t), _p(src), _p(kind), int(zero_new), _p(out), stream))
            return out

        new_params = {k: gather(params[k], False) for k in PARAMS if k in params}
        if n_norm:
            if normals is None:
                normals = normal_source(n_norm, dev) if normal_source is not None else torch.randn(n_norm, 3, device=dev)
            if tuple(normals.shape) != (n_norm, 3):
                raise RuntimeError(f"normals must have dimensions ({n_norm}, 3)")
            check(L.sgr_densify_split_children(n_out, int(n_split), _p(src), _p(kind), _p(srow), _p(f32(params["xyz"])),
                                               _p(sc), _p(f32(params["rotation"])), _p(f32(normals)), _p(new_params["xyz"]),
                                               _p(new_params["scaling"]), stream))
        new_states = None
        if states is not None:
            new_states = {k: (gather(a, True), gather(b, True)) for k, (a, b) in states.items()}
    scalars = {"points_total": int(counts[0]), "points_clone": int(counts[1]), "points_split": int(counts[2]),
               "points_pruned": int(counts[3])}
    return new_params, new_states, scalars, {"src": src, "kind": kind}


def _gather(L, t, src, kind, n_out, zero_new, stream, N):
    t = t.detach().to(torch.float32).contiguous()
    width = t[0].numel() if N else 0
    out = _alloc.empty((n_out,) + tuple(t.shape[1:]), torch.float32, t.device)
    check(L.sgr_densify_gather(n_out, width, _p(t), _p(src), _p(kind), int(zero_new), _p(out), stream))
    return out


def _densify_two_step(params, xyz_gradient_accum, denom, *, max_grad, min_opacity, extent, percent_dense, percent_big_ws,
                      prune_big, states, grad_column, n_split, normals, variant, sphere_center, sphere_radius, box_min,
                      box_max, box_normals, normal_source=None):
    xyz = params["xyz"]
    if not xyz.is_cuda:
        raise SgrError("densify_and_prune needs HIP (cuda) tensors: there is no CPU path")
    dev, N = xyz.device, xyz.shape[0]
    L = _native.lib()
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    cp = _CParams(float(max_grad), float(min_opacity), float(extent), float(percent_dense), float(percent_big_ws),
                  int(bool(prune_big)), int(grad_column), int(n_split), 1)
    counts = (C.c_int64 * 6)()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    acc, den, sc, op = f32(xyz_gradient_accum), f32(denom), f32(params["scaling"]), f32(params["opacity"])
    with torch.cuda.device(dev):
        work = torch.empty(L.sgr_densify_work_bytes(N), dtype=torch.uint8, device=dev)
        check(L.sgr_densify_plan(N, C.byref(cp), _p(acc), _p(den), _p(sc), _p(op), _p(work), counts, stream))
        n_cand, n_norm = int(counts[4]), int(counts[5])
        src = torch.empty(n_cand, dtype=torch.int32, device=dev)
        kind = torch.empty(n_cand, dtype=torch.uint8, device=dev)
        srow = torch.empty(n_cand, dtype=torch.int32, device=dev)
        check(L.sgr_densify_map(N, C.byref(cp), _p(work), _p(src), _p(kind), _p(srow), stream))
        # the candidates' geometry: gathered rows, split children computed
        cand = {k: _gather(L, params[k], src, kind, n_cand, False, stream, N) for k in ("xyz", "scaling", "rotation", "opacity")}
        if n_norm:
            if normals is None:
                normals = normal_source(n_norm, dev) if normal_source is not None else torch.randn(n_norm, 3, device=dev)
            if tuple(normals.shape) != (n_norm, 3):
                raise RuntimeError(f"normals must have dimensions ({n_norm}, 3)")
            check(L.sgr_densify_split_children(n_cand, int(n_split), _p(src), _p(kind), _p(srow), _p(f32(params["xyz"])),
                                               _p(sc), _p(f32(params["rotation"])), _p(f32(normals)), _p(cand["xyz"]),
                                               _p(cand["scaling"]), stream))
        sphere = box = None
        if variant == "bkgd":
            if sphere_center is None or sphere_radius is None:
                raise ValueError('variant "bkgd" needs sphere_center and sphere_radius')
            c = [float(v) for v in torch.as_tensor(sphere_center).flatten().tolist()]
            sphere = (C.c_float * 4)(c[0], c[1], c[2], float(torch.as_tensor(sphere_radius).flatten()[0]))
        if variant == "actor" and prune_big:
            if box_min is None or box_max is None:
                raise ValueError('variant "actor" needs box_min and box_max')
            lo = [float(v) for v in torch.as_tensor(box_min).flatten().tolist()]
            hi = [float(v) for v in torch.as_tensor(box_max).flatten().tolist()]
            box = (C.c_float * 6)(*lo, *hi)
            if box_normals is None:
                box_normals = (normal_source(n_cand * 2, dev).view(n_cand, 2, 3) if normal_source is not None
                               else torch.randn(n_cand, 2, 3, device=dev))
            if tuple(box_normals.shape) != (n_cand, 2, 3):
                raise RuntimeError(f"box_normals must have dimensions ({n_cand}, 2, 3)")
            box_normals = f32(box_normals)
        prune = torch.empty(n_cand, dtype=torch.uint8, device=dev)
        pc = (C.c_int64 * 4)()
        check(L.sgr_densify_prune_mask(n_cand, C.byref(cp), _VARIANTS[variant], _p(cand["xyz"]), _p(cand["scaling"]),
                                       _p(cand["rotation"]), _p(cand["opacity"]), sphere, box,
                                       _p(box_normals) if box is not None else None, _p(prune), pc, stream))
        sel = torch.empty(n_cand, dtype=torch.int32, device=dev)
        n_out = C.c_int64(0)
        work2 = torch.empty(L.sgr_densify_work_bytes(n_cand), dtype=torch.uint8, device=dev)
        check(L.sgr_densify_compact(n_cand, _p(prune), _p(work2), _p(sel), C.byref(n_out), stream))
        n_out = int(n_out.value)
        sel = sel[:n_out]
        sel64 = sel.long()
        src_f, kind_f = src[sel64].contiguous(), kind[sel64].contiguous()
        keep0 = torch.zeros(n_out, dtype=torch.uint8, device=dev)  # rows copied from the candidate arrays as they are
        new_params = {}
        for k in PARAMS:
            if k not in params:
                continue
            if k in cand:
                new_params[k] = _gather(L, cand[k], sel, keep0, n_out, False, stream, n_cand)
                if params[k].dim() != new_params[k].dim():
                    new_params[k] = new_params[k].reshape((n_out,) + tuple(params[k].shape[1:]))
            else:
                new_params[k] = _gather(L, params[k], src_f, kind_f, n_out, False, stream, N)
        new_states = None
        if states is not None:
            new_states = {k: (_gather(L, a, src_f, kind_f, n_out, True, stream, N), _gather(L, b, src_f, kind_f, n_out, True, stream, N))
                          for k, (a, b) in states.items()}
    scalars = {"points_total": int(counts[0]), "points_clone": int(counts[1]), "points_split": int(counts[2]),
               "points_pruned": int(pc[3])}
    if variant == "bkgd":
        scalars["points_below_min_opacity"] = int(pc[0])
        if prune_big:
            scalars["points_big_ws"] = int(pc[1])
    return new_params, new_states, scalars, {"src": src_f, "kind": kind_f}


def reset_opacity(opacity: torch.Tensor, state: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """GaussianModel.reset_opacity (gaussian_model.py:410-414): returns inverse_sigmoid(min(sigmoid(opacity), 0.01)) as a
    new tensor; ``state`` = the group's (exp_avg, exp_avg_sq), zero-filled IN PLACE like reset_optimizer (:344-361)."""
    if not opacity.is_cuda:
        raise SgrError("reset_opacity needs a HIP (cuda) tensor: there is no CPU path")
    dev = opacity.device
    out = opacity.detach().to(torch.float32).contiguous().clone()
    a = b = None
    if state is not None:
        a, b = state
        for t in (a, b):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != out.numel():
                raise SgrError("the Adam moments must be contiguous float32 tensors shaped like opacity")
    with torch.cuda.device(dev):
        check(_native.lib().sgr_reset_opacity(out.numel(), _p(out), _p(a), _p(b),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out
