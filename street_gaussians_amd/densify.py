"""Adaptive density control as plan + gather (include/sgr_densify.h).

``densify_and_prune`` computes what ``GaussianModel.densify_and_prune`` of the reference computes
(/root/reference/lib/models/gaussian_model.py:522-553: densify_and_clone :494-520, densify_and_split :448-492,
prune_points :409-427, with the Adam-state surgery of cat_optimizer / prune_optimizer :363-407) on plain tensors:
it returns the new parameter tensors, the new Adam moments and the reference's ``scalar_dict`` counters.  Wrapping the
results back into ``nn.Parameter`` / ``optimizer.state`` stays with the caller.

The split draws ``samples = normal(0, std)``; here the caller may pass the standard normals (``normals``), otherwise
they are drawn with ``torch.randn`` -- same distribution, not the same random stream.  GPU only."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _native
from ._native import SgrError, check

PARAMS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "semantic")  # optimiser group names (:409-412)


class _CParams(C.Structure):
    _fields_ = [("max_grad", C.c_float), ("min_opacity", C.c_float), ("extent", C.c_float), ("percent_dense", C.c_float),
                ("percent_big_ws", C.c_float), ("prune_big", C.c_int32), ("grad_column", C.c_int32), ("n_split", C.c_int32)]


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def densify_and_prune(params: Dict[str, torch.Tensor], xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, *,
                      max_grad: float, min_opacity: float, extent: float, percent_dense: float, percent_big_ws: float,
                      prune_big: bool, states: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]] = None,
                      grad_column: int = 0, n_split: int = 2, normals: Optional[torch.Tensor] = None):
    """params: {'xyz' [N,3], 'f_dc' [N,C,3], 'f_rest' [N,M-1,3], 'opacity' [N,1], 'scaling' [N,3], 'rotation' [N,4],
    'semantic' [N,S]} raw parameters; states: optional {name: (exp_avg, exp_avg_sq)} shaped like the parameters.
    Returns (new_params, new_states, scalars, index) with scalars = {'points_total', 'points_clone', 'points_split',
    'points_pruned'} and index = {'src', 'kind'} (source row and 0 keep / 1 clone / 2 split child per result row)."""
    xyz = params["xyz"]
    if not xyz.is_cuda:
        raise SgrError("densify_and_prune needs HIP (cuda) tensors: there is no CPU path")
    dev, N = xyz.device, xyz.shape[0]
    L = _native.lib()
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    cp = _CParams(float(max_grad), float(min_opacity), float(extent), float(percent_dense), float(percent_big_ws),
                  int(bool(prune_big)), int(grad_column), int(n_split))
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
            out = torch.empty((n_out,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev)
            check(L.sgr_densify_gather(n_out, width, _p(t), _p(src), _p(kind), int(zero_new), _p(out), stream))
            return out

        new_params = {k: gather(params[k], False) for k in PARAMS if k in params}
        if n_norm:
            if normals is None:
                normals = torch.randn(n_norm, 3, device=dev)
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
