"""Scene-graph rows next to the rasterizer (include/sgr_scene.h).

``compose`` flattens a street_gaussians scene graph -- one static background model plus one rigid, per-frame posed
model per visible actor -- into the rasterizer's inputs, as one autograd op backed by HIP kernels.  It computes what
``StreetGaussianModel.get_xyz / get_rotation / get_scaling / get_opacity / get_features / get_semantic`` compute
with per-attribute ``torch.cat`` / ``einsum`` / quaternion products (/root/reference/lib/models/
street_gaussian_model.py:287-449, gaussian_model.py:224-251, gaussian_model_actor.py:62-80), and back-propagates to
every raw parameter and to each actor's pose.  ``densification_stats`` is the per-model scatter of
``add_densification_stats`` + ``set_max_radii2D`` (:551-571) in one pass.

There is no CPU implementation: tensors must live on the GPU and the HIP library must be built.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

from . import _native
from ._native import ALLOC_FN, SgrError, check

SEG_STATIC, SEG_ACTOR = 0, 1
SEM_LOGITS, SEM_PROBABILITIES = 0, 1
_SEM = {"logits": SEM_LOGITS, "probabilities": SEM_PROBABILITIES}


class _CSeg(C.Structure):
    _fields_ = [("count", C.c_int32), ("kind", C.c_int32), ("fourier_dim", C.c_int32), ("class_label", C.c_int32),
                ("sem_mode", C.c_int32), ("flip_axis", C.c_int32), ("flip_quat", C.c_float * 4),
                ("xyz", C.c_void_p), ("rotation", C.c_void_p), ("scaling", C.c_void_p), ("opacity", C.c_void_p),
                ("features_dc", C.c_void_p), ("features_rest", C.c_void_p), ("semantic", C.c_void_p),
                ("flip_mask", C.c_void_p), ("pose", C.c_void_p), ("idft", C.c_void_p)]


class _CSegGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("xyz", "rotation", "scaling", "opacity", "features_dc", "features_rest",
                                         "semantic", "pose")]


class _CStatSeg(C.Structure):
    _fields_ = [("count", C.c_int32), ("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p),
                ("max_radii2D", C.c_void_p)]


_TENSORS = ("xyz", "rotation", "scaling", "opacity", "features_dc", "features_rest", "semantic", "pose")


@dataclass
class Segment:
    """One sub-model, raw (pre-activation) parameters as the reference stores them."""
    xyz: torch.Tensor            # [n, 3]
    rotation: torch.Tensor       # [n, 4]
    scaling: torch.Tensor        # [n, 3]
    opacity: torch.Tensor        # [n, 1]
    features_dc: torch.Tensor    # [n, fourier_dim, 3]  (fourier_dim = 1 for the background)
    features_rest: torch.Tensor  # [n, M-1, 3]
    semantic: Optional[torch.Tensor] = None   # background [n, S]; actor [n, 1]
    pose: Optional[torch.Tensor] = None       # actor: [7] = obj_rot (w, x, y, z), obj_trans; None = static model
    idft: Optional[torch.Tensor] = None       # actor: [fourier_dim]
    flip_mask: Optional[torch.Tensor] = None  # actor, training: [n] bool
    class_label: int = 0
    semantic_mode: str = "logits"
    flip_axis: int = 1
    flip_quat: Sequence[float] = field(default_factory=lambda: (0.0, 0.0, 1.0, 0.0))

    @property
    def kind(self) -> int:
        return SEG_ACTOR if self.pose is not None else SEG_STATIC


class _Grow:
    def __init__(self, device):
        self.device = device
        self.tensor = None
        self.cb = ALLOC_FN(self._alloc)

    def _alloc(self, nbytes, _user):
        self.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self.tensor.data_ptr()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise SgrError(f"{name} must be a HIP (cuda) tensor: there is no CPU path")
    if t.dtype == torch.float32 and t.is_contiguous():
        return t  # only the data pointer is used
    return t.detach().to(torch.float32).contiguous()


def _pack(segs: List[Segment], tensors: List[Optional[torch.Tensor]]):
    arr = (_CSeg * len(segs))()
    keep = []
    it = iter(tensors)
    for c, s in zip(arr, segs):
        vals = {n: _f32c(next(it), n) for n in _TENSORS}
        keep.append(vals)
        n = vals["xyz"].shape[0]
        c.count, c.kind = n, s.kind
        c.fourier_dim = vals["features_dc"].shape[1] if vals["features_dc"].dim() == 3 else 1
        c.class_label, c.sem_mode, c.flip_axis = int(s.class_label), _SEM[s.semantic_mode], int(s.flip_axis)
        c.flip_quat = (C.c_float * 4)(*[float(v) for v in s.flip_quat])
        for name in _TENSORS:
            setattr(c, name, vals[name].data_ptr() if vals[name] is not None and vals[name].numel() else None)
        fm = None
        if s.flip_mask is not None:
            fm = s.flip_mask
            # a bool mask is one byte per element already: reinterpret, no conversion kernel
            fm = fm.view(torch.uint8) if fm.dtype == torch.bool and fm.is_contiguous() else fm.to(torch.uint8).contiguous()
            keep.append(fm)
        c.flip_mask = fm.data_ptr() if fm is not None and fm.numel() else None
        idft = _f32c(s.idft, "idft")
        keep.append(idft)
        c.idft = idft.data_ptr() if idft is not None else None
    return arr, keep


class _Compose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, segs, M, S, *tensors):
        dev = tensors[0].device
        arr, keep = _pack(segs, list(tensors))
        N = sum(int(c.count) for c in arr)
        f = dict(dtype=torch.float32, device=dev)
        outs = [torch.empty(N, 3, **f), torch.empty(N, 4, **f), torch.empty(N, 3, **f), torch.empty(N, 1, **f),
                torch.empty(N, M, 3, **f), torch.empty(N, S, **f)]
        grow = _Grow(dev)
        with torch.cuda.device(dev):
            check(_native.lib().sgr_scene_compose_forward(
                len(segs), arr, int(M), int(S), *[_ptr(o) if o.numel() else None for o in outs], grow.cb, None,
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        ctx.segs, ctx.M, ctx.S = segs, int(M), int(S)
        ctx.save_for_backward(*[t for t in tensors if t is not None])
        ctx.present = [t is not None for t in tensors]
        ctx.packed = (arr, keep)  # same storages in backward (autograd forbids in-place changes of saved tensors)
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_means, d_rot, d_scale, d_opac, d_shs, d_sem):
        saved = iter(ctx.saved_tensors)
        tensors = [next(saved) if p else None for p in ctx.present]
        segs, M, S = ctx.segs, ctx.M, ctx.S
        dev = tensors[0].device
        arr, keep = ctx.packed
        garr = (_CSegGrads * len(segs))()
        need = ctx.needs_input_grad[3:]
        # one allocation for all gradients (16-byte aligned slices), carved into per-parameter views with ONE split
        # call + one view per parameter: this runs on the autograd thread and is pure host time next to ~0.4 ms of kernels
        want = [t is not None and need[i] for i, t in enumerate(tensors)]
        idx = [i for i, w in enumerate(want) if w]
        sizes = [((tensors[i].numel() + 3) // 4) * 4 for i in idx]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        grads: List[Optional[torch.Tensor]] = [None] * len(tensors)
        for i, piece, size in zip(idx, flat.split_with_sizes(sizes) if idx else (), sizes):
            t = tensors[i]
            grads[i] = (piece if size == t.numel() else piece[:t.numel()]).view(t.shape)
        nt = len(_TENSORS)
        for k, g in enumerate(garr):
            for j, name in enumerate(_TENSORS):
                out = grads[k * nt + j]
                if out is not None and out.numel():
                    setattr(g, name, out.data_ptr())
        dz = lambda t: None if t is None else _f32c(t, "grad")
        ins = [dz(d_means), dz(d_rot), dz(d_scale), dz(d_opac), dz(d_shs), dz(d_sem) if S else None]
        grow = _Grow(dev)
        with torch.cuda.device(dev):
            check(_native.lib().sgr_scene_compose_backward(
                len(segs), arr, garr, M, S, *[_ptr(t) if t is not None and t.numel() else None for t in ins], grow.cb,
                None, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        del keep
        # kernels work in float32; hand autograd the dtype of each input
        grads = [g if g is None or g.dtype == t.dtype else g.to(t.dtype) for g, t in zip(grads, tensors)]
        return (None, None, None) + tuple(grads)


def compose(segments: List[Segment], max_sh_coeffs: int, num_classes: int):
    """Returns (means3D [N,3], rotations [N,4], scales [N,3], opacities [N,1], shs [N,M,3], semantics [N,S]) for the
    concatenation of ``segments`` (background first, then the visible actors, like ``parse_camera`` orders them)."""
    if not segments:
        raise ValueError("need at least one segment")
    flat = []
    for s in segments:
        flat += [getattr(s, n) for n in _TENSORS]
    return _Compose.apply(list(segments), int(max_sh_coeffs), int(num_classes), *flat)


# ---- flat-parameter mode ---------------------------------------------------------------------------------------------
# `compose` takes one leaf tensor per (sub-model, attribute): a street scene with 20 actors crosses the autograd boundary
# with 167 leaves, and the engine's per-leaf work (an AccumulateGrad node each) costs more than the kernels (1.56 ms wall
# for 0.71 ms of kernels at 2 M Gaussians).  In flat mode the scene's parameters live in ONE leaf tensor per attribute
# (all sub-models concatenated, the per-model tensors are views of it) and every actor's pose in one [n_actors, 7] leaf: 8
# leaves whatever the number of actors.  The kernels are the same -- they take per-segment pointers, which here point into
# the flat storages.
_FLAT = ("xyz", "rotation", "scaling", "opacity", "features_dc", "features_rest", "semantic")


class FlatScene:
    """The raw parameters of a scene graph as one leaf tensor per attribute.

    ``FlatScene.from_segments(segments)`` concatenates the segments' tensors (rows of features_dc are [fourier_dim * 3]
    wide per Gaussian and of semantic [S] for the background / [1] for an actor, so those two are flattened to 1-D); the
    result's ``xyz, rotation, scaling, opacity, features_dc, features_rest, semantic`` and ``poses`` [n_actors, 7] require
    grad and are what an optimiser holds; ``views()`` gives the per-model tensors (views: densification, I/O).
    ``compose(M, S, flip_masks=None, idfts=None)`` is ``scene.compose`` for the current values."""

    def __init__(self, meta, tensors, poses):
        self.meta = meta              # per segment: dict(count, kind, fourier_dim, sem_width, class_label, semantic_mode, flip_*)
        self.tensors = tensors        # attr -> flat leaf
        self.poses = poses            # [n_actors, 7] leaf or None
        for k in _FLAT:
            setattr(self, k, tensors[k])

    @classmethod
    def from_segments(cls, segments: Sequence[Segment], requires_grad: bool = True) -> "FlatScene":
        meta, parts = [], {k: [] for k in _FLAT}
        poses = []
        dev = segments[0].xyz.device
        for s in segments:
            n = int(s.xyz.shape[0])
            C = int(s.features_dc.shape[1]) if s.features_dc.dim() == 3 else 1
            sw = 0 if s.semantic is None or s.semantic.numel() == 0 else int(s.semantic.shape[1])
            meta.append(dict(count=n, kind=s.kind, fourier_dim=C, sem_width=sw, class_label=int(s.class_label),
                             semantic_mode=s.semantic_mode, flip_axis=int(s.flip_axis), flip_quat=tuple(s.flip_quat),
                             idft=None if s.idft is None else s.idft.detach().float().to(dev)))
            for k in _FLAT:
                t = getattr(s, k)
                if t is None:
                    # row-wise attributes share ONE cumulative row offset (views(), the kernel pointers): a segment
                    # without one of them would shift every later segment's block.  Only `semantic` has its own offset
                    # (sem_width may be 0).
                    if k != "semantic":
                        raise ValueError(f"FlatScene: segment {len(meta) - 1} has no `{k}`; every segment must carry "
                                         "xyz, rotation, scaling, opacity, features_dc and features_rest")
                    continue
                t = t.detach().float()
                parts[k].append(t.reshape(-1) if k in ("features_dc", "semantic") else t.reshape(n, -1))
            if s.pose is not None:
                poses.append(s.pose.detach().float().reshape(7))
        tensors = {}
        for k in _FLAT:
            t = torch.cat(parts[k], 0).contiguous() if parts[k] else torch.zeros(0, device=dev)
            tensors[k] = t.requires_grad_(requires_grad)
        n_actors = sum(1 for m in meta if m["kind"] == SEG_ACTOR)
        if len(poses) != n_actors:
            raise ValueError(f"FlatScene: {n_actors} actor segments but {len(poses)} poses (every actor needs its "
                             "[7] pose: obj_rot wxyz + obj_trans)")
        P = torch.stack(poses).contiguous().requires_grad_(requires_grad) if poses else None
        return cls(meta, tensors, P)

    def _offsets(self):
        """Per segment: element offsets of its block inside each flat tensor."""
        offs, row, dc, sem = [], 0, 0, 0
        for m in self.meta:
            offs.append(dict(row=row, dc=dc, sem=sem))
            row += m["count"]
            dc += m["count"] * m["fourier_dim"] * 3
            sem += m["count"] * m["sem_width"]
        return offs

    def views(self) -> List[dict]:
        out, a = [], 0
        for m, o in zip(self.meta, self._offsets()):
            n = m["count"]
            d = {k: self.tensors[k][o["row"]:o["row"] + n] for k in ("xyz", "rotation", "scaling", "opacity", "features_rest")}
            d["features_rest"] = d["features_rest"].view(n, -1, 3)
            d["features_dc"] = self.tensors["features_dc"][o["dc"]:o["dc"] + n * m["fourier_dim"] * 3].view(n, m["fourier_dim"], 3)
            d["semantic"] = self.tensors["semantic"][o["sem"]:o["sem"] + n * m["sem_width"]].view(n, m["sem_width"])
            if m["kind"] == SEG_ACTOR:
                d["pose"] = self.poses[a]
                a += 1
            out.append(d)
        return out

    def compose(self, max_sh_coeffs: int, num_classes: int, flip_masks: Optional[Sequence] = None,
                poses: Optional[torch.Tensor] = None, idfts: Optional[Sequence] = None):
        """Same result as ``scene.compose(segments, M, S)``; gradients arrive in the 7 flat leaves and in the poses.

        Per frame (street_gaussian_model.py:230-330): ``poses`` [n_actors, 7] (obj_rot wxyz + obj_trans in world space) is
        usually a NON-leaf the caller derives from its tracking-pose refinements and the ego pose -- pass it here and
        the gradient flows on to those; default: the ``poses`` leaf held by this object.  ``idfts`` = one IDFT row
        [fourier_dim] per segment (None for the background): it depends on the frame's timestamp
        (gaussian_model_actor.py:71-80); default: the rows the segments carried at construction.  ``flip_masks`` = one
        bool [n] mask per segment or None (training-time symmetry flips, :270-283)."""
        fm = list(flip_masks) if flip_masks is not None else [None] * len(self.meta)
        P = self.poses if poses is None else poses
        n_actors = sum(1 for m in self.meta if m["kind"] == SEG_ACTOR)
        if n_actors and (P is None or P.shape[0] != n_actors):
            raise ValueError(f"FlatScene.compose: {n_actors} actor segments need a [{n_actors}, 7] pose tensor")
        if P is not None and (P.dtype != torch.float32 or not P.is_contiguous()):
            P = P.float().contiguous()
        ids = None
        if idfts is not None:
            ids = [None if t is None else torch.as_tensor(t, dtype=torch.float32, device=self.xyz.device).contiguous() for t in idfts]
        args = [self.tensors[k] for k in _FLAT] + [P]
        return _ComposeFlat.apply(self, (fm, ids), int(max_sh_coeffs), int(num_classes), *args)


def _pack_flat(fs: FlatScene, tensors, frame, grads=None):
    """_CSeg array (and, with `grads`, the _CSegGrads array) whose pointers address the segments' blocks of the flat tensors.
    frame = (flip masks, IDFT rows or None) of this call."""
    flip_masks, idfts = frame
    arr = (_CSeg * len(fs.meta))()
    garr = (_CSegGrads * len(fs.meta))() if grads is not None else None
    keep = []
    base = {k: (t.data_ptr() if t is not None and t.numel() else 0) for k, t in zip(_FLAT + ("poses",), tensors)}
    gbase = None
    if grads is not None:
        gbase = {k: (t.data_ptr() if t is not None and t.numel() else 0) for k, t in zip(_FLAT + ("poses",), grads)}
    width = {"xyz": 3, "rotation": 4, "scaling": 3, "opacity": 1}
    rest_w = tensors[5].shape[1] if tensors[5] is not None and tensors[5].dim() == 2 else 0
    a = 0
    for i, (m, o) in enumerate(zip(fs.meta, fs._offsets())):
        c = arr[i]
        c.count, c.kind, c.fourier_dim = m["count"], m["kind"], m["fourier_dim"]
        c.class_label, c.sem_mode, c.flip_axis = m["class_label"], _SEM[m["semantic_mode"]], m["flip_axis"]
        c.flip_quat = (C.c_float * 4)(*[float(v) for v in m["flip_quat"]])

        def addr(b, name):
            if not b[name] or not m["count"]:
                return None
            if name in width:
                return b[name] + 4 * o["row"] * width[name]
            if name == "features_rest":
                return b[name] + 4 * o["row"] * rest_w if rest_w else None
            if name == "features_dc":
                return b[name] + 4 * o["dc"]
            return b[name] + 4 * o["sem"] if m["sem_width"] else None  # semantic
        for name in _FLAT:
            setattr(c, name, addr(base, name))
            if garr is not None:
                setattr(garr[i], name, addr(gbase, name))
        if m["kind"] == SEG_ACTOR:
            c.pose = base["poses"] + 4 * 7 * a
            if garr is not None and gbase["poses"]:
                garr[i].pose = gbase["poses"] + 4 * 7 * a
            a += 1
        fmk = flip_masks[i]
        if fmk is not None:
            fmk = fmk.view(torch.uint8) if fmk.dtype == torch.bool and fmk.is_contiguous() else fmk.to(torch.uint8).contiguous()
            keep.append(fmk)
            c.flip_mask = fmk.data_ptr() if fmk.numel() else None
        idft = m["idft"] if idfts is None else idfts[i]
        if idft is not None:
            keep.append(idft)
            c.idft = idft.data_ptr()
    return arr, garr, keep


class _ComposeFlat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fs, flip_masks, M, S, *tensors):
        dev = tensors[0].device
        for t in tensors:
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
                raise SgrError("flat scene parameters must be contiguous float32 HIP tensors")
        arr, _, keep = _pack_flat(fs, tensors, flip_masks)
        N = sum(m["count"] for m in fs.meta)
        f = dict(dtype=torch.float32, device=dev)
        outs = [torch.empty(N, 3, **f), torch.empty(N, 4, **f), torch.empty(N, 3, **f), torch.empty(N, 1, **f),
                torch.empty(N, M, 3, **f), torch.empty(N, S, **f)]
        grow = _Grow(dev)
        with torch.cuda.device(dev):
            check(_native.lib().sgr_scene_compose_forward(
                len(fs.meta), arr, int(M), int(S), *[_ptr(o) if o.numel() else None for o in outs], grow.cb, None,
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        ctx.fs, ctx.flip_masks, ctx.M, ctx.S = fs, flip_masks, int(M), int(S)
        ctx.save_for_backward(*[t for t in tensors if t is not None])
        ctx.present = [t is not None for t in tensors]
        ctx.keep = keep
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_means, d_rot, d_scale, d_opac, d_shs, d_sem):
        saved = iter(ctx.saved_tensors)
        tensors = [next(saved) if p else None for p in ctx.present]
        fs, M, S = ctx.fs, ctx.M, ctx.S
        dev = tensors[0].device
        need = ctx.needs_input_grad[4:]
        grads = [torch.empty_like(t) if (t is not None and need[i]) else None for i, t in enumerate(tensors)]
        arr, garr, keep = _pack_flat(fs, tensors, ctx.flip_masks, grads)
        dz = lambda t: None if t is None else _f32c(t, "grad")
        ins = [dz(d_means), dz(d_rot), dz(d_scale), dz(d_opac), dz(d_shs), dz(d_sem) if S else None]
        grow = _Grow(dev)
        with torch.cuda.device(dev):
            check(_native.lib().sgr_scene_compose_backward(
                len(fs.meta), arr, garr, M, S, *[_ptr(t) if t is not None and t.numel() else None for t in ins], grow.cb,
                None, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        del keep
        return (None, None, None, None) + tuple(grads)


def densification_stats(models: Sequence[dict], dL_dmeans2D: torch.Tensor, radii: torch.Tensor) -> None:
    """In-place update of every model's ``xyz_gradient_accum`` [n,2], ``denom`` [n,1] and ``max_radii2D`` [n] from one
    view's screen-space gradient [N,3] and radii [N] (models in concatenation order; dict keys as the attribute
    names of the reference's GaussianModel)."""
    dev = dL_dmeans2D.device
    if not dL_dmeans2D.is_cuda:
        raise SgrError("dL_dmeans2D must be a HIP (cuda) tensor: there is no CPU path")
    arr = (_CStatSeg * len(models))()
    for c, m in zip(arr, models):
        for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
            t = m[k]
            if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise SgrError(f"{k} must be a contiguous float32 HIP tensor (updated in place)")
        c.count = m["denom"].shape[0]
        c.xyz_gradient_accum, c.denom, c.max_radii2D = (m["xyz_gradient_accum"].data_ptr(), m["denom"].data_ptr(),
                                                        m["max_radii2D"].data_ptr())
    g = dL_dmeans2D.detach().to(torch.float32).contiguous()
    r = radii.to(torch.int32).contiguous()
    grow = _Grow(dev)
    with torch.cuda.device(dev):
        check(_native.lib().sgr_scene_densification_stats(len(models), arr, _ptr(g), _ptr(r), grow.cb, None,
                                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


class FlatStats:
    """The sub-models' densification statistics (``xyz_gradient_accum`` [n,2], ``denom`` [n,1], ``max_radii2D`` [n] of
    the reference's GaussianModel) kept as VIEWS of three flat tensors in concatenation order -- the layout the
    rasterizer's backward updates in place when ``GaussianRasterizer.stats_sink = stats.sink()`` is set
    (include/sgr.h: sgr_backward_ex), so that ``set_max_radii2D`` + ``add_densification_stats``
    (street_gaussian_model.py:551-571) cost nothing extra.  A frame that renders only some of the models passes ``sink(models=[...])``.  After a densification step the
    point counts change:
    build a new FlatStats (the reference re-creates the three tensors as zeros there as well, gaussian_model.py:545-547).
    """

    def __init__(self, counts: Sequence[int], device):
        self.counts = [int(c) for c in counts]
        n = sum(self.counts)
        self.xyz_gradient_accum = torch.zeros(n, 2, dtype=torch.float32, device=device)
        self.denom = torch.zeros(n, 1, dtype=torch.float32, device=device)
        self.max_radii2D = torch.zeros(n, dtype=torch.float32, device=device)

    def sink(self, models: Optional[Sequence[int]] = None):
        """The value for ``GaussianRasterizer.stats_sink``.  ``models`` = the indices (into ``counts``) of the sub-models
        this frame renders, in rasterization order -- the reference rebuilds that list per frame
        (street_gaussian_model.py:230-250: background + the actors visible at the timestamp), so the rasterized set is
        in general a subset / re-ordering of the persistent models.  None = every model, in order (a static graph)."""
        if models is None:
            return self.xyz_gradient_accum, self.denom, self.max_radii2D
        models = [int(m) for m in models]
        if len(set(models)) != len(models):  # two segments on the same persistent rows: a racy read-modify-write
            raise ValueError("FlatStats.sink: a sub-model is rendered at most once per frame")
        starts = [0]
        for c in self.counts:
            starts.append(starts[-1] + c)
        segs, src = [], 0
        for m in models:
            c = self.counts[m]
            if c:
                segs.append((src, c, starts[m]))
            src += c
        return self.xyz_gradient_accum, self.denom, self.max_radii2D, segs

    def views(self) -> List[dict]:
        out, start = [], 0
        for c in self.counts:
            out.append({"xyz_gradient_accum": self.xyz_gradient_accum[start:start + c], "denom": self.denom[start:start + c],
                        "max_radii2D": self.max_radii2D[start:start + c]})
            start += c
        return out
