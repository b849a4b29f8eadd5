"""Multi-GPU use of the rasterizer: camera views shard across ranks, Gaussians are replicated.

The reference is single-process / single-GPU; this module is what BASELINE.json's north_star
adds: each rank runs an independent ``rasterizer(...)`` + ``.backward()`` for its own view (no data-path
collective), then ONE exchange step sums the per-Gaussian parameter gradients over RCCL/xGMI
(`torch.distributed` backend "nccl" is RCCL on ROCm).  Gradients are packed into a single contiguous fp32
bucket so a step issues one large collective (236 B per Gaussian at SH degree 3) instead of
one per tensor; on the 8-GPU xGMI mesh a reduce-scatter + all-gather pair keeps all seven links busy, so that
is what ``mode="rs_ag"`` issues explicitly (the default lets RCCL choose).

``FactoredGradReducer`` cuts the exchange to a quarter: 48 of those 59 floats are dL/dSH, and the SH gradient of one
view is rank-1, ``dL/dSH[k][c] = Y_k(dir) * dRGB[c]`` (cuda_rasterizer/backward.cu:46-105).  Ranks all-gather the
3 floats of dRGB per Gaussian (+ their camera centre) and rebuild ``sum_v Y(dir_v) (x) dRGB_v`` locally with one HIP
kernel (csrc/sgr_multiview.hip): 44 B all-reduced + 12 B all-gathered per Gaussian instead of 236 B all-reduced.

Overlap.  ``begin()`` issues the exchange on a side stream behind an event of the compute stream and returns at once;
``wait()`` makes the compute stream wait for it and installs the reduced gradients.  ``all_reduce()`` = both, back to
back (the blocking form).  What the exchange can overlap with depends on the schedule: with several views per rank
(``views_per_rank > 1``) the all-gather of view j's dRGB runs under view j+1's forward + backward; with one view per
rank it can only run under the NEXT step's forward (0.76 ms of the 1.95 ms step at 1 M Gaussians), which in training
means the optimiser consumes gradients one step late (``bench.py --exchange overlap`` measures exactly that schedule
and says so; ``--exchange blocking`` keeps everything inside the step).

Also combines the densification statistics the training loop derives from the rasterizer outputs
(/root/reference/lib/models/street_gaussian_model.py:551-571): sums for the view-space gradient accumulators,
max for the screen radii.  Works with the gloo backend on CPU tensors (tests/test_multiview_gloo.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


class GradReducer:
    """All-reduce (sum) of the ``.grad`` of a fixed list of parameters through one flat bucket."""

    def __init__(self, params: Iterable[torch.Tensor], group: Optional[dist.ProcessGroup] = None, mode: str = "all_reduce",
                 average: bool = False, force: bool = False):
        self.params: List[torch.Tensor] = list(params)
        self.group = group
        self.mode = mode
        self.average = average
        self.force = force  # run the collective even with one rank (testing)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        ref = self.params[0]
        self._side = torch.cuda.Stream(ref.device) if ref.is_cuda else None  # the exchange's own stream
        self._works = []
        self._inflight = False
        self._store = None
        self._layout()
        # the rasterizer backward writes the gradients of parameters it is handed DIRECTLY into the bucket (no packing copy)
        self.direct = ref.is_cuda
        if self.direct and type(self)._register_sink:
            from . import rasterizer as _rast
            import weakref
            self._rast_mod = _rast
            _rast.BACKWARD_SINKS.append(weakref.WeakMethod(self._sink))

    _register_sink = True
    _GRAD_NAME = {"means3D": "means3D", "scales": "scales", "rotations": "rotations", "sh": "sh", "semantics": "semantics",
                  "colors": "colors", "cov3D": "cov3D"}

    def _sink(self, inputs, opacities_key, num_points):
        """rasterizer.BACKWARD_SINKS provider: for every input of this rasterizer call that IS one of the bucket's parameters
        (same storage address and shape, no gradient accumulated yet, no exchange in flight) the matching slice of the
        bucket becomes the backward's output tensor -- autograd then installs that very tensor as ``p.grad`` and ``begin()``
        has nothing to copy."""
        if not self.direct or self._inflight or (self.world == 1 and not self.force):
            return None
        by_ptr = {}
        off = 0
        given = self.__dict__.setdefault("_direct_given", set())
        for i, (p, n) in enumerate(zip(self.params, self._numel)):
            # (a slice is handed out ONCE per exchange: two rasterizer calls over the same leaves in one autograd pass both
            # run before either gradient has been installed -- the second one must take the ordinary route and be accumulated)
            if p.grad is None and p.is_leaf and p.requires_grad and i not in given:
                by_ptr[p.data_ptr()] = (off, n, tuple(p.shape), i)
            off += n
        out = {}
        cand = [(self._GRAD_NAME[k], t.data_ptr(), tuple(t.shape)) for k, t in inputs.items()
                if isinstance(t, torch.Tensor) and t.numel() and t.is_cuda]
        if opacities_key is not None:
            cand.append(("opacity", opacities_key[0], opacities_key[1]))
        for name, ptr, shape in cand:
            hit = by_ptr.get(ptr)
            if hit is not None and hit[2] == shape:
                out[name] = self.flat[hit[0]:hit[0] + hit[1]].view(shape)  # a FRESH view: autograd may adopt it as p.grad
                given.add(hit[3])
                by_ptr.pop(ptr)
        return {"out": out} if out else None

    def close(self) -> None:
        m = getattr(self, "_rast_mod", None)
        if m is not None:
            m.remove_sink(self._sink)

    def _layout(self) -> None:
        """(Re)builds the flat bucket and its per-parameter views for ``self.params``.  The storage is kept when it is
        large enough (it is allocated with 1/8 head-room): a densify step changes every size, and on some hosts a fresh
        device allocation costs tens of ms."""
        self._numel = [p.numel() for p in self.params]
        total = sum(self._numel)
        pad = (-total) % max(self.world, 1)  # reduce_scatter needs equal shards
        ref = self.params[0]
        need = total + pad
        if self._store is None or self._store.numel() < need or self._store.device != ref.device:
            self._store = torch.zeros(need + need // 8, dtype=torch.float32, device=ref.device)
        self.flat = self._store[:need]
        self._views = []
        off = 0
        for p, n in zip(self.params, self._numel):
            self._views.append(self.flat[off:off + n].view(p.shape))
            off += n

    def rebuild(self, params: Iterable[torch.Tensor]) -> None:
        """The parameter set changed size (densify / prune, train.py:187-210): joins an exchange still in flight and lays the
        bucket out for the new tensors.  Every rank must call it with parameters of the same shapes (replicated densify)."""
        if self._inflight:
            self.wait()
        for p, v in zip(self.params, self._views):  # old gradients that alias the old bucket are meaningless now
            if p.grad is not None and p.grad.data_ptr() == v.data_ptr():
                p.grad = None
        self.params = list(params)
        self._layout()

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def warm_up(self, rounds: int = 3) -> None:
        """Run the collective on the (zeroed) bucket a few times: RCCL builds communicators, channels and its
        algorithm tables lazily on first use -- seconds that belong to set-up, not to a training step."""
        if self.world == 1 and not self.force:
            return
        self.flat.zero_()
        for _ in range(rounds):
            self._collective()
        if self.flat.is_cuda:
            torch.cuda.synchronize(self.flat.device)

    def _collective(self, async_op: bool = False):
        """Issues the collective(s) on the current stream; returns the work handles when async_op."""
        works = []
        if self.mode == "rs_ag":
            shard = self.flat.numel() // self.world
            rank = dist.get_rank(self.group)
            mine = self.flat[rank * shard:(rank + 1) * shard]
            w = dist.reduce_scatter_tensor(mine, self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                w.wait()  # orders the all-gather behind the reduce-scatter on this stream (no host block on RCCL)
            works.append(dist.all_gather_into_tensor(self.flat, mine.clone(), group=self.group, async_op=async_op))
        else:
            works.append(dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
        return works if async_op else []

    def begin(self) -> None:
        """Non-blocking start of the exchange: gradients are packed and the collective is issued on the side stream,
        ordered behind everything the compute stream has queued so far.  Pair with ``wait()``."""
        if self._inflight:
            raise RuntimeError("begin() called twice without wait()")
        if self.world == 1 and not self.force:
            return
        self.__dict__.setdefault("_direct_given", set()).clear()
        grads = [p.grad if p.grad is not None else torch.zeros_like(v) for p, v in zip(self.params, self._views)]
        # gradients the rasterizer backward wrote straight into the bucket (self._sink) are already in place
        todo = [(v, g) for v, g in zip(self._views, grads) if g.data_ptr() != v.data_ptr()]
        self.copied_last = len(todo)
        views_c, grads_c = [v for v, _ in todo], [g for _, g in todo]
        if self._side is not None:
            cur = torch.cuda.current_stream(self.flat.device)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                if todo:
                    torch._foreach_copy_(views_c, grads_c)
                for g in grads_c:
                    g.record_stream(self._side)  # the caching allocator must not hand these out before the copy ran
                self._works = self._collective(async_op=True)
                if self.average:
                    for w in self._works:
                        w.wait()
                    self.flat.div_(self.world)
        else:
            if todo:
                torch._foreach_copy_(views_c, grads_c)
            self._works = self._collective(async_op=True)
        # a gradient that still IS a view of the bucket (installed by the previous wait()) must not be accumulated into
        # while the collective runs on the bucket: detach it, the next backward allocates a fresh one
        for p, v in zip(self.params, self._views):
            if p.grad is not None and p.grad.data_ptr() == v.data_ptr():
                p.grad = None
        self._inflight = True

    def wait(self) -> None:
        """The compute stream waits for the exchange begun by ``begin()``; ``p.grad`` becomes a view of the reduced
        bucket.  No host synchronisation on RCCL (work.wait() orders streams)."""
        if not self._inflight:
            return
        if self._side is not None:
            with torch.cuda.stream(self._side):
                for w in self._works:
                    w.wait()
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
        else:
            for w in self._works:
                w.wait()
            if self.average:
                self.flat.div_(self.world)
        self._works = []
        self._inflight = False
        for p, v in zip(self.params, self._views):
            p.grad = v

    def all_reduce(self) -> None:
        """Sum gradients over all ranks (blocking form); ``p.grad`` is replaced by a view of the reduced bucket."""
        if self.world == 1 and not self.force:
            return
        self.begin()
        self.wait()


@dataclass
class SHSegment:
    """One sub-model of a scene graph for the factored exchange: its SH LEAF parameters as the reference stores them
    (features_dc [n, C, 3] with C = 1 for the background and C = fourier_dim for an actor, features_rest [n, M-1, 3];
    gaussian_model.py:120-123, gaussian_model_actor.py:45-60) and where its Gaussians are.

    means is not None   a static model: world positions [n, 3], replicated on every rank (view directions are
                        recomputed from them locally);
    means is None       a posed model (actor): its world positions differ per frame
                        (street_gaussian_model.py:287-330), so the positions of the view travel with its dRGB
                        (24 B instead of 12 B per Gaussian and view -- actors are a few % of a street scene)."""
    features_dc: torch.Tensor
    features_rest: Optional[torch.Tensor]
    means: Optional[torch.Tensor] = None

    @property
    def n(self) -> int:
        return int(self.features_dc.shape[0])

    @property
    def fourier_dim(self) -> int:
        return int(self.features_dc.shape[1])

    @property
    def posed(self) -> bool:
        return self.means is None


class FactoredGradReducer:
    """Exchange step for view-sharded training with SH colours: dense parameters through a flat all-reduce bucket, the
    SH gradient through an all-gather of per-view dRGB + a local rebuild (module docstring).

    dense_params  parameters whose ``.grad`` is all-reduced as is (means3D, scales, rotations, opacities, semantics ...)
    shs           the SH coefficient parameter [P, M, 3], or the reference's pair (features_dc [P, 1, 3],
                  features_rest [P, M-1, 3]) (gaussian_model.py:120-123); ``.grad`` of it / of both is REPLACED by the
                  rebuilt sum over all views.  Leaf tensors only: a non-leaf SH tensor (e.g. the output of
                  scene.compose) has already sent this view's dL/dSH down the graph.
    means3D       positions [P, 3] (view directions are recomputed from them on every rank)
    segments      instead of (shs, means3D): the scene graph as a list of ``SHSegment`` -- background + actors, whose
                  composed SH tensor is a ``cat`` over sub-models with per-frame actor poses and a per-frame Fourier mix
                  of the actors' DC term (street_gaussian_model.py:287-449).  The rasterizer then sees a NON-leaf SH
                  tensor; detach it before the rasterizer call (``shs.detach()``) so that no dL/dSH flows down the
                  graph, and declare every frame with ``set_frame`` -- the exchange stays at 12 (static) / 24 (posed)
                  bytes per Gaussian and view instead of the 192 B/Gaussian of a flat bucket.
    views_per_rank  rasterizer backward calls every rank makes per step (equal on all ranks)

    The per-view dL/dcolour, geometry buffer, camera centre and positions are picked up from the rasterizer's backward
    through ``rasterizer.BACKWARD_OBSERVERS``; call ``close()`` (or use it as a context manager) to detach.
    ``mask_fn`` / ``rebuild_fn`` default to the HIP kernels and exist so the collective logic can be tested on CPU
    tensors with the gloo backend; there is no CPU implementation in the product.

    The payload buffer is double-buffered and guarded by events: a rasterizer backward that runs between ``begin()`` and
    ``wait()`` (the overlap schedule, or k > 1 accumulation) writes the OTHER buffer, and a buffer is only reused once
    the all-gather that read it has completed on the side stream.
    """

    def __init__(self, dense_params: Iterable[torch.Tensor], shs=None, means3D: Optional[torch.Tensor] = None,
                 group: Optional[dist.ProcessGroup] = None, views_per_rank: int = 1, mode: str = "all_reduce",
                 force: bool = False, mask_fn=None, rebuild_fn=None, segments: Optional[Sequence[SHSegment]] = None):
        from . import rasterizer as _rast
        self.dense = GradReducer(dense_params, group=group, mode=mode, force=force)
        self.group, self.force = group, force
        self.k = int(views_per_rank)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._mask_fn, self._rebuild_fn = mask_fn, rebuild_fn
        self._pending = []
        self._buf_store = [None, None]
        self._all_store = None
        self._configure(shs, means3D, segments)
        self._rast = _rast
        self._direct_slot = None
        _rast.BACKWARD_OBSERVERS.append(self._observe)
        import weakref
        _rast.BACKWARD_SINKS.append(weakref.WeakMethod(self._sink))

    def rebuild(self, dense_params: Iterable[torch.Tensor], shs=None, means3D: Optional[torch.Tensor] = None,
                segments: Optional[Sequence[SHSegment]] = None) -> None:
        """After a densify / prune step (train.py:187-210: every parameter changes its length): joins an exchange that is
        still in flight, then lays the dense bucket and the per-view payload out for the new parameters -- same arguments
        as the constructor.  Storage is reused when it is large enough.  All ranks must rebuild with the same shapes: with
        replicated Gaussians the densify step itself has to be identical everywhere (``densify_replicated``)."""
        if getattr(self, "_begun", False):
            self.wait()
        if self._pending:
            raise RuntimeError("rebuild() between a rasterizer backward and its exchange: run the exchange first")
        self.dense.rebuild(dense_params)
        self._configure(shs, means3D, segments)

    def _configure(self, shs, means3D, segments) -> None:
        self._single = None
        if segments is None:
            if shs is None or means3D is None:
                raise ValueError("give (shs, means3D) or segments")
            parts = list(shs) if isinstance(shs, (tuple, list)) else [shs]
            if len(parts) > 2:
                raise ValueError("shs is one [P, M, 3] parameter or the reference's pair (features_dc, features_rest); "
                                 f"got {len(parts)} parts (extra parts would silently get no gradient)")
            if len(parts) == 1:  # one [P, M, 3] parameter: its gradient is the rebuilt tensor as a whole
                self._single = parts[0]
                segments = [SHSegment(parts[0][:, :1, :], parts[0][:, 1:, :] if parts[0].shape[1] > 1 else None, means3D)]
            else:
                segments = [SHSegment(parts[0], parts[1], means3D)]
        self.segments = list(segments)
        for sg in self.segments:
            for t in ([self._single] if self._single is not None else [sg.features_dc, sg.features_rest]):
                if t is not None and not t.is_leaf:
                    raise ValueError("FactoredGradReducer needs LEAF SH parameters (their .grad is replaced)")
            if not sg.posed and sg.fourier_dim != 1:
                raise ValueError("a static segment has one DC row (features_dc [n, 1, 3])")
        self.M = 1 + (int(self.segments[0].features_rest.shape[1]) if self.segments[0].features_rest is not None else 0)
        for i, sg in enumerate(self.segments):  # one rebuild kernel, one row width: every segment must have the same layout
            dc, rest = sg.features_dc, sg.features_rest
            if dc.dim() != 3 or dc.shape[2] != 3:
                raise ValueError(f"segment {i}: features_dc must be [n, C, 3], got {tuple(dc.shape)}")
            m_i = 1 + (int(rest.shape[1]) if rest is not None else 0)
            if m_i != self.M:
                raise ValueError(f"segment {i} has {m_i} SH coefficients, segment 0 has {self.M}: all segments must share M")
            if rest is not None and (rest.dim() != 3 or rest.shape[0] != dc.shape[0] or rest.shape[2] != 3):
                raise ValueError(f"segment {i}: features_rest must be [n, M-1, 3] with the same n as features_dc")
        first = self.segments[0].features_dc
        dev = first.device
        # payload row of one view: [campos 3 | dRGB of every segment, statics first then posed | positions of the posed
        # segments | Fourier mix (idft) of every posed segment]
        self._static = [i for i, sg in enumerate(self.segments) if not sg.posed]
        self._posed = [i for i, sg in enumerate(self.segments) if sg.posed]
        off = 3
        self._drgb_off = {}
        for i in self._static + self._posed:
            self._drgb_off[i] = off
            off += 3 * self.segments[i].n
        self._posed_drgb0 = self._drgb_off[self._posed[0]] if self._posed else off
        self._n_posed = sum(self.segments[i].n for i in self._posed)
        self._pos_off = {}
        for i in self._posed:
            self._pos_off[i] = off
            off += 3 * self.segments[i].n
        self._pos0 = self._pos_off[self._posed[0]] if self._posed else off
        self._idft_off = {}
        for i in self._posed:
            self._idft_off[i] = off
            off += self.segments[i].fourier_dim
        self._row = off

        def fit(store, rows):  # a [rows, row] view of a store that is kept across rebuilds (1/8 head-room)
            need = rows * self._row
            if store is None or store.numel() < need or store.device != dev:
                store = torch.zeros(need + need // 8, dtype=torch.float32, device=dev)
            else:
                store[:need].zero_()
            return store, store[:need].view(rows, self._row)
        self._bufs = []
        for b in range(2):
            self._buf_store[b], v = fit(self._buf_store[b], self.k)
            self._bufs.append(v)
        self._free_ev = [None, None]  # event on the side stream: "the all-gather that read buffer b is done"
        self._cur = 0
        self._mine = self._bufs[0]
        self._all_store, self._all = fit(self._all_store, self.world * self.k)
        self._frame = (list(range(len(self.segments))), {})

    # legacy attribute (bench.py, tests): the single static model's positions
    @property
    def means3D(self):
        return self.segments[0].means

    def set_frame(self, models: Sequence[int], idft: Optional[dict] = None) -> None:
        """Declares the NEXT rasterizer backward's frame: the indices (into ``segments``) of the sub-models it renders,
        in rasterization order (the reference's per-frame graph_obj_list, street_gaussian_model.py:230-250), and for
        posed segments with fourier_dim > 1 their Fourier mix of this frame, ``idft[i]`` = [fourier_dim]
        (gaussian_model_actor.py:71-80: features_dc is mixed over its fourier dimension by the frame's IDFT row)."""
        models = [int(m) for m in models]
        for m in models:
            if not 0 <= m < len(self.segments):
                raise ValueError(f"set_frame: segment index {m} out of range (have {len(self.segments)})")
        if len(set(models)) != len(models):
            raise ValueError("set_frame: a segment is rendered at most once per frame")
        self._frame = (models, dict(idft or {}))

    @property
    def frame_num_points(self) -> int:
        """Gaussians of the declared frame: what the next rasterizer backward over this reducer's set must report."""
        return sum(self.segments[m].n for m in self._frame[0])

    # -- rasterizer backward hooks --
    def _sink(self, inputs, opacities_key, num_points):
        """rasterizer.BACKWARD_SINKS provider (called BEFORE the native backward of a frame of this reducer's set): the
        backward does not write this view's dL/dSH at all (``wait()`` rebuilds the sum over all views and replaces the
        gradient), and -- when the frame is the whole single static model, the layout bench.py and a plain GaussianModel
        have -- its row-sum stage writes the clamp-masked colour gradient straight into this view's payload slot
        (sgr_backward_extras.masked_color_out) instead of a mask launch + a 12 B/Gaussian copy afterwards."""
        self._direct_slot = None
        models, _ = self._frame
        if int(num_points) != sum(self.segments[m].n for m in models) or len(self._pending) >= self.k:
            return None
        if self._mask_fn is not None or not self._bufs[0].is_cuda:  # (CPU test backend: everything through _pack)
            return None
        sg0 = self.segments[0]
        if len(self.segments) == 1 and sg0.means is not None:
            # (shs, means3D) form: the frame must be a render of THESE positions (the hook list is process-wide, and a
            # reducer somebody forgot to close() must not take the SH gradient away from an unrelated render of equal size)
            m = inputs.get("means3D")
            if m is None or m.data_ptr() != sg0.means.data_ptr():
                return None
        res = {"skip_sh_grad": True}
        if len(self.segments) == 1 and models == [0] and not self._posed:
            buf = self._bufs[self._cur]
            ev = self._free_ev[self._cur]
            if ev is not None and len(self._pending) == 0:
                torch.cuda.current_stream(buf.device).wait_event(ev)  # the exchange that read this buffer has finished
                self._free_ev[self._cur] = None
            slot = buf[len(self._pending)]
            o, n = self._drgb_off[0], self.segments[0].n
            self._direct_slot = slot[o:o + 3 * n]
            res["masked_color_out"] = self._direct_slot
        return res

    def _observe(self, grad_colors, geomBuffer, campos, sh_degree, num_points, means3D=None, color_ready=None):
        # the hook list is process-wide: passes over another Gaussian set (the reference's training step also renders
        # single objects, street_gaussian_renderer.render_object) are not this reducer's business
        models, idft = self._frame
        if int(num_points) != sum(self.segments[m].n for m in models):
            return
        if len(self._pending) >= self.k:
            raise RuntimeError(f"more than views_per_rank={self.k} rasterizer backward passes since the last exchange")
        # dL/dcolour is final as soon as the backward's row-sum stage has run (`color_ready`, recorded between that stage
        # and the per-Gaussian stage): the payload is packed on the exchange's side stream behind that event, so that the
        # all-gather of begin() can start while the compute stream still runs K12 + K13 and whatever the caller queues next
        side = self.dense._side
        early = color_ready is not None and side is not None and (self.world > 1 or self.force)
        if early:
            side.wait_event(color_ready)
            ctx = torch.cuda.stream(side)
            for t in (grad_colors, geomBuffer, campos, means3D):
                if t is not None and t.is_cuda:
                    t.record_stream(side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        self._packed_on_side = early
        with ctx:
            self._pack(grad_colors, geomBuffer, campos, sh_degree, num_points, means3D, models, idft)

    def _pack(self, grad_colors, geomBuffer, campos, sh_degree, num_points, means3D, models, idft):
        direct = self._direct_slot is not None  # the backward's row sum already wrote the masked dRGB into the slot (_sink)
        self._direct_slot = None
        if direct:
            self._bufs[self._cur][len(self._pending)][:3].copy_(campos.reshape(3))
            self._pending.append(int(sh_degree))
            return
        if self._mask_fn is not None:
            drgb = self._mask_fn(geomBuffer, grad_colors, num_points)
        else:
            from . import _C
            drgb = _C.masked_color_grad(geomBuffer, grad_colors, num_points)
        buf = self._bufs[self._cur]
        ev = self._free_ev[self._cur]
        if ev is not None and len(self._pending) == 0:
            torch.cuda.current_stream(buf.device).wait_event(ev)  # the exchange that read this buffer has finished
            self._free_ev[self._cur] = None
        slot = buf[len(self._pending)]
        whole = len(models) == len(self.segments) and models == self._static + self._posed and not self._posed
        if not whole:
            slot.zero_()  # segments this frame does not render contribute zeros
        slot[:3].copy_(campos.reshape(3))
        src = 0
        for m in models:
            sg = self.segments[m]
            n = sg.n
            o = self._drgb_off[m]
            slot[o:o + 3 * n].copy_(drgb[src:src + n].reshape(-1))
            if sg.posed:
                if means3D is None:
                    raise RuntimeError("posed segments need the rasterizer's means3D (world positions of this frame)")
                o = self._pos_off[m]
                slot[o:o + 3 * n].copy_(means3D.detach()[src:src + n].reshape(-1))
                o, C = self._idft_off[m], sg.fourier_dim
                if C == 1 and m not in idft:
                    slot[o:o + 1].fill_(1.0)
                else:
                    if m not in idft:
                        raise RuntimeError(f"set_frame: segment {m} has fourier_dim {C} and needs its idft row")
                    slot[o:o + C].copy_(torch.as_tensor(idft[m], dtype=torch.float32).reshape(C))
            src += n
        self._pending.append(int(sh_degree))

    @property
    def nbytes(self) -> int:
        """Bytes this rank contributes to the collectives per step."""
        return self.dense.nbytes + self._mine.numel() * 4

    def warm_up(self, rounds: int = 3) -> None:
        """Set-up time collectives (see GradReducer.warm_up)."""
        if self.world == 1 and not self.force:
            return
        self.dense.warm_up(rounds)
        if dist.is_initialized():
            for _ in range(rounds):
                dist.all_gather_into_tensor(self._all, self._mine, group=self.group)
        if self._all.is_cuda:
            torch.cuda.synchronize(self._all.device)

    def begin(self) -> None:
        """Non-blocking start: the dense bucket's all-reduce and the all-gather of the per-view dRGB go to the side stream."""
        if getattr(self, "_begun", False):
            raise RuntimeError("begin() called twice without wait()")
        if len(self._pending) != self.k:
            raise RuntimeError(f"expected {self.k} rasterizer backward passes before the exchange, saw {len(self._pending)}")
        self._degree = self._pending[0]
        self._pending = []
        self._gather_works = []
        self._begun = True
        mine = self._mine = self._bufs[self._cur]
        sent = self._cur
        self._cur ^= 1  # rasterizer backward passes from now on fill the other buffer
        if self.world == 1 and not self.force:
            self._all.copy_(mine)  # one rank: its own views are all the views
            return
        side = self.dense._side
        early = side is not None and getattr(self, "_packed_on_side", False)
        if not early:
            self.dense.begin()
        if self.world > 1 or dist.is_initialized():
            if side is not None:
                if not early:  # the payload was packed on the compute stream: order the gather behind it
                    side.wait_stream(torch.cuda.current_stream(mine.device))
                with torch.cuda.stream(side):
                    self._gather_works = [dist.all_gather_into_tensor(self._all, mine, group=self.group, async_op=True)]
                    for w in self._gather_works:
                        w.wait()  # stream order on RCCL, no host block
                    ev = torch.cuda.Event()
                    ev.record(side)
                    self._free_ev[sent] = ev
            else:
                self._gather_works = [dist.all_gather_into_tensor(self._all, mine, group=self.group, async_op=True)]
        else:
            if early:  # the payload was packed on the side stream: the compute stream's copy must come after it
                torch.cuda.current_stream(mine.device).wait_stream(side)
            self._all.copy_(mine)
        if early:
            # the all-gather is already queued behind the packed payload; the dense bucket follows it on the side stream
            # once the compute stream has produced every gradient
            self.dense.begin()

    def _rebuild(self, n, means, means_stride, col0, V):
        """sum_v Y(dir_v) (x) dRGB_v for the n Gaussians whose dRGB sits at column col0 of the gathered rows -> [n, M, 3]."""
        A, row = self._all, self._row
        if self._rebuild_fn is not None:  # test path (CPU tensors): contiguous copies, same arithmetic
            drgb = A[:, col0:col0 + 3 * n].reshape(V, n, 3)
            m = means.detach() if means_stride == 0 else A[:, means:means + 3 * n].reshape(V, n, 3)
            return self._rebuild_fn(m, A[:, :3].contiguous(), drgb, self._degree, self.M).view(n, self.M, 3)
        from . import _C
        base = A.data_ptr()
        if means_stride == 0:
            mt = means.detach()
            mt = mt if (mt.dtype == torch.float32 and mt.is_contiguous()) else mt.float().contiguous()
            mp = mt.data_ptr()
        else:
            mp = base + 4 * means
        return _C.sh_grad_from_rows(n, self._degree, self.M, V, mp, means_stride, base, row, base + 4 * col0, row, A.device)

    def wait(self) -> None:
        """Joins the exchange and rebuilds dL/dSH = sum over all views of Y(dir_v) (x) dRGB_v on the compute stream."""
        if not getattr(self, "_begun", False):
            return
        self._begun = False
        if self.world > 1 or self.force:
            side = self.dense._side
            if side is None:
                for w in self._gather_works:
                    w.wait()
            self._gather_works = []
            self.dense.wait()  # also makes the compute stream wait for the side stream
        V = self._all.shape[0]
        M = self.M
        # static models: one launch each (their positions are local leaves); every posed model in ONE launch over the
        # concatenated posed block (positions per view from the payload)
        grads = {}
        for i in self._static:
            sg = self.segments[i]
            grads[i] = self._rebuild(sg.n, sg.means, 0, self._drgb_off[i], V)
        if self._posed:
            g_all = self._rebuild(self._n_posed, self._pos0, self._row, self._posed_drgb0, V)
            o = 0
            for i in self._posed:
                grads[i] = g_all[o:o + self.segments[i].n]
                o += self.segments[i].n
        for i, sg in enumerate(self.segments):
            g = grads[i]
            if self._single is not None:
                self._single.grad = g.view_as(self._single)
                continue
            if sg.features_rest is not None:
                sg.features_rest.grad = g[:, 1:, :]
            if not sg.posed:
                sg.features_dc.grad = g[:, :1, :]
                continue
            # posed segment: features_dc is mixed over its fourier dimension by the frame's IDFT row, so
            # dL/dfeatures_dc[:, f, :] = sum_v idft_v[f] * Y_0 * dRGB_v (fourier_dim 1 with idft = 1 is the plain DC row)
            C = sg.fourier_dim
            o, d0 = self._idft_off[i], self._drgb_off[i]
            mix = self._all[:, o:o + C]                                   # [V, C]
            drgb = self._all[:, d0:d0 + 3 * sg.n].reshape(V, sg.n, 3)      # [V, n, 3]
            sg.features_dc.grad = 0.28209479177387814 * torch.einsum("vf,vnc->nfc", mix, drgb)

    def all_reduce(self) -> None:
        self.begin()
        self.wait()

    def close(self) -> None:
        if self._observe in self._rast.BACKWARD_OBSERVERS:
            self._rast.BACKWARD_OBSERVERS.remove(self._observe)
        self._rast.remove_sink(self._sink)
        self.dense.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def reduce_densification_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor,
                               group: Optional[dist.ProcessGroup] = None) -> None:
    """In-place combination of the per-view densification statistics across ranks: sums for the gradient-norm
    accumulators and their denominators, max for the radii."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    packed = torch.cat([xyz_gradient_accum.reshape(-1).float(), denom.reshape(-1).float()])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    n = xyz_gradient_accum.numel()
    xyz_gradient_accum.copy_(packed[:n].view_as(xyz_gradient_accum))
    denom.copy_(packed[n:].view_as(denom))
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


class ReplicatedNormals:
    """Standard normals that are IDENTICAL on every rank: the source of the split's samples when Gaussians are
    replicated (the reference draws ``torch.normal(mean=0, std=stds)`` inside densify_and_split,
    /root/reference/lib/models/gaussian_model.py:469-473 -- per process, so replicas would drift apart at the first split).

    mode "broadcast" (default): rank 0 draws, everybody receives (exact by construction, 12 bytes per new point);
    mode "seeded": every rank draws from a generator seeded with (seed, call index) -- no traffic, identical as long as the
    ranks run the same generator implementation on the same device type.
    Call it with (rows, device); it returns a [rows, 3] float32 tensor.  Pass it as ``densify_and_prune(normal_source=...)``."""

    def __init__(self, seed: int = 0, group: Optional[dist.ProcessGroup] = None, mode: str = "broadcast"):
        if mode not in ("broadcast", "seeded"):
            raise ValueError("mode must be 'broadcast' or 'seeded'")
        self.seed, self.group, self.mode, self.calls = int(seed), group, mode, 0

    def __call__(self, rows: int, device, cols: int = 3) -> torch.Tensor:
        self.calls += 1
        multi = dist.is_initialized() and dist.get_world_size(self.group) > 1
        if self.mode == "seeded" or not multi:
            g = torch.Generator(device=device)
            g.manual_seed((self.seed * 1_000_003 + self.calls) & 0x7FFFFFFFFFFFFFFF)
            return torch.randn(int(rows), cols, generator=g, device=device, dtype=torch.float32)
        if dist.get_rank(self.group) == 0:
            g = torch.Generator(device=device)
            g.manual_seed((self.seed * 1_000_003 + self.calls) & 0x7FFFFFFFFFFFFFFF)
            t = torch.randn(int(rows), cols, generator=g, device=device, dtype=torch.float32)
        else:
            t = torch.empty(int(rows), cols, device=device, dtype=torch.float32)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return t


def replicas_identical(tensors: Sequence[torch.Tensor], group: Optional[dist.ProcessGroup] = None) -> bool:
    """True when every tensor of the list is BIT-identical on all ranks (shapes included): one int64 checksum per tensor
    (the sum of its 32-bit words + its length), reduced with MIN and MAX.  Collective: every rank must call it."""
    sums = []
    for t in tensors:
        w = t.detach().contiguous().view(-1)
        w = w.view(torch.int32) if w.element_size() == 4 else w.to(torch.float32).view(torch.int32)
        # position-weighted, so that two rows swapped do not cancel
        idx = torch.arange(1, w.numel() + 1, device=w.device, dtype=torch.int64)
        sums.append((w.to(torch.int64) * (idx % 65521 + 1)).sum() + w.numel())
    if not sums:
        return True
    v = torch.stack(sums)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return True
    lo, hi = v.clone(), v.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


def densify_replicated(params, xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor, *,
                       normals: ReplicatedNormals, states=None, group: Optional[dist.ProcessGroup] = None,
                       densify_fn=None, **kw):
    """One densify / prune step of view-sharded training (SURVEY 8e "what does NOT shard"; the reference's
    train.py:187-210 on one process): the per-view densification statistics are combined across ranks (sums / max,
    street_gaussian_model.py:551-571 as if one process had rendered every view), then EVERY rank runs the same
    ``densify_and_prune`` on the same inputs with the same normals, so that the replicas stay bit-identical without
    moving a single parameter.  Returns what ``densify.densify_and_prune`` returns; afterwards re-create the leaf
    parameters and call ``rebuild`` on the reducers.  ``densify_fn`` replaces densify.densify_and_prune (CPU tests)."""
    reduce_densification_stats(xyz_gradient_accum, denom, max_radii2D, group=group)
    if densify_fn is None:
        from . import densify as _d
        densify_fn = _d.densify_and_prune
    return densify_fn(params, xyz_gradient_accum, denom, states=states, normal_source=normals, **kw)


def view_for_rank(views: list, step: int, rank: Optional[int] = None, world: Optional[int] = None):
    """Round-robin view assignment: at step s rank r renders view (s*world + r) mod len(views)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return views[(step * world + rank) % len(views)]
