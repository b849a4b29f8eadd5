"""world_size-2 and -4 gloo tests (CPU) of the view-sharding exchange step (street_gaussians_amd/multiview.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from street_gaussians_amd import multiview
    g = torch.Generator().manual_seed(100)
    shapes = [(1000, 3), (1000, 16, 3), (1000, 1), (1000, 4), (7,)]
    params = [torch.zeros(s, requires_grad=True) for s in shapes]
    all_grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(world)]
    for p, gr in zip(params, all_grads[rank]):
        p.grad = gr.clone()
    params[4].grad = None if rank == 1 else params[4].grad  # a rank without a gradient contributes zeros
    red = multiview.GradReducer(params, mode=mode)
    red.all_reduce()
    ok = True
    for i, p in enumerate(params):
        exp = sum(all_grads[r][i] for r in range(world))
        if i == 4:
            exp = sum(all_grads[r][4] for r in range(world) if r != 1)
        ok &= torch.allclose(p.grad, exp, atol=1e-5)
    # begin() / wait(): the exchange carries the gradients as they were at begin(); what the next step does to .grad
    # in between does not leak into it, and wait() installs the reduced values
    for p, gr in zip(params, all_grads[rank]):
        p.grad = gr.clone()
    red.begin()
    try:
        red.begin()
        ok = False  # a second begin() without wait() is a usage error
    except RuntimeError:
        pass
    for p in params:
        p.grad = None if p.grad is None else torch.full_like(p.grad, 123.0)
    red.wait()
    for i, p in enumerate(params):
        ok &= torch.allclose(p.grad, sum(all_grads[r][i] for r in range(world)), atol=1e-5)
    red.wait()  # idempotent
    acc = torch.full((50, 2), float(rank + 1))
    den = torch.full((50, 1), 1.0)
    rad = torch.arange(50, dtype=torch.float32) * (1 if rank == 0 else -1)
    multiview.reduce_densification_stats(acc, den, rad)
    ok &= bool((acc == world * (world + 1) / 2).all() and (den == float(world)).all() and
               (rad == torch.arange(50).clamp(min=0)).all())
    ok &= multiview.view_for_rank(list(range(5)), step=1) == (world + rank) % 5
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])  # 4: the flat bucket's length is not a multiple of the world size (rs_ag pads)
@pytest.mark.parametrize("mode", ["all_reduce", "rs_ag"])
def test_grad_reducer_world2(mode, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def _rebuild_ref(means3D, campos, drgb, degree, M):
    """float64 autograd reference of sgr_sh_grad_from_views: d/dSH of sum_v <SH->RGB(dir_v), dRGB_v>."""
    import torch_ref
    P = means3D.shape[0]
    shs = torch.zeros(P, M, 3, dtype=torch.float64, requires_grad=True)
    total = 0.0
    for v in range(campos.shape[0]):
        d = means3D.double() - campos[v].double()
        d = d / d.norm(dim=1, keepdim=True)
        total = total + (torch_ref.sh_to_rgb(degree, shs, d) * drgb[v].double()).sum()
    total.backward()
    return shs.grad.float()


def _factored_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from street_gaussians_amd import multiview, rasterizer
    P, M, deg, k = 257, 16, 3, 2
    g = torch.Generator().manual_seed(7)
    means3D = torch.randn(P, 3, generator=g) * 3 + torch.tensor([0.0, 0.0, 10.0])
    # the reference keeps the SH coefficients as two parameters (gaussian_model.py:120-123)
    f_dc, f_rest = torch.zeros(P, 1, 3, requires_grad=True), torch.zeros(P, M - 1, 3, requires_grad=True)
    dense = [torch.zeros(P, 3, requires_grad=True), torch.zeros(P, 1, requires_grad=True)]
    campos = torch.randn(world * k, 3, generator=g)
    colors = torch.randn(world * k, P, 3, generator=g)
    clamp = torch.rand(world * k, P, 3, generator=g) < 0.2
    dense_g = [[torch.randn(p.shape, generator=g) for p in dense] for _ in range(world)]
    with multiview.FactoredGradReducer(dense, (f_dc, f_rest), means3D, views_per_rank=k,
                                       mask_fn=lambda geom, gc, n: gc * (~geom).float(),
                                       rebuild_fn=_rebuild_ref) as red:
        ok = len(rasterizer.BACKWARD_OBSERVERS) == 1
        for j in range(k):  # what the rasterizer's backward would report for this rank's views
            v = rank * k + j
            for obs in list(rasterizer.BACKWARD_OBSERVERS):
                obs(grad_colors=colors[v], geomBuffer=clamp[v], campos=campos[v], sh_degree=deg, num_points=P)
                # a pass over another Gaussian set (render_object) is ignored, not an error
                obs(grad_colors=colors[v][:5], geomBuffer=clamp[v][:5], campos=campos[v], sh_degree=deg, num_points=5)
        for p, gr in zip(dense, dense_g[rank]):
            p.grad = gr.clone()
        f_dc.grad = torch.full_like(f_dc, float("nan"))  # must be replaced, not accumulated into
        red.begin()
        red.wait()
        exp = _rebuild_ref(means3D, campos, colors * (~clamp).float(), deg, M)
        ok &= torch.allclose(f_dc.grad, exp[:, :1], atol=1e-6) and torch.allclose(f_rest.grad, exp[:, 1:], atol=1e-6)
        for i, p in enumerate(dense):
            ok &= torch.allclose(p.grad, sum(dense_g[r][i] for r in range(world)), atol=1e-6)
        try:  # a second exchange without new backward passes is a usage error
            red.all_reduce()
            ok = False
        except RuntimeError:
            pass
    ok &= len(rasterizer.BACKWARD_OBSERVERS) == 0
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_factored_grad_reducer_world2(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_factored_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def _segments_worker(rank, world, port, q):
    """A scene graph of one static model + two posed models (fourier_dim 2); the ranks render DIFFERENT frames: other
    actor poses, other Fourier mixes, and rank 1's frame does not contain actor 2 at all."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch_ref
    from street_gaussians_amd import multiview, rasterizer
    M, deg, C = 16, 3, 2
    n = [301, 57, 40]
    g = torch.Generator().manual_seed(21)
    bk_means = torch.randn(n[0], 3, generator=g) * 3 + torch.tensor([0.0, 0.0, 10.0])
    local = [None, torch.randn(n[1], 3, generator=g), torch.randn(n[2], 3, generator=g)]
    leaf = lambda *s: torch.zeros(*s, requires_grad=True)
    segs = [multiview.SHSegment(leaf(n[0], 1, 3), leaf(n[0], M - 1, 3), bk_means),
            multiview.SHSegment(leaf(n[1], C, 3), leaf(n[1], M - 1, 3), None),
            multiview.SHSegment(leaf(n[2], C, 3), leaf(n[2], M - 1, 3), None)]
    dense = [torch.zeros(sum(n), 3, requires_grad=True)]
    # per rank (= per view): frame layout, actor translations, idft rows, camera, upstream colour gradients, clamp masks
    frames = [[0, 1, 2], [0, 1]]
    trans = torch.randn(world, 3, 3, generator=g) * 2 + torch.tensor([0.0, 0.0, 8.0])
    idft = torch.randn(world, 3, C, generator=g)
    campos = torch.randn(world, 3, generator=g)
    ok = True
    exp_rest = [torch.zeros(k, M - 1, 3, dtype=torch.float64) for k in n]
    exp_dc = [torch.zeros(n[0], 1, 3, dtype=torch.float64), torch.zeros(n[1], C, 3, dtype=torch.float64),
              torch.zeros(n[2], C, 3, dtype=torch.float64)]
    views = []
    for v in range(world):
        Pf = sum(n[m] for m in frames[v])
        colors = torch.randn(Pf, 3, generator=g)
        clamp = torch.rand(Pf, 3, generator=g) < 0.2
        pos = torch.cat([bk_means if m == 0 else local[m] + trans[v, m] for m in frames[v]])
        views.append((colors, clamp, pos))
        # float64 reference: d/d(leaves) of sum <SH->RGB(dir), dRGB> with the composed SH = cat over the frame's models,
        # actors' DC mixed by the frame's idft row (gaussian_model_actor.py:71-80)
        src = 0
        for m in frames[v]:
            dc = torch.zeros(n[m], exp_dc[m].shape[1], 3, dtype=torch.float64, requires_grad=True)
            rest = torch.zeros(n[m], M - 1, 3, dtype=torch.float64, requires_grad=True)
            dc_mixed = dc if m == 0 else (idft[v, m].double()[None, :, None] * dc).sum(1, keepdim=True)
            shs = torch.cat([dc_mixed, rest], 1)
            d = pos[src:src + n[m]].double() - campos[v].double()
            d = d / d.norm(dim=1, keepdim=True)
            drgb = (colors * (~clamp).float())[src:src + n[m]].double()
            (torch_ref.sh_to_rgb(deg, shs, d) * drgb).sum().backward()
            exp_dc[m] += dc.grad
            exp_rest[m] += rest.grad
            src += n[m]
    with multiview.FactoredGradReducer(dense, segments=segs, mask_fn=lambda geom, gc, k: gc * (~geom).float(),
                                       rebuild_fn=_rebuild_ref_pv) as red:
        for rnd in range(2):  # twice: the payload is double-buffered
            colors, clamp, pos = views[rank]
            red.set_frame(frames[rank], idft={m: idft[rank, m] for m in frames[rank] if m != 0})
            for obs in list(rasterizer.BACKWARD_OBSERVERS):
                obs(grad_colors=colors, geomBuffer=clamp, campos=campos[rank], sh_degree=deg, num_points=colors.shape[0],
                    means3D=pos)
            dense[0].grad = torch.full_like(dense[0], float(rank + 1))
            red.begin()
            if rnd == 0:  # a backward between begin() and wait() (overlap schedule) fills the OTHER buffer
                for obs in list(rasterizer.BACKWARD_OBSERVERS):
                    obs(grad_colors=colors * 0 + 77.0, geomBuffer=clamp, campos=campos[rank], sh_degree=deg,
                        num_points=colors.shape[0], means3D=pos)
                red._pending = []  # (that pass is discarded by this test)
            red.wait()
            for m, sg in enumerate(segs):
                ok &= torch.allclose(sg.features_rest.grad.double(), exp_rest[m], atol=2e-5)
                ok &= torch.allclose(sg.features_dc.grad.double(), exp_dc[m], atol=2e-5)
            ok &= bool((dense[0].grad == 3.0).all())
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _rebuild_ref_pv(means3D, campos, drgb, degree, M):
    """_rebuild_ref with per-view positions ([V, n, 3]) for posed models."""
    if means3D.dim() == 2:
        return _rebuild_ref(means3D, campos, drgb, degree, M)
    out = 0
    for v in range(campos.shape[0]):
        out = out + _rebuild_ref(means3D[v], campos[v:v + 1], drgb[v:v + 1], degree, M)
    return out


def test_factored_grad_reducer_scene_segments_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_segments_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


# ---------------------------------------------------------------------------------------------------------------------
# Replicated densify (BASELINE configs[4]: Gaussians replicated, views sharded, densify / prune ACTIVE): per-view statistics are
# combined across ranks, every rank runs the same densify_and_prune with the same normals, the reducers are rebuilt for the
# new P, and the next exchange works on the new parameters -- which must stay BIT-identical on all ranks.
def _densify_cpu(params, accum, denom, *, states, normal_source, max_grad, min_opacity, extent, percent_dense, percent_big_ws,
                 prune_big, n_split=2, **_):
    """densify.densify_and_prune's contract on CPU tensors through the step-by-step torch restatement of the reference
    (tests/torch_ref_densify.py, pinned to GaussianModel.densify_and_prune in tests/test_densify_cpu.py)."""
    import torch_ref_densify as trd
    m = trd.Model(params, states, accum, denom)
    # the split's normals: one row per child, asked from the source once the count is known (it is a function of the
    # statistics, i.e. identical on every rank after the reduce)
    grads = (accum[:, 0:1] / denom).nan_to_num(0.0).squeeze(-1)
    big = torch.exp(params["scaling"]).max(dim=1).values > percent_dense * extent
    n_norm = n_split * int(((grads >= max_grad) & big).sum())
    normals = normal_source(n_norm, params["xyz"].device)
    sc = m.densify_and_prune(max_grad, min_opacity, extent, percent_dense, percent_big_ws, prune_big, normals, N=n_split)
    return m.p, m.s, sc, None


def _densify_worker(rank, world, port, q, mode):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from street_gaussians_amd import multiview, rasterizer
    import torch_ref_densify as trd
    P, M, deg = 301, 16, 3
    g = torch.Generator().manual_seed(11)  # identical replicas to start with
    raw = {"xyz": torch.randn(P, 3, generator=g) * 3 + torch.tensor([0.0, 0.0, 10.0]), "f_dc": torch.randn(P, 1, 3, generator=g),
           "f_rest": torch.randn(P, M - 1, 3, generator=g) * 0.1, "opacity": torch.randn(P, 1, generator=g),
           "scaling": torch.randn(P, 3, generator=g) * 0.5 - 1.0, "rotation": torch.randn(P, 4, generator=g),
           "semantic": torch.zeros(P, 0)}
    states = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in raw.items()}
    failed = []

    def chk(i, cond):
        if not cond:
            failed.append(i)
        return cond

    def leaves(p):
        return {k: v.clone().requires_grad_(True) for k, v in p.items() if v.numel()}

    def fake_backward(par, red, step):
        """What this rank's rasterizer backward reports for ITS view: rank-dependent gradients and colour gradients."""
        n = par["xyz"].shape[0]
        gr = torch.Generator().manual_seed(1000 * step + rank)
        for k in ("xyz", "opacity", "scaling", "rotation"):
            par[k].grad = torch.randn(par[k].shape, generator=gr)
        campos = torch.randn(3, generator=gr)
        for obs in list(rasterizer.BACKWARD_OBSERVERS):
            obs(grad_colors=torch.randn(n, 3, generator=gr), geomBuffer=torch.rand(n, 3, generator=gr) < 0.2, campos=campos,
                sh_degree=deg, num_points=n)

    par = leaves(raw)
    dense = lambda p: [p[k] for k in ("xyz", "opacity", "scaling", "rotation")]
    rn = multiview.ReplicatedNormals(seed=5, mode=mode)
    with multiview.FactoredGradReducer(dense(par), (par["f_dc"], par["f_rest"]), par["xyz"],
                                       mask_fn=lambda geom, gc, n: gc * (~geom).float(), rebuild_fn=_rebuild_ref) as red:
        plain = multiview.GradReducer(dense(par))
        for step in range(3):
            fake_backward(par, red, step)
            red.all_reduce()
            n = par["xyz"].shape[0]
            chk(1, par["f_rest"].grad.shape == (n, M - 1, 3) and par["xyz"].grad.shape == (n, 3))
            chk(2, multiview.replicas_identical([par[k].grad for k in par], None))  # summed gradients: the same everywhere
            # an "optimiser step" on the summed gradients keeps the replicas identical
            with torch.no_grad():
                for k in par:
                    raw[k] = (par[k] - 0.01 * par[k].grad).detach()
            # this rank's per-view densification statistics (different on every rank)
            gs = torch.Generator().manual_seed(77 * step + rank)
            accum = torch.rand(n, 2, generator=gs) * 2e-3
            denom = torch.ones(n, 1) + (torch.rand(n, 1, generator=gs) < 0.5).float()
            radii = torch.rand(n, generator=gs) * 30
            alone = accum.clone()
            raw["semantic"] = torch.zeros(n, 0)
            new_p, new_s, sc, _ = multiview.densify_replicated(
                raw, accum, denom, radii, normals=rn, states=states, densify_fn=_densify_cpu, max_grad=2.2e-3 / 1.5,
                min_opacity=0.2, extent=1.0, percent_dense=0.35, percent_big_ws=1e9, prune_big=False)
            chk(3, not torch.equal(alone, accum))  # the statistics were combined ...
            chk(4, multiview.replicas_identical([accum, denom, radii]))  # ... to the same values on every rank
            # (the first step clones, splits and prunes; later ones find fewer candidates among the now smaller points)
            chk(5, (sc["points_clone"] > 0 and sc["points_split"] > 0 and sc["points_pruned"] > 0) if step == 0
                else sc["points_clone"] + sc["points_split"] > 0) or failed.append(dict(sc))
            raw, states = {k: v.detach() for k, v in new_p.items()}, new_s
            chk(6, raw["xyz"].shape[0] != n)  # P changed
            # the replicas: parameters and Adam moments bit-identical on all ranks
            chk(7, multiview.replicas_identical([raw[k] for k in trd.NAMES] + [t for k in trd.NAMES for t in states[k]]))
            par = leaves(raw)
            red.rebuild(dense(par), (par["f_dc"], par["f_rest"]), par["xyz"])
            plain.rebuild(dense(par))
            # the flat bucket follows too
            for p_ in dense(par):
                p_.grad = torch.full(p_.shape, float(rank + 1))
            plain.all_reduce()
            chk(8, all(torch.equal(p_.grad, torch.full(p_.shape, float(sum(range(1, world + 1))))) for p_ in dense(par)))
        # a divergent replica is detected
        bad = raw["xyz"].clone()
        if rank == 1:
            bad[3, 1] += 1e-7
        chk(9, not multiview.replicas_identical([bad]))
    q.put((rank, failed))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["broadcast", "seeded"])
def test_replicated_densify_keeps_the_replicas_identical_world2(mode):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_densify_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(not f for _, f in res), res


# ---- bench.py's `--densify-loop` rank program (BASELINE configs[4] as worded: replicated Gaussians, one view per rank, the
# gradient exchange every iteration, replicated densify / prune every k-th) on two gloo ranks.  bench.DensifyLoop runs
# UNCHANGED -- step, statistics reduce, thresholds, densify_and_prune with the replicated normals, re-activation, reducer
# rebuild, the timing / report code -- on a CPU backend: a differentiable rasterizer made of the C oracle's tile lists and
# the float64 torch restatement (tests/torch_ref.py) that reports to the backward observers and the statistics sink the way
# GaussianRasterizer does, and the torch restatement of the reference's densify_and_prune (tests/torch_ref_densify.py).
class _TorchRefRasterizer:
    def __init__(self, st, backend):
        self.st, self.be, self.stats_sink = st, backend, None

    def __call__(self, means3D, means2D, opacities, shs=None, scales=None, rotations=None, **_):
        import numpy as np
        import torch_ref
        from oracle import oracle
        from street_gaussians_amd import rasterizer
        st = self.st
        H, W = int(st.image_height), int(st.image_width)
        d = lambda t: t.detach().float()
        fw = oracle.forward(means3D=d(means3D), opacities=d(opacities), viewmatrix=st.viewmatrix, projmatrix=st.projmatrix,
                            campos=st.campos, bg=st.bg, tanfovx=st.tanfovx, tanfovy=st.tanfovy, image_height=H, image_width=W,
                            sh_degree=st.sh_degree, shs=d(shs), scales=d(scales), rotations=d(rotations))
        self.be.num_rendered = int(fw.num_rendered)
        radii = torch.from_numpy(np.asarray(fw.radii).astype(np.int32))
        point_list = torch.from_numpy(np.asarray(fw.point_list).astype(np.int64)) if fw.num_rendered else torch.zeros(0, dtype=torch.int64)
        ranges = torch.from_numpy(np.asarray(fw.ranges).astype(np.int64).reshape(-1, 2))
        fw.free()
        pre = torch_ref.preprocess(means3D.double(), st.viewmatrix, st.projmatrix, st.campos, st.tanfovx, st.tanfovy, W, H,
                                   st.sh_degree, opacities.double(), shs=shs.double(), scales=scales.double(),
                                   rotations=rotations.double())
        # the screen-space handle: its gradient is dL/dmean2D (the reference's `screenspace_points`)
        pre["pix"] = pre["pix"] + means2D[:, :2].double()
        n, vis = means3D.shape[0], radii > 0
        if pre["rgb"].requires_grad:
            clamped = pre["clamped"]
            pre["rgb"].register_hook(lambda g: [obs(grad_colors=g.float(), geomBuffer=clamped, campos=st.campos,
                                                    sh_degree=st.sh_degree, num_points=n, means3D=means3D.detach())
                                                for obs in list(rasterizer.BACKWARD_OBSERVERS)] and None)
        if self.stats_sink is not None and means2D.requires_grad:
            acc, den, mr = self.stats_sink[:3]

            def stats(g):  # set_max_radii2D + add_densification_stats (street_gaussian_model.py:551-571)
                acc[vis, 0] += g[vis, :2].norm(dim=1).float()
                acc[vis, 1] += g[vis, 2].abs().float()
                den[vis] += 1.0
                mr[vis] = torch.maximum(mr[vis], radii[vis].float())
            means2D.register_hook(stats)
        color, depth, alpha, _, _ = torch_ref.render(pre, point_list, ranges, W, H, st.bg)
        return color.float(), radii, depth.float(), alpha.float(), None


class _CpuLoopBackend:
    """bench.GpuLoopBackend's members on CPU tensors."""

    def __init__(self):
        self.device, self.num_rendered = torch.device("cpu"), 0
        self.reducer_kw = dict(mask_fn=lambda geom, gc, n: gc * (~geom).float(), rebuild_fn=_rebuild_ref)

    def settings(self, **kw):
        from types import SimpleNamespace
        return SimpleNamespace(**kw)

    def rasterizer(self, st):
        return _TorchRefRasterizer(st, self)

    def densify_and_prune(self, params, accum, denom, **kw):
        # (like the fused kernels, the result carries no autograd history: xyz doubles as the rasterizer's leaf input)
        params = {k: v.detach() for k, v in params.items()}
        params.setdefault("semantic", torch.zeros(params["xyz"].shape[0], 0))
        states = {k: (a.detach(), b.detach()) for k, (a, b) in kw.pop("states").items()}
        states.setdefault("semantic", (torch.zeros_like(params["semantic"]), torch.zeros_like(params["semantic"])))
        with torch.no_grad():
            new_p, new_s, sc, idx = _densify_cpu(params, accum, denom, states=states, **kw)
        new_p, new_s = dict(new_p), dict(new_s)
        new_p.pop("semantic", None), new_s.pop("semantic", None)
        return {k: v.detach().clone() for k, v in new_p.items()}, {k: (a.detach().clone(), b.detach().clone()) for k, (a, b) in new_s.items()}, sc, idx

    def last_num_rendered(self):
        return self.num_rendered

    def sync(self):
        pass

    def mark(self):
        import time
        return time.perf_counter()

    def elapsed_ms(self, a, b):
        return 1e3 * (b - a)

    def reserve(self, n_points):
        return 0

    def allocator(self):
        return {}

    def host_wait_us(self, reset):
        return 0


def _loop_worker(rank, world, port, q):
    import sys
    from types import SimpleNamespace
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import bench
    args = SimpleNamespace(width=64, height=48, reduce="factored", exchange="blocking")
    failed = []
    try:
        loop = bench.DensifyLoop(args, 220, torch.device("cpu"), 2, dist=dist, rank=rank, force_dist=False,
                                 backend=_CpuLoopBackend())
        p0 = {k: v.clone() for k, v in loop.params.items()}
        res = loop.run(lambda: dist.barrier(), n_densify=2)
        if not (res["ranks"] == world and res["densify_steps"] == 2 and res["steps"] == 4):
            failed.append(("shape of the run", {k: res[k] for k in ("ranks", "densify_steps", "steps")}))
        if res["replicas_identical"] is not True:
            failed.append(("replicas diverged", res["replicas_identical"]))
        if res["gaussians_end"] == res["gaussians_start"] or not all(x["points_clone"] + x["points_split"] > 0 for x in res["densify_log"]):
            failed.append(("densify did nothing", res["densify_log"]))
        if loop.params["xyz"].shape[0] != res["gaussians_end"] or loop.inputs["shs"].shape[0] != res["gaussians_end"]:
            failed.append(("inputs not re-activated", loop.inputs["shs"].shape))
        if res["max_num_rendered_R"] <= 0:
            failed.append(("nothing rendered", res["max_num_rendered_R"]))
        # the exchange really summed over the two views: one more step, then every rank holds the same gradients, and the SH
        # gradient is the sum of the two views' (not this rank's own)
        loop.step()
        grads = [loop.inputs[k].grad for k in ("means3D", "scales", "rotations", "opacities", "shs")]
        from street_gaussians_amd import multiview
        if not multiview.replicas_identical(grads):
            failed.append(("summed gradients differ between ranks", None))
        del p0
    except Exception as ex:  # noqa: BLE001 -- reported to the parent
        import traceback
        failed.append(("exception", traceback.format_exc()[-1500:]))
    q.put((rank, failed))
    dist.destroy_process_group()


def test_bench_densify_loop_rank_program_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(not f for _, f in res), res
