"""world_size-2 gloo tests (CPU) of the view-sharding exchange step (street_gaussians_amd/multiview.py)."""
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
            exp = all_grads[0][4]
        ok &= torch.allclose(p.grad, exp, atol=1e-6)
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
        ok &= torch.allclose(p.grad, sum(all_grads[r][i] for r in range(world)), atol=1e-6)
    red.wait()  # idempotent
    acc = torch.full((50, 2), float(rank + 1))
    den = torch.full((50, 1), 1.0)
    rad = torch.arange(50, dtype=torch.float32) * (1 if rank == 0 else -1)
    multiview.reduce_densification_stats(acc, den, rad)
    ok &= bool((acc == 3.0).all() and (den == 2.0).all() and (rad == torch.arange(50).clamp(min=0)).all())
    ok &= multiview.view_for_rank(list(range(5)), step=1) == (world + rank) % 5
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["all_reduce", "rs_ag"])
def test_grad_reducer_world2(mode):
    world, port = 2, _free_port()
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


def test_factored_grad_reducer_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_factored_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
