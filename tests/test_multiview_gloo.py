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
