"""`-m gpu` tests of the view-sharded exchange step (street_gaussians_amd/multiview.py + csrc/sgr_multiview.hip):
the SH gradient rebuilt from per-view dRGB must equal the sum of the per-view dL/dSH tensors it replaces."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from gpu_utils import dev, npy, settings
from helpers import oracle_kwargs
from oracle import oracle
from street_gaussians_amd import multiview, rasterizer
from street_gaussians_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _views(V):
    return [syn.make_camera(480, 320, fx=512.5, yaw_deg=4.0 * v) for v in range(V)]


@pytest.mark.parametrize("deg", [3, 1])
def test_factored_sh_gradient_equals_sum_over_views(deg):
    from diff_gaussian_rasterization import GaussianRasterizer
    V = 3
    cams = _views(V)
    sc = syn.make_scene(20000, cams[0], S=0, seed=11)
    sc.shs[::3, 0, :] -= 2.0  # plenty of clamped channels
    t = {k: dev(getattr(sc, k)).requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "shs"]}

    def run_views():
        for v, cam in enumerate(cams):
            st = settings(cam)._replace(sh_degree=deg)
            color, radii, depth, alpha, sem = GaussianRasterizer(st)(t["means3D"], None, t["opacities"], shs=t["shs"],
                                                                     scales=t["scales"], rotations=t["rotations"])
            w = syn.loss_weights(cam, seed=3 + v)
            torch.autograd.backward([color, depth, alpha], [dev(w["color"]), dev(w["depth"]), dev(w["alpha"])])

    run_views()  # plain autograd accumulation: shs.grad = sum of the per-view dL/dSH
    want = {k: p.grad.clone() for k, p in t.items()}
    for p in t.values():
        p.grad = None
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port.getsockname()[1])
    port.close()
    dist.init_process_group("nccl", rank=0, world_size=1)  # one-rank RCCL group: the collectives really run
    try:
        dense = [t[k] for k in ["means3D", "scales", "rotations", "opacities"]]
        with multiview.FactoredGradReducer(dense, t["shs"], t["means3D"], views_per_rank=V, force=True) as red:
            run_views()
            red.all_reduce()
            assert red.nbytes == 4 * (sum(p.numel() for p in dense) + V * (3 + 3 * sc.P))
        assert len(rasterizer.BACKWARD_OBSERVERS) == 0
    finally:
        dist.destroy_process_group()
    got, ref = npy(t["shs"].grad), npy(want["shs"])
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-6 * scale, np.abs(got - ref).max() / scale
    assert (got[:, (deg + 1) ** 2:, :] == 0).all()
    for k in ["means3D", "scales", "rotations", "opacities"]:
        assert torch.equal(t[k].grad, want[k]), k


def test_masked_color_grad_and_rebuild_match_oracle():
    """One view, through the raw entry points: dRGB = dL/dcolour with the oracle's clamp mask, and the rebuilt dL/dSH
    equals the oracle's (backward.cu:46-105)."""
    from gpu_utils import raw_backward, raw_forward
    from street_gaussians_amd import _C
    cam = _views(1)[0]
    sc = syn.make_scene(5000, cam, S=0, seed=12)
    sc.shs[::2, 0, :] -= 2.0
    kw = oracle_kwargs(cam, sc, deg=2)
    wts = syn.loss_weights(cam)
    fw = oracle.forward(**kw)
    ref = oracle.backward(fw, wts["color"], wts["depth"], wts["alpha"], None)
    res, _ = raw_forward(kw)
    g = raw_backward(kw, res, wts)
    drgb = _C.masked_color_grad(res["geom"], g["colors"], sc.P)
    mask = fw.clamped.reshape(sc.P, 3) != 0
    assert (npy(drgb)[mask] == 0).all() and (npy(drgb)[~mask] == npy(g["colors"])[~mask]).all()
    dsh = _C.sh_grad_from_views(dev(sc.means3D), dev(cam.campos).reshape(1, 3), drgb.reshape(1, sc.P, 3), 2, 16)
    want = ref["sh"].reshape(sc.P, 16, 3)
    assert np.abs(npy(dsh) - want).max() <= 1e-5 * np.abs(want).max()
    assert torch.allclose(dsh, g["sh"].reshape(sc.P, 16, 3), rtol=0, atol=2e-6 * float(g["sh"].abs().max()))
    fw.free()


def test_bench_runs_the_rccl_exchange_on_one_rank():
    """bench.py with a forced one-rank RCCL group (SGR_BENCH_FORCE_DIST=1): the factored exchange, overlapped and
    blocking, runs end to end on the GPU and prints the contract's JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("overlap", "blocking"):
        env = dict(os.environ, SGR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                              "--gaussians", "100000", "--no-cpu-baseline", "--no-other-configs", "--exchange", mode],
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().split("\n")[-1])
        assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["exchange_bytes_per_rank"] > 0
        assert (mode == "overlap") == ("overlapped" in line["config"]["exchange_schedule"])


def test_async_exchange_matches_blocking_exchange():
    """GradReducer / FactoredGradReducer: begin() + wait() give the same gradients as all_reduce(), also when the next
    step's work is queued in between (one-rank RCCL group)."""
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port.getsockname()[1])
    port.close()
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        _async_body()
    finally:
        dist.destroy_process_group()


def _async_body():
    g = torch.Generator().manual_seed(1)
    params = [torch.zeros(50000, 3, device="cuda", requires_grad=True), torch.zeros(50000, 1, device="cuda", requires_grad=True)]
    grads = [torch.randn(p.shape, generator=g).cuda() for p in params]
    red = multiview.GradReducer(params, force=True)
    for p, gr in zip(params, grads):
        p.grad = gr.clone()
    red.begin()
    junk = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")  # "the next forward"
    for p in params:
        p.grad = torch.zeros_like(p)
    red.wait()
    torch.cuda.synchronize()
    for p, gr in zip(params, grads):
        assert torch.equal(p.grad, gr)
    assert torch.isfinite(junk).all()
