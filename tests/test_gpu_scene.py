"""`-m gpu` parity tests of the scene-graph rows (street_gaussians_amd/scene.py + csrc/sgr_scene.hip) against the
torch restatement of the reference's flattening code (tests/torch_ref_scene.py, evaluated in float64 on the CPU)."""
import numpy as np
import pytest
import torch

import torch_ref_scene as ref
from street_gaussians_amd import scene

pytestmark = pytest.mark.gpu
NAMES = ["xyz", "rotation", "scaling", "opacity", "features_dc", "features_rest", "semantic", "pose"]


def _make(counts, M, S, seed, fourier=(3, 1), train=True):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    segs = []
    for k, n in enumerate(counts):
        d = dict(xyz=r(n, 3) * 3, rotation=r(n, 4), scaling=r(n, 3) * 0.5 - 1, opacity=r(n, 1) * 2,
                 features_rest=r(n, M - 1, 3))
        if k == 0:
            d.update(features_dc=r(n, 1, 3), semantic=r(n, S) if S else None,
                     semantic_mode="probabilities" if seed % 2 else "logits")
        else:
            C = fourier[(k - 1) % len(fourier)]
            d.update(features_dc=r(n, C, 3), semantic=r(n, 1) if S else None, pose=r(7) * torch.tensor([1, 1, 1, 1, 5, 5, 5.0]),
                     idft=r(C) if C > 1 else None, class_label=(k * 2) % max(S, 1),
                     semantic_mode="logits" if k % 2 else "probabilities")
            if train and k % 2:
                d["flip_mask"] = torch.rand(n, generator=g) < 0.5
        segs.append(d)
    return segs


def _to_gpu(segs):
    out = []
    leaves = []
    for d in segs:
        kw = {}
        for n in NAMES:
            t = d.get(n)
            kw[n] = None if t is None else t.float().cuda().requires_grad_(True)
            if kw[n] is not None:
                leaves.append(kw[n])
        kw["idft"] = None if d.get("idft") is None else d["idft"].float().cuda()
        kw["flip_mask"] = None if d.get("flip_mask") is None else d["flip_mask"].cuda()
        out.append(scene.Segment(class_label=d.get("class_label", 0), semantic_mode=d.get("semantic_mode", "logits"), **kw))
    return out, leaves


def _close(a, b, rel, name):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    assert a.shape == b.shape, name
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= rel * scale, f"{name}: {err / scale:.3g}"


@pytest.mark.parametrize("counts,M,S,seed", [((1000,), 16, 0, 1), ((700, 300, 257), 16, 5, 2), ((5, 1, 3, 513), 4, 19, 3),
                                             ((0, 40), 16, 2, 4), ((300, 129), 1, 0, 5)])
def test_compose_forward_backward_match_reference_flattening(counts, M, S, seed):
    segs = _make(counts, M, S, seed)
    gsegs, gleaves = _to_gpu(segs)
    outs = scene.compose(gsegs, M, S)
    # float64 reference with autograd
    rleaves = []
    for d in segs:
        for n in NAMES:
            if d.get(n) is not None:
                d[n] = d[n].clone().requires_grad_(True)
                rleaves.append(d[n])
    routs = ref.compose(segs, M, S)
    names = ["means3D", "rotations", "scales", "opacities", "shs", "semantics"]
    for o, r, n in zip(outs, routs, names):
        _close(o.detach().cpu().numpy(), r.detach().numpy(), 3e-6, n)
    g = torch.Generator().manual_seed(100 + seed)
    ups = [torch.randn(r.shape, generator=g, dtype=torch.float64) for r in routs]
    sel = [i for i, r in enumerate(routs) if r.requires_grad]  # semantics [N, 0] carries no graph when S == 0

    def bwd(os_, cast):
        torch.autograd.backward([os_[i] for i in sel], [cast(ups[i]) for i in sel])

    bwd(routs, lambda u: u)
    bwd(outs, lambda u: u.float().cuda())
    assert len(gleaves) == len(rleaves)
    for i, (a, b) in enumerate(zip(gleaves, rleaves)):
        _close(a.grad.cpu().numpy(), b.grad.numpy(), 2e-5, f"grad[{i}] shape {tuple(b.shape)}")
    # deterministic pose gradients (fixed-order chunk sums)
    for p in gleaves:
        p.grad = None
    bwd(scene.compose(gsegs, M, S), lambda u: u.float().cuda())
    again = [p.grad.clone() for p in gleaves]
    for p in gleaves:
        p.grad = None
    bwd(scene.compose(gsegs, M, S), lambda u: u.float().cuda())
    for a, p in zip(again, gleaves):
        assert torch.equal(a, p.grad)


def test_compose_feeds_the_rasterizer_and_partial_grads():
    """End to end: compose -> rasterizer -> backward reaches the raw parameters and the actor pose; inputs that do not
    require grad get none."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gpu_utils import dev, settings
    from street_gaussians_amd import synthetic as syn
    cam = syn.make_camera(320, 200, fx=300.0)
    sc = syn.make_scene(3000, cam, S=0, seed=2)
    raw = lambda t: t.cuda().requires_grad_(True)
    bk = scene.Segment(xyz=raw(sc.means3D), rotation=raw(sc.rotations * 2.0), scaling=raw(sc.scales.log()),
                       opacity=raw(torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4))), features_dc=raw(sc.shs[:, :1]),
                       features_rest=raw(sc.shs[:, 1:]))
    n = 500
    pose = torch.tensor([0.9, 0.1, -0.2, 0.3, 0.5, -0.2, 6.0], device="cuda", requires_grad=True)
    act = scene.Segment(xyz=raw(torch.randn(n, 3) * 0.5), rotation=raw(torch.randn(n, 4)), scaling=raw(torch.full((n, 3), -3.0)),
                        opacity=raw(torch.zeros(n, 1)), features_dc=raw(torch.randn(n, 2, 3) * 0.3),
                        features_rest=sc.shs[:n, 1:].cuda(), pose=pose, idft=torch.tensor([0.7, -0.4], device="cuda"))
    means3D, rot, scales, opac, shs, sem = scene.compose([bk, act], 16, 0)
    assert means3D.shape == (3500, 3) and sem.shape == (3500, 0)
    color, radii, depth, alpha, _ = GaussianRasterizer(settings(cam))(means3D, None, opac, shs=shs, scales=scales, rotations=rot)
    (color.sum() + depth.sum() + alpha.sum()).backward()
    assert pose.grad is not None and torch.isfinite(pose.grad).all() and float(pose.grad.abs().max()) > 0
    assert bk.xyz.grad.shape == (3000, 3) and float(bk.features_rest.grad.abs().max()) > 0
    assert act.features_dc.grad.shape == (n, 2, 3) and act.features_rest.grad is None


def test_densification_stats_match_reference_scatter():
    g = torch.Generator().manual_seed(9)
    counts = [700, 0, 300, 257]
    N = sum(counts)
    grad = torch.randn(N, 3, generator=g)
    radii = torch.randint(-2, 40, (N,), generator=g, dtype=torch.int32)
    mk = lambda n: dict(xyz_gradient_accum=torch.rand(n, 2, generator=g), denom=torch.rand(n, 1, generator=g).round(),
                        max_radii2D=torch.rand(n, generator=g) * 30)
    cpu = [mk(n) for n in counts]
    gpu = [{k: v.clone().cuda() for k, v in m.items()} for m in cpu]
    ref.densification_stats(cpu, grad, radii)
    scene.densification_stats(gpu, grad.cuda(), radii.cuda())
    for a, b in zip(gpu, cpu):
        for k in a:
            assert torch.allclose(a[k].cpu(), b[k], rtol=1e-6, atol=1e-7), k
