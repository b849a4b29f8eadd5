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


def test_backward_updates_flat_densification_stats_in_place():
    """GaussianRasterizer.stats_sink (sgr_backward_ex): the backward's per-Gaussian epilogue applies set_max_radii2D +
    add_densification_stats (street_gaussian_model.py:551-571) to flat statistics whose slices are the sub-models'
    tensors; two views accumulate.  Checked against the separate scatter kernel and against torch ops."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gpu_utils import dev, settings
    from street_gaussians_amd import synthetic as syn
    counts = [1500, 300, 200]
    P = sum(counts)
    fused = scene.FlatStats(counts, "cuda")
    sep = [{k: torch.zeros_like(v) for k, v in m.items()} for m in fused.views()]
    want_acc, want_den, want_max = torch.zeros(P, 2), torch.zeros(P, 1), torch.zeros(P)
    for view in range(2):
        cam = syn.make_camera(320, 200, fx=300.0, yaw_deg=4.0 * view)
        sc = syn.make_scene(P, cam, S=0, seed=3, scale_px=0.006)
        t = {k: dev(getattr(sc, k)).requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "shs"]}
        m2d = torch.zeros(P, 3, device="cuda", requires_grad=True)
        rast = GaussianRasterizer(settings(cam))
        rast.stats_sink = fused.sink()
        color, radii, depth, alpha, _ = rast(t["means3D"], m2d, t["opacities"], shs=t["shs"], scales=t["scales"],
                                             rotations=t["rotations"])
        w = syn.loss_weights(cam)
        ((color * dev(w["color"])).sum() + (depth * dev(w["depth"])).sum()).backward()
        scene.densification_stats(sep, m2d.grad, radii)
        g, r = m2d.grad.cpu(), radii.cpu()
        vis = r > 0
        want_acc[vis, 0] += torch.norm(g[vis, :2], dim=-1)
        want_acc[vis, 1] += torch.norm(g[vis, 2:], dim=-1)
        want_den[vis] += 1
        want_max[vis] = torch.max(want_max[vis], r[vis].float())
    assert float(want_den.sum()) > 0
    for a, b in zip(fused.views(), sep):
        for k in a:
            assert torch.allclose(a[k], b[k], rtol=1e-6, atol=0), k
    assert torch.allclose(fused.xyz_gradient_accum.cpu(), want_acc, rtol=2e-6, atol=1e-12)
    assert torch.equal(fused.denom.cpu(), want_den) and torch.equal(fused.max_radii2D.cpu(), want_max)


@pytest.mark.parametrize("source", ["ply", "pth"])
def test_scene_file_in_the_reference_layout_renders_like_the_oracle(source):
    """n4 end to end: the fixture written with the reference's make_ply / state_dict (tests/golden/
    make_scene_fixture.py) -> plyio / checkpoint -> scene.Segment -> compose -> rasterizer, against the C oracle fed
    with the torch restatement of the reference's flattening of the same parameters."""
    import os
    from diff_gaussian_rasterization import GaussianRasterizer
    from gpu_utils import image_close, npy, settings
    from oracle import oracle
    from street_gaussians_amd import checkpoint, plyio, synthetic as syn
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene")
    if source == "ply":
        models = plyio.read_scene_ply(os.path.join(gold, "scene_ref_layout.ply"))
    else:
        models = checkpoint.load(os.path.join(gold, "scene_ref_state.pth"))
    pose = torch.tensor([0.95, 0.05, 0.25, -0.1, 1.0, -0.5, 9.0])
    idft = torch.tensor([0.8, 0.3, -0.2])
    segs = checkpoint.segments(models, "cuda", poses={"obj_001": pose}, idfts={"obj_001": idft})
    assert [s.kind for s in segs] == [scene.SEG_STATIC, scene.SEG_ACTOR]
    S, M = 5, 16
    means3D, rot, scales, opac, shs, sem = scene.compose(segs, M, S)
    cam = syn.make_camera(320, 200, fx=260.0)
    color, radii, depth, alpha, semo = GaussianRasterizer(settings(cam))(means3D, None, opac, shs=shs, scales=scales,
                                                                         rotations=rot, semantics=sem)
    t = lambda v: torch.as_tensor(v).float()
    flat = ref.compose([dict(xyz=t(m["xyz"]), rotation=t(m["rotation"]), scaling=t(m["scaling"]), opacity=t(m["opacity"]),
                             features_dc=t(m["features_dc"]), features_rest=t(m["features_rest"]), semantic=t(m["semantic"]),
                             **(dict(pose=pose, idft=idft) if name == "obj_001" else {}))
                        for name, m in models.items()], M, S)
    fw = oracle.forward(means3D=flat[0], opacities=flat[3], viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
                        campos=cam.campos, bg=torch.zeros(3), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                        image_height=cam.image_height, image_width=cam.image_width, sh_degree=3, shs=flat[4],
                        scales=flat[2], rotations=flat[1], semantics=flat[5])
    assert int((fw.radii > 0).sum()) > 300
    assert (npy(radii) != fw.radii).mean() < 2e-3  # compose differs from the float32 torch flattening in the last ulp
    image_close(npy(color), fw.color, rel=2e-3, name="scene color", max_outliers=50)
    image_close(npy(alpha), fw.alpha, rel=2e-3, name="scene alpha", max_outliers=50)
    image_close(npy(semo), fw.semantic, rel=2e-3, name="scene semantic", max_outliers=200)
    fw.free()


@pytest.mark.parametrize("counts,M,S,seed", [((700, 300, 257), 16, 5, 2), ((5, 1, 3, 513), 4, 19, 3), ((1000,), 16, 0, 1)])
def test_flat_parameter_mode_equals_per_model_compose(counts, M, S, seed):
    """scene.FlatScene: the same kernels fed from ONE leaf tensor per attribute (8 autograd leaves instead of 8 per
    sub-model) -- outputs and every gradient bit-identical to scene.compose on the per-model leaves."""
    segs = _make(counts, M, S, seed)
    gsegs, gleaves = _to_gpu(segs)
    outs = scene.compose(gsegs, M, S)
    flat = scene.FlatScene.from_segments(gsegs)
    fouts = flat.compose(M, S, flip_masks=[s.flip_mask for s in gsegs])
    for a, b in zip(outs, fouts):
        assert torch.equal(a, b)
    g = torch.Generator().manual_seed(100 + seed)
    ups = [torch.randn(o.shape, generator=g).cuda() for o in outs]
    sel = [i for i, o in enumerate(outs) if o.requires_grad]
    torch.autograd.backward([outs[i] for i in sel], [ups[i] for i in sel])
    torch.autograd.backward([fouts[i] for i in sel], [ups[i] for i in sel])
    views = flat.views()
    fgrads = {k: flat.tensors[k].grad for k in scene._FLAT}
    offs = flat._offsets()
    for k, (seg, m, o) in enumerate(zip(gsegs, flat.meta, offs)):
        n = m["count"]
        for name in ("xyz", "rotation", "scaling", "opacity"):
            assert torch.equal(getattr(seg, name).grad, fgrads[name][o["row"]:o["row"] + n]), (k, name)
        assert torch.equal(seg.features_rest.grad.reshape(n, -1), fgrads["features_rest"][o["row"]:o["row"] + n]), k
        assert torch.equal(seg.features_dc.grad.reshape(-1), fgrads["features_dc"][o["dc"]:o["dc"] + n * m["fourier_dim"] * 3]), k
        if S and seg.semantic is not None:
            assert torch.equal(seg.semantic.grad.reshape(-1), fgrads["semantic"][o["sem"]:o["sem"] + n * m["sem_width"]]), k
    a = 0
    for seg in gsegs:
        if seg.pose is not None:
            assert torch.equal(seg.pose.grad, flat.poses.grad[a])
            a += 1
    assert len(views) == len(counts) and views[0]["xyz"].shape[0] == counts[0]
    # per-frame inputs: a NON-leaf pose tensor (the caller derives it from its tracking refinements) and explicit IDFT rows
    if flat.poses is not None:
        want_pose = flat.poses.grad.clone()
        for t in list(flat.tensors.values()) + [flat.poses]:
            t.grad = None
        raw = flat.poses.detach().clone().requires_grad_(True)
        fouts2 = flat.compose(M, S, flip_masks=[s.flip_mask for s in gsegs], poses=raw * 1.0, idfts=[s.idft for s in gsegs])
        for a, b in zip(fouts, fouts2):
            assert torch.equal(a, b)
        torch.autograd.backward([fouts2[i] for i in sel], [ups[i] for i in sel])
        assert torch.equal(raw.grad, want_pose) and flat.poses.grad is None
