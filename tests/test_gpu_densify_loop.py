"""`-m gpu` test of BASELINE.json configs[4] as written: the densify / prune step ACTIVE between rasterizer iterations
(/root/reference/train.py:187-210, lib/models/gaussian_model.py:522-553), so that P, R and the three scratch buffers
change between consecutive forward calls.

One loop = k iterations of forward + backward with the fused statistics sink (set_max_radii2D +
add_densification_stats, street_gaussian_model.py:551-571), then densify_and_prune on the raw parameters, then the next
forward on the NEW set.  Checked:
  * the statistics the sink accumulated == the reference's formulas applied to the per-view dL/dmeans2D and radii;
  * densify_and_prune == the step-by-step torch reference (tests/torch_ref_densify.py) fed with those statistics;
  * the render of the post-densify set == the C oracle on the same tensors -- every integer output bit-exact, images and
    gradients within the parity gates -- also when R grows past the binning buffer's 25 % head-room (second request);
  * reset_opacity (train.py:207-208) followed by a render;
  * the sink with a per-frame segment map (a frame renders a subset of the sub-models) scatters into persistent rows;
  * two backward passes over one forward (retain_graph) == fresh forward + backward (the row flags live in the forward's
    binning buffer and are re-marked, not accumulated)."""
import numpy as np
import pytest
import torch

import torch_ref_densify as ref
from gpu_utils import grad_close, image_close, npy, raw_forward, restrict_oracle, settings
from helpers import oracle_kwargs
from oracle import oracle
from street_gaussians_amd import densify, rasterizer, scene
from street_gaussians_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _raw(sc):
    op = sc.opacities.clamp(1e-6, 1 - 1e-6)
    return {"xyz": sc.means3D.clone(), "f_dc": sc.shs[:, :1, :].contiguous(), "f_rest": sc.shs[:, 1:, :].contiguous(),
            "opacity": torch.log(op / (1 - op)), "scaling": torch.log(sc.scales), "rotation": sc.rotations.clone(),
            "semantic": sc.semantics.clone()}


def _activate(p):
    return syn.Scene(p["xyz"].contiguous(), torch.exp(p["scaling"]).contiguous(),
                     torch.nn.functional.normalize(p["rotation"]).contiguous(), torch.sigmoid(p["opacity"]).contiguous(),
                     torch.cat([p["f_dc"], p["f_rest"]], 1).contiguous(), p["semantic"].contiguous())


def _render(cam, sc, wts, sink=None, S=0):
    """One iteration through the drop-in API; returns (outputs, leaf tensors, radii)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    t = {k: getattr(sc, k).cuda().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "shs"]}
    sem = sc.semantics.cuda().requires_grad_(True) if S else None
    m2d = torch.zeros(sc.P, 3, device="cuda", requires_grad=True)
    r = GaussianRasterizer(settings(cam, bg=torch.tensor([0.1, 0.2, 0.3])))
    r.stats_sink = sink
    color, radii, depth, alpha, semantic = r(t["means3D"], m2d, t["opacities"], shs=t["shs"], scales=t["scales"],
                                             rotations=t["rotations"], semantics=sem)
    outs, grads = [color, depth, alpha], [wts["color"].cuda(), wts["depth"].cuda(), wts["alpha"].cuda()]
    if S:
        outs.append(semantic)
        grads.append(wts["semantic"].cuda())
    torch.autograd.backward(outs, grads)
    t["means2D"] = m2d
    return dict(color=color, depth=depth, alpha=alpha, semantic=semantic), t, radii


def _check_against_oracle(cam, sc, wts, S):
    kw = oracle_kwargs(cam, sc, bg=torch.tensor([0.1, 0.2, 0.3]))
    fw = oracle.forward(**kw)
    res, internal = raw_forward(kw)
    b = restrict_oracle(internal, fw, kw)  # the oracle's binning arrays restricted to the emitted tile rects
    assert res["R"] == b.num_rendered
    assert (npy(res["radii"]) == fw.radii).all()
    assert (npy(internal("tiles_touched")).view(np.uint32) == b.tiles_touched).all()
    assert (npy(internal("keys")).view(np.uint64) == b.keys).all()
    assert (npy(internal("point_list")).view(np.uint32) == b.point_list).all()
    assert (npy(internal("ranges")).view(np.uint32) == b.ranges).all()
    for k in ["color", "depth", "alpha"] + (["semantic"] if S else []):
        image_close(npy(res[k]), getattr(fw, k), name=f"post-densify {k}")
    out, t, radii = _render(cam, sc, wts, S=S)
    g = oracle.backward(fw, wts["color"], wts["depth"], wts["alpha"], wts["semantic"] if S else None)
    for name, key in [("means3D", "means3D"), ("shs", "sh"), ("opacities", "opacity"), ("scales", "scales"),
                      ("rotations", "rotations"), ("means2D", "means2D")]:
        grad_close(npy(t[name].grad).reshape(g[key].shape), g[key], name=f"post-densify dL/d{name}", rel=2e-4, abs_frac=3e-4)
    R = fw.num_rendered
    fw.free()
    return R


@pytest.mark.parametrize("P0,S,grow", [(30000, 0, 0.10), (12000, 3, 0.75)])
def test_render_densify_render_loop(P0, S, grow):
    cam = syn.make_camera(320, 208, fx=340.0, yaw_deg=2.0)
    sc0 = syn.make_scene(P0, cam, S=S, seed=3, scale_px=0.004)
    params = _raw(sc0)
    g = torch.Generator().manual_seed(11)
    states = {k: (torch.randn(v.shape, generator=g), torch.randn(v.shape, generator=g).abs()) for k, v in params.items()}
    wts = syn.loss_weights(cam, S=S)
    R_prev = None
    for interval in range(2):
        sc = _activate(params)
        P = sc.P
        stats = scene.FlatStats([P], "cuda")
        acc_ref = torch.zeros(P, 2)
        den_ref = torch.zeros(P, 1)
        rad_ref = torch.zeros(P)
        for it in range(3):
            w_it = syn.loss_weights(cam, S=S, seed=5 + it)
            out, t, radii = _render(cam, sc, w_it, sink=stats.sink(), S=S)
            if R_prev is not None and it == 0:
                # the forward after a densify step ran on buffers sized for the OLD set (the binning buffer is requested
                # with the previous R + 25 % before R is known): still the oracle's result
                assert rasterizer.last_num_rendered() != R_prev
            gm = t["means2D"].grad.cpu()
            vis = (radii > 0).cpu()
            # street_gaussian_model.py:551-571 / gaussian_model.py:555-560
            acc_ref[vis, 0:1] += torch.norm(gm[vis, :2], dim=-1, keepdim=True)
            acc_ref[vis, 1:2] += torch.norm(gm[vis, 2:], dim=-1, keepdim=True)
            den_ref[vis] += 1
            rad_ref[vis] = torch.max(rad_ref[vis], radii.cpu()[vis].float())
        assert torch.allclose(stats.xyz_gradient_accum.cpu(), acc_ref, rtol=1e-6, atol=1e-12)
        assert torch.equal(stats.denom.cpu(), den_ref)
        assert torch.equal(stats.max_radii2D.cpu(), rad_ref)
        R_prev = rasterizer.last_num_rendered()

        # ---- densify_and_prune on the accumulated statistics: quantile thresholds (clone + split ~ `grow` of the points)
        acc, den = stats.xyz_gradient_accum.cpu(), stats.denom.cpu()
        gr = (acc[:, 0:1] / den).nan_to_num(0.0).flatten()
        max_grad = float(torch.quantile(gr, 1.0 - grow))
        big = torch.exp(params["scaling"]).max(dim=1).values
        dense = float(torch.quantile(big[gr >= max_grad], 0.5))
        kw = dict(max_grad=max_grad, min_opacity=float(torch.quantile(torch.sigmoid(params["opacity"]), 0.05)), extent=1.0,
                  percent_dense=dense, percent_big_ws=1e9)
        normals = torch.randn(2 * P, 3, generator=g)
        m = ref.Model(params, states, acc, den)
        want = m.densify_and_prune(prune_big=False, normals=normals, **kw)
        new_p, new_s, got, index = densify.densify_and_prune(
            {k: v.cuda() for k, v in params.items()}, stats.xyz_gradient_accum, stats.denom, prune_big=False,
            states={k: (a.cuda(), b.cuda()) for k, (a, b) in states.items()}, normals=normals[:2 * want["points_split"]].cuda(),
            **kw)
        assert got == want
        assert want["points_clone"] > 0 and want["points_split"] > 0 and want["points_pruned"] > 0
        keep = (index["kind"] != 2).cpu()
        for k in params:
            if k in ("xyz", "scaling"):
                assert torch.allclose(new_p[k].cpu(), m.p[k], rtol=2e-6, atol=2e-6), k
                assert torch.equal(new_p[k].cpu()[keep], m.p[k][keep]), k
            else:
                assert torch.equal(new_p[k].cpu(), m.p[k]), k
            for j in range(2):
                assert torch.equal(new_s[k][j].cpu(), m.s[k][j]), (k, j)
        params = {k: v.cpu() for k, v in new_p.items()}
        states = {k: (a.cpu(), b.cpu()) for k, (a, b) in new_s.items()}
        assert params["xyz"].shape[0] != P

        # ---- the next iteration's render, on the changed P, against the oracle on the same tensors
        R_new = _check_against_oracle(cam, _activate(params), wts, S)
        if grow > 0.5 and interval == 0:
            # R grew: when it passes the 25 % head-room of the binning buffer (requested before R is known, from the
            # previous forward's R) the exact-size re-request runs (sgr_api.hip: r_hint)
            assert R_new > R_prev

    # ---- reset_opacity (train.py:207-208, gaussian_model.py:410-414), then a render on the reset set
    st = (states["opacity"][0].cuda(), states["opacity"][1].cuda())
    new_op = densify.reset_opacity(params["opacity"].cuda(), st)
    m2 = ref.Model(params, states, torch.zeros(1, 2), torch.ones(1, 1))
    m2.reset_opacity()
    assert torch.allclose(new_op.cpu(), m2.p["opacity"], rtol=1e-6, atol=1e-6)
    assert float(st[0].abs().max()) == 0.0 and float(st[1].abs().max()) == 0.0
    params["opacity"] = new_op.cpu()
    _check_against_oracle(cam, _activate(params), wts, S)


def test_stats_sink_with_a_per_frame_segment_map():
    """Three persistent models; frame A renders models (0, 2), frame B renders (2, 1) -- a subset and a different order,
    as the reference's per-frame graph_obj_list does (street_gaussian_model.py:230-250)."""
    cam = syn.make_camera(320, 208, fx=340.0, yaw_deg=2.0)
    counts = [5000, 1200, 800]
    full = syn.make_scene(sum(counts), cam, S=0, seed=9, scale_px=0.004)
    starts = np.cumsum([0] + counts)
    stats = scene.FlatStats(counts, "cuda")
    ref_acc, ref_den, ref_rad = torch.zeros(sum(counts), 2), torch.zeros(sum(counts), 1), torch.zeros(sum(counts))
    wts = syn.loss_weights(cam, S=0)
    for order in [(0, 2), (2, 1), (0, 1, 2)]:
        rows = torch.cat([torch.arange(starts[m], starts[m + 1]) for m in order])
        sub = syn.Scene(*[t[rows].contiguous() for t in (full.means3D, full.scales, full.rotations, full.opacities, full.shs,
                                                         full.semantics)])
        out, t, radii = _render(cam, sub, wts, sink=stats.sink(models=order))
        gm, vis = t["means2D"].grad.cpu(), (radii > 0).cpu()
        ref_acc[rows[vis], 0] += torch.norm(gm[vis, :2], dim=-1)
        ref_acc[rows[vis], 1] += gm[vis, 2].abs()
        ref_den[rows[vis]] += 1
        ref_rad[rows[vis]] = torch.max(ref_rad[rows[vis]], radii.cpu()[vis].float())
    assert torch.allclose(stats.xyz_gradient_accum.cpu(), ref_acc, rtol=1e-6, atol=1e-12)
    assert torch.equal(stats.denom.cpu(), ref_den)
    assert torch.equal(stats.max_radii2D.cpu(), ref_rad)
    views = stats.views()
    assert [v["denom"].shape[0] for v in views] == counts
    # a sink without a map must still cover the whole call
    from street_gaussians_amd._native import SgrError
    with pytest.raises(SgrError):
        _render(cam, syn.Scene(*[t[:100].contiguous() for t in (full.means3D, full.scales, full.rotations, full.opacities,
                                                                 full.shs, full.semantics)]), wts, sink=stats.sink())


def test_two_backward_passes_over_one_forward():
    """retain_graph: the second backward (different upstream gradients, then all-zero ones) over the same forward state
    must equal a fresh forward + backward -- the rows a backward writes are a function of the forward's hit record alone
    (include/sgr.h, binning buffer contract), so the row flags are re-marked, never accumulated."""
    from diff_gaussian_rasterization import GaussianRasterizer
    cam = syn.make_camera(320, 208, fx=340.0, yaw_deg=2.0)
    sc = syn.make_scene(9000, cam, S=3, seed=4, scale_px=0.004)
    t = {k: getattr(sc, k).cuda().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "shs", "semantics"]}
    m2d = torch.zeros(sc.P, 3, device="cuda", requires_grad=True)
    leaves = list(t.values()) + [m2d]
    r = GaussianRasterizer(settings(cam))

    def fwd():
        return r(t["means3D"], m2d, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"],
                 semantics=t["semantics"])

    def bwd(outs, w, retain):
        for x in leaves:
            x.grad = None
        color, radii, depth, alpha, sem = outs
        torch.autograd.backward([color, depth, alpha, sem], [w["color"].cuda(), w["depth"].cuda(), w["alpha"].cuda(),
                                                            w["semantic"].cuda()], retain_graph=retain)
        return [x.grad.clone() for x in leaves]

    wa, wb = syn.loss_weights(cam, S=3, seed=1), syn.loss_weights(cam, S=3, seed=2)
    wz = {k: torch.zeros_like(v) for k, v in wa.items()}
    outs = fwd()
    bwd(outs, wa, True)
    g_b = bwd(outs, wb, True)
    g_z = bwd(outs, wz, True)
    g_a2 = bwd(outs, wa, False)
    fresh_b = bwd(fwd(), wb, False)
    fresh_a = bwd(fwd(), wa, False)
    for x, y in zip(g_b, fresh_b):
        assert torch.equal(x, y)
    for x, y in zip(g_a2, fresh_a):
        assert torch.equal(x, y)
    for x in g_z:
        assert float(x.abs().max()) == 0.0
