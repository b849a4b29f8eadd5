"""`-m gpu` parity test of the plan + gather density control (street_gaussians_amd/densify.py, csrc/sgr_densify.hip)
against the step-by-step torch reference (tests/torch_ref_densify.py, float32 on the CPU)."""
import pytest
import torch

import torch_ref_densify as ref
from street_gaussians_amd import densify

pytestmark = pytest.mark.gpu


def _setup(N, C, M, S, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    params = {"xyz": r(N, 3) * 5, "f_dc": r(N, C, 3), "f_rest": r(N, M - 1, 3), "opacity": r(N, 1) * 3,
              "scaling": r(N, 3) * 1.2 - 3.5, "rotation": r(N, 4), "semantic": r(N, S)}
    states = {k: (r(*v.shape), r(*v.shape).abs()) for k, v in params.items()}
    accum = torch.rand(N, 2, generator=g) * 0.002
    denom = torch.randint(0, 4, (N, 1), generator=g).float()  # zeros -> 0/0 = NaN -> 0 and x/0 = inf -> selected
    return params, states, accum, denom, g


@pytest.mark.parametrize("N,C,M,S,prune_big,col,seed", [(5000, 1, 16, 3, True, 0, 1), (777, 5, 16, 1, False, 1, 2),
                                                      (1, 1, 4, 0, True, 0, 3), (20000, 1, 16, 19, True, 0, 4)])
def test_densify_and_prune_matches_reference_steps(N, C, M, S, prune_big, col, seed):
    params, states, accum, denom, g = _setup(N, C, M, S, seed)
    kw = dict(max_grad=0.0004, min_opacity=0.05, extent=3.0, percent_dense=0.01, percent_big_ws=0.1)
    normals = torch.randn(2 * N, 3, generator=g)
    m = ref.Model(params, states, accum, denom)
    want = m.densify_and_prune(prune_big=prune_big, normals=normals, grad_column=col, **kw)
    n_norm = 2 * want["points_split"]
    dev = lambda t: t.cuda()
    new_p, new_s, got, index = densify.densify_and_prune(
        {k: dev(v) for k, v in params.items()}, dev(accum), dev(denom), prune_big=prune_big, grad_column=col,
        states={k: (dev(a), dev(b)) for k, (a, b) in states.items()}, normals=dev(normals[:n_norm]), **kw)
    assert got == want
    assert want["points_clone"] > 0 or N == 1
    for k in ref.NAMES:
        assert new_p[k].shape == m.p[k].shape, k
        if k in ("xyz", "scaling"):  # split children are computed, not copied
            assert torch.allclose(new_p[k].cpu(), m.p[k], rtol=2e-6, atol=2e-6), k
            keep = (index["kind"] != 2).cpu()
            assert torch.equal(new_p[k].cpu()[keep], m.p[k][keep]), k
        else:
            assert torch.equal(new_p[k].cpu(), m.p[k]), k
        for j in range(2):
            assert torch.equal(new_s[k][j].cpu(), m.s[k][j]), (k, j)
    kinds = index["kind"].cpu()
    assert (kinds[:-1] <= kinds[1:]).all()  # originals, then clones, then split children


def test_densify_without_states_and_random_normals():
    params, states, accum, denom, g = _setup(3000, 1, 16, 2, 7)
    new_p, new_s, scal, index = densify.densify_and_prune({k: v.cuda() for k, v in params.items()}, accum.cuda(), denom.cuda(),
                                                          max_grad=0.0004, min_opacity=0.05, extent=3.0, percent_dense=0.01,
                                                          percent_big_ws=0.1, prune_big=False)
    assert new_s is None and new_p["xyz"].shape[0] == index["src"].shape[0]
    assert torch.isfinite(new_p["xyz"]).all() and scal["points_split"] > 0


def _check(new_p, new_s, got, index, m, want, keys):
    for k in keys:
        assert got[k] == want[k], (k, got[k], want[k])
    for k in ref.NAMES:
        assert new_p[k].shape == m.p[k].shape, k
        if k in ("xyz", "scaling"):  # split children are computed, not copied (R (z * s) + mu with s up to ~50 here)
            assert torch.allclose(new_p[k].cpu(), m.p[k], rtol=2e-5, atol=2e-5), k
            keep = (index["kind"] != 2).cpu()
            assert torch.equal(new_p[k].cpu()[keep], m.p[k][keep]), k
        else:
            assert torch.equal(new_p[k].cpu(), m.p[k]), k
        for j in range(2):
            assert torch.equal(new_s[k][j].cpu(), m.s[k][j]), (k, j)


KW = dict(max_grad=0.0004, min_opacity=0.05, extent=3.0, percent_dense=0.01, percent_big_ws=0.1)


@pytest.mark.parametrize("N,S,col,seed", [(6000, 3, 0, 11), (50000, 19, 1, 12), (3, 0, 0, 13)])
def test_background_model_prune_rule(N, S, col, seed):
    """GaussianModelBkgd.densify_and_prune (gaussian_model_bkgd.py:74-114): big points far from the sphere are exempt."""
    params, states, accum, denom, g = _setup(N, 1, 16, S, seed)
    params["scaling"][::7] += 3.0
    normals = torch.randn(2 * N, 3, generator=g)
    center, radius = torch.tensor([0.5, -1.0, 0.3]), 2.5
    m = ref.Model(params, states, accum, denom)
    want = m.densify_and_prune(prune_big=True, normals=normals, grad_column=col, variant="bkgd", sphere_center=center,
                               sphere_radius=radius, **KW)
    dev = lambda t: t.cuda()
    new_p, new_s, got, index = densify.densify_and_prune(
        {k: dev(v) for k, v in params.items()}, dev(accum), dev(denom), prune_big=True, grad_column=col,
        states={k: (dev(a), dev(b)) for k, (a, b) in states.items()}, normals=dev(normals[:2 * want["points_split"]]),
        variant="bkgd", sphere_center=center, sphere_radius=radius, **KW)
    _check(new_p, new_s, got, index, m, want, ["points_total", "points_clone", "points_split", "points_below_min_opacity",
                                              "points_big_ws", "points_pruned"])
    if N > 100:
        assert want["points_big_ws"] > 0


@pytest.mark.parametrize("N,prune_big,seed", [(4000, True, 21), (4000, False, 22), (30000, True, 23)])
def test_actor_model_prune_rule(N, prune_big, seed):
    """GaussianModelActor.densify_and_prune (gaussian_model_actor.py:204-261): sampled extent against the tracking box."""
    params, states, accum, denom, g = _setup(N, 3, 16, 1, seed)
    params["xyz"] *= 0.3
    normals = torch.randn(2 * N, 3, generator=g)
    lo, hi = torch.tensor([-1.2, -0.9, -1.5]), torch.tensor([1.1, 1.0, 1.4])
    probe = ref.Model(params, states, accum, denom)
    probe.densify_and_prune(prune_big=False, normals=normals, **dict(KW, min_opacity=-1.0))
    n_cand = probe.p["xyz"].shape[0]
    zb = torch.randn(n_cand, 2, 3, generator=g)
    m = ref.Model(params, states, accum, denom)
    want = m.densify_and_prune(prune_big=prune_big, normals=normals, variant="actor", box_min=lo, box_max=hi,
                               box_normals=zb, **KW)
    dev = lambda t: t.cuda()
    new_p, new_s, got, index = densify.densify_and_prune(
        {k: dev(v) for k, v in params.items()}, dev(accum), dev(denom), prune_big=prune_big,
        states={k: (dev(a), dev(b)) for k, (a, b) in states.items()}, normals=dev(normals[:2 * want["points_split"]]),
        variant="actor", box_min=lo, box_max=hi, box_normals=dev(zb), **KW)
    _check(new_p, new_s, got, index, m, want, ["points_total", "points_clone", "points_split", "points_pruned"])


def test_reset_opacity_matches_reference():
    g = torch.Generator().manual_seed(3)
    op = torch.randn(10000, 1, generator=g) * 4
    a, b = torch.randn(10000, 1, generator=g), torch.rand(10000, 1, generator=g)
    m = ref.Model({k: (op if k == "opacity" else torch.zeros(10000, 1)) for k in ref.NAMES},
                  {k: (a, b) for k in ref.NAMES}, torch.zeros(10000, 2), torch.zeros(10000, 1))
    m.reset_opacity()
    ag, bg = a.cuda().clone(), b.cuda().clone()
    new = densify.reset_opacity(op.cuda(), (ag, bg))
    assert torch.allclose(new.cpu(), m.p["opacity"], rtol=2e-6, atol=2e-6)
    assert float(ag.abs().max()) == 0.0 and float(bg.abs().max()) == 0.0
    assert float(torch.sigmoid(new).max()) <= 0.01 * (1 + 1e-5)
