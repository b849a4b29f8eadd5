"""CPU pin of tests/torch_ref_densify.py (the oracle of the density-control row) against the reference's
OWN methods: densify_and_clone / densify_and_split / densification_postfix / cat_optimizer / prune_optimizer /
prune_points / reset_optimizer / reset_opacity of GaussianModel and the densify_and_prune overrides of
GaussianModelBkgd and GaussianModelActor are cut out of the reference's source files and executed in place on a stub
object that carries a real torch.optim.Adam.  Only three textual substitutions are made so that the code runs here:
device="cuda" -> "cpu", .cuda() dropped, and torch.normal(mean, std) routed to pre-drawn standard normals (the only way
to compare two implementations of a sampling step)."""
import os
import re
import types

import pytest
import torch
from torch import nn

import torch_ref_densify as ref

REF = "/root/reference/lib/models"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present on this machine")

NAMES = ref.NAMES


def _method_src(path, cls_hint, name):
    src = open(path).read()
    m = re.search(rf"^    def {name}\(self.*?(?=^    def |^    @|^class |\Z)", src, re.S | re.M)
    assert m, (path, name)
    body = m.group(0)
    body = body.replace('device="cuda"', 'device="cpu"').replace(".cuda()", "")
    return "\n".join(ln[4:] if ln.startswith("    ") else ln for ln in body.split("\n"))


class _Normals:
    """torch.normal(mean=..., std=...) replacement: mean + std * z with z taken from a queue of pre-drawn tensors."""

    def __init__(self):
        self.queue = []

    def __call__(self, mean=None, std=None):
        z = self.queue.pop(0)
        assert z.shape == std.shape, (z.shape, std.shape)
        return mean + std * z.to(std.dtype)


def _reference_class(kind):
    normal = _Normals()
    tproxy = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    tproxy.normal = normal

    def quaternion_to_matrix(q):  # lib/utils/general_utils.py:125-146 (executed from the reference's source below)
        return ns_q["quaternion_to_matrix"](q)
    ns_q = {"torch": torch, "F": torch.nn.functional}
    gu = open("/root/reference/lib/utils/general_utils.py").read()
    exec(re.search(r"^def quaternion_to_matrix\(.*?(?=^def |\Z)", gu, re.S | re.M).group(0).replace("device='cuda'", "device='cpu'"), ns_q)
    exec(re.search(r"^def inverse_sigmoid\(.*?(?=^def |\Z)", gu, re.S | re.M).group(0), ns_q)
    cfg = types.SimpleNamespace(optim={})
    ns = {"torch": tproxy, "nn": nn, "quaternion_to_matrix": quaternion_to_matrix, "inverse_sigmoid": ns_q["inverse_sigmoid"],
          "cfg": cfg}
    base = os.path.join(REF, "gaussian_model.py")
    methods = {}
    for name in ("prune_optimizer", "cat_optimizer", "reset_optimizer", "densification_postfix", "densify_and_split",
                 "densify_and_clone", "prune_points", "reset_opacity", "densify_and_prune"):
        exec(_method_src(base, "GaussianModel", name), ns)
        methods[name] = ns[name]
    if kind == "bkgd":
        exec(_method_src(os.path.join(REF, "gaussian_model_bkgd.py"), "GaussianModelBkgd", "densify_and_prune"), ns)
        methods["densify_and_prune"] = ns["densify_and_prune"]
    if kind == "actor":
        exec(_method_src(os.path.join(REF, "gaussian_model_actor.py"), "GaussianModelActor", "densify_and_prune"), ns)
        methods["densify_and_prune"] = ns["densify_and_prune"]

    class Stub:
        get_xyz = property(lambda self: self._xyz)
        get_scaling = property(lambda self: torch.exp(self._scaling))
        get_opacity = property(lambda self: torch.sigmoid(self._opacity))
        get_rotation = property(lambda self: torch.nn.functional.normalize(self._rotation))
        scaling_inverse_activation = staticmethod(torch.log)

    for k, f in methods.items():
        setattr(Stub, k, f)
    return Stub, normal, cfg


def _setup(N, C, M, S, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    params = {"xyz": r(N, 3) * 5, "f_dc": r(N, C, 3), "f_rest": r(N, M - 1, 3), "opacity": r(N, 1) * 3,
              "scaling": r(N, 3) * 1.2 - 3.5, "rotation": r(N, 4), "semantic": r(N, S)}
    states = {k: (r(*v.shape), r(*v.shape).abs()) for k, v in params.items()}
    accum = torch.rand(N, 2, generator=g) * 0.002
    denom = torch.randint(0, 4, (N, 1), generator=g).float()
    return params, states, accum, denom, g


def _make_stub(Stub, params, states, accum, denom, percent_dense, percent_big_ws):
    m = Stub()
    attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
            "scaling": "_scaling", "rotation": "_rotation", "semantic": "_semantic"}
    groups = []
    for k in NAMES:
        p = nn.Parameter(params[k].clone().requires_grad_(True))
        setattr(m, attr[k], p)
        groups.append({"params": [p], "lr": 1e-3, "name": k})
    m.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    for gdict, k in zip(m.optimizer.param_groups, NAMES):
        m.optimizer.state[gdict["params"][0]] = {"step": torch.tensor(3.0), "exp_avg": states[k][0].clone(),
                                                 "exp_avg_sq": states[k][1].clone()}
    m.xyz_gradient_accum, m.denom = accum.clone(), denom.clone()
    m.max_radii2D = torch.zeros(accum.shape[0])
    m.percent_dense, m.percent_big_ws = percent_dense, percent_big_ws
    m.scalar_dict, m.tensor_dict = {}, {}
    m._attr = attr
    return m


def _compare(m, mine, scal_ref, scal_mine, keys):
    for k in keys:
        assert scal_ref[k] == scal_mine[k], (k, scal_ref[k], scal_mine[k])
    for gdict, k in zip(m.optimizer.param_groups, NAMES):
        p = gdict["params"][0]
        assert torch.equal(p.detach(), mine.p[k]), k
        st = m.optimizer.state[p]
        assert torch.equal(st["exp_avg"], mine.s[k][0]) and torch.equal(st["exp_avg_sq"], mine.s[k][1]), k


KW = dict(max_grad=0.0004, min_opacity=0.05, extent=3.0, percent_dense=0.01, percent_big_ws=0.1)


@pytest.mark.parametrize("prune_big,seed", [(True, 1), (False, 2)])
def test_base_model_restatement_matches_the_reference_methods(prune_big, seed):
    Stub, normal, _ = _reference_class("base")
    params, states, accum, denom, g = _setup(1500, 1, 16, 3, seed)
    z = torch.randn(2 * 1500, 3, generator=g)
    m = _make_stub(Stub, params, states, accum, denom, KW["percent_dense"], KW["percent_big_ws"])
    mine = ref.Model(params, states, accum, denom)
    want = mine.densify_and_prune(prune_big=prune_big, normals=z, **KW)
    normal.queue = [z[:2 * want["points_split"]]]
    scal, _ = m.densify_and_prune(KW["max_grad"], KW["min_opacity"], KW["extent"], 20 if prune_big else None)
    _compare(m, mine, scal, want, ["points_total", "points_clone", "points_split", "points_pruned"])
    # reset_opacity on the result
    m.reset_opacity()
    mine.reset_opacity()
    _compare(m, mine, {}, {}, [])


def test_background_model_restatement_matches_the_reference_methods():
    Stub, normal, cfg = _reference_class("bkgd")
    params, states, accum, denom, g = _setup(2000, 1, 16, 2, 5)
    params["scaling"][::7] += 3.0  # some big points, near and far
    z = torch.randn(2 * 2000, 3, generator=g)
    center, radius = torch.tensor([0.5, -1.0, 0.3]), 2.5
    m = _make_stub(Stub, params, states, accum, denom, KW["percent_dense"], KW["percent_big_ws"])
    m.scene_radius = torch.tensor([KW["extent"]])
    m.sphere_center, m.sphere_radius = center, torch.tensor([radius])
    mine = ref.Model(params, states, accum, denom)
    want = mine.densify_and_prune(prune_big=True, normals=z, variant="bkgd", sphere_center=center, sphere_radius=radius, **KW)
    normal.queue = [z[:2 * want["points_split"]]]
    scal, _ = m.densify_and_prune(KW["max_grad"], KW["min_opacity"], True)
    assert want["points_big_ws"] > 0 and want["points_below_min_opacity"] > 0
    _compare(m, mine, scal, want, ["points_total", "points_clone", "points_split", "points_below_min_opacity",
                                  "points_big_ws", "points_pruned"])
    # the "abs" gradient column (cfg.optim.densify_grad_abs_bkgd, gaussian_model_bkgd.py:76-79)
    cfg.optim = {"densify_grad_abs_bkgd": True}
    m2 = _make_stub(Stub, params, states, accum, denom, KW["percent_dense"], KW["percent_big_ws"])
    m2.scene_radius, m2.sphere_center, m2.sphere_radius = m.scene_radius, center, torch.tensor([radius])
    mine2 = ref.Model(params, states, accum, denom)
    want2 = mine2.densify_and_prune(prune_big=True, normals=z, variant="bkgd", sphere_center=center, sphere_radius=radius,
                                    grad_column=1, **KW)
    normal.queue = [z[:2 * want2["points_split"]]]
    scal2, _ = m2.densify_and_prune(KW["max_grad"], KW["min_opacity"], True)
    _compare(m2, mine2, scal2, want2, ["points_clone", "points_split", "points_pruned"])


def test_actor_model_restatement_matches_the_reference_methods():
    Stub, normal, cfg = _reference_class("actor")
    params, states, accum, denom, g = _setup(1200, 3, 16, 1, 9)
    params["xyz"] *= 0.3
    z = torch.randn(2 * 1200, 3, generator=g)
    m = _make_stub(Stub, params, states, accum, denom, KW["percent_dense"], KW["percent_big_ws"])
    m.random_initialization, m.deformable, m.extent = False, False, torch.tensor([KW["extent"]])
    m.min_xyz, m.max_xyz = torch.tensor([-1.2, -0.9, -1.5]), torch.tensor([1.1, 1.0, 1.4])
    mine = ref.Model(params, states, accum, denom)
    # candidates after clone + split, before pruning = rows the box samples are drawn for
    probe = ref.Model(params, states, accum, denom)
    n_cand = params["xyz"].shape[0]
    s0 = probe.densify_and_prune(prune_big=False, normals=z, **dict(KW, min_opacity=-1.0))  # nothing pruned
    n_cand = probe.p["xyz"].shape[0]
    zb = torch.randn(n_cand, 2, 3, generator=g)
    want = mine.densify_and_prune(prune_big=True, normals=z, variant="actor", box_min=m.min_xyz, box_max=m.max_xyz,
                                  box_normals=zb, **KW)
    normal.queue = [z[:2 * want["points_split"]], zb]
    m.densify_and_prune(KW["max_grad"], KW["min_opacity"], True)
    assert s0["points_split"] == want["points_split"] and want["points_pruned"] > 0
    _compare(m, mine, {}, {}, [])
