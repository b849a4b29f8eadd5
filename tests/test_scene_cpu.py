"""CPU tests for the scene-graph rows: the torch restatement (tests/torch_ref_scene.py) is pinned against the
reference's own quaternion helpers when /root/reference is present, its analytic structure is checked with float64
finite differences, and the C ABI of include/sgr_scene.h must be exported by the built library."""
import ctypes as C
import importlib.util
import os
import re

import pytest
import torch

import torch_ref_scene as ref

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GU = "/root/reference/lib/utils/general_utils.py"


def _reference_fn(name):
    """Extracts ONE pure function from the reference's general_utils.py (the module itself imports CUDA-only code)."""
    src = open(GU).read()
    m = re.search(rf"^def {name}\(.*?(?=^def |\Z)", src, re.S | re.M)
    ns = {"torch": torch, "F": torch.nn.functional}
    helper = re.search(r"^def _sqrt_positive_part\(.*?(?=^def |\Z)", src, re.S | re.M)
    if helper:
        exec(helper.group(0), ns)
    exec(m.group(0).replace("device='cuda'", "device=r.device"), ns)
    return ns[name]


@pytest.mark.skipif(not os.path.exists(GU), reason="reference checkout not present on this machine")
def test_restated_quaternion_helpers_match_the_reference():
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(50, 4, generator=g), torch.randn(50, 4, generator=g)
    assert torch.equal(ref.quaternion_raw_multiply(a, b), _reference_fn("quaternion_raw_multiply")(a, b))
    assert torch.equal(ref.quaternion_to_matrix(a), _reference_fn("quaternion_to_matrix")(a))
    # the flip quaternion the reference derives (street_gaussian_model.py:58-61) is what Segment defaults to
    flip = torch.eye(3) * -1
    flip[1, 1] = 1
    q = _reference_fn("matrix_to_quaternion")(flip.unsqueeze(0))[0]
    assert torch.allclose(q, torch.tensor([0.0, 0.0, 1.0, 0.0]))


def _segments(dtype=torch.float64, S=3, M=4, seed=1):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=dtype)
    bk = dict(xyz=r(7, 3), rotation=r(7, 4), scaling=r(7, 3) * 0.3, opacity=r(7, 1), features_dc=r(7, 1, 3),
              features_rest=r(7, M - 1, 3), semantic=r(7, S), semantic_mode="probabilities")
    a1 = dict(xyz=r(5, 3), rotation=r(5, 4), scaling=r(5, 3) * 0.3, opacity=r(5, 1), features_dc=r(5, 3, 3),
              features_rest=r(5, M - 1, 3), semantic=r(5, 1), pose=r(7), idft=r(3), class_label=2,
              flip_mask=torch.tensor([True, False, True, False, False]), semantic_mode="logits")
    a2 = dict(xyz=r(3, 3), rotation=r(3, 4), scaling=r(3, 3) * 0.3, opacity=r(3, 1), features_dc=r(3, 1, 3),
              features_rest=r(3, M - 1, 3), semantic=r(3, 1), pose=r(7), class_label=0, semantic_mode="probabilities")
    return [bk, a1, a2]


def test_restatement_shapes_and_gradcheck():
    segs = _segments()
    outs = ref.compose(segs, 4, 3)
    assert [tuple(o.shape) for o in outs] == [(15, 3), (15, 4), (15, 3), (15, 1), (15, 4, 3), (15, 3)]
    assert torch.allclose(outs[1].norm(dim=1), torch.ones(15, dtype=torch.float64))
    assert torch.allclose(outs[5][:7].sum(1), torch.ones(7, dtype=torch.float64))        # softmax rows
    assert (outs[5][7:12, :2] == 0).all() and (outs[5][12:, 1:] == 0).all()              # actors: one column each
    names = ["xyz", "rotation", "scaling", "opacity", "features_dc", "features_rest", "semantic", "pose"]
    leaves = []
    for s in segs:
        for n in names:
            if s.get(n) is not None:
                s[n] = s[n].clone().requires_grad_(True)
                leaves.append(s[n])

    def fn(*ts):
        it = iter(ts)
        cur = []
        for s in segs:
            d = dict(s)
            for n in names:
                if s.get(n) is not None:
                    d[n] = next(it)
            cur.append(d)
        return ref.compose(cur, 4, 3)

    assert torch.autograd.gradcheck(fn, leaves, eps=1e-6, atol=1e-5)


def test_scene_abi_symbols_are_exported():
    from street_gaussians_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip("libsgr_hip.so not built")
    L = C.CDLL(_native.LIB_PATH)
    decl = open(os.path.join(ROOT, "include", "sgr_scene.h")).read()
    names = re.findall(r"^int (sgr_scene_\w+)\(", decl, re.M)
    assert len(names) == 3
    for n in names:
        assert hasattr(L, n), n


# ---- pin of the flattening itself: the reference's OWN property getters, cut out of its source and executed on stub
# objects (street_gaussian_model.py:287-449, gaussian_model.py:224-251, gaussian_model_actor.py:55-80) ----------------
SGM = "/root/reference/lib/models/street_gaussian_model.py"
GMP = "/root/reference/lib/models/gaussian_model.py"
GMA = "/root/reference/lib/models/gaussian_model_actor.py"


def _getters(path, names, ns, prop=True):
    src = open(path).read()
    out = {}
    for name in names:
        pat = rf"^    @property\n    def {name}\(self.*?(?=^    def |^    @|\Z)" if prop else rf"^    def {name}\(self.*?(?=^    def |^    @|\Z)"
        m = re.search(pat, src, re.S | re.M)
        assert m, (path, name)
        body = "\n".join(ln[4:] if ln.startswith("    ") else ln for ln in m.group(0).split("\n"))
        body = body.replace("@property\n", "").replace(".cuda()", "")
        exec(body, ns)
        out[name] = property(ns[name]) if prop else ns[name]
    return out


@pytest.mark.skipif(not os.path.exists(SGM), reason="reference checkout not present on this machine")
@pytest.mark.parametrize("train,sem_mode", [(True, "logits"), (False, "probabilities")])
def test_restated_flattening_matches_the_reference_getters(train, sem_mode):
    import types
    S, M = 4, 16
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g)
    cfg = types.SimpleNamespace(mode="train" if train else "eval")
    idft_rows = {}

    def IDFT(time, dim):  # lib/utils/general_utils.py: only its result is used here; the row is an input of the test
        return idft_rows[dim][None]
    ns = {"torch": torch, "cfg": cfg, "quaternion_raw_multiply": _reference_fn("quaternion_raw_multiply"),
          "quaternion_to_matrix": _reference_fn("quaternion_to_matrix"), "IDFT": IDFT,
          "GaussianModel": object, "GaussianModelActor": object}

    Base = type("Base", (), _getters(GMP, ["get_scaling", "get_rotation", "get_xyz", "get_features", "get_semantic", "get_opacity"], ns))
    actor_ns = dict(ns)
    Actor = type("Actor", (Base,), {**_getters(GMA, ["get_semantic"], actor_ns),
                                    **_getters(GMA, ["get_features_fourier"], actor_ns, prop=False)})
    Street = type("Street", (), _getters(SGM, ["get_scaling", "get_rotation", "get_xyz", "get_features", "get_semantic", "get_opacity"], dict(ns)))

    def model(cls, n, C, sem_cols):
        m = cls()
        m._xyz, m._rotation, m._scaling, m._opacity = r(n, 3), r(n, 4), r(n, 3) * 0.4, r(n, 1)
        m._features_dc, m._features_rest, m._semantic = r(n, C, 3), r(n, M - 1, 3), r(n, sem_cols)
        m.scaling_activation, m.opacity_activation = torch.exp, torch.sigmoid
        m.rotation_activation = torch.nn.functional.normalize
        m.semantic_mode = sem_mode
        return m

    bk = model(Base, 40, 1, S)
    actors = []
    for i, (n, C, label) in enumerate([(17, 3, 2), (9, 1, 0)]):
        a = model(Actor, n, C, 1)
        a.num_classes_global, a.obj_class_label = S, label
        a.start_frame, a.end_frame, a.fourier_scale, a.fourier_dim = 0, 10, 1.0, C
        idft_rows[C] = r(C)
        actors.append(a)
    st = Street()
    st.background = bk
    st.get_visibility = lambda name: True
    st.graph_obj_list = ["obj_0", "obj_1"]
    st.obj_0, st.obj_1 = actors
    st.use_pose_correction = False
    st.frame = 3
    poses = [r(7), r(7)]
    st.obj_rots = torch.cat([p[:4].unsqueeze(0).expand(a._xyz.shape[0], -1) for p, a in zip(poses, actors)], 0)
    st.obj_trans = torch.cat([p[4:].unsqueeze(0).expand(a._xyz.shape[0], -1) for p, a in zip(poses, actors)], 0)
    masks = [torch.rand(a._xyz.shape[0], generator=g) < 0.4 for a in actors]
    st.flip_mask = torch.cat(masks, 0)
    st.flip_axis = 1
    st.flip_matrix = torch.tensor([[0.0, 0.0, 1.0, 0.0]])  # matrix_to_quaternion of diag(-1, 1, -1), checked above

    seg = lambda m: dict(xyz=m._xyz, rotation=m._rotation, scaling=m._scaling, opacity=m._opacity,
                         features_dc=m._features_dc, features_rest=m._features_rest, semantic=m._semantic,
                         semantic_mode=sem_mode)
    segs = [seg(bk)]
    for a, p, fm in zip(actors, poses, masks):
        d = seg(a)
        d.update(pose=p, idft=idft_rows[a.fourier_dim], class_label=a.obj_class_label, flip_mask=fm if train else None)
        segs.append(d)
    xyz, rot, scale, opac, feats, sem = ref.compose(segs, M, S)
    assert torch.equal(xyz, st.get_xyz)
    assert torch.equal(rot, st.get_rotation)
    assert torch.equal(scale, st.get_scaling)
    assert torch.equal(opac, st.get_opacity)
    assert torch.equal(feats, st.get_features)
    assert torch.equal(sem, st.get_semantic)
