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
