"""CPU tests of the reference's multi-element PLY scene format (street_gaussians_amd/plyio.py)."""
import os
import struct
from collections import OrderedDict

import numpy as np
import pytest

from street_gaussians_amd import plyio


def _model(n, C, M, S, seed):
    r = np.random.default_rng(seed)
    return dict(xyz=r.normal(size=(n, 3)).astype(np.float32), features_dc=r.normal(size=(n, C, 3)).astype(np.float32),
                features_rest=r.normal(size=(n, M - 1, 3)).astype(np.float32), opacity=r.normal(size=(n, 1)).astype(np.float32),
                scaling=r.normal(size=(n, 3)).astype(np.float32), rotation=r.normal(size=(n, 4)).astype(np.float32),
                semantic=r.normal(size=(n, S)).astype(np.float32))


def test_round_trip_and_layout(tmp_path):
    models = OrderedDict([("background", _model(11, 1, 16, 5, 0)), ("obj_003", _model(4, 5, 16, 1, 1)),
                          ("obj_017", _model(0, 5, 16, 1, 2))])
    path = str(tmp_path / "scene.ply")
    plyio.write_scene_ply(path, models)
    back = plyio.read_scene_ply(path)
    assert list(back) == ["background", "obj_003", "obj_017"]
    for name, m in models.items():
        for k in plyio.FIELDS:
            assert back[name][k].shape == m[k].shape, (name, k)
            assert (back[name][k] == m[k]).all(), (name, k)
    # header and byte layout as plyfile writes them for the reference (gaussian_model.py:80-101, 327-341)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n")].decode().split("\n")
    assert head[:3] == ["ply", "format binary_little_endian 1.0", "element vertex_background 11"]
    props = [ln.split()[2] for ln in head[3:3 + 6 + 3 + 45 + 1 + 3 + 4 + 5]]
    assert props[:6] == ["x", "y", "z", "nx", "ny", "nz"] and props[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert props[9] == "f_rest_0" and props[54] == "opacity" and props[55:58] == ["scale_0", "scale_1", "scale_2"]
    assert props[58:62] == ["rot_0", "rot_1", "rot_2", "rot_3"] and props[62] == "semantic_0"
    body = raw[raw.index(b"end_header\n") + 11:]
    first = struct.unpack("<67f", body[:67 * 4])
    bk = models["background"]
    assert first[:3] == tuple(bk["xyz"][0]) and first[3:6] == (0.0, 0.0, 0.0)
    # channel-major features: f_rest_{c*15 + k} = features_rest[:, k, c]
    assert first[9 + 1 * 15 + 2] == bk["features_rest"][0, 2, 1]
    assert first[54] == bk["opacity"][0, 0]


def test_reads_single_model_files_and_other_scalar_types(tmp_path):
    # a file as GaussianModel.save_ply writes it: one element called "vertex"; with an extra uchar column in front
    n = 3
    m = _model(n, 1, 4, 2, 5)
    cols = plyio._columns(m)
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {n}", "property uchar flag"]
    head += [f"property float {c}" for c, _ in cols] + ["end_header"]
    rec = np.empty(n, dtype=[("flag", "u1")] + [(c, "<f4") for c, _ in cols])
    rec["flag"] = 7
    for c, v in cols:
        rec[c] = v
    p = tmp_path / "single.ply"
    p.write_bytes(("\n".join(head) + "\n").encode() + rec.tobytes())
    back = plyio.read_scene_ply(str(p))
    assert list(back) == [""] and (back[""]["features_rest"] == m["features_rest"]).all()
    assert (back[""]["rotation"] == m["rotation"]).all() and back[""]["semantic"].shape == (n, 2)


# ---- fixtures written in the reference's layout by the reference's own make_ply / state_dict
# (tests/golden/make_scene_fixture.py) --------------------------------------------------------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene")


def _raw():
    z = np.load(os.path.join(GOLD, "scene_ref_params.npz"))
    out = {}
    for key in z.files:
        name, k = key.split("/")
        out.setdefault(name, {})[k] = z[key]
    return out


def test_reads_the_layout_the_reference_writes():
    models = plyio.read_scene_ply(os.path.join(GOLD, "scene_ref_layout.ply"))
    raw = _raw()
    assert list(models) == ["background", "obj_001"]
    names = {"xyz": "_xyz", "features_dc": "_features_dc", "features_rest": "_features_rest", "opacity": "_opacity",
             "scaling": "_scaling", "rotation": "_rotation", "semantic": "_semantic"}
    for name, m in models.items():
        for ours, theirs in names.items():
            assert m[ours].shape == raw[name][theirs].shape, (name, ours)
            assert np.array_equal(m[ours], raw[name][theirs]), (name, ours)
    # ... and writes it back byte for byte
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "again.ply")
        plyio.write_scene_ply(out, models)
        assert open(out, "rb").read() == open(os.path.join(GOLD, "scene_ref_layout.ply"), "rb").read()


def test_checkpoint_layout_of_the_reference():
    import torch
    from street_gaussians_amd import checkpoint
    path = os.path.join(GOLD, "scene_ref_state.pth")
    models = checkpoint.load(path)
    raw = _raw()
    assert list(models) == ["background", "obj_001"]
    for name, m in models.items():
        for theirs, ours in checkpoint.PARAM_KEYS.items():
            src = {"xyz": "_xyz", "feature_dc": "_features_dc", "feature_rest": "_features_rest", "scaling": "_scaling",
                   "rotation": "_rotation", "opacity": "_opacity", "semantic": "_semantic"}[theirs]
            assert np.array_equal(m[ours].numpy(), raw[name][src]), (name, ours)
    # the PLY and the checkpoint describe the same scene
    ply = plyio.read_scene_ply(os.path.join(GOLD, "scene_ref_layout.ply"))
    for name in models:
        for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "semantic"):
            assert np.array_equal(ply[name][k], models[name][k].numpy()), (name, k)
    # round trip through the reference's dictionary layout (train.py:218-223), extras kept when not final
    st = torch.load(path, map_location="cpu", weights_only=False)
    assert st["iter"] == 30000
    models["background"]["denom"] = torch.ones(700, 1)
    again = checkpoint.state_from_models(models, is_final=False, iteration=7)
    assert again["iter"] == 7 and torch.equal(again["background"]["denom"], torch.ones(700, 1))
    assert set(again["obj_001"]) == set(st["obj_001"])
    for k in st["obj_001"]:
        assert torch.equal(again["obj_001"][k], st["obj_001"][k]), k
    final = checkpoint.state_from_models(models, is_final=True)
    assert "denom" not in final["background"]


@pytest.mark.skipif(not os.path.exists("/root/reference/lib/models/gaussian_model.py"), reason="reference checkout not present")
def test_saved_checkpoint_is_loadable_by_the_reference_load_state_dict(tmp_path):
    """checkpoint.save writes nn.Parameter entries (requires_grad) like the reference's state_dict does: the reference's
    OWN GaussianModel.load_state_dict (gaussian_model.py:157-180), cut out of its source and executed on a stub in train
    mode, assigns them as the model's parameters and builds an optimiser over them."""
    import re
    import types
    import torch
    from street_gaussians_amd import checkpoint
    models = checkpoint.load(os.path.join(GOLD, "scene_ref_state.pth"))
    out = str(tmp_path / "ours.pth")
    checkpoint.save(out, models, is_final=True, iteration=12)
    st = torch.load(out, weights_only=False)  # as train.py / render.py read it: no map_location
    src = open("/root/reference/lib/models/gaussian_model.py").read()
    m = re.search(r"^    def load_state_dict\(self.*?(?=^    def |\Z)", src, re.S | re.M).group(0)
    body = "\n".join(ln[4:] if ln.startswith("    ") else ln for ln in m.split("\n"))
    ns = {"cfg": types.SimpleNamespace(mode="train"), "torch": torch}
    exec(body, ns)

    class Stub:
        def training_setup(self):  # gaussian_model.py:258-286: an Adam over the seven parameters
            self.optimizer = torch.optim.Adam([{"params": [p], "name": n} for n, p in (
                ("xyz", self._xyz), ("f_dc", self._features_dc), ("f_rest", self._features_rest), ("opacity", self._opacity),
                ("scaling", self._scaling), ("rotation", self._rotation), ("semantic", self._semantic))], lr=1e-3)

    for name in ("background", "obj_001"):
        stub = Stub()
        ns["load_state_dict"](stub, st[name])
        for attr, ours in (("_xyz", "xyz"), ("_features_dc", "features_dc"), ("_features_rest", "features_rest"),
                           ("_scaling", "scaling"), ("_rotation", "rotation"), ("_opacity", "opacity"), ("_semantic", "semantic")):
            p = getattr(stub, attr)
            assert isinstance(p, torch.nn.Parameter) and p.requires_grad and p.is_leaf
            assert torch.equal(p.detach(), models[name][ours].float())
        # the optimiser training_setup builds can take a step on them
        (stub._xyz.sum() + stub._opacity.sum()).backward()
        stub.optimizer.step()
    assert st["iter"] == 12


@pytest.mark.skipif(not os.path.exists("/root/reference/make_ply.py"), reason="reference checkout not present")
def test_viewer_ply_matches_the_reference_make_ply_script(tmp_path):
    """plyio.write_viewer_ply against the reference's make_ply.py:37-72 (the block that turns a composed frame into the
    structured array of the single `vertex` element), cut out of the script by line content and executed on a stub model."""
    import re
    import textwrap
    import types
    import torch
    src = open("/root/reference/make_ply.py").read()
    a = src.index("    xyz = gaussians.get_xyz.detach().cpu().numpy()")
    b = src.index("    save_dir = os.path.join(cfg.model_path")
    block = textwrap.dedent(src[a:b])
    g = torch.Generator().manual_seed(4)
    n, M = 500, 16
    stub = types.SimpleNamespace(get_xyz=torch.randn(n, 3, generator=g), get_features=torch.randn(n, M, 3, generator=g),
                                 get_opacity=torch.rand(n, 1, generator=g), get_scaling=torch.rand(n, 3, generator=g) + 0.01,
                                 get_rotation=torch.nn.functional.normalize(torch.randn(n, 4, generator=g)))
    stub.get_opacity[:3] = torch.tensor([[0.0], [1.0], [0.5]])  # the clip is active at both ends
    ns = {"np": np, "gaussians": stub, "inverse_opacity": lambda x: np.log(x / (1 - x)), "inverse_scale": lambda x: np.log(x)}
    exec(block, ns)
    want = ns["elements"]  # structured array, f4 columns in the script's order
    out = str(tmp_path / "viewer.ply")
    plyio.write_viewer_ply(out, stub.get_xyz.numpy(), stub.get_features.numpy(), stub.get_opacity.numpy(),
                           stub.get_scaling.numpy(), stub.get_rotation.numpy())
    data = open(out, "rb").read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    header = data[:end].decode().split("\n")
    assert header[2] == f"element vertex {n}"
    assert [ln.split()[2] for ln in header if ln.startswith("property")] == list(want.dtype.names)
    got = np.frombuffer(data, dtype=want.dtype, count=n, offset=end)
    for c in want.dtype.names:
        assert np.array_equal(got[c], want[c]), c
    # and it reads back as a single-model scene
    m = plyio.read_scene_ply(out)[""]
    assert np.allclose(m["xyz"], stub.get_xyz.numpy()) and m["features_rest"].shape == (n, M - 1, 3)


def test_checkpoint_save_drops_the_size_ladders_slack(tmp_path):
    """Tensors backed by ladder-sized storages (street_gaussians_amd/_alloc.py: the densify step's outputs) must reach the
    file with storages of exactly their size: torch.save writes whole storages."""
    import torch
    from street_gaussians_amd import checkpoint
    big = torch.zeros(1 << 19, dtype=torch.float32)          # 2 MiB storage
    view = big[:300_000].view(100_000, 3)                    # 1.2 MB of it: what _alloc.empty hands out
    assert view.untyped_storage().nbytes() > view.numel() * 4
    out = checkpoint.exact_storage({"a": view, "b": [view, 3], "c": torch.ones(4)})
    assert out["a"].untyped_storage().nbytes() == view.numel() * 4 and torch.equal(out["a"], view)
    assert out["b"][0].untyped_storage().nbytes() == view.numel() * 4 and out["b"][1] == 3
    assert out["c"].untyped_storage().nbytes() == 16
    p = tmp_path / "t.pth"
    torch.save(out, p)
    assert p.stat().st_size < 2 * view.numel() * 4 + 20_000 + view.numel() * 4  # three exact copies at most, no 2 MiB storages
