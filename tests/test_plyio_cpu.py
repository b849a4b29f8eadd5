"""CPU tests of the reference's multi-element PLY scene format (street_gaussians_amd/plyio.py)."""
import struct
from collections import OrderedDict

import numpy as np

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
