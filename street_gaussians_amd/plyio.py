"""The reference's on-disk scene format: a binary little-endian PLY with ONE ELEMENT PER
SUB-MODEL, named ``vertex_<model_name>`` (/root/reference/lib/models/street_gaussian_model.py:94-117), whose float32
columns are ``x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_* semantic_*``
(lib/models/gaussian_model.py:80-95, 327-341).  Feature columns are channel-major: ``f_dc_{c*C + k}`` holds
``features_dc[:, k, c]`` (the reference flattens ``features.transpose(1, 2)``, :83-84, and undoes it on load, :126-127).

The reference goes through the ``plyfile`` package; this is a dependency-free reader / writer of exactly that layout
(numpy structured arrays), so that scenes trained with the reference can be loaded into this repository's tools and
written back.  Host-side I/O: nothing here touches the GPU."""
from __future__ import annotations

import re
from collections import OrderedDict
from typing import Dict

import numpy as np

FIELDS = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "semantic")
_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def _columns(m: Dict[str, np.ndarray]):
    """(name, [n] float32 column) pairs in the reference's order (construct_list_of_attributes)."""
    n = m["xyz"].shape[0]
    dc, rest = np.asarray(m["features_dc"], np.float32), np.asarray(m["features_rest"], np.float32)
    if dc.ndim != 3 or rest.ndim != 3 or dc.shape[2] != 3 or rest.shape[2] != 3:
        raise ValueError("features_dc / features_rest must have dimensions (n, coefficients, 3)")
    cols = [(k, np.asarray(m["xyz"], np.float32)[:, i]) for i, k in enumerate("xyz")]
    cols += [(k, np.zeros(n, np.float32)) for k in ("nx", "ny", "nz")]
    for name, f in (("f_dc", dc), ("f_rest", rest)):
        flat = np.ascontiguousarray(f.transpose(0, 2, 1)).reshape(n, 3 * f.shape[1])  # channel-major
        cols += [(f"{name}_{i}", flat[:, i]) for i in range(flat.shape[1])]
    cols.append(("opacity", np.asarray(m["opacity"], np.float32).reshape(n)))
    for name, key in (("scale", "scaling"), ("rot", "rotation"), ("semantic", "semantic")):
        a = np.asarray(m.get(key, np.zeros((n, 0))), np.float32)
        a = a.reshape(n, a.shape[1] if a.ndim == 2 else 0)
        cols += [(f"{name}_{i}", a[:, i]) for i in range(a.shape[1])]
    return cols


def write_scene_ply(path: str, models: "OrderedDict[str, Dict[str, np.ndarray]]") -> None:
    """models: name -> {xyz [n,3], features_dc [n,C,3], features_rest [n,M-1,3], opacity [n,1], scaling [n,3],
    rotation [n,4], semantic [n,S]} (raw parameters, as the reference stores them)."""
    header = ["ply", "format binary_little_endian 1.0"]
    blobs = []
    for name, m in models.items():
        cols = _columns(m)
        n = m["xyz"].shape[0]
        header.append(f"element vertex_{name} {n}")
        header += [f"property float {c}" for c, _ in cols]
        rec = np.empty(n, dtype=[(c, "<f4") for c, _ in cols])
        for c, v in cols:
            rec[c] = v
        blobs.append(rec.tobytes())
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        for b in blobs:
            f.write(b)


def _group(rec, n, prefix):
    names = [c for c in rec.dtype.names if re.fullmatch(prefix + r"_\d+", c)]
    names.sort(key=lambda c: int(c.rsplit("_", 1)[1]))  # gaussian_model.py:113-116: sorted by the numeric suffix
    out = np.zeros((n, len(names)), np.float32)
    for i, c in enumerate(names):
        out[:, i] = rec[c]
    return out


def read_scene_ply(path: str) -> "OrderedDict[str, Dict[str, np.ndarray]]":
    """Inverse of write_scene_ply; also reads a single-model file whose element is called ``vertex``
    (gaussian_model.py:97-101) under the name ''."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").split("\n")
    if lines[0].strip() != "ply" or "binary_little_endian" not in lines[1]:
        raise ValueError("not a binary little-endian PLY file")
    elements = []
    for ln in lines[2:]:
        t = ln.split()
        if not t:
            continue
        if t[0] == "element":
            elements.append((t[1], int(t[2]), []))
        elif t[0] == "property":
            if t[1] == "list":
                raise ValueError("list properties are not part of the scene format")
            elements[-1][2].append((t[2], _PLY_TYPES[t[1]]))
    out = OrderedDict()
    off = end
    for name, n, props in elements:
        dt = np.dtype(props)
        rec = np.frombuffer(data, dtype=dt, count=n, offset=off)
        off += n * dt.itemsize
        if not name.startswith("vertex"):
            continue
        dc, rest = _group(rec, n, "f_dc"), _group(rec, n, "f_rest")
        out[name[7:] if name.startswith("vertex_") else ""] = {
            "xyz": np.stack([rec["x"], rec["y"], rec["z"]], 1).astype(np.float32),
            "features_dc": np.ascontiguousarray(dc.reshape(n, 3, dc.shape[1] // 3).transpose(0, 2, 1)),        # :126, :143
            "features_rest": np.ascontiguousarray(rest.reshape(n, 3, rest.shape[1] // 3).transpose(0, 2, 1)),  # :127, :144
            "opacity": np.asarray(rec["opacity"], np.float32)[:, None],
            "scaling": _group(rec, n, "scale"),
            "rotation": _group(rec, n, "rot"),
            "semantic": _group(rec, n, "semantic"),
        }
    return out


def write_viewer_ply(path: str, means3D, shs, opacities, scales, rotations) -> None:
    """The single-element ``vertex`` PLY the reference's ``make_ply.py`` writes for a composed frame (the vanilla 3D-GS
    viewer layout; /root/reference/make_ply.py:37-79): inputs are the FLATTENED, post-activation rasterizer inputs of one
    frame (``scene.compose`` / the model's getters: means3D [n,3], shs [n,M,3], opacities [n,1] in (0,1), scales [n,3] > 0,
    rotations [n,4]); the file holds x y z, zero normals, f_dc_* / f_rest_* channel-major, opacity as logit of the value
    clipped to [1e-6, 1 - 1e-6], log scales, rotations -- float32, in that column order."""
    xyz = np.asarray(means3D, np.float32)
    n = xyz.shape[0]
    f = np.ascontiguousarray(np.asarray(shs, np.float32).transpose(0, 2, 1))       # [n, 3, M]   (:40)
    f_dc = f[..., :1].reshape(n, -1)                                                # (:41)
    f_rest = f[..., 1:].reshape(n, -1)                                              # (:42)
    op = np.clip(np.asarray(opacities, np.float32).reshape(n, 1), 1e-6, 1. - 1e-6)  # (:43-44)
    op = np.log(op / (1 - op))                                                      # (:45)
    sc = np.log(np.asarray(scales, np.float32))                                     # (:47-48)
    rot = np.asarray(rotations, np.float32)
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(f_dc.shape[1])] + \
            [f"f_rest_{i}" for i in range(f_rest.shape[1])] + ["opacity"] + [f"scale_{i}" for i in range(sc.shape[1])] + \
            [f"rot_{i}" for i in range(rot.shape[1])]
    attributes = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, op, sc, rot), axis=1).astype(np.float32)
    rec = np.empty(n, dtype=[(c, "<f4") for c in names])
    for i, c in enumerate(names):
        rec[c] = attributes[:, i]
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"] + [f"property float {c}" for c in names] + \
             ["end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        fh.write(rec.tobytes())
