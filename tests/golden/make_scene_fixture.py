"""Generates tests/golden/scene_ref_layout.ply and scene_ref_state.pth: a two-model scene written in the layout the
REFERENCE writes, by executing the reference's own code where that is possible in this container:

  * per sub-model, ``GaussianModel.make_ply`` and ``construct_list_of_attributes``
    (/root/reference/lib/models/gaussian_model.py:80-95, 327-341) are cut out of the reference's source and run on a
    stub object -> the structured array (column names, order, channel-major feature flattening) is the reference's;
  * ``StreetGaussianModel.save_ply`` (street_gaussian_model.py:94-105) then hands these arrays to the ``plyfile``
    package as elements named ``vertex_<model>``.  plyfile is not installed here, so the file is emitted by the
    15-line writer below, which follows plyfile's output for this case: header lines ``ply`` /
    ``format binary_little_endian 1.0`` / ``element <name> <count>`` / one ``property float <column>`` per f4 column /
    ``end_header``, then each element's records, packed, in order;
  * per sub-model, ``GaussianModel.state_dict`` (gaussian_model.py:182-205) is executed the same way and the result is
    stored under the model's name together with ``iter``, as train.py:218-223 does.

Run once where /root/reference exists:  python tests/golden/make_scene_fixture.py"""
import os
import re
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
GM = "/root/reference/lib/models/gaussian_model.py"


def reference_methods():
    src = open(GM).read()
    ns = {"torch": torch, "np": np, "nn": nn}
    for name in ("make_ply", "construct_list_of_attributes", "state_dict"):
        m = re.search(rf"^    def {name}\(self.*?(?=^    def |\Z)", src, re.S | re.M)
        body = "\n".join(ln[4:] if ln.startswith("    ") else ln for ln in m.group(0).split("\n"))
        exec(body, ns)
    return ns


def make_models(seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    models = {}
    for name, n, C, S in (("background", 700, 1, 5), ("obj_001", 150, 3, 1)):
        models[name] = {"_xyz": r(n, 3) * 4 + torch.tensor([0.0, 0.0, 12.0]) * (name == "background"),
                        "_features_dc": r(n, C, 3) * 0.3 + 0.4, "_features_rest": r(n, 15, 3) * 0.05,
                        "_opacity": r(n, 1) * 2, "_scaling": r(n, 3) * 0.5 - 2.5, "_rotation": r(n, 4), "_semantic": r(n, S)}
    return models


class Stub:
    pass


def write_ply(path, elements):
    """plyfile's PlyData([...]).write for float32-only elements (see the module docstring)."""
    header = ["ply", "format binary_little_endian 1.0"]
    for name, rec in elements:
        header.append(f"element {name} {rec.shape[0]}")
        header += [f"property float {c}" for c in rec.dtype.names]
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        for _, rec in elements:
            f.write(rec.astype(rec.dtype.newbyteorder("<")).tobytes())


def main():
    ns = reference_methods()
    models = make_models()
    elements, state, raw = [], {}, {}
    for name, p in models.items():
        m = Stub()
        for k, v in p.items():
            setattr(m, k, nn.Parameter(v.clone()))
        m.construct_list_of_attributes = lambda m=m: ns["construct_list_of_attributes"](m)
        elements.append((f"vertex_{name}", ns["make_ply"](m)))           # street_gaussian_model.py:98-103
        state[name] = {k: v.detach().clone() for k, v in ns["state_dict"](m, is_final=True).items()}
        for k, v in p.items():
            raw[f"{name}/{k}"] = v.numpy()
    state["iter"] = 30000                                                # train.py:221
    write_ply(os.path.join(HERE, "scene", "scene_ref_layout.ply"), elements)
    torch.save(state, os.path.join(HERE, "scene", "scene_ref_state.pth"))
    np.savez_compressed(os.path.join(HERE, "scene", "scene_ref_params.npz"), **raw)
    print("wrote", [os.path.getsize(os.path.join(HERE, "scene", f)) for f in ("scene_ref_layout.ply", "scene_ref_state.pth", "scene_ref_params.npz")])


if __name__ == "__main__":
    sys.exit(main())
