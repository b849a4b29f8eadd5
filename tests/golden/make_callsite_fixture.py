"""Captures what the REFERENCE's own call sites pass to ``GaussianRasterizer`` and writes it to
tests/golden/callsite/*.npz -- the argument conventions of the drop-in boundary pinned by the reference's code instead of
by a restatement.

Two call sites, both executed FROM THE REFERENCE'S SOURCE in this container (no GPU, no CUDA extension):

  * ``StreetGaussianRenderer.render_kernel`` (/root/reference/lib/models/street_gaussian_renderer.py:122-260) together with
    ``make_rasterizer`` (lib/utils/camera_utils.py:194-227) and ``eval_sh`` (lib/utils/sh_utils.py): cut out of their
    files and run on a stub ``pc`` / camera / cfg, with ``diff_gaussian_rasterization`` replaced by a RECORDING
    rasterizer that stores the settings tuple and the keyword arguments of every call and returns outputs of the right
    shapes.  Variants: train mode with normals + semantics (the rasterizer's ``semantics`` input is their ``cat``),
    eval mode (``means2D=None``), and the python SH / covariance path (``colors_precomp`` + ``cov3D_precomp``).
  * ``script/test_gaussian_rasterization.py`` (the reference's only test, BASELINE.json configs[0]): its ``__main__`` body
    with ``torch.manual_seed(0)`` in front; both calls (S = 0 and S = 15) are captured.

The only textual changes are ``.cuda()`` / ``device="cuda"`` (there is no GPU here).  Run where /root/reference exists:

    python tests/golden/make_callsite_fixture.py
"""
import math
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "callsite")
REF = "/root/reference"
SETTINGS_FIELDS = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
                   "sh_degree", "campos", "prefiltered", "debug")
CALL_ARGS = ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp", "semantics")


class AttrDict(dict):
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):
        return dict.get(self, k, d)


def _nocuda(src):
    return src.replace(".cuda()", "").replace('device="cuda"', 'device="cpu"').replace("device='cuda'", "device='cpu'")


class Recorder:
    """Stands in for the diff_gaussian_rasterization package: records settings + keyword arguments of every call."""

    def __init__(self):
        self.calls = []
        rec = self

        class GaussianRasterizationSettings(tuple):
            def __new__(cls, **kw):
                assert tuple(kw) == SETTINGS_FIELDS or set(kw) == set(SETTINGS_FIELDS), kw.keys()
                self_ = tuple.__new__(cls, [kw[k] for k in SETTINGS_FIELDS])
                self_.kw = kw
                return self_

        class GaussianRasterizer:
            def __init__(self, raster_settings):
                self.raster_settings = raster_settings

            def __call__(self, *args, **kw):
                assert not args, "the reference calls the rasterizer with keyword arguments only"
                st = self.raster_settings.kw
                rec.calls.append((st, dict(kw)))
                P = kw["means3D"].shape[0]
                H, W = st["image_height"], st["image_width"]
                S = kw["semantics"].shape[1] if kw.get("semantics") is not None else 0
                z = lambda *s: torch.zeros(*s)
                return z(3, H, W), torch.ones(P, dtype=torch.int32), z(1, H, W), z(1, H, W), z(S, H, W)

        self.GaussianRasterizationSettings = GaussianRasterizationSettings
        self.GaussianRasterizer = GaussianRasterizer


def _func(path, name, method=False):
    src = open(path).read()
    pat = rf"^    def {name}\(.*?(?=^    def |\Z)" if method else rf"^def {name}\(.*?(?=^def |^class |\Z)"
    m = re.search(pat, src, re.S | re.M)
    assert m, (path, name)
    body = m.group(0)
    if method:
        body = "\n".join(ln[4:] if ln.startswith("    ") else ln for ln in body.split("\n"))
    return _nocuda(body)


def make_pc(P, M, n_sem, seed):
    """Stub of StreetGaussianModel: the getters render_kernel reads (post-activation values, like the real getters)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    pc = types.SimpleNamespace()
    pc.get_xyz = r(P, 3) * torch.tensor([6.0, 2.0, 6.0]) + torch.tensor([0.0, 0.0, 14.0])
    pc.get_opacity = torch.sigmoid(r(P, 1) * 2)
    pc.get_scaling = torch.exp(r(P, 3) * 0.5 - 2.3)
    pc.get_rotation = torch.nn.functional.normalize(r(P, 4))
    f = r(P, M, 3) * 0.1
    f[:, 0] += 0.6
    pc.get_features = f
    pc.get_semantic = r(P, n_sem)
    pc.max_sh_degree = int(math.isqrt(M)) - 1
    pc.active_sh_degree = pc.max_sh_degree
    nrm = torch.nn.functional.normalize(r(P, 3))
    pc.get_normals = lambda cam: nrm
    L = torch.diag_embed(pc.get_scaling)

    def get_covariance(scaling_modifier=1.0):  # symmetric 3x3 -> the 6 upper-triangular entries (gaussian_model.py:207-222)
        q = pc.get_rotation
        w, x, y, z = q.unbind(-1)
        R = torch.stack([torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)), -1),
                         torch.stack((2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)), -1),
                         torch.stack((2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), -1)], -2)
        A = R @ (scaling_modifier * L)
        Sg = A @ A.transpose(1, 2)
        return torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], -1)
    pc.get_covariance = get_covariance
    return pc


def make_camera(W, H, fx):
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from street_gaussians_amd import synthetic as syn
    c = syn.make_camera(W, H, fx=fx, yaw_deg=3.0, translation=(0.1, -0.05, 0.2))
    return types.SimpleNamespace(image_height=H, image_width=W, FoVx=2 * math.atan(c.tanfovx), FoVy=2 * math.atan(c.tanfovy),
                                 world_view_transform=c.viewmatrix, full_proj_transform=c.projmatrix, camera_center=c.campos)


def capture_render_kernel(mode, render_normal, use_semantic, convert_SHs_python, compute_cov3D_python, white_bg, seed):
    rec = Recorder()
    cfg = AttrDict(mode=mode, render=AttrDict(scaling_modifier=1.0, convert_SHs_python=convert_SHs_python,
                                             compute_cov3D_python=compute_cov3D_python, debug=False, render_normal=render_normal),
                   data=AttrDict(white_background=white_bg, use_semantic=use_semantic),
                   model=AttrDict(gaussian=AttrDict(semantic_mode="logits")))
    ns = {"torch": torch, "math": math, "cfg": cfg, "GaussianRasterizationSettings": rec.GaussianRasterizationSettings,
          "GaussianRasterizer": rec.GaussianRasterizer, "Camera": object, "StreetGaussianModel": object}
    shu = {}
    exec(_nocuda(open(os.path.join(REF, "lib/utils/sh_utils.py")).read()), shu)
    ns["eval_sh"] = shu["eval_sh"]
    exec(_func(os.path.join(REF, "lib/utils/camera_utils.py"), "make_rasterizer"), ns)
    exec(_func(os.path.join(REF, "lib/models/street_gaussian_renderer.py"), "render_kernel", method=True), ns)
    renderer = types.SimpleNamespace(cfg=cfg.render)
    pc = make_pc(3000, 16, 16, seed)
    cam = make_camera(320, 208, 340.0)
    out = ns["render_kernel"](renderer, cam, pc)
    assert len(rec.calls) == 1 and "rgb" in out
    return rec.calls[0]


def capture_smoke_script():
    rec = Recorder()
    src = open(os.path.join(REF, "script/test_gaussian_rasterization.py")).read()
    body = src[src.index("if __name__ == '__main__':"):]
    body = "\n".join(ln[4:] for ln in body.split("\n")[1:])
    ns = {"torch": torch, "math": math, "time": __import__("time"),
          "GaussianRasterizationSettings": rec.GaussianRasterizationSettings, "GaussianRasterizer": rec.GaussianRasterizer}
    torch.cuda.synchronize = lambda *a, **k: None  # the script brackets its calls with synchronize()
    torch.manual_seed(0)
    exec(_nocuda(body), ns)
    assert len(rec.calls) == 2
    return rec.calls


def save(name, call):
    st, kw = call
    assert set(kw) == set(CALL_ARGS), sorted(kw)
    arrays, meta = {}, {"none": [], "requires_grad": [], "settings_scalars": {}}
    for k in SETTINGS_FIELDS:
        v = st[k]
        if torch.is_tensor(v):
            arrays["st_" + k] = v.detach().numpy()
        else:
            meta["settings_scalars"][k] = v
    for k in CALL_ARGS:
        v = kw[k]
        if v is None:
            meta["none"].append(k)
        else:
            arrays["kw_" + k] = v.detach().numpy()
            if v.requires_grad:
                meta["requires_grad"].append(k)
    arrays["meta"] = np.frombuffer(repr(meta).encode(), dtype=np.uint8)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    return meta


def load(path):
    """-> (settings dict, kwargs dict of torch tensors / None, meta)."""
    z = np.load(path)
    meta = eval(bytes(z["meta"]).decode())  # noqa: S307 -- our own repr of a small dict
    st = dict(meta["settings_scalars"])
    kw = {k: None for k in meta["none"]}
    for k in z.files:
        if k.startswith("st_"):
            st[k[3:]] = torch.from_numpy(z[k].copy())
        elif k.startswith("kw_"):
            kw[k[3:]] = torch.from_numpy(z[k].copy())
    return st, kw, meta


VARIANTS = {
    "render_kernel_train_normals_semantics": dict(mode="train", render_normal=True, use_semantic=True, convert_SHs_python=False,
                                                  compute_cov3D_python=False, white_bg=False, seed=1),
    "render_kernel_eval_plain": dict(mode="eval", render_normal=False, use_semantic=False, convert_SHs_python=False,
                                     compute_cov3D_python=False, white_bg=True, seed=2),
    "render_kernel_train_python_sh_cov": dict(mode="train", render_normal=False, use_semantic=True, convert_SHs_python=True,
                                              compute_cov3D_python=True, white_bg=False, seed=3),
}


def capture_all():
    calls = {name: capture_render_kernel(**v) for name, v in VARIANTS.items()}
    smoke = capture_smoke_script()
    calls["smoke_script_call1"], calls["smoke_script_call2_sem15"] = smoke
    return calls


if __name__ == "__main__":
    for name, call in capture_all().items():
        meta = save(name, call)
        print(name, {k: tuple(v.shape) for k, v in call[1].items() if v is not None}, "None:", meta["none"],
              "requires_grad:", meta["requires_grad"], meta["settings_scalars"])
