"""`-m gpu` replay of the reference's own call sites: the argument sets tests/golden/callsite/*.npz hold -- captured from
StreetGaussianRenderer.render_kernel (/root/reference/lib/models/street_gaussian_renderer.py:122-260) and from
script/test_gaussian_rasterization.py by executing those sources with a recording rasterizer
(tests/golden/make_callsite_fixture.py, pinned by tests/test_callsite_cpu.py) -- go through the drop-in
GaussianRasterizationSettings / GaussianRasterizer exactly as captured (keyword call, None where the reference passes None,
means2D a [P,3] tensor that requires grad) and are compared with the C oracle on the same inputs."""
import glob
import os

import numpy as np
import pytest
import torch

from golden import make_callsite_fixture as mk
from gpu_utils import grad_close, image_close, npy
from oracle import oracle
from street_gaussians_amd import rasterizer as rast_mod

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIX = sorted(glob.glob(os.path.join(HERE, "golden", "callsite", "*.npz")))


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_replay_reference_call_site(path):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    st, kw, meta = mk.load(path)
    cu = lambda t: None if t is None else t.cuda()
    settings = GaussianRasterizationSettings(**{k: (cu(v) if torch.is_tensor(v) else v) for k, v in st.items()})
    args = {k: cu(v) for k, v in kw.items()}
    train = "means2D" in meta["requires_grad"]
    leaves = {}
    if train:
        # as render_kernel builds it: zeros(requires_grad=True) + 0 with retain_grad() -- a non-leaf whose .grad is read
        base = torch.zeros_like(args["means2D"], requires_grad=True)
        args["means2D"] = base + 0
        args["means2D"].retain_grad()
        for k in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp", "semantics"):
            if args[k] is not None:
                args[k] = args[k].clone().requires_grad_(True)
                leaves[k] = args[k]
    r = GaussianRasterizer(raster_settings=settings)
    color, radii, depth, alpha, semantic = r(**args)
    assert rast_mod.last_num_rendered() > 0

    fw = oracle.forward(means3D=kw["means3D"], opacities=kw["opacities"], viewmatrix=st["viewmatrix"],
                        projmatrix=st["projmatrix"], campos=st["campos"], bg=st["bg"], tanfovx=st["tanfovx"],
                        tanfovy=st["tanfovy"], image_height=st["image_height"], image_width=st["image_width"],
                        sh_degree=st["sh_degree"], scale_modifier=st["scale_modifier"], shs=kw["shs"],
                        colors_precomp=kw["colors_precomp"], scales=kw["scales"], rotations=kw["rotations"],
                        cov3D_precomp=kw["cov3D_precomp"], semantics=kw["semantics"])
    assert 0 < rast_mod.last_num_rendered() <= fw.num_rendered  # (tile rects cut down to where alpha >= 1/255 is possible)
    assert (npy(radii) == fw.radii).all()
    S = 0 if kw["semantics"] is None else kw["semantics"].shape[1]
    assert semantic.shape == (S, st["image_height"], st["image_width"])
    for k, t in (("color", color), ("depth", depth), ("alpha", alpha), ("semantic", semantic)):
        image_close(npy(t), getattr(fw, k), name=f"{os.path.basename(path)} {k}")
    if train:
        g = torch.Generator().manual_seed(5)
        w = {k: torch.rand(t.shape, generator=g) - 0.5 for k, t in (("color", color), ("depth", depth), ("alpha", alpha),
                                                                      ("semantic", semantic))}
        torch.autograd.backward([color, depth, alpha, semantic], [w[k].cuda() for k in ("color", "depth", "alpha", "semantic")])
        go = oracle.backward(fw, w["color"], w["depth"], w["alpha"], w["semantic"] if S else None)
        # viewspace_points.grad: what add_densification_stats reads (street_gaussian_model.py:551-571)
        grad_close(npy(args["means2D"].grad), go["means2D"], name="viewspace_points.grad", rel=2e-4, abs_frac=3e-4)
        names = {"means3D": "means3D", "opacities": "opacity", "shs": "sh", "colors_precomp": "colors", "scales": "scales",
                 "rotations": "rotations", "cov3D_precomp": "cov3D", "semantics": "semantics"}
        for k, t in leaves.items():
            grad_close(npy(t.grad).reshape(go[names[k]].shape), go[names[k]], name=f"dL/d{k}", rel=2e-4, abs_frac=3e-4)
    fw.free()
