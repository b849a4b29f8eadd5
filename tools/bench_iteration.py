#!/usr/bin/env python
"""One training iteration's GPU-side pipeline at BASELINE.json configs[2] scale, end to end:
    scene graph -> flat rasterizer inputs (n1) -> rasterizer forward -> colour loss (n3) -> backward through all of it,
2 M Gaussians (1.8 M background + 20 posed actors), SH degree 3, 19 semantic classes, 1920x1280.
Both variants use THIS repository's rasterizer; they differ in the rows next to it: the fused HIP ops
(street_gaussians_amd/scene.py, losses.py) vs the reference's torch-op formulation (tests/torch_ref_scene.py,
tests/torch_ref_loss.py run on the GPU).  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_ref_loss as ref_loss  # noqa: E402
import torch_ref_scene as ref_scene  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from street_gaussians_amd import losses, scene  # noqa: E402
from street_gaussians_amd import synthetic as syn  # noqa: E402

M, S, C, H, W = 16, 19, 5, 1280, 1920
dev = torch.device("cuda")
cam = syn.make_camera(W, H, fx=2050.0)
NB, NA, A = 1_800_000, 10_000, 20
bk = syn.make_scene(NB, cam, S=S, seed=0)
g = torch.Generator().manual_seed(1)
P = lambda t: t.to(dev).requires_grad_(True)
raw = lambda sc: dict(xyz=P(sc.means3D), rotation=P(sc.rotations), scaling=P(sc.scales.log()),
                      opacity=P(torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4))), features_rest=P(sc.shs[:, 1:]))
dicts = [dict(raw(bk), features_dc=P(bk.shs[:, :1]), semantic=P(bk.semantics), semantic_mode="logits")]
for k in range(A):  # actors: small clouds in their own frames, posed somewhere in front of the camera
    sc = syn.make_scene(NA, cam, S=0, seed=10 + k, zmin=4.0, zmax=6.0, margin=0.05)
    sc.means3D = (sc.means3D - sc.means3D.mean(0)) * 0.5
    z = 5.0 + 3.0 * k
    pose = torch.tensor([1.0, 0.0, 0.02 * k, 0.0, (k % 5 - 2) * 0.3 * z, 0.1 * z, z])
    dicts.append(dict(raw(sc), features_dc=P(sc.shs[:, :1].repeat(1, C, 1) / C), semantic=P(torch.randn(NA, 1, generator=g)),
                      pose=P(pose), idft=torch.ones(C).to(dev), class_label=k % S, semantic_mode="logits",
                      flip_mask=(torch.rand(NA, generator=g) < 0.5).to(dev)))
segs = [scene.Segment(**d) for d in dicts]
st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                   bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.viewmatrix.to(dev),
                                   projmatrix=cam.projmatrix.to(dev), sh_degree=3, campos=cam.campos.to(dev),
                                   prefiltered=False, debug=False)
rast = GaussianRasterizer(st)
gt = torch.rand(3, H, W, generator=g).to(dev)
mask = (torch.rand(1, H, W, generator=g) < 0.95).to(dev)
leaves = [t for d in dicts for t in d.values() if torch.is_tensor(t) and t.requires_grad]


def iteration(fused):
    for t in leaves:
        t.grad = None
    flat = scene.compose(segs, M, S) if fused else ref_scene.compose(dicts, M, S)
    means3D, rot, scales, opac, shs, sem = flat
    m2d = torch.zeros(means3D.shape[0], 3, device=dev, requires_grad=True)
    color, radii, depth, alpha, semantic = rast(means3D, m2d, opac, shs=shs, scales=scales, rotations=rot, semantics=sem)
    if fused:  # one forward pass + one backward pass over the image (losses.color_loss)
        loss = losses.color_loss(color, gt, mask, lambda_dssim=0.2, lambda_l1=1.0)
    else:
        loss = 0.8 * ref_loss.l1_loss(color, gt, mask) + 0.2 * (1.0 - ref_loss.ssim(color, gt, mask=mask))
    loss.backward()
    return loss


def timeit(fused, n=10):
    for _ in range(3):
        iteration(fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        iteration(fused)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


lf, lt = iteration(True).item(), iteration(False).item()
tf, tt = timeit(True), timeit(False)
print(json.dumps({"what": "scene graph -> rasterizer -> colour loss, forward+backward, 2M Gaussians, 1920x1280, SH3, S=19",
                  "loss_fused": lf, "loss_torch_ops": lt, "fused_ms": round(tf, 3), "torch_ops_rows_ms": round(tt, 3),
                  "speedup": round(tt / tf, 2), "iters_per_s_fused": round(1e3 / tf, 1)}))
