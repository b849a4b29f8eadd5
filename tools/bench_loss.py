#!/usr/bin/env python
"""Measurement of the colour-loss row on one MI355X: train.py:100-104's
    (1 - lambda_dssim) * lambda_l1 * l1_loss(image, gt, mask) + lambda_dssim * (1 - ssim(image, gt, mask=mask))
forward + backward at 1920x1280, with the fused kernels (street_gaussians_amd/losses.py) and with the reference's
torch ops (tests/torch_ref_loss.py on the GPU).  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_ref_loss as ref  # noqa: E402
from street_gaussians_amd import losses  # noqa: E402

H, W, C = 1280, 1920, 3
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
gt = torch.rand(C, H, W, generator=g).to(dev)
img = (gt + 0.1 * torch.randn(C, H, W, generator=g).to(dev)).clamp(0, 1).requires_grad_(True)
mask = (torch.rand(1, H, W, generator=g) < 0.9).to(dev)
acc = torch.rand(1, H, W, generator=g).to(dev).requires_grad_(True)
depth = (torch.rand(1, H, W, generator=g) * 40).to(dev).requires_grad_(True)
sky = (torch.rand(1, H, W, generator=g) < 0.2).to(dev)
lidar = torch.where(torch.rand(1, H, W, generator=g) < 0.3, torch.rand(1, H, W, generator=g) * 60, torch.zeros(1, H, W)).to(dev)


def step(mod):
    img.grad = acc.grad = depth.grad = None
    l1 = mod.l1_loss(img, gt, mask)
    loss = 0.8 * l1 + 0.2 * (1.0 - mod.ssim(img, gt, mask=mask))                  # train.py:100-104
    if FULL:
        loss = loss + 0.05 * mod.sky_loss(acc, sky)                               # :106-112
        loss = loss + 0.1 * mod.lidar_depth_loss(depth, acc, lidar, mask)         # :124-131
    loss.backward()
    return loss


def timeit(mod, n=20):
    for _ in range(3):
        step(mod)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(mod)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


FULL = False
a, b = step(losses).item(), step(ref).item()
fused, torch_ops = timeit(losses), timeit(ref)
FULL = True
a2, b2 = step(losses).item(), step(ref).item()
fused2, torch_ops2 = timeit(losses), timeit(ref)
px = C * H * W * 4
alg = (2 * px + 3 * px) + (3 * px + 2 * px + px) + (2 * px) + (2 * px + px)  # ssim fwd, ssim bwd, l1 fwd, l1 bwd
print(json.dumps({"what": "colour loss forward+backward (SURVEY 8f n3), 1920x1280", "loss_fused": a, "loss_torch_ops": b,
                  "fused_ms": round(fused, 3), "torch_ops_ms": round(torch_ops, 3), "speedup": round(torch_ops / fused, 2),
                  "algorithmic_bytes": alg, "fused_GBps": round(alg / fused / 1e6, 1), "hbm_peak_GBps": 8000.0,
                  "with_sky_and_lidar_terms": {"loss_fused": a2, "loss_torch_ops": b2, "fused_ms": round(fused2, 3),
                                               "torch_ops_ms": round(torch_ops2, 3), "speedup": round(torch_ops2 / fused2, 2)}}))
