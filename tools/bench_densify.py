#!/usr/bin/env python
"""Measurement of adaptive density control at BASELINE.json configs[4] scale: one
densify_and_prune over 5 M Gaussians (SH3, 19 classes, Adam moments for all seven parameter groups) with the plan +
gather kernels (street_gaussians_amd/densify.py) and with the reference's step-by-step torch ops
(tests/torch_ref_densify.py on the GPU).  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_ref_densify as ref  # noqa: E402
from street_gaussians_amd import densify  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5_000_000
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g, device=dev)
params = {"xyz": r(N, 3) * 5, "f_dc": r(N, 1, 3), "f_rest": r(N, 15, 3), "opacity": r(N, 1) * 3,
          "scaling": r(N, 3) * 1.2 - 3.5, "rotation": r(N, 4), "semantic": r(N, 19)}
states = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in params.items()}
accum = torch.rand(N, 2, generator=g, device=dev) * 0.002
denom = torch.randint(0, 4, (N, 1), generator=g, device=dev).float()
kw = dict(max_grad=0.0008, min_opacity=0.05, extent=3.0, percent_dense=0.01, percent_big_ws=0.1)
normals = torch.randn(2 * N, 3, generator=g, device=dev)


def fused():
    return densify.densify_and_prune(params, accum, denom, prune_big=True, states=states, normals=None, **kw)[2]


def torch_ops():
    m = ref.Model(params, states, accum, denom)
    return m.densify_and_prune(prune_big=True, normals=normals, **kw)


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, out


def fused_bkgd():
    return densify.densify_and_prune(params, accum, denom, prune_big=True, states=states, normals=None, variant="bkgd",
                                     sphere_center=torch.zeros(3), sphere_radius=20.0, **kw)[2]


tf, sf = timeit(fused)
tb, sb = timeit(fused_bkgd)
if "--no-torch" in sys.argv:
    tt, st = float("nan"), None
else:
    tt, st = timeit(torch_ops)
moved = sum(v.numel() * 4 * 3 * 2 for v in params.values())  # parameters + two Adam moments, read + written once
print(json.dumps({"what": "densify_and_prune (SURVEY 8f n2)", "gaussians": N, "fused_ms": round(tf, 2), "torch_ops_ms": round(tt, 2),
                  "speedup": round(tt / tf, 1), "fused_bkgd_variant_ms": round(tb, 2),
                  "row_bytes_moved_gb": round(moved / 1e9, 2), "fused_effective_tb_s": round(moved / 1e12 / (tf * 1e-3), 2),
                  "scalars_fused": sf, "scalars_bkgd": sb, "scalars_torch_ops": st}))
