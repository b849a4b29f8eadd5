#!/usr/bin/env python
"""The exchange step's local rebuild, dL/dSH = sum_v Y(dir_v) (x) dRGB_v (sgr_sh_grad_from_views, csrc/sgr_multiview.hip),
timed on ONE GPU with a synthetic all-gathered payload of V views: what every rank runs on its compute stream after the
all-gather at N = V GPUs (the exposed part of the 8-GPU step budget)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from street_gaussians_amd import _C  # noqa: E402

P, M, deg = int(os.environ.get("SGR_BENCH_P", 1_000_000)), 16, 3
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
means = torch.randn(P, 3, device=dev, generator=g) * 20
out = {}
for V in (1, 2, 4, 8):
    row = 3 + 3 * P  # [campos | dRGB] per view, as FactoredGradReducer lays the payload out
    A = torch.randn(V, row, device=dev, generator=g)
    A[:, 3:].view(V, P, 3)[torch.rand(V, P, device=dev, generator=g) < 0.14] = 0  # Gaussians a view does not see
    base = A.data_ptr()
    f = lambda: _C.sh_grad_from_rows(P, deg, M, V, means.data_ptr(), 0, base, row, base + 12, row, dev)
    for _ in range(5):
        r = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        r = f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byts = V * 12 * P + 12 * P + 192 * P
    out[f"V{V}"] = {"ms": round(ms, 4), "algorithmic_GB": round(byts / 1e9, 3), "TBps": round(byts / ms / 1e9, 2)}
print(json.dumps({"gaussians": P, "sh_grad_from_views": out}))
