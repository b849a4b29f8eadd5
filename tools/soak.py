#!/usr/bin/env python
"""Soak test on the GPU box: many repetitions of the forward and the backward on dense scenes, every result compared
bit for bit with the first (the kernels have no unordered float accumulation, so ANY difference is a race -- the
missing s_waitcnt of DESIGN.md section 3 showed up as one wrong row in thousands of launches).

    python tools/soak.py [repetitions]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_utils import raw_backward, raw_forward  # noqa: E402
from helpers import oracle_kwargs  # noqa: E402
from street_gaussians_amd import synthetic as syn  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
only = [int(x) for x in os.environ["SOAK_CONFIGS"].split(",")] if os.environ.get("SOAK_CONFIGS") else None  # indices into the list below
cam = syn.make_camera(1920, 1280, fx=2050.0)
bad = 0
for ci, (P, S) in enumerate([(1_000_000, 0), (400_000, 19), (300_000, 3), (300_000, 8), (300_000, 12), (300_000, 16), (200_000, 24), (200_000, 32)]):
    if only is not None and ci not in only:
        continue
    sc = syn.make_scene(P, cam, S=S, seed=3)
    kw = oracle_kwargs(cam, sc)
    wts = syn.loss_weights(cam, S=S)
    res, internal = raw_forward(kw)
    g0 = raw_backward(kw, res, wts)
    nf = nb = 0
    for i in range(reps):
        g = raw_backward(kw, res, wts)
        nb += any(not torch.equal(g[k], g0[k]) for k in g0)
        if i % 4 == 0:
            r2, _ = raw_forward(kw)
            nf += any(not torch.equal(r2[k], res[k]) for k in ["color", "depth", "alpha", "semantic", "radii"])
            g2 = raw_backward(kw, r2, wts)  # a fresh forward state (new hit record, new row flags)
            nb += any(not torch.equal(g2[k], g0[k]) for k in g0)
    torch.cuda.synchronize()
    print(f"P={P} S={S}: {reps} backward + {reps // 4 + 1} forward repetitions, mismatching backward {nb}, forward {nf}", flush=True)
    bad += nb + nf
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
