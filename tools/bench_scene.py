#!/usr/bin/env python
"""Measurement of the scene-graph rows on one MI355X: the fused compose op (csrc/sgr_scene.hip)
against the same computation written with the reference's torch ops (tests/torch_ref_scene.py, run on the GPU), at
BASELINE.json configs[2] scale: 2 M Gaussians = background + posed actors, SH degree 3, 19 semantic classes.

    python tools/bench_scene.py [--actors 20] [--actor-gaussians 10000] [--background 1800000]

Prints one JSON line: ms per forward+backward for both, and the fused op's HBM rate on its algorithmic bytes
(every raw parameter read once + every output written once, forward; upstream gradients read once + every parameter
gradient written once + parameters re-read, backward)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_ref_scene as ref  # noqa: E402  (the torch-op baseline; test infrastructure used as the thing to beat)
from street_gaussians_amd import scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--background", type=int, default=1_800_000)
ap.add_argument("--actors", type=int, default=20)
ap.add_argument("--actor-gaussians", type=int, default=10_000)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
M, S, C = 16, 19, 5
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev).requires_grad_(True)
counts = [args.background] + [args.actor_gaussians] * args.actors
segs, dicts = [], []
for k, n in enumerate(counts):
    d = dict(xyz=r(n, 3), rotation=r(n, 4), scaling=r(n, 3), opacity=r(n, 1), features_rest=r(n, M - 1, 3))
    if k == 0:
        d.update(features_dc=r(n, 1, 3), semantic=r(n, S), semantic_mode="logits")
    else:
        d.update(features_dc=r(n, C, 3), semantic=r(n, 1), pose=r(7), idft=torch.randn(C, generator=g).to(dev),
                 class_label=k % S, semantic_mode="logits", flip_mask=(torch.rand(n, generator=g) < 0.5).to(dev))
    dicts.append(d)
    segs.append(scene.Segment(**d))
N = sum(counts)
ups = None


leaves = [v for d in dicts for v in d.values() if torch.is_tensor(v) and v.requires_grad]


def run(fn):
    global ups
    for t in leaves:  # optimizer.zero_grad(set_to_none=True): the op's gradients become .grad, no accumulation kernels
        t.grad = None
    outs = fn()
    if ups is None:
        ups = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(list(outs), ups)


def timeit(fn):
    for _ in range(3):
        run(fn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run(fn)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / args.steps


fused = timeit(lambda: scene.compose(segs, M, S))
# flat-parameter mode: the same kernels, 8 autograd leaves instead of 8 per sub-model
flat = scene.FlatScene.from_segments(segs)
masks = [s.flip_mask for s in segs]
per_model_leaves = leaves
leaves = [t for t in flat.tensors.values()] + [flat.poses]
ups = None
flat_ms = timeit(lambda: flat.compose(M, S, flip_masks=masks))
leaves, ups = per_model_leaves, None
torch_ops = timeit(lambda: ref.compose(dicts, M, S))
per_g = 3 + 4 + 3 + 1 + 3 * M + S      # output floats per Gaussian
raw_bk = 3 + 4 + 3 + 1 + 3 * M + S      # raw floats per background Gaussian
raw_ac = 3 + 4 + 3 + 1 + 3 * C + 3 * (M - 1) + 1
n_ac = N - counts[0]
fwd = 4 * (counts[0] * raw_bk + n_ac * raw_ac + N * per_g)
bwd = 4 * (N * per_g + counts[0] * 2 * raw_bk + n_ac * 2 * raw_ac)
print(json.dumps({"what": "scene compose forward+backward (SURVEY 8f n1)", "gaussians": N, "actors": args.actors,
                  "fused_ms": round(fused, 3), "fused_flat_parameters_ms": round(flat_ms, 3), "autograd_leaves": len(per_model_leaves),
                  "autograd_leaves_flat": 8, "torch_ops_ms": round(torch_ops, 3),
                  "speedup": round(torch_ops / fused, 2), "algorithmic_bytes": fwd + bwd,
                  "fused_GBps": round((fwd + bwd) / fused / 1e6, 1), "hbm_peak_GBps": 8000.0,
                  "frac": round((fwd + bwd) / fused / 1e6 / 8000.0, 3)}))
