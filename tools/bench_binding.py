#!/usr/bin/env python
"""Host-side cost of the two bindings of the C ABI (pybind module built with torch.utils.cpp_extension vs ctypes): wall
time of rasterizer forward + backward at a size where the GPU work is small, so the step is bound by the binding, the
allocator callbacks and autograd; and at the benchmark size, where the GPU hides it.
    python tools/bench_binding.py   -> one JSON object"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from street_gaussians_amd import _C, synthetic as syn  # noqa: E402


def run(P, W, H, steps):
    dev = torch.device("cuda")
    cam = syn.make_camera(W, H, fx=2050.0 * W / 1920.0)
    sc = syn.make_scene(P, cam, seed=0)
    t = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "shs"]}
    w = {k: v.to(dev) for k, v in syn.loss_weights(cam).items()}
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                       bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.viewmatrix.to(dev),
                                       projmatrix=cam.projmatrix.to(dev), sh_degree=3, campos=cam.campos.to(dev),
                                       prefiltered=False, debug=False)
    rast = GaussianRasterizer(st)

    def step():
        for p in t.values():
            p.grad = None
        c, r, d, a, s = rast(t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([c, d, a], [w["color"], w["depth"], w["alpha"]])
    out = {}
    for b in ("ctypes", "pybind", "ctypes", "pybind"):
        _C.set_binding(b)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        out.setdefault(b, []).append(round(1e3 * (time.perf_counter() - t0) / steps, 4))
    return {k: min(v) for k, v in out.items()}


if __name__ == "__main__":
    res = {"what": "ms per rasterizer forward+backward (wall, best of 2 runs), pybind (torch.utils.cpp_extension) vs ctypes binding",
           "host_bound_2k_gaussians_320x200": run(2000, 320, 200, 300),
           "bench_1M_gaussians_1920x1280": run(1_000_000, 1920, 1280, 100)}
    print(json.dumps(res))
