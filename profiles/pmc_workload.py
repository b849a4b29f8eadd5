"""Workload for the rocprofv3 PMC passes (profiles/README.md): a calibration copy of known size followed by
three forward+backward steps of the bench workload.  Run as
    rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d <dir> -o fetch -- python profiles/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE  --kernel-trace --output-format csv -d <dir> -o write -- python profiles/pmc_workload.py
(counters in their own passes; FETCH_SIZE needs 3 of the 4 TCC slots)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from street_gaussians_amd import synthetic as syn  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

from street_gaussians_amd import _C as native_c  # noqa: E402

# the mode bench.py's headline runs in (bench.py --mode): strict = parity arithmetic on the reference's tile rects
MODE = os.environ.get("SGR_PMC_MODE", "strict")
native_c.test_switches((native_c.test_switches(-1) & ~(native_c.EXACT | native_c.REF_RECT)) |
                       {"strict": native_c.EXACT | native_c.REF_RECT, "exact": native_c.EXACT, "fast": 0}[MODE])
dev = torch.device("cuda")
# calibration: 1 GiB float4-wide device copy (reads 2^30 B, writes 2^30 B) -- scale past the 256 MiB Infinity Cache
a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()

P = int(os.environ.get("SGR_BENCH_P", "1000000"))
cam = syn.make_camera(1920, 1280, fx=2050.0)
sc = syn.make_scene(P, cam, seed=0)
t = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "shs"]}
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
w = {k: v.to(dev) for k, v in syn.loss_weights(cam).items()}
st = GaussianRasterizationSettings(image_height=1280, image_width=1920, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                   bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.viewmatrix.to(dev),
                                   projmatrix=cam.projmatrix.to(dev), sh_degree=3, campos=cam.campos.to(dev),
                                   prefiltered=False, debug=False)
rast = GaussianRasterizer(st)
for _ in range(3):
    for p in list(t.values()) + [m2d]:
        p.grad = None
    color, radii, depth, alpha, sem = rast(t["means3D"], m2d, t["opacities"], shs=t["shs"], scales=t["scales"],
                                           rotations=t["rotations"])
    ((color * w["color"]).sum() + (depth * w["depth"]).sum() + (alpha * w["alpha"]).sum()).backward()
torch.cuda.synchronize()
print("pmc workload done")
