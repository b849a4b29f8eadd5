#!/bin/bash
run() { timeout 900 python bench.py --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'R', d['config']['num_rendered_R'], d['roofline']['stages_ms'])"; }
echo "== 5M"; run --gaussians 5000000 --steps 5 --warmup 2
echo "== 4K image 1M"; run --width 3840 --height 2160 --steps 5 --warmup 2
echo "== small image 100k"; run --width 320 --height 200 --gaussians 100000 --steps 10 --warmup 3
echo "== 2M S=19"; run --gaussians 2000000 --semantics 19 --steps 5 --warmup 2
echo "== 1M S=3"; run --semantics 3 --steps 5 --warmup 2
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from street_gaussians_amd import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
# one giant splat covering the whole 1920x1280 image + many tiny ones: fat rect in the duplicate kernel, long tile lists
cam = syn.make_camera(1920, 1280, fx=2050.0)
sc = syn.make_scene(200000, cam, seed=3)
sc.scales[:50] = 30.0; sc.means3D[:50, 2] = 20.0; sc.means3D[:50, :2] *= 0.1
dev = "cuda"
st = GaussianRasterizationSettings(1280, 1920, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, True)
t = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ["means3D","scales","rotations","opacities","shs"]}
out = GaussianRasterizer(st)(t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
(out[0].sum() + out[2].sum()).backward()
torch.cuda.synchronize()
print("giant splats ok: radii max", int(out[1].max()), "grad finite", bool(torch.isfinite(t["means3D"].grad).all()), "alpha max", float(out[3].max()))
PY
