import os, sys, time
sys.path.insert(0, '.')
import torch
from street_gaussians_amd import synthetic as syn
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cam = syn.make_camera(1920, 1280, fx=2050.0)
scene = syn.make_scene(P, cam, seed=0)
params = {k: getattr(scene, k).to(dev).requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "shs"]}
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
w = {k: v.to(dev) for k, v in syn.loss_weights(cam).items()}
st = GaussianRasterizationSettings(image_height=1280, image_width=1920, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev),
    scale_modifier=1.0, viewmatrix=cam.viewmatrix.to(dev), projmatrix=cam.projmatrix.to(dev), sh_degree=3, campos=cam.campos.to(dev), prefiltered=False, debug=False)
rast = GaussianRasterizer(st)
def sync(): torch.cuda.synchronize()
for it in range(12):
    for p in list(params.values()) + [m2d]: p.grad = None
    sync(); t0 = time.perf_counter()
    color, radii, depth, alpha, sem = rast(params["means3D"], m2d, params["opacities"], shs=params["shs"], scales=params["scales"], rotations=params["rotations"])
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    loss = (color * w["color"]).sum() + (depth * w["depth"]).sum() + (alpha * w["alpha"]).sum()
    sync(); t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter(); sync(); t5 = time.perf_counter()
    print(f"it {it}: fwd host {1e3*(t1-t0):.2f} ms, fwd total {1e3*(t2-t0):.2f}; loss {1e3*(t3-t2):.2f}; bwd host {1e3*(t4-t3):.2f}, bwd total {1e3*(t5-t3):.2f}; step {1e3*(t5-t0):.2f}", flush=True)
print("mem allocated GB", torch.cuda.memory_allocated()/1e9, "reserved", torch.cuda.memory_reserved()/1e9)
# raw native calls without autograd
from street_gaussians_amd import _C
e = torch.Tensor([])
for it in range(5):
    sync(); t0 = time.perf_counter()
    out = _C.rasterize_gaussians(st.bg, params["means3D"].detach(), e, torch.zeros(P, 0, device=dev), params["opacities"].detach(), params["scales"].detach(),
        params["rotations"].detach(), 1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, 1280, 1920, params["shs"].detach(), 3, st.campos, False, False)
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    print(f"raw fwd host {1e3*(t1-t0):.2f} total {1e3*(t2-t0):.2f}", flush=True)
