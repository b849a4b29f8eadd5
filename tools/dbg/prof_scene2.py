import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from street_gaussians_amd import scene
exec(open(os.path.join(ROOT, "tools/dbg/prof_scene.py")).read().split("outs = scene.compose")[0].split("from street_gaussians_amd import scene")[1])
T = {}
orig_bwd = scene._Compose.backward
def timed_bwd(ctx, *a):
    t0 = time.perf_counter(); r = orig_bwd(ctx, *a); T["bwd_py"] = T.get("bwd_py", 0) + time.perf_counter() - t0; return r
scene._Compose.backward = staticmethod(timed_bwd)
orig_call = scene._native.lib().sgr_scene_compose_backward
outs = scene.compose(segs, M, S)
ups = [torch.randn_like(o) for o in outs]
def step():
    for t in leaves: t.grad = None
    t0 = time.perf_counter()
    outs = scene.compose(segs, M, S)
    t1 = time.perf_counter()
    torch.autograd.backward(list(outs), ups)
    t2 = time.perf_counter()
    T["fwd"] = T.get("fwd", 0) + t1 - t0; T["bwd"] = T.get("bwd", 0) + t2 - t1
for _ in range(5): step()
torch.cuda.synchronize(); T.clear()
n = 50
for _ in range(n): step()
torch.cuda.synchronize()
print({k: round(1e3 * v / n, 3) for k, v in T.items()})
# the same with a trivial custom function over 168 leaves: what autograd itself costs
class Id(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *ts):
        ctx.shapes = [t.shape for t in ts]
        return torch.zeros(4, device=ts[0].device)
    @staticmethod
    def backward(ctx, g):
        flat = torch.empty(sum(int(torch.Size(s).numel()) for s in ctx.shapes), device=g.device)
        return tuple(p.view(s) for p, s in zip(flat.split_with_sizes([int(torch.Size(s).numel()) for s in ctx.shapes]), ctx.shapes))
def step2():
    for t in leaves: t.grad = None
    t0 = time.perf_counter()
    o = Id.apply(*leaves)
    t1 = time.perf_counter()
    o.backward(torch.ones_like(o))
    t2 = time.perf_counter()
    T["fwd"] = T.get("fwd", 0) + t1 - t0; T["bwd"] = T.get("bwd", 0) + t2 - t1
T.clear()
for _ in range(5): step2()
torch.cuda.synchronize(); T.clear()
for _ in range(n): step2()
torch.cuda.synchronize()
print("identity op over", len(leaves), "leaves:", {k: round(1e3 * v / n, 3) for k, v in T.items()})
