import cProfile, pstats, io, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from street_gaussians_amd import scene
M, S, C = 16, 19, 5
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev).requires_grad_(True)
counts = [1_800_000] + [10_000] * 20
segs = []
leaves = []
for k, n in enumerate(counts):
    d = dict(xyz=r(n, 3), rotation=r(n, 4), scaling=r(n, 3), opacity=r(n, 1), features_rest=r(n, M - 1, 3))
    if k == 0:
        d.update(features_dc=r(n, 1, 3), semantic=r(n, S), semantic_mode="logits")
    else:
        d.update(features_dc=r(n, C, 3), semantic=r(n, 1), pose=r(7), idft=torch.randn(C, generator=g).to(dev),
                 class_label=k % S, semantic_mode="logits", flip_mask=(torch.rand(n, generator=g) < 0.5).to(dev))
    leaves += [v for v in d.values() if torch.is_tensor(v) and v.requires_grad]
    segs.append(scene.Segment(**d))
outs = scene.compose(segs, M, S)
ups = [torch.randn_like(o) for o in outs]
def step(sync=False):
    for t in leaves: t.grad = None
    t0 = time.perf_counter()
    outs = scene.compose(segs, M, S)
    if sync: torch.cuda.synchronize()
    t1 = time.perf_counter()
    torch.autograd.backward(list(outs), ups)
    if sync: torch.cuda.synchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1
for _ in range(5): step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print("async ms/iter", 1e3 * (time.perf_counter() - t0) / n)
f = b = 0
for _ in range(n):
    a, c = step(True); f += a; b += c
print("sync fwd ms", 1e3 * f / n, "bwd ms", 1e3 * b / n)
# host-only cost: time to ISSUE (no sync) forward and backward
f = b = 0
torch.cuda.synchronize()
for _ in range(n):
    a, c = step(False); f += a; b += c
torch.cuda.synchronize()
print("issue fwd ms", 1e3 * f / n, "issue bwd ms", 1e3 * b / n)
pr = cProfile.Profile(); pr.enable()
for _ in range(n): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25); print(s.getvalue()[:6000])
