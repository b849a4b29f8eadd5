import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from gpu_utils import raw_forward, raw_backward, npy, switches
from helpers import oracle_kwargs
from oracle import oracle
from street_gaussians_amd import synthetic as syn, _C

cam = syn.make_camera(256, 256, fx=280.0)
sc = syn.make_scene(5000, cam, S=19, seed=5, scale_px=0.005)
kw = oracle_kwargs(cam, sc, deg=1)
wts = syn.loss_weights(cam, S=19)
fw = oracle.forward(**kw)
ref = oracle.backward(fw, wts["color"], wts["depth"], wts["alpha"], wts["semantic"])
res, internal = raw_forward(kw)
def rep(tag, g):
    out = []
    for k in ["means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations", "semantics"]:
        a = npy(g[k]).reshape(ref[k].shape).astype(np.float64); b = ref[k].astype(np.float64)
        s = np.abs(b).max() + 1e-30
        out.append(f"{k}:{np.abs(a-b).max()/s:.2e}")
    print(tag, " ".join(out), flush=True)
rep("default", raw_backward(kw, res, wts))
for name, m in [("NO_HITS", _C.NO_HITS), ("NO_DET", _C.NO_DET), ("NO_DPP", _C.NO_DPP), ("NO_HITS|NO_DET", _C.NO_HITS | _C.NO_DET), ("NO_CULL", _C.NO_CULL)]:
    with switches(m):
        rep(name, raw_backward(kw, res, wts))
nc = npy(internal("n_contrib")).view(np.uint32).reshape(256, 256)
print("n_contrib diff", (nc != fw.n_contrib).sum())
