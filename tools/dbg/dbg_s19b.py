import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from gpu_utils import raw_forward, raw_backward, npy, switches
from helpers import oracle_kwargs
from street_gaussians_amd import synthetic as syn, _C
for S in [8, 12, 20]:
    cam = syn.make_camera(256, 256, fx=280.0)
    sc = syn.make_scene(5000, cam, S=S, seed=5, scale_px=0.005)
    kw = oracle_kwargs(cam, sc, deg=1)
    wts = syn.loss_weights(cam, S=S)
    res, internal = raw_forward(kw)
    g0 = raw_backward(kw, res, wts)
    g0b = raw_backward(kw, res, wts)
    nrep = sum(1 for _ in range(20) if not all(torch.equal(g0[k], v) for k, v in raw_backward(kw, res, wts).items()))
    with switches(_C.NO_HITS):
        gn = raw_backward(kw, res, wts)
        nrep_nohits = sum(1 for _ in range(20) if not all(torch.equal(gn[k], v) for k, v in raw_backward(kw, res, wts).items()))
    with switches(_C.NO_HITS):
        g1 = raw_backward(kw, res, wts)
    out = []
    for k in g0:
        a, b = npy(g0[k]).astype(np.float64), npy(g1[k]).astype(np.float64)
        if a.size == 0: continue
        s = np.abs(b).max() + 1e-30
        out.append(f"{k}:{np.abs(a-b).max()/s:.1e}/{(a!=b).sum()}")
    rep = all(torch.equal(g0[k], g0b[k]) for k in g0)
    print("S", S, "repeatable", rep, "differing runs of 20:", nrep, "no-hits:", nrep_nohits, " ".join(out), flush=True)
