import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from test_gpu_parity import CASES, _kw
from gpu_utils import raw_forward, raw_backward, npy
from oracle import oracle
from street_gaussians_amd import synthetic as syn

def fwd_debug(name):
    cam, sc, kw = _kw(name)
    fw = oracle.forward(**kw)
    for cull in (1, 0):
        if not cull: os.environ["SGR_NO_CULL"] = "1"
        res, internal = raw_forward(kw)
        os.environ.pop("SGR_NO_CULL", None)
        c = npy(res["color"]); 
        bad = ~np.isfinite(c) | (np.abs(c - fw.color) > 1e-3)
        badpix = bad.any(0)
        ys, xs = np.nonzero(badpix)
        print(f"[{name}] cull={cull} bad pixels {badpix.sum()} / {badpix.size}; alpha bad {(np.abs(npy(res['alpha'])-fw.alpha)>1e-3).sum()} depth bad {(np.abs(npy(res['depth'])-fw.depth)>1e-2).sum()}")
        nc = npy(internal("n_contrib")).reshape(cam.image_height, cam.image_width)
        print("   n_contrib mismatches", (nc != fw.n_contrib).sum())
        for y, x in list(zip(ys, xs))[:12]:
            print(f"   pix ({x},{y}) tile ({x//16},{y//16}) quad ({(x%16)//8},{(y%16)//8}) lane {(y%8)*8+(x%8)} got {c[:,y,x]} want {fw.color[:,y,x]} nc {nc[y,x]} vs {fw.n_contrib[y,x]} alpha {npy(res['alpha'])[0,y,x]} vs {fw.alpha[0,y,x]}")
        # histogram of bad pixels by lane index within quadrant
        lanes = (ys % 8) * 8 + (xs % 8)
        print("   bad by lane:", np.bincount(lanes, minlength=64).tolist())
        print("   bad by quadrant:", np.bincount(((ys % 16) // 8) * 2 + (xs % 16) // 8, minlength=4).tolist())
    fw.free()

def bwd_debug(name):
    os.environ["SGR_TRACE"] = "1"
    cam, sc, kw = _kw(name)
    wts = syn.loss_weights(cam, S=sc.semantics.shape[1])
    res, internal = raw_forward(kw)
    print(f"[{name}] forward done R={res['R']}", flush=True)
    g = raw_backward(kw, res, wts)
    torch.cuda.synchronize()
    print("backward done", {k: float(v.abs().max()) for k, v in g.items()}, flush=True)

if sys.argv[1] == "fwd": fwd_debug(sys.argv[2])
else: bwd_debug(sys.argv[2])
