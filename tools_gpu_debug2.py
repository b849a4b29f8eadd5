import os, sys
os.environ["SGR_TRACE"] = "1"
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
print("import ok", flush=True)
from test_gpu_parity import CASES, _kw
from gpu_utils import raw_forward, raw_backward, npy
from oracle import oracle
name = sys.argv[1]
cam, sc, kw = _kw(name)
fw = oracle.forward(**kw)
print("oracle ok R", fw.num_rendered, flush=True)
for it in range(3):
    for cull in (1, 0):
        if not cull: os.environ["SGR_NO_CULL"] = "1"
        res, internal = raw_forward(kw)
        torch.cuda.synchronize()
        os.environ.pop("SGR_NO_CULL", None)
        c = npy(res["color"])
        bad = ~np.isfinite(c) | (np.abs(c - fw.color) > 1e-3)
        badpix = bad.any(0)
        ys, xs = np.nonzero(badpix)
        nc = npy(internal("n_contrib")).reshape(cam.image_height, cam.image_width)
        print(f"[{name}] it={it} cull={cull} bad pixels {badpix.sum()} / {badpix.size}; n_contrib mismatches {(nc != fw.n_contrib).sum()} max nc {nc.max()} vs {fw.n_contrib.max()}", flush=True)
        for y, x in list(zip(ys, xs))[:6]:
            print(f"   pix ({x},{y}) tile ({x//16},{y//16}) quad {((y%16)//8)*2+(x%16)//8} lane {(y%8)*8+(x%8)} got {c[:,y,x]} want {fw.color[:,y,x]} nc {nc[y,x]} vs {fw.n_contrib[y,x]}", flush=True)
        if len(ys):
            lanes = (ys % 8) * 8 + (xs % 8)
            print("   bad by lane:", np.bincount(lanes, minlength=64).tolist(), flush=True)
            print("   bad by quadrant:", np.bincount(((ys % 16) // 8) * 2 + (xs % 16) // 8, minlength=4).tolist(), flush=True)
            tiles = (ys // 16) * ((cam.image_width + 15) // 16) + xs // 16
            ut = np.unique(tiles)
            rg = fw.ranges
            print("   bad tiles:", len(ut), "list lens of bad tiles:", (rg[ut, 1] - rg[ut, 0])[:20].tolist(), "max list len overall", (rg[:, 1] - rg[:, 0]).max(), flush=True)
