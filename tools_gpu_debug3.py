import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from test_gpu_parity import CASES, _kw
from gpu_utils import raw_forward, npy
from oracle import oracle
name = sys.argv[1]
cam, sc, kw = _kw(name)
fw = oracle.forward(**kw)
T = fw.ranges.shape[0]
dbg = torch.zeros(T * 20, dtype=torch.int64, device="cuda")
os.environ["SGR_DBG_PTR"] = str(dbg.data_ptr())
os.environ["SGR_NO_CULL"] = "1"
res, internal = raw_forward(kw)
torch.cuda.synchronize()
d = dbg.cpu().numpy().view(np.uint64).reshape(T, 20)
nc = npy(internal("n_contrib")).reshape(cam.image_height, cam.image_width)
lens = fw.ranges[:, 1] - fw.ranges[:, 0]
bad = 0
for t in range(T):
    L = int(lens[t])
    if L == 0: continue
    for w in range(4):
        for c in range(4):
            m = int(d[t, w * 4 + c])
            lo = c * 64
            expect = 0
            for b in range(64):
                if lo + b < L: expect |= (1 << b)
            if m != expect:
                bad += 1
                if bad < 15:
                    print(f"tile {t} wave {w} chunk {c}: mask {m:016x} expected {expect:016x} len {L} range dbg {d[t,16]} {d[t,17]} oracle {fw.ranges[t]}")
print("bad masks", bad, "tiles", T)
c = npy(res["color"]); badpix = (np.abs(c - fw.color) > 1e-3).any(0)
print("bad pixels", badpix.sum(), "n_contrib mismatch", (nc != fw.n_contrib).sum())
ys, xs = np.nonzero(badpix)
gx = (cam.image_width + 15) // 16
for t in np.unique((ys // 16) * gx + xs // 16):
    print("bad tile", t, "len", lens[t], "masks", [f"{int(v):x}" for v in d[t, :16]], "nc in tile max", nc[(t//gx)*16:(t//gx)*16+16, (t%gx)*16:(t%gx)*16+16].max())
