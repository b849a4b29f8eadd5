import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from gpu_utils import raw_forward, npy
from helpers import oracle_kwargs
from oracle import oracle
from street_gaussians_amd import synthetic as syn

def expected_hits(fw, W, H):
    gx, gy = (W + 15) // 16, (H + 15) // 16
    R = fw.num_rendered
    exp = np.zeros(R, np.uint8); valid = np.zeros(R, bool)
    ys, xs = np.mgrid[0:16, 0:16]
    quad = ((ys >> 3) * 2 + (xs >> 3))
    for t in range(gx * gy):
        r0, r1 = fw.ranges[t]
        if r1 <= r0: continue
        tx, ty = t % gx, t // gx
        px, py = tx * 16 + xs, ty * 16 + ys
        inside = (px < W) & (py < H)
        nc = np.zeros((16, 16), np.int64)
        nc[inside] = fw.n_contrib[py[inside], px[inside]]
        maxc = nc.max()
        for pos in range(int(maxc)):
            g = fw.point_list[r0 + pos]
            X, Y = fw.means2D[g]; A, B, C, O = fw.conic_opacity[g]
            dx = np.float32(X) - px.astype(np.float32); dy = np.float32(Y) - py.astype(np.float32)
            pw = np.float32(-0.5) * (A * dx * dx + C * dy * dy) - B * dx * dy
            al = np.minimum(np.float32(0.99), O * np.exp(pw))
            hit = inside & (pw <= 0) & (al >= np.float32(1 / 255)) & (pos < nc)
            b = 0
            for q in range(4):
                if hit[quad == q].any(): b |= 1 << q
            exp[r0 + pos] = b; valid[r0 + pos] = True
    return exp, valid

for name, S, deg in [("S0", 0, 3), ("S3", 3, 3), ("S19", 19, 1), ("S8", 8, 1), ("S12", 12, 1)]:
    cam = syn.make_camera(256, 256, fx=280.0)
    sc = syn.make_scene(5000, cam, S=S, seed=5, scale_px=0.005)
    kw = oracle_kwargs(cam, sc, deg=deg)
    fw = oracle.forward(**kw)
    res, internal = raw_forward(kw)
    hits = npy(internal("hits"))
    exp, valid = expected_hits(fw, 256, 256)
    d = (hits != exp) & valid
    print(name, "R", fw.num_rendered, "valid", valid.sum(), "mismatch", d.sum(), "missing bits", ((exp & ~hits) != 0)[valid].sum(), "extra bits", ((hits & ~exp) != 0)[valid].sum(), flush=True)
    if d.sum():
        i = np.nonzero(d)[0][:10]
        print("  idx", i, "exp", exp[i], "got", hits[i])
