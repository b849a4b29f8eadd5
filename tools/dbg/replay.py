import sys, os, ctypes as C, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from helpers import oracle_kwargs
from oracle import oracle
from street_gaussians_amd import synthetic as syn
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cam = syn.make_camera(1920, 1280, fx=2050.0)
sc = syn.make_scene(P, cam, S=0, seed=0)
kw = oracle_kwargs(cam, sc)
t = time.time(); fw = oracle.forward(**kw); print("oracle fwd", time.time() - t, "R", fw.num_rendered, flush=True)
L = C.CDLL("/tmp/replay.so")
p = lambda a: a.ctypes.data_as(C.c_void_p)
for batch in (128, 256):
    out = np.zeros(16)
    arrs = [np.ascontiguousarray(x) for x in (fw.ranges.astype(np.uint32), fw.point_list.astype(np.uint32), fw.means2D.astype(np.float32), fw.conic_opacity.astype(np.float32), fw.n_contrib.astype(np.uint32))]
    t = time.time(); L.replay(1920, 1280, *[p(a) for a in arrs], batch, p(out)); 
    names = ["quadrant_visits", "max4_4x4_per_batch", "max4_8x2_per_batch", "max2_8x4_per_batch", "lane_hits", "sum_4x4_visits", "max4_4x4_nobatch", "rounds", "max16_per_batch", "sum_8x4_visits"]
    print("batch", batch, {n: float(out[i]) for i, n in enumerate(names)}, "t", time.time() - t, flush=True)
