#!/usr/bin/env python
"""How many of a quadrant's 64 pixels hit, per (quadrant, instance) visit of the blend walk, on the benchmark scene (no GPU):
replays the oracle's per-tile lists (tools/replay_visits.c: replay_lane_hist) and prints the cumulative histogram quoted in
DESIGN.md section 10.  Measurement tooling, not part of the product.

    python tools/lane_hist.py [gaussians]
"""
import subprocess, sys, os, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'tests')); sys.path.insert(0, os.path.join(HERE, '..'))
import numpy as np
from helpers import oracle_kwargs
from oracle import oracle
from street_gaussians_amd import synthetic as syn
cam = syn.make_camera(1920, 1280, fx=2050.0)
sc = syn.make_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, cam, S=0, seed=0)
fw = oracle.forward(**oracle_kwargs(cam, sc))
subprocess.check_call(["gcc","-O2","-fopenmp","-shared","-fPIC",os.path.join(HERE, "replay_visits.c"),"-o","/tmp/replay_visits.so","-lm"])
L=C.CDLL("/tmp/replay_visits.so"); p=lambda a:a.ctypes.data_as(C.c_void_p)
arrs=[np.ascontiguousarray(x) for x in (fw.ranges.astype(np.uint32), fw.point_list.astype(np.uint32), fw.means2D.astype(np.float32), fw.conic_opacity.astype(np.float32), fw.n_contrib.astype(np.uint32))]
h=np.zeros(65)
L.replay_lane_hist(1920,1280,*[p(a) for a in arrs],p(h))
tot=h.sum(); print("visits",tot)
cum=np.cumsum(h)/tot
for k in (1,2,3,4,6,8,12,16,24,32,48,64): print(k, round(cum[k],4))
print("mean lanes", (h*np.arange(65)).sum()/tot)
if len(sys.argv) > 2:  # full histogram for tools/valu_model.py
    import json
    json.dump({"gaussians": int(sys.argv[1]), "visits": float(tot), "hist_hit_lanes": [float(x) for x in h],
               "note": "visits by number of hitting lanes (0..64), CPU replay of the benchmark frame (tools/replay_visits.c)"},
              open(sys.argv[2], "w"))
