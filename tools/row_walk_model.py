#!/usr/bin/env python
"""Cost model of a ROW-GRANULAR blend-backward walk on the benchmark frame (no GPU): every 16-lane row
of a quadrant's wave owns a 4x4 block (shape 0) or an 8x2 strip (shape 1) of pixels and walks only the instances that hit
it.  Replays the oracle's lists (tools/replay_visits.c: replay_row_walk) and prints today's quadrant visits, the block
visits, the steps of the row walk (per round and wave: its busiest row) and both with the barrier skew of a round.
Measurement tooling, not part of the product.

    python tools/row_walk_model.py
"""
import subprocess, sys, os, ctypes as C, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'tests')); sys.path.insert(0, os.path.join(HERE, '..'))
import numpy as np
from helpers import oracle_kwargs
from oracle import oracle
from street_gaussians_amd import synthetic as syn
cam = syn.make_camera(1920, 1280, fx=2050.0)
sc = syn.make_scene(1_000_000, cam, S=0, seed=0)
t0=time.time()
fw = oracle.forward(**oracle_kwargs(cam, sc))
print("oracle fwd", time.time()-t0)
subprocess.check_call(["gcc","-O2","-fopenmp","-shared","-fPIC",os.path.join(HERE, "replay_visits.c"),"-o","/tmp/replay_visits.so","-lm"])
L=C.CDLL("/tmp/replay_visits.so"); p=lambda a:a.ctypes.data_as(C.c_void_p)
arrs=[np.ascontiguousarray(x) for x in (fw.ranges.astype(np.uint32), fw.point_list.astype(np.uint32), fw.means2D.astype(np.float32), fw.conic_opacity.astype(np.float32), fw.n_contrib.astype(np.uint32))]
for shape in (0,1):
    for rn in (64,128,256,100000):
        o=np.zeros(7)
        t0=time.time()
        L.replay_row_walk(1920,1280,*[p(a) for a in arrs],shape,rn,p(o))
        print("shape",shape,"round",rn,"visits %.3gM blockvisits %.3gM steps %.3gM  skewed: today %.3gM  rowwalk %.3gM  hit entries %.3gM  rounds over the row cap %.4gM"%tuple(o/1e6), "t", round(time.time()-t0,1))
