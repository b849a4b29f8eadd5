import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from test_gpu_parity import CASES, _kw, _color_mag
from gpu_utils import raw_forward, raw_backward, npy
from oracle import oracle, ref
from street_gaussians_amd import synthetic as syn
name = "huge_splats"
cam, sc, kw = _kw(name)
wts = syn.loss_weights(cam, S=0)
fw = oracle.forward(**kw)
go = oracle.backward(fw, wts["color"], wts["depth"], wts["alpha"], wts["semantic"])
rf = ref.forward(**kw)
gr = ref.backward(rf, wts["color"], wts["depth"], wts["alpha"], wts["semantic"])
gr2 = ref.backward(rf, wts["color"], wts["depth"], wts["alpha"], wts["semantic"])
res, _ = raw_forward(kw)
gh = raw_backward(kw, res, wts)
mag = _color_mag(fw, wts)["colors"]
for k in ["colors", "means2D", "opacity", "means3D", "sh", "scales", "rotations", "cov3D"]:
    o = go[k].astype(np.float64); r = npy(gr[k]).reshape(o.shape).astype(np.float64); r2 = npy(gr2[k]).reshape(o.shape).astype(np.float64); h = npy(gh[k]).reshape(o.shape).astype(np.float64)
    sc_ = np.abs(o).max()
    print(f"{k:10s} scale {sc_:.4g} | hip-oracle max {np.abs(h-o).max():.3g} mean {np.abs(h-o).mean():.3g} | hip-ref max {np.abs(h-r).max():.3g} mean {np.abs(h-r).mean():.3g} | oracle-ref max {np.abs(o-r).max():.3g} mean {np.abs(o-r).mean():.3g} | ref-ref(run2) max {np.abs(r-r2).max():.3g}")
o = go["colors"].astype(np.float64); h = npy(gh["colors"]).astype(np.float64); r = npy(gr["colors"]).astype(np.float64)
i = np.unravel_index(np.argmax(np.abs(h - o)), o.shape)
print("worst colors idx", i, "hip", h[i], "oracle", o[i], "ref", r[i], "mag", mag[i], "tiles_touched", fw.tiles_touched[i[0]], "radius", fw.radii[i[0]])
# contributions count
print("R", fw.num_rendered, "max list", (fw.ranges[:,1]-fw.ranges[:,0]).max(), "n_contrib max", fw.n_contrib.max())
