"""Generates the golden fixtures tests/golden/*.npz from the REFERENCE'S OWN KERNELS (oracle/_ref:
the untouched CUDA sources of /root/reference compiled for gfx950 by oracle/ref_build.sh) run on an
MI355X box:

    gpurun -- 'python tests/golden/make_golden.py gpurun_out/golden'   # then copy the .npz files here

Each file holds the inputs (seeded synthetic scene + camera), the per-pixel loss weights and the
reference's outputs: images, radii, num_rendered, sorted point_list, ranges and every gradient.
tests/test_gpu_parity.py::test_golden_fixture checks the HIP path against them and
tests/test_golden_cpu.py checks the C oracle against them (no GPU needed) -- this is what pins the oracle.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENE_KEYS = ["means3D", "opacities", "shs", "scales", "rotations", "semantics", "viewmatrix", "projmatrix", "campos", "bg"]
GRADS = ["means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations", "semantics"]


def cases():
    from street_gaussians_amd import synthetic as syn
    out = {}
    cam = syn.make_camera(112, 80, fx=120.0, yaw_deg=4.0, translation=(0.05, -0.02, 0.1))
    out["g1_sh3_sem2"] = (cam, syn.make_scene(1500, cam, S=2, seed=21, zmax=20.0, scale_px=0.006), 3, [0.3, 0.1, 0.7])
    cam = syn.make_camera(96, 64, fx=90.0)
    sc = syn.make_scene(1200, cam, S=0, seed=22, zmax=8.0, scale_px=0.03, margin=1.5)
    sc.opacities.fill_(0.9)
    sc.shs[::2, 0, :] -= 2.0
    out["g2_dense_clamped_offscreen"] = (cam, sc, 2, [0.0, 0.0, 0.0])
    cam = syn.make_camera(80, 80, fx=70.0, yaw_deg=-6.0)
    out["g3_deg0_sem5_whitebg"] = (cam, syn.make_scene(1000, cam, S=5, seed=23, zmax=15.0, scale_px=0.01), 0, [1.0, 1.0, 1.0])
    return out


def to_kw(d):
    S = d["semantics"].shape[1]
    kw = dict(means3D=torch.from_numpy(d["means3D"]), opacities=torch.from_numpy(d["opacities"]),
              viewmatrix=torch.from_numpy(d["viewmatrix"]), projmatrix=torch.from_numpy(d["projmatrix"]),
              campos=torch.from_numpy(d["campos"]), bg=torch.from_numpy(d["bg"]), tanfovx=float(d["tanfovx"]),
              tanfovy=float(d["tanfovy"]), image_height=int(d["H"]), image_width=int(d["W"]),
              sh_degree=int(d["deg"]), shs=torch.from_numpy(d["shs"]), scales=torch.from_numpy(d["scales"]),
              rotations=torch.from_numpy(d["rotations"]))
    if S:
        kw["semantics"] = torch.from_numpy(d["semantics"])
    return kw


def load_case(path):
    d = dict(np.load(path))
    kw = to_kw(d)
    wts = {k: torch.from_numpy(d["w_" + k]) for k in ["color", "depth", "alpha", "semantic"]}
    return kw, wts, d


def main(outdir):
    from oracle import ref
    from street_gaussians_amd import synthetic as syn
    os.makedirs(outdir, exist_ok=True)
    for name, (cam, sc, deg, bg) in cases().items():
        S = sc.semantics.shape[1]
        d = dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations,
                 semantics=sc.semantics, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, campos=cam.campos,
                 bg=torch.tensor(bg))
        d = {k: v.numpy().astype(np.float32) for k, v in d.items()}
        d.update(tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, H=cam.image_height, W=cam.image_width, deg=deg)
        kw = to_kw(d)
        wts = syn.loss_weights(cam, S=S)
        rf = ref.forward(**kw)
        g = ref.backward(rf, wts["color"], wts["depth"], wts["alpha"], wts["semantic"])
        c = lambda t: t.detach().cpu().numpy()
        d.update(color=c(rf.color), depth=c(rf.depth), alpha=c(rf.alpha), semantic=c(rf.semantic),
                 radii=c(rf.radii), num_rendered=rf.num_rendered,
                 point_list=c(rf.internal("point_list")).view(np.uint32), ranges=c(rf.internal("ranges")).view(np.uint32),
                 n_contrib=c(rf.internal("n_contrib")).view(np.uint32),
                 tiles_touched=c(rf.internal("tiles_touched")).view(np.uint32))
        for k in GRADS:
            d["g_" + k] = c(g[k])
        for k, v in wts.items():
            d["w_" + k] = v.numpy().astype(np.float32)
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **d)
        print(name, "R", rf.num_rendered, "visible", int((rf.radii > 0).sum()), "bytes",
              os.path.getsize(os.path.join(outdir, name + ".npz")))
        rf.free()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden"))
