"""Shared helpers for the parity tests."""
import numpy as np
import torch

from street_gaussians_amd import synthetic as syn


def small_case(P=60, W=48, H=40, S=2, seed=0, fx=None, deg=3, zmin=1.0, zmax=20.0, scale_px=0.004):
    """Tiny scene with ragged edge tiles (W, H not multiples of 16)."""
    fx = fx if fx is not None else W * 1.1
    cam = syn.make_camera(W, H, fx=fx, yaw_deg=3.0, translation=(0.05, -0.02, 0.1))
    sc = syn.make_scene(P, cam, sh_degree_max=3, S=S, seed=seed, zmin=zmin, zmax=zmax, scale_px=scale_px)
    return cam, sc


def oracle_kwargs(cam, sc, deg=3, bg=None, use_sh=True, use_cov_precomp=False, semantics=True, colors=None,
                  cov3D=None, scale_modifier=1.0):
    kw = dict(means3D=sc.means3D, opacities=sc.opacities, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
              campos=cam.campos, bg=torch.zeros(3) if bg is None else bg, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
              image_height=cam.image_height, image_width=cam.image_width, sh_degree=deg,
              scale_modifier=scale_modifier)
    if use_sh:
        kw["shs"] = sc.shs
    else:
        kw["colors_precomp"] = colors
    if use_cov_precomp:
        kw["cov3D_precomp"] = cov3D
    else:
        kw["scales"], kw["rotations"] = sc.scales, sc.rotations
    if semantics and sc.semantics.shape[1] > 0:
        kw["semantics"] = sc.semantics
    return kw


def rel_err(a, b, floor=1e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), floor)


def assert_close(a, b, rel=1e-4, floor=1e-5, name="", max_outlier_frac=0.0):
    """|a-b| <= rel*max(|a|,|b|,floor_scale) -- the SURVEY 8c float gate.  floor is scaled by the
    tensor's own magnitude so exact zeros compare cleanly."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{name}: shape {a.shape} vs {b.shape}"
    if a.size == 0:
        return
    scale = max(np.abs(b).max(), 1e-30)
    tol = rel * np.maximum(np.maximum(np.abs(a), np.abs(b)), floor * scale / rel if False else floor * scale)
    bad = np.abs(a - b) > tol
    frac = bad.mean()
    if frac > max_outlier_frac:
        i = np.unravel_index(np.argmax(np.abs(a - b) / tol), a.shape)
        raise AssertionError(f"{name}: {bad.sum()}/{a.size} outside rel={rel} (worst at {i}: {a[i]} vs {b[i]})")
