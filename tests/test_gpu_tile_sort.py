"""`-m gpu`: the per-tile LDS sort form of the binning chain (sgr_test_switches bit 12, csrc/sgr_tile_sort.hip: index-order
emission, stable tile sort, one radix sort by depth per tile in LDS) must produce the SAME sorted lists as the default chain
(depth pre-sort of the Gaussians, depth-order emission, stable tile sort) -- which the rest of the suite holds entry for
entry against the oracle and the reference's kernels (rasterizer_impl.cu:70-138,286-321): keys, point_list, ranges,
n_contrib, every image and every gradient bit for bit.  Covers the three code paths of the kernel (one-wave rank sort for
lists up to 64 entries, 256-thread LDS sort up to 2048, 1024-thread LDS sort up to 7168, the HBM ping-pong beyond), equal
depths (the tie order is ascending Gaussian id), far depths (32-bit keys) and the reference's tile rects."""
import numpy as np
import pytest
import torch

from gpu_utils import npy, raw_backward, raw_forward, switches
from helpers import oracle_kwargs
from street_gaussians_amd import _C
from street_gaussians_amd import synthetic as syn

pytestmark = pytest.mark.gpu

INTERNALS = [("keys", np.uint64), ("point_list", np.uint32), ("ranges", np.uint32), ("n_contrib", np.uint32),
             ("point_offsets", np.uint32), ("tiles_touched", np.uint32), ("hits", np.uint8)]


def _run(kw, wts, mask, hits_defined=None):
    with switches(mask):
        res, internal = raw_forward(kw)
        g = raw_backward(kw, res, wts)
        torch.cuda.synchronize()
        out = {"R": res["R"]}
        for k in ["color", "depth", "alpha", "semantic", "radii"]:
            out[k] = npy(res[k]).copy()
        for k, dt in INTERNALS:
            out["i_" + k] = npy(internal(k)).view(dt).copy()
        for k, v in g.items():
            out["g_" + k] = npy(v).copy()
    return out


def _same(a, b, label):
    assert a["R"] == b["R"], label
    # the hit record is defined up to the last batch a tile processed (include/sgr.h, export 15): compare where both forms
    # processed the list -- positions below the tile's highest n_contrib
    for k in a:
        if k == "R" or k == "i_hits":
            continue
        assert np.array_equal(a[k], b[k]), (label, k, int((np.asarray(a[k]) != np.asarray(b[k])).sum()))


def _scene(name):
    if name == "mid":  # a few hundred entries per tile: the 256-thread LDS sort and the one-wave rank sort
        cam = syn.make_camera(480, 320, fx=512.5, yaw_deg=2.0)
        return cam, syn.make_scene(40000, cam, S=3, seed=3), 3
    if name == "fat_tiles":  # 12 x 8 tiles, ~60 k splats of ~10 px: thousands of entries per tile -> the 1024-thread LDS sort
        cam = syn.make_camera(192, 128, fx=200.0)
        return cam, syn.make_scene(60000, cam, S=0, seed=5, scale_px=0.02, zmin=1.0, zmax=20.0), 0
    if name == "beyond_lds":  # > 7168 entries in a tile: the HBM ping-pong
        cam = syn.make_camera(64, 48, fx=70.0)
        return cam, syn.make_scene(50000, cam, S=0, seed=6, scale_px=0.03, zmin=1.0, zmax=12.0), 0
    if name == "equal_depths":  # a wall of splats at a handful of depths: almost every comparison is a tie
        cam = syn.make_camera(320, 208, fx=340.0)
        sc = syn.make_scene(30000, cam, S=0, seed=7, scale_px=0.01, zmin=2.0, zmax=9.0)
        z = torch.round(sc.means3D[:, 2])
        sc.means3D[:, 0] *= z / sc.means3D[:, 2]
        sc.means3D[:, 1] *= z / sc.means3D[:, 2]
        sc.means3D[:, 2] = z
        return cam, sc, 0
    if name == "far":  # depths beyond the 27-bit keys: the kernel sorts on all 32 bits
        cam = syn.make_camera(160, 96, fx=170.0)
        sc = syn.make_scene(4000, cam, S=0, seed=31, scale_px=0.02, zmin=1.0, zmax=8.0)
        far = sc.means3D.clone()
        n = far[::7].shape[0]
        far[::7, :] = far[::7, :] * torch.tensor([1.0, 1.0, 0.0]) * 3000.0 + torch.tensor([0.0, 0.0, 1.0]) * torch.linspace(13000.0, 90000.0, n)[:, None]
        sc.means3D.copy_(far)
        sc.scales[::7] *= 4000.0
        return cam, sc, 0
    raise KeyError(name)


@pytest.mark.parametrize("mode", ["default", "strict"])
@pytest.mark.parametrize("name", ["mid", "fat_tiles", "beyond_lds", "equal_depths", "far"])
def test_tile_sort_form_gives_the_default_chains_lists(name, mode):
    cam, sc, S = _scene(name)
    kw = oracle_kwargs(cam, sc, bg=torch.tensor([0.2, 0.1, 0.4]))
    wts = syn.loss_weights(cam, S=S)
    base = (_C.EXACT | _C.REF_RECT) if mode == "strict" else 0
    a = _run(kw, wts, base)
    b = _run(kw, wts, base | _C.TILE_SORT)
    _same(a, b, f"{name}/{mode}")
    # the test must have exercised what its name says
    rg = a["i_ranges"].reshape(-1, 2).astype(np.int64)
    longest = int((rg[:, 1] - rg[:, 0]).max())
    want = {"mid": (65, 2048), "fat_tiles": (2049, 7168), "beyond_lds": (7169, 10 ** 9), "equal_depths": (65, 10 ** 9),
            "far": (2, 10 ** 9)}[name]
    assert want[0] <= longest <= want[1], (name, longest)
    if name == "equal_depths":
        keys = a["i_keys"]
        assert int((keys[1:] == keys[:-1]).sum()) > 1000  # real ties in the 64-bit keys


def test_tile_sort_form_is_deterministic_over_repeated_forwards():
    cam, sc, S = _scene("fat_tiles")
    kw = oracle_kwargs(cam, sc)
    wts = syn.loss_weights(cam, S=S)
    first = _run(kw, wts, _C.TILE_SORT)
    for _ in range(3):
        _same(first, _run(kw, wts, _C.TILE_SORT), "repeat")


@pytest.mark.parametrize("mode", ["default", "strict"])
def test_longest_first_tile_order_changes_nothing(mode):
    """The tile order of the blend launches (XCD-aware supertiles / longest list first / decided per frame: sgr_test_switches
    bits 15 / 14 / default) changes no output and no internal array, bit for bit (tiles are independent) -- on a street-like
    scene with an empty sky and actor clusters composed by scene.compose."""
    from street_gaussians_amd import scene as sg
    cam = syn.make_camera(480, 320, fx=512.5)
    raw = syn.make_street_segments(60000, cam, n_actors=5, S=3, seed=2)
    segs = [sg.Segment(**{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items() if v is not None}) for d in raw]
    with torch.no_grad():
        means, rot, scl, opa, shs, sem = sg.compose(segs, 16, 3)
    sc = syn.Scene(means.cpu(), scl.cpu(), rot.cpu(), opa.cpu(), shs.cpu(), sem.cpu())
    kw = oracle_kwargs(cam, sc)
    wts = syn.loss_weights(cam, S=3)
    base = (_C.EXACT | _C.REF_RECT) if mode == "strict" else 0
    a = _run(kw, wts, base | _C.NO_LPT)          # XCD-aware supertile order, no look at the lists
    b = _run(kw, wts, base | _C.LPT)             # always longest-first
    c = _run(kw, wts, base)                      # decided per frame (this scene: longest-first)
    _same(a, b, "lpt/" + mode)
    _same(a, c, "adaptive/" + mode)
    rg = a["i_ranges"].reshape(-1, 2).astype(np.int64)
    n = rg[:, 1] - rg[:, 0]
    assert (n == 0).sum() > 10 and n.max() > 4 * n.mean()  # empty sky tiles and tiles several times the mean


@pytest.mark.parametrize("mode", ["default", "strict"])
def test_32_bit_tile_keys_give_the_same_lists_as_the_16_bit_default(mode):
    """The instance list's tile keys are 16-bit whenever the frame has fewer than 65535 tiles (the tile sort then moves 6
    instead of 8 bytes per pair and pass); switch bit 18 keeps the 32-bit keys that larger frames take anyway.  Same keys,
    lists, ranges, images and gradients bit for bit -- also through the per-tile sort form, whose scratch shares the key
    buffers."""
    cam, sc, S = _scene("mid")
    kw = oracle_kwargs(cam, sc, bg=torch.tensor([0.2, 0.1, 0.4]))
    wts = syn.loss_weights(cam, S=S)
    base = (_C.EXACT | _C.REF_RECT) if mode == "strict" else 0
    a = _run(kw, wts, base)
    b = _run(kw, wts, base | _C.KEY32)
    _same(a, b, "key32/" + mode)
    c = _run(kw, wts, base | _C.KEY32 | _C.TILE_SORT)
    _same(a, c, "key32+tile_sort/" + mode)
    assert int(a["i_keys"].max() >> 32) < 65535  # the frame is one the default sorts on 16-bit keys
