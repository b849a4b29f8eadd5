"""`-m gpu`: the forward's COMPACT hit list (include/sgr.h exports 19 / 20) and the blend backward that walks it.

The reference's backward steps through every entry of a tile's list behind the tile's largest n_contrib and lets each pixel
test "contributor < my last contributor" (cuda_rasterizer/backward.cu:476-545).  Here the forward leaves, next to the hit
record (export 15), the ascending list of the positions whose hit byte is non-zero and, per pixel, n_contrib counted in
entries of that list; the backward stages only those entries.  The tests hold
  * the list against its definition from the hit record and n_contrib (numpy, entry for entry),
  * the gradients of the compact walk against the positional walk (switch bit 16) bit for bit -- the positional walk is what
    the rest of the suite holds against the oracle and the reference's kernels -- in the three modes, with semantics, with an
    image that is not a multiple of the tile, with lists that end inside a round and tiles that saturate early."""
import numpy as np
import pytest
import torch

from gpu_utils import npy, raw_backward, raw_forward, switches
from helpers import oracle_kwargs
from street_gaussians_amd import _C
from street_gaussians_amd import synthetic as syn

pytestmark = pytest.mark.gpu

MODES = {"fast": 0, "exact": _C.EXACT, "strict": _C.EXACT | _C.REF_RECT}


def _scene(name):
    if name == "mid":
        cam = syn.make_camera(480, 320, fx=512.5, yaw_deg=2.0)
        return cam, syn.make_scene(40000, cam, S=3, seed=3), 3
    if name == "ragged":  # 150 x 90: partial tiles on two edges
        cam = syn.make_camera(150, 90, fx=160.0)
        return cam, syn.make_scene(9000, cam, S=0, seed=11, scale_px=0.02), 0
    if name == "saturating":  # fat opaque splats: tiles finish long before their lists end
        cam = syn.make_camera(192, 128, fx=200.0)
        sc = syn.make_scene(60000, cam, S=0, seed=5, scale_px=0.02, zmin=1.0, zmax=20.0)
        sc.opacities.fill_(0.97)
        return cam, sc, 0
    if name == "sparse":  # a few hundred splats: most tiles hold less than one round, many are empty
        cam = syn.make_camera(320, 208, fx=340.0)
        return cam, syn.make_scene(400, cam, S=5, seed=17, scale_px=0.03), 5
    raise KeyError(name)


def _run(kw, wts, mask):
    with switches(mask):
        res, internal = raw_forward(kw)
        g = raw_backward(kw, res, wts)
        torch.cuda.synchronize()
        out = {k: npy(v).copy() for k, v in g.items()}
        arrays = {k: npy(internal(k)).copy() for k in ["ranges", "n_contrib", "hits", "hit_list", "n_contrib_k"]}
    return out, arrays, (kw["image_height"], kw["image_width"])


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", ["mid", "ragged", "saturating", "sparse"])
def test_compact_walk_equals_positional_walk_and_the_list_is_what_it_says(name, mode):
    cam, sc, S = _scene(name)
    kw = oracle_kwargs(cam, sc, bg=torch.tensor([0.3, 0.0, 0.1]))
    wts = syn.loss_weights(cam, S=S)
    g_new, arr, (H, W) = _run(kw, wts, MODES[mode] | _C.HLIST_ALWAYS)  # (default: with the reference's rects only)
    g_old, _, _ = _run(kw, wts, MODES[mode] | _C.NO_HLIST)
    for k in g_old:
        assert np.array_equal(g_new[k], g_old[k]), (name, mode, k, int((g_new[k] != g_old[k]).sum()))

    # the list against its definition
    ranges = arr["ranges"].view(np.uint32).reshape(-1, 2).astype(np.int64)
    nc = arr["n_contrib"].view(np.uint32).reshape(H, W).astype(np.int64)
    nk = arr["n_contrib_k"].view(np.uint32).reshape(H, W).astype(np.int64)
    hits = arr["hits"]
    hl = arr["hit_list"].view(np.uint32).astype(np.int64)
    gx = (W + 15) // 16
    checked = 0
    for t, (a, b) in enumerate(ranges):
        ty, tx = divmod(t, gx)
        blk_c = nc[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
        blk_k = nk[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
        top = int(blk_c.max()) if blk_c.size else 0
        if top == 0:
            assert int(blk_k.max(initial=0)) == 0
            continue
        pos = np.nonzero(hits[a:a + top])[0]  # the record is defined for the batches the tile processed: covers [0, top)
        assert np.array_equal(hl[a:a + len(pos)], pos), (name, mode, t)
        # per pixel: entries of the list in front of its last contributor + 1 ... and the last contributor is in the list
        want = np.searchsorted(pos, blk_c, side="left")  # = #{entries with position < n_contrib}
        assert np.array_equal(blk_k, want), (name, mode, t)
        lastpos = blk_c[blk_c > 0] - 1
        assert np.all(hits[a + lastpos] != 0)
        checked += 1
    assert checked > 5


def test_compact_walk_is_what_runs_by_default_and_shortens_the_walk():
    """In the strict mode about 40 % of the entries are marked dead and more lie behind saturated pixels: the compact list
    must be clearly shorter than the positional range the round-5 backward stepped through."""
    cam, sc, S = _scene("mid")
    kw = oracle_kwargs(cam, sc)
    wts = syn.loss_weights(cam, S=S)
    _, arr, (H, W) = _run(kw, wts, MODES["strict"])
    nc = arr["n_contrib"].view(np.uint32).reshape(H, W).astype(np.int64)
    nk = arr["n_contrib_k"].view(np.uint32).reshape(H, W).astype(np.int64)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pos_walk = sum(int(nc[y * 16:(y + 1) * 16, x * 16:(x + 1) * 16].max()) for y in range(gy) for x in range(gx))
    cmp_walk = sum(int(nk[y * 16:(y + 1) * 16, x * 16:(x + 1) * 16].max()) for y in range(gy) for x in range(gx))
    assert 0 < cmp_walk < 0.8 * pos_walk, (cmp_walk, pos_walk)
    assert not (_C.test_switches(-1) & _C.NO_HLIST)


def test_backward_follows_what_its_frames_forward_did():
    """Whether a frame has a compact hit list is recorded in its own buffers (word 7 of the geometry header): a backward that
    runs under other switches than its forward -- list wanted, frame has none; list not wanted, frame has one -- must give
    the same gradients as the matching pair, not gather through a list that is not there."""
    cam, sc, S = _scene("mid")
    kw = oracle_kwargs(cam, sc)
    wts = syn.loss_weights(cam, S=S)
    with switches(_C.EXACT):  # cut-down rects: no list by default
        res, _ = raw_forward(kw)
        want = {k: npy(v).copy() for k, v in raw_backward(kw, res, wts).items()}
    with switches(_C.EXACT | _C.HLIST_ALWAYS):  # the backward would like a list; this frame has none
        got = {k: npy(v).copy() for k, v in raw_backward(kw, res, wts).items()}
    for k in want:
        assert np.array_equal(want[k], got[k]), k
    with switches(_C.EXACT | _C.HLIST_ALWAYS):  # a frame WITH a list ...
        res2, _ = raw_forward(kw)
    with switches(_C.EXACT):  # ... walked by a backward that would not have asked for one
        got2 = {k: npy(v).copy() for k, v in raw_backward(kw, res2, wts).items()}
    with switches(_C.EXACT | _C.NO_HLIST):  # ... and by one that refuses it
        got3 = {k: npy(v).copy() for k, v in raw_backward(kw, res2, wts).items()}
    for k in want:
        assert np.array_equal(want[k], got2[k]), k
        assert np.array_equal(want[k], got3[k]), k
