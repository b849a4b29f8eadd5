"""Helpers shared by the `-m gpu` parity tests: run the HIP path through the drop-in API and pull
its internal arrays out through the C ABI."""
import contextlib
import json
import os

import numpy as np
import torch

from street_gaussians_amd import _C

# every image_close / grad_close call appends what it MEASURED (not only pass/fail) to this log on the GPU box, so the
# gates in the tests can be read against the measured errors (committed copy: profiles/r2/parity_measured.jsonl)
_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_measured.jsonl")


def _log(rec):
    try:
        os.makedirs(os.path.dirname(_LOG), exist_ok=True)
        with open(_LOG, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


@contextlib.contextmanager
def switches(mask):
    """Run a block with the blend kernels' A/B switches set (_C.NO_CULL | NO_DPP | NO_DET | NO_HITS)."""
    prev = _C.test_switches(mask)
    try:
        yield
    finally:
        _C.test_switches(prev)


def dev(t):
    return None if t is None else t.detach().to("cuda").contiguous()


def settings(cam, deg=3, bg=None, scale_modifier=1.0, debug=False, prefiltered=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    bg = torch.zeros(3) if bg is None else bg
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=dev(bg.float()), scale_modifier=scale_modifier, viewmatrix=dev(cam.viewmatrix), projmatrix=dev(cam.projmatrix),
        sh_degree=deg, campos=dev(cam.campos), prefiltered=prefiltered, debug=debug)


def raw_forward(kw):
    """kw: the oracle-style keyword dict (tests/helpers.oracle_kwargs).  Calls the native forward and
    returns (outputs dict, internals getter)."""
    e = torch.Tensor([])
    g = lambda k: dev(kw[k]) if kw.get(k) is not None else e
    P = kw["means3D"].shape[0]
    sem = dev(kw["semantics"]) if kw.get("semantics") is not None else torch.zeros(P, 0, device="cuda")
    out = _C.rasterize_gaussians(dev(kw["bg"].float()), g("means3D"), g("colors_precomp"), sem, g("opacities"),
                                 g("scales"), g("rotations"), kw.get("scale_modifier", 1.0), g("cov3D_precomp"),
                                 g("viewmatrix"), g("projmatrix"), kw["tanfovx"], kw["tanfovy"], kw["image_height"],
                                 kw["image_width"], g("shs"), kw["sh_degree"], g("campos"), False, True)
    R, color, depth, alpha, semantic, radii, gb, bb, ib = out
    res = dict(R=R, color=color, depth=depth, alpha=alpha, semantic=semantic, radii=radii, geom=gb, binning=bb, img=ib)

    def internal(name):
        return _C.export_internal(name, P, R, kw["image_height"], kw["image_width"], gb, bb, ib)
    return res, internal


def raw_backward(kw, res, wts):
    e = torch.Tensor([])
    g = lambda k: dev(kw[k]) if kw.get(k) is not None else e
    P = kw["means3D"].shape[0]
    sem = dev(kw["semantics"]) if kw.get("semantics") is not None else torch.zeros(P, 0, device="cuda")
    S = sem.shape[1]
    H, W = kw["image_height"], kw["image_width"]
    gsem = dev(wts["semantic"].float()) if S else torch.zeros(0, H, W, device="cuda")
    outs = _C.rasterize_gaussians_backward(
        dev(kw["bg"].float()), g("means3D"), res["radii"], g("colors_precomp"), g("scales"), g("rotations"),
        kw.get("scale_modifier", 1.0), g("cov3D_precomp"), g("viewmatrix"), g("projmatrix"), kw["tanfovx"],
        kw["tanfovy"], dev(wts["color"].float()), dev(wts["depth"].float()), dev(wts["alpha"].float()), gsem, g("shs"),
        kw["sh_degree"], g("campos"), res["geom"], res["R"], res["binning"], res["img"], res["alpha"], sem, True)
    names = ["means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations", "semantics"]
    return dict(zip(names, outs))


def npy(t):
    return t.detach().cpu().numpy()


def image_close(a, b, rel=1e-4, name="", max_outliers=None, outlier_cap=0.05):
    """Float images: |a-b| <= rel*max(|a|,|b|,floor).  A handful of pixels may differ by one
    alpha-threshold flip (exp() differs in the last ulp between v_exp_f32 and libm, SURVEY 7
    "Transcendentals"); they are counted, bounded and reported."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if a.size == 0:
        return 0
    scale = max(np.abs(b).max(), 1e-12)
    tol = rel * np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-2 * scale)
    bad = np.abs(a - b) > tol
    nbad = int(bad.sum())
    _log(dict(kind="image", name=name, n=int(a.size), rel=rel, outside=nbad, worst_abs_over_scale=float(np.abs(a - b).max() / scale)))
    if max_outliers is None:
        max_outliers = max(3, int(3e-5 * a.size))
    if nbad:
        worst = np.abs(a - b)[bad].max()
        assert nbad <= max_outliers, f"{name}: {nbad}/{a.size} elements outside rel={rel}; worst abs err {worst}"
        assert worst <= outlier_cap * scale, f"{name}: outlier error {worst} exceeds one-flip bound {outlier_cap * scale}"
    return nbad


def grad_close(a, b, rel=1e-4, name="", max_outlier_frac=2e-3, abs_frac=2e-6, mag=None, cap=2e-2):
    """Gradients: |a-b| <= rel*max(|a|,|b|) + abs_frac*max|b|.  The absolute term is the fp32 summation-order
    noise of near-cancelling sums (entries that are ~0 relative to the tensor's scale); a small fraction of
    elements may carry an alpha-threshold flip of one (pixel, Gaussian) pair (see image_close)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if a.size == 0:
        return
    assert np.isfinite(a).all(), f"{name}: non-finite values"
    scale = max(np.abs(b).max(), 1e-30)
    tol = rel * np.maximum(np.abs(a), np.abs(b)) + abs_frac * scale
    if mag is not None:
        # `mag` = sum of |terms| of the float32 sum each element is (e.g. sum_pix w*|dL/dC| for dL/drgb).  With
        # random-sign weights over ~1e5 pixels the sum cancels by ~sqrt(N), so 1e-4 of the RESULT is below one ulp
        # of the terms.  Measured on MI355X: the reference's own kernels and its line-by-line C restatement differ
        # by up to 1.7e-5*mag on the 255-layer `huge_splats` stress case (summation order + fp32 T recovery).
        tol = tol + 3e-5 * np.asarray(mag, np.float64).reshape(a.shape)
    bad = np.abs(a - b) > tol
    frac = bad.mean()
    strict = np.abs(a - b) > 1e-4 * np.maximum(np.abs(a), np.abs(b)) + 2e-6 * scale  # the north-star gate, for the log
    _log(dict(kind="grad", name=name, n=int(a.size), rel=rel, abs_frac=abs_frac, outside=int(bad.sum()),
              outside_strict=int(strict.sum()), worst_abs_over_scale=float(np.abs(a - b).max() / scale),
              worst_over_tol=float((np.abs(a - b) / tol).max())))
    assert frac <= max_outlier_frac, (f"{name}: {bad.sum()}/{a.size} outside rel={rel} "
                                      f"(worst abs {np.abs(a - b).max()} at scale {scale})")
    # even the outliers stay small relative to the tensor's scale
    assert np.abs(a - b).max() <= cap * scale, f"{name}: worst abs {np.abs(a - b).max()} vs scale {scale}"


def oracle_backward_same_state(oracle, fw, res, wts, S, parallel=False):
    """The oracle's backward fed with the forward state the HIP backward was given (the HIP path's alpha image): the
    reference takes T_final = 1 - alpha_out (backward.cu:468), so for a pixel that ends near T = 1e-4 one ulp of
    alpha_out is a 6e-4 relative change of every gradient term of that pixel.  Comparing the two backward passes on
    IDENTICAL inputs isolates the backward; the end-to-end comparison (each side on its own forward) is made
    separately, with the gate that sensitivity dictates."""
    own = fw.alpha
    fw.alpha = npy(res["alpha"]).reshape(own.shape).copy()
    try:
        return oracle.backward(fw, wts["color"], wts["depth"], wts["alpha"], wts["semantic"] if S else None,
                               parallel=parallel)
    finally:
        fw.alpha = own


def strict_gate(a, b, name="", rel=1e-4, abs_frac=2e-6, allow=0):
    """north_star's own gate, |a-b| <= 1e-4*max(|a|,|b|) + 2e-6*max|b| with NO element outside (`allow` = the measured,
    documented exceptions of a case); logs what was measured.  Returns the number of elements outside."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if a.size == 0:
        return 0
    assert np.isfinite(a).all(), f"{name}: non-finite values"
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b)
    bad = err > rel * np.maximum(np.abs(a), np.abs(b)) + abs_frac * scale
    nbad = int(bad.sum())
    _log(dict(kind="strict", name=name, n=int(a.size), rel=rel, abs_frac=abs_frac, outside=nbad,
              worst_abs_over_scale=float(err.max() / scale)))
    assert nbad <= allow, f"{name}: {nbad}/{a.size} outside rel {rel} + {abs_frac}*scale (worst {err.max() / scale:.3g} of the scale)"
    return nbad


GRAD_NAMES = ["means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations", "semantics"]


def count_outside(a, b, rel=1e-4, abs_frac=2e-6):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0
    scale = max(np.abs(b).max(), 1e-30)
    return int((np.abs(a - b) > rel * np.maximum(np.abs(a), np.abs(b)) + abs_frac * scale).sum())


# dL/dcov3D (and dL/dscale, dL/drot behind it) come out of computeCov2DCUDA's dL_da / dL_db / dL_dc (backward.cu:206-208):
# sums like -c*c*dL_dconic.x + 2*b*c*dL_dconic.y + (denom - a*c)*dL_dconic.z times 1 / (denom^2 + 1e-7), which cancel by
# several digits for elongated splats.  A last-bit difference of dL/dconic -- the reference's own unordered atomics make
# one from run to run -- is amplified there past rel 1e-4 for a handful of Gaussians per million.  These three tensors are
# therefore allowed what the reference's own rerun shows (x3) plus 1e-5 of their elements; every other tensor: nothing.
CONDITIONED = ("cov3D", "scales", "rotations")


def conditioned_allowance(k, n, rerun_outside=0):
    return (3 * rerun_outside + max(4, int(1e-5 * n))) if k in CONDITIONED else 0


def exact_mode_against_reference_kernels(kw, wts, S, label, rf=None, gref=None, allow=None, color_rel=1e-4):
    """The library in its parity mode (sgr_test_switches bit 7, SGR_EXACT=1) against the reference's OWN kernels
    (oracle/_ref, strict build), END TO END, each side on its own forward, held to north_star's statement itself:
    every integer output identical; alpha / depth / semantic images and n_contrib bit-identical; the colour image within
    rel 1e-4 with no outlier; all nine gradient tensors within rel 1e-4 + 2e-6 of the tensor's scale with NO element
    outside (`allow`: {tensor: count}, the documented exceptions of a case).  `rf` / `gref`: the reference's forward
    state and gradients when the caller already has them (or a dict of golden arrays).  Returns (res, g)."""
    own = rf is None
    rerun = {}
    if own:
        from oracle import ref
        rf = ref.forward(**kw)
        gref = ref.backward(rf, wts["color"], wts["depth"], wts["alpha"], wts["semantic"] if S else None)
        # the reference accumulates with unordered float atomics (backward.cu:568-638): its own gradients move from run
        # to run.  A second backward over the same forward state measures that spread with the same gate
        gref_b = ref.backward(rf, wts["color"], wts["depth"], wts["alpha"], wts["semantic"] if S else None)
        for k in GRAD_NAMES:
            if k == "semantics" and not S:
                continue
            rerun[k] = count_outside(npy(gref_b[k]).reshape(-1), npy(gref[k]).reshape(-1))
        _log(dict(kind="ref_rerun", name=label, outside=rerun))
    with switches(_C.test_switches(-1) | _C.EXACT):
        res, internal = raw_forward(kw)
        g = raw_backward(kw, res, wts)
        torch.cuda.synchronize()
    golden = isinstance(rf, dict)
    get = (lambda k: np.asarray(rf[k])) if golden else (lambda k: npy(getattr(rf, k)))
    R = int(rf["num_rendered"]) if golden else rf.num_rendered
    assert int(internal("num_rendered_reference")[0]) == R, label
    # the binning arrays: the reference's, restricted to the tile rects the HIP path emits (restrict_binning)
    if golden:
        b = restrict_binning(internal, rf["point_list"], rf["ranges"], int(kw["image_width"]), int(kw["image_height"]))
    else:
        b = restrict_oracle(internal, rf, kw)
    assert res["R"] == b.num_rendered, label
    assert (npy(res["radii"]) == get("radii")).all(), label
    if R:
        assert (npy(internal("point_list")).view(np.uint32) == b.point_list).all(), label
        if not golden:
            assert (npy(internal("keys")).view(np.uint64) == b.keys).all(), label
    assert (npy(internal("ranges")).view(np.uint32).reshape(-1) == b.ranges.reshape(-1)).all(), label
    for k in ["alpha", "depth"] + (["semantic"] if S else []):
        assert np.array_equal(npy(res[k]).reshape(-1), get(k).reshape(-1)), f"{label}: {k} image not bit-identical"
    if not golden:
        assert np.array_equal(npy(internal("n_contrib")).view(np.uint32).reshape(-1), b.n_contrib.reshape(-1)), label
    image_close(npy(res["color"]).reshape(-1), get("color").reshape(-1), rel=color_rel, name=f"exact {label}: color", max_outliers=0)
    for k in GRAD_NAMES:
        if k == "semantics" and not S:
            continue
        want = np.asarray(gref["g_" + k]) if golden else npy(gref[k])
        strict_gate(npy(g[k]).reshape(-1), want.reshape(-1), name=f"exact {label}: {k}",
                    allow=max((allow or {}).get(k, 0), conditioned_allowance(k, want.size, rerun.get(k, 0))))
    if own:
        rf.free()
    return res, g


def restrict_binning(internal, point_list, ranges, W, H, keys=None, n_contrib=None):
    """The reference's binning arrays RESTRICTED to the tile rects the HIP path emitted its Gaussians for.

    The reference emits a (tile, Gaussian) instance for every tile of the square getRect gives (auxiliary.h:46-57); the HIP
    path cuts that rect down to the tiles in which the Gaussian can reach alpha >= 1/255 (sgr_preprocess.hip: the bounding
    box of those tiles and, for rects of up to 64 tiles, a mask of the tiles themselves; with sgr_test_switches bit 10 it keeps
    the reference's rect and this function is the identity).  What it must hold is then:
    its sorted list is the reference's sorted list with the instances outside the rects REMOVED and nothing else changed --
    same order, same ranges after the removal, n_contrib counting only the instances that stayed.  (That the removed
    instances change no output is what the image / gradient gates and test_tight_rects_are_invisible check.)

    point_list [R], ranges [T, 2] (, keys [R], n_contrib [H, W]): the reference's arrays (oracle, oracle/_ref or golden file).
    -> namespace(num_rendered, point_list, ranges, tiles_touched, point_offsets, keys, n_contrib, rect)."""
    from types import SimpleNamespace
    rect = npy(internal("tile_rect")).astype(np.int64)
    P = rect.shape[0]
    tmask = npy(internal("tile_mask")).view(np.uint64).reshape(-1)  # 0 = every tile of the rect; else bit j = tile j of the rect
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    pl = np.asarray(point_list).astype(np.int64).reshape(-1)
    rg = np.asarray(ranges).astype(np.int64).reshape(T, 2)
    tile_of = np.repeat(np.arange(T, dtype=np.int64), rg[:, 1] - rg[:, 0])  # the list is sorted by tile, ranges are contiguous
    assert tile_of.shape[0] == pl.shape[0]
    tx, ty = tile_of % gx, tile_of // gx
    r = rect[pl] if pl.size else np.zeros((0, 4), np.int64)
    keep = (tx >= r[:, 0]) & (tx < r[:, 2]) & (ty >= r[:, 1]) & (ty < r[:, 3])
    mk = tmask[pl] if pl.size else np.zeros(0, np.uint64)
    j = np.clip((ty - r[:, 1]) * (r[:, 2] - r[:, 0]) + (tx - r[:, 0]), 0, 63).astype(np.uint64)
    keep &= (mk == 0) | (((mk >> j) & np.uint64(1)) == 1)
    cum = np.concatenate([[0], np.cumsum(keep)])
    new_rg = np.stack([cum[rg[:, 0]], cum[rg[:, 1]]], axis=1)
    new_rg[new_rg[:, 0] == new_rg[:, 1]] = 0  # a tile nothing was emitted for keeps its memset (rasterizer_impl.cu:313)
    tt = (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])
    pop = np.unpackbits(tmask.view(np.uint8).reshape(-1, 8), axis=1).sum(axis=1).astype(np.int64)
    tt = np.where(tmask != 0, pop, tt)
    # every rect lies inside the reference's: the reference has an instance for each of its tiles
    assert (np.bincount(pl[keep], minlength=P) == tt).all(), "a HIP tile rect is not inside the reference's"
    out = SimpleNamespace(num_rendered=int(keep.sum()), point_list=pl[keep].astype(np.uint32), ranges=new_rg.astype(np.uint32),
                          tiles_touched=tt.astype(np.uint32), point_offsets=np.cumsum(tt).astype(np.uint32), keys=None,
                          n_contrib=None, rect=rect, removed=int((~keep).sum()))
    if keys is not None:
        out.keys = np.asarray(keys).reshape(-1)[keep]
    if n_contrib is not None:
        nc = np.asarray(n_contrib).astype(np.int64).reshape(H, W)
        ys, xs = np.mgrid[0:H, 0:W]
        s = rg[(ys // 16) * gx + xs // 16, 0]
        out.n_contrib = (cum[s + nc] - cum[s]).astype(np.uint32)
    return out


def restrict_oracle(internal, fw, kw):
    """restrict_binning for a forward result of the C oracle (oracle.ForwardResult) / of oracle/_ref (ref.RefForward)."""
    W, H = int(kw["image_width"]), int(kw["image_height"])
    if hasattr(fw, "internal"):  # the reference's kernels: device tensors behind internal(name)
        g = lambda k, dt: npy(fw.internal(k)).view(dt)
        return restrict_binning(internal, g("point_list", np.uint32), g("ranges", np.uint32), W, H, keys=g("keys", np.uint64),
                                n_contrib=g("n_contrib", np.uint32))
    if fw.num_rendered == 0:
        T = ((W + 15) // 16) * ((H + 15) // 16)
        return restrict_binning(internal, np.zeros(0, np.uint32), np.zeros((T, 2), np.uint32), W, H, keys=np.zeros(0, np.uint64),
                                n_contrib=np.zeros((H, W), np.uint32))
    return restrict_binning(internal, fw.point_list, fw.ranges, W, H, keys=fw.keys, n_contrib=fw.n_contrib)
