"""The maths header the gfx950 kernels run (street_gaussians_amd/csrc/sgr_math.h), compiled for the
host, against the oracle: integer outputs and the geometry floats must be BIT-EXACT (contraction is
off on both sides), per-Gaussian gradients within 1e-4.  No GPU needed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from helpers import assert_close, oracle_kwargs, small_case
from oracle import oracle
from street_gaussians_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_math", "host_math.hip")
HDR = os.path.join(HERE, "..", "street_gaussians_amd", "csrc", "sgr_math.h")
LIB = os.path.join(HERE, "_build", "libsgr_hostmath.so")


@pytest.fixture(scope="module")
def hm():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-fPIC", "-shared",
                               "-std=c++17", SRC, "-o", LIB])
    L = C.CDLL(LIB)
    L.hm_power2.restype = C.c_float
    L.hm_power2.argtypes = [C.c_float] * 5
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(t):
    return np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)


@pytest.mark.parametrize("deg,margin,scale_px", [(3, 1.1, 0.01), (1, 1.7, 0.03), (0, 1.1, 0.002)])
def test_forward_geometry_bit_exact_and_backward_close(hm, deg, margin, scale_px):
    cam = syn.make_camera(200, 120, fx=210.0, yaw_deg=7.0, translation=(0.1, -0.05, 0.3))
    sc = syn.make_scene(5000, cam, S=0, seed=11, margin=margin, zmin=0.15, zmax=30.0, scale_px=scale_px)
    sc.shs[::3, 0, :] -= 2.5  # some clamped colours
    fw = oracle.forward(**oracle_kwargs(cam, sc, deg=deg))
    P, M = sc.P, sc.shs.shape[1]
    W, H = cam.image_width, cam.image_height
    means, scales, rots, opac, shs = map(_f, (sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs))
    view, proj, campos = map(_f, (cam.viewmatrix, cam.projmatrix, cam.campos))
    radii = np.zeros(P, np.int32); m2d = np.zeros((P, 2), np.float32); depths = np.zeros(P, np.float32)
    conic = np.zeros((P, 3), np.float32); cov3D = np.zeros((P, 6), np.float32); tiles = np.zeros(P, np.uint32)
    rgb = np.zeros((P, 3), np.float32); clamped = np.zeros((P, 3), np.uint8); ext = np.zeros((P, 2), np.float32)
    hm.hm_forward(P, deg, M, _p(means), _p(scales), _p(rots), _p(opac), _p(shs), _p(view), _p(proj), _p(campos),
                  C.c_float(cam.tanfovx), C.c_float(cam.tanfovy), W, H, C.c_float(1.0), _p(radii), _p(m2d), _p(depths),
                  _p(conic), _p(cov3D), _p(tiles), _p(rgb), _p(clamped), _p(ext))
    vis = fw.radii > 0
    assert vis.sum() > 1000 and (~vis).sum() > 10
    assert (radii == fw.radii).all()
    assert (tiles == fw.tiles_touched).all()
    assert (m2d[vis] == fw.means2D[vis]).all()
    assert (depths[vis] == fw.depths[vis]).all()
    assert (conic[vis] == fw.conic_opacity[vis, :3]).all()
    front = depths > 0.2
    assert (cov3D[front] == fw.cov3D[front]).all()
    assert (clamped[vis] == fw.clamped[vis]).all() and fw.clamped.sum() > 0
    assert np.abs(rgb[vis] - fw.rgb[vis]).max() <= 1e-6

    # conservative extents: every (pixel, Gaussian) pair the oracle's blend accepts lies inside the box
    pl, rg = fw.point_list, fw.ranges
    gx = (W + 15) // 16
    worst = 0.0
    for t in range(rg.shape[0]):
        ids = pl[rg[t, 0]:rg[t, 1]]
        if len(ids) == 0:
            continue
        ty, tx = divmod(t, gx)
        ys, xs = np.meshgrid(np.arange(ty * 16, min(ty * 16 + 16, H)), np.arange(tx * 16, min(tx * 16 + 16, W)), indexing="ij")
        dx = fw.means2D[ids, 0][:, None, None] - xs[None].astype(np.float32)
        dy = fw.means2D[ids, 1][:, None, None] - ys[None].astype(np.float32)
        co = fw.conic_opacity[ids]
        power = -0.5 * (co[:, 0, None, None] * dx * dx + co[:, 2, None, None] * dy * dy) - co[:, 1, None, None] * dx * dy
        alpha = np.minimum(0.99, co[:, 3, None, None] * np.exp(power))
        acc = (power <= 0) & (alpha >= 1.0 / 255.0)
        outside = (np.abs(dx) > ext[ids, 0][:, None, None]) | (np.abs(dy) > ext[ids, 1][:, None, None])
        assert not (acc & outside).any()
        if acc.any():
            worst = max(worst, (np.abs(dx)[acc] / ext[ids, 0][:, None, None].repeat(dx.shape[1], 1).repeat(dx.shape[2], 2)[acc]).max())
    assert 0.5 < worst <= 1.0  # the box is conservative but not vacuous

    # quadrant cull (box + exact ellipse-vs-rectangle): never drops a quadrant that holds an accepted pair, and is
    # tight enough to be worth running (most set bits correspond to a quadrant with an accepted pair)
    set_bits = needed_bits = 0
    for t in range(rg.shape[0]):
        ids = pl[rg[t, 0]:rg[t, 1]]
        if len(ids) == 0:
            continue
        ty, tx = divmod(t, gx)
        masks = np.zeros(len(ids), np.uint32)
        m2 = np.ascontiguousarray(fw.means2D[ids]); co = np.ascontiguousarray(fw.conic_opacity[ids])
        ex = np.ascontiguousarray(ext[ids])
        hm.hm_quadrant_masks(len(ids), _p(m2), _p(co), _p(ex), C.c_float(tx * 16.0), C.c_float(ty * 16.0), _p(masks))
        ys, xs = np.meshgrid(np.arange(ty * 16, ty * 16 + 16), np.arange(tx * 16, tx * 16 + 16), indexing="ij")
        dx = m2[:, 0][:, None, None] - xs[None].astype(np.float32)
        dy = m2[:, 1][:, None, None] - ys[None].astype(np.float32)
        power = -0.5 * (co[:, 0, None, None] * dx * dx + co[:, 2, None, None] * dy * dy) - co[:, 1, None, None] * dx * dy
        alpha = np.minimum(0.99, co[:, 3, None, None] * np.exp(power))
        acc = (power <= 0) & (alpha >= 1.0 / 255.0)
        for q in range(4):
            qa = acc[:, (q >> 1) * 8:(q >> 1) * 8 + 8, (q & 1) * 8:(q & 1) * 8 + 8].any(axis=(1, 2))
            bit = (masks >> q) & 1
            assert not (qa & (bit == 0)).any(), "cull dropped a quadrant with an accepted pair"
            set_bits += int(bit.sum()); needed_bits += int(qa.sum())
    assert needed_bits / max(set_bits, 1) > 0.6

    wts = syn.loss_weights(cam)
    g = oracle.backward(fw, wts["color"], wts["depth"], wts["alpha"], None)
    dmean3D = np.zeros((P, 3), np.float32); dcov = np.zeros((P, 6), np.float32); dscale = np.zeros((P, 3), np.float32)
    drot = np.zeros((P, 4), np.float32); dsh = np.zeros((P, M, 3), np.float32)
    hm.hm_backward(P, deg, M, _p(means), _p(scales), _p(rots), _p(shs), _p(fw.cov3D), _p(fw.clamped), _p(fw.radii),
                   _p(view), _p(proj), _p(campos), C.c_float(cam.tanfovx), C.c_float(cam.tanfovy), W, H, C.c_float(1.0),
                   _p(g["means2D"]), _p(g["conic"].reshape(P, 4)), _p(g["colors"]), _p(g["depths"].reshape(P)),
                   _p(dmean3D), _p(dcov), _p(dscale), _p(drot), _p(dsh))
    for name, a, b in [("means3D", dmean3D, g["means3D"]), ("cov3D", dcov, g["cov3D"]), ("scales", dscale, g["scales"]),
                       ("rotations", drot, g["rotations"]), ("sh", dsh, g["sh"])]:
        assert_close(a, b, rel=1e-4, floor=1e-6, name=name)
    fw.free()


def test_power2_matches_reference_expression(hm):
    rng = np.random.default_rng(0)
    LOG2E = np.float32(1.4426950408889634)
    for _ in range(2000):
        A, C_, B = rng.uniform(0.001, 3.0), rng.uniform(0.001, 3.0), rng.uniform(-1, 1)
        dx, dy = rng.uniform(-20, 20, 2)
        ref = (-0.5 * (A * dx * dx + C_ * dy * dy) - B * dx * dy) * 1.4426950408889634
        qa, qb, qc = np.float32(-0.5) * LOG2E * np.float32(A), -LOG2E * np.float32(B), np.float32(-0.5) * LOG2E * np.float32(C_)
        got = hm.hm_power2(C.c_float(qa), C.c_float(qb), C.c_float(qc), C.c_float(dx), C.c_float(dy))
        assert abs(got - ref) <= 2e-6 * max(1.0, A * dx * dx + C_ * dy * dy + abs(B * dx * dy))


def test_parity_mode_staged_power_and_quotient_identities(hm):
    """SGR_EXACT mode stages the conic as (-0.5 cx, -cy, -0.5 cz) and forms T / (1 - alpha) from a refined reciprocal:
    the staged power must have the bits of the reference's expression (forward.cu:420), and the quotient must be the
    correctly rounded one whatever the last bit of the reciprocal seed is (v_rcp_f32 is accurate to 1 ulp; the GPU test
    sgr_test_exact_math compares against hipcc's own `/`)."""
    rng = np.random.default_rng(3)
    n = 2_000_000
    c = np.stack([np.exp(rng.uniform(-9, 2, n)), rng.normal(0, 0.3, n) * np.exp(rng.uniform(-9, 1, n)),
                  np.exp(rng.uniform(-9, 2, n))], 1).astype(np.float32)
    d = rng.uniform(-40, 40, (n, 2)).astype(np.float32)
    ref, staged = np.zeros(n, np.float32), np.zeros(n, np.float32)
    hm.hm_power_ref_pair(n, _p(c), _p(d), _p(ref), _p(staged))
    assert (ref.view(np.int32) == staged.view(np.int32)).all()
    # numerators: T and T_final = 1 - alpha_out, i.e. 0 or >= 2^-24 (a subnormal numerator would need v_div_scale)
    a = np.concatenate([rng.uniform(0, 1, n // 2), np.maximum(rng.uniform(0, 1, n // 2) ** 8, 2.0 ** -24)]).astype(np.float32)
    a[:3] = [0.0, 1.0, 2.0 ** -24]
    b = rng.uniform(0.01, 1 - 1 / 255, n).astype(np.float32)
    want = (a.astype(np.float64) / b.astype(np.float64)).astype(np.float32)
    y = (np.float32(1) / b).astype(np.float32)
    for seed in (y, np.nextafter(y, np.float32(2)), np.nextafter(y, np.float32(0))):
        q = np.zeros(n, np.float32)
        hm.hm_div_by_seed(n, _p(a), _p(b), _p(np.ascontiguousarray(seed, dtype=np.float32)), _p(q))
        assert (q == want).all(), int((q != want).sum())


def test_tile_masks_are_conservative_and_the_bit_utilities_invert_each_other(hm):
    """sgr_tile_mask (which tiles of its rect a Gaussian is emitted for): a tile that holds a pixel the blend accepts
    (power <= 0 and alpha >= 1/255, forward.cu:425-430) always has its bit; the mask is worth having (most set bits are
    needed).  sgr_select_bit / the rank inside sgr_row_of: select enumerates the set bits in order, rank inverts it."""
    cam = syn.make_camera(400, 256, fx=420.0)
    sc = syn.make_scene(3000, cam, S=0, seed=9, scale_px=0.012, zmin=1.0, zmax=12.0)
    sc.scales[::3, 0] *= 6.0  # some needles: their bounding boxes have empty corners
    fw = oracle.forward(**oracle_kwargs(cam, sc))
    gx = (cam.image_width + 15) // 16
    tile = (fw.keys >> np.uint64(32)).astype(np.int64)
    tx, ty = tile % gx, tile // gx
    ids, rects = [], []
    for g in np.unique(fw.point_list):
        m = fw.point_list == g
        r = [tx[m].min(), ty[m].min(), tx[m].max() + 1, ty[m].max() + 1]
        if 2 <= (r[2] - r[0]) * (r[3] - r[1]) <= 64:
            ids.append(g); rects.append(r)
    ids = np.array(ids); rects = np.ascontiguousarray(rects, dtype=np.uint32)
    assert len(ids) > 500
    m2 = np.ascontiguousarray(fw.means2D[ids]); co = np.ascontiguousarray(fw.conic_opacity[ids])
    masks = np.zeros(len(ids), np.uint64)
    hm.hm_tile_masks(len(ids), _p(m2), _p(co), _p(rects), _p(masks))
    set_bits = needed = total = 0
    for i in range(len(ids)):
        x0, y0, x1, y1 = [int(v) for v in rects[i]]
        w = x1 - x0
        ys, xs = np.meshgrid(np.arange(y0 * 16, y1 * 16), np.arange(x0 * 16, x1 * 16), indexing="ij")
        dx = m2[i, 0] - xs.astype(np.float32); dy = m2[i, 1] - ys.astype(np.float32)
        power = -0.5 * (co[i, 0] * dx * dx + co[i, 2] * dy * dy) - co[i, 1] * dx * dy
        acc = (power <= 0) & (np.minimum(0.99, co[i, 3] * np.exp(power)) >= 1.0 / 255.0)
        for j in range(w * (y1 - y0)):
            jy, jx = divmod(j, w)
            need = acc[jy * 16:jy * 16 + 16, jx * 16:jx * 16 + 16].any()
            bit = (int(masks[i]) >> j) & 1
            assert bit or not need, "the mask dropped a tile with an accepted pixel"
            set_bits += bit; needed += int(need); total += 1
    assert needed / set_bits > 0.75 and set_bits / total < 0.95
    # the row-wise form the preprocess runs against the per-tile definition: the same region up to the margins -- it may
    # keep a few tiles more (the per-tile test knows both band limits of a tile, the row form one band) or, at a margin, fewer
    ref_masks = np.zeros(len(ids), np.uint64)
    hm.hm_tile_masks_per_tile(len(ids), _p(m2), _p(co), _p(rects), _p(ref_masks))
    pop = lambda a: int(np.unpackbits(a.view(np.uint8)).sum())
    assert abs(pop(masks) - pop(ref_masks)) <= 0.02 * pop(ref_masks), (pop(masks), pop(ref_masks))
    hm.hm_select_bit.restype = C.c_uint32
    hm.hm_select_bit.argtypes = [C.c_uint64, C.c_uint32]
    hm.hm_row_of.restype = C.c_uint32
    hm.hm_row_of.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(3)
    u0 = np.array([1000], np.uint32)
    for _ in range(300):
        m = int(rng.integers(1, 2 ** 63, dtype=np.uint64)) & int(rng.integers(1, 2 ** 63, dtype=np.uint64)) | (1 << int(rng.integers(0, 64)))
        bits = [b for b in range(64) if (m >> b) & 1]
        tm = np.array([m], np.uint64)
        for k, b in enumerate(bits):
            assert hm.hm_select_bit(m, k) == b
            # a masked rect of width 8 at (3, 5): tile index b -> row u0 + rank
            rect = 3 | (5 << 10) | (8 << 20) | 0x80000000
            assert hm.hm_row_of(rect, 3 + b % 8, 5 + b // 8, _p(u0), _p(tm), 0) == 1000 + k
    rect = 3 | (5 << 10) | (8 << 20)  # without a mask: the index inside the rect
    assert hm.hm_row_of(rect, 3 + 5, 5 + 2, _p(u0), None, 0) == 1000 + 2 * 8 + 5
    fw.free()


def test_tile_mask_row_form_never_drops_a_needed_tile_for_extreme_splats(hm):
    """Randomised sweep of sgr_tile_mask (the O(rows) form the preprocess runs) over the cases the advisor flagged: extreme
    anisotropy (needles whose conic determinant nearly cancels), axis-aligned conics (B == 0 exactly), centres far off the
    screen, opacities at both ends.  Ground truth = the blend's own per-pixel test (power <= 0 and alpha >= 1/255,
    forward.cu:425-430) evaluated on every pixel of every tile of the rect: a tile with an accepted pixel must have its bit.
    (A lost tile is the one way the tile masks could change an image.)"""
    rng = np.random.default_rng(11)
    n = 4000
    # covariance from eigenvalues spanning 1e-1 .. 1e5 px^2 and a random angle; a third axis-aligned (B == 0 exactly)
    l1 = 10.0 ** rng.uniform(-1, 5, n)
    l2 = 10.0 ** rng.uniform(-1, 1.5, n)
    th = rng.uniform(0, np.pi, n)
    th[::3] = rng.integers(0, 2, len(th[::3])) * (np.pi / 2)
    c, s = np.cos(th), np.sin(th)
    a = l1 * c * c + l2 * s * s + 0.3
    b = (l1 - l2) * c * s
    b[::3] = 0.0
    cc = l1 * s * s + l2 * c * c + 0.3
    det = a * cc - b * b
    conic = np.stack([cc / det, -b / det, a / det], 1)
    opac = np.where(rng.random(n) < 0.3, rng.uniform(0.004, 0.02, n), rng.uniform(0.02, 1.0, n))
    # rects of 2..64 tiles somewhere on a 64 x 64 tile screen, the centre inside, at the border of, or far outside the rect
    w = rng.integers(1, 9, n); h = rng.integers(1, 9, n)
    w = np.where(w * h < 2, 2, w)
    x0 = rng.integers(0, 56, n); y0 = rng.integers(0, 56, n)
    cx = (x0 + w * rng.uniform(-1.5, 2.5, n)) * 16.0
    cy = (y0 + h * rng.uniform(-1.5, 2.5, n)) * 16.0
    m2 = np.ascontiguousarray(np.stack([cx, cy], 1), dtype=np.float32)
    co = np.ascontiguousarray(np.concatenate([conic, opac[:, None]], 1), dtype=np.float32)
    rects = np.ascontiguousarray(np.stack([x0, y0, x0 + w, y0 + h], 1), dtype=np.uint32)
    masks = np.zeros(n, np.uint64)
    hm.hm_tile_masks(n, _p(m2), _p(co), _p(rects), _p(masks))
    needed_total = 0
    for i in range(n):
        X0, Y0, X1, Y1 = [int(v) for v in rects[i]]
        ww = X1 - X0
        ys, xs = np.meshgrid(np.arange(Y0 * 16, Y1 * 16), np.arange(X0 * 16, X1 * 16), indexing="ij")
        dx = m2[i, 0] - xs.astype(np.float32); dy = m2[i, 1] - ys.astype(np.float32)
        with np.errstate(over="ignore", invalid="ignore"):
            power = np.float32(-0.5) * (co[i, 0] * dx * dx + co[i, 2] * dy * dy) - co[i, 1] * dx * dy
            acc = (power <= 0) & (np.minimum(np.float32(0.99), co[i, 3] * np.exp(power)) >= np.float32(1.0 / 255.0))
        need = acc.reshape(Y1 - Y0, 16, ww, 16).any(axis=(1, 3)).reshape(-1)
        bits = np.array([(int(masks[i]) >> j) & 1 for j in range(need.size)], bool)
        assert not (need & ~bits).any(), (i, m2[i], co[i], rects[i], hex(int(masks[i])))
        needed_total += int(need.sum())
    assert needed_total > n // 4  # the sweep does exercise tiles that are needed
