"""GPU tests of the hand-written primitives (scan, stable radix sort, DPP wave reduction) through the
C ABI self-test entry points (include/sgr.h)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from street_gaussians_amd import _native
    return _native.lib(), _native.check


def _vp(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("n", [1, 7, 2047, 2048, 2049, 100000, 4 * 1024 * 1024 + 5])
@pytest.mark.parametrize("inclusive", [0, 1])
def test_scan(n, inclusive):
    L, check = _lib()
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 50, (n,), generator=g, dtype=torch.int32)
    xd = x.cuda()
    out = torch.zeros_like(xd)
    tmp = torch.zeros(L.sgr_test_scan_tmp_words(n), dtype=torch.int32, device="cuda")
    check(L.sgr_test_scan(_vp(xd), _vp(out), n, inclusive, _vp(tmp), None))
    torch.cuda.synchronize()
    ref = np.cumsum(x.numpy().astype(np.int64))
    if not inclusive:
        ref = ref - x.numpy()
    assert (out.cpu().numpy().astype(np.int64) == ref).all()
    # in place
    check(L.sgr_test_scan(_vp(xd), _vp(xd), n, inclusive, _vp(tmp), None))
    torch.cuda.synchronize()
    assert (xd.cpu().numpy().astype(np.int64) == ref).all()


@pytest.mark.parametrize("n,end_bit,dup", [(1, 46, False), (100, 46, True), (2048, 46, True), (2049, 40, False),
                                           (70001, 46, True), (3_000_000, 46, False), (1_000_003, 32, True),
                                           (500_000, 41, True)])
@pytest.mark.parametrize("one_sweep", [False, True])
def test_sort_pairs_stable(n, end_bit, dup, one_sweep, sort_mode):
    L, check = _lib()
    sort_mode(one_sweep)
    rng = np.random.default_rng(n)
    hi = 1 << end_bit
    if dup:  # few distinct keys -> stability is exercised
        keys = rng.integers(0, 97, n, dtype=np.uint64) * np.uint64(hi // 128 + 1)
    else:
        keys = rng.integers(0, hi, n, dtype=np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    k0 = torch.from_numpy(keys.view(np.int64)).cuda()
    v0 = torch.from_numpy(vals.view(np.int32)).cuda()
    k1, v1 = torch.zeros_like(k0), torch.zeros_like(v0)
    # the work area arrives dirty (torch.empty in production): here, with everything a stale run could have left
    hist = torch.full((L.sgr_test_sort_hist_words(n),), -1, dtype=torch.int32, device="cuda")
    tmp = torch.zeros(L.sgr_test_scan_tmp_words(hist.numel()), dtype=torch.int32, device="cuda")
    cur = check(L.sgr_test_sort(_vp(k0), _vp(k1), _vp(v0), _vp(v1), n, end_bit, _vp(hist), _vp(tmp), None))
    torch.cuda.synchronize()
    ks = (k1 if cur else k0).cpu().numpy().view(np.uint64)
    vs = (v1 if cur else v0).cpu().numpy().view(np.uint32)
    order = np.argsort(keys, kind="stable")
    assert (ks == keys[order]).all()
    assert (vs == vals[order]).all()


# max_bits 9: digits of NINE bits (512 bins, two per thread in the scatter's prefix section) -- the forward's depth sort runs
# on 27 key bits in three such passes below 750 k Gaussians and in four 7-bit passes above
@pytest.mark.parametrize("n,end_bit,dup,max_bits", [(1, 14, False, 8), (4097, 14, True, 8), (2_500_000, 14, True, 8),
                                                    (1_000_000, 32, False, 8), (20_000_003, 15, False, 8),
                                                    (1_000_001, 27, False, 8), (1_000_001, 27, False, 9),
                                                    (300_000, 27, True, 9), (70_001, 18, True, 9), (5000, 9, False, 9),
                                                    (5000, 9, False, 8), (40_000, 32, False, 9)])
@pytest.mark.parametrize("one_sweep", [False, True])
def test_sort_pairs32_stable(n, end_bit, dup, max_bits, one_sweep, sort_mode):
    L, check = _lib()
    sort_mode(one_sweep)
    rng = np.random.default_rng(n + 1)
    hi = 1 << end_bit
    keys = (rng.integers(0, 61, n, dtype=np.uint64) * (hi // 64) if dup else rng.integers(0, hi, n, dtype=np.uint64)).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    k0 = torch.from_numpy(keys.view(np.int32)).cuda()
    v0 = torch.from_numpy(vals.view(np.int32)).cuda()
    k1, v1 = torch.zeros_like(k0), torch.zeros_like(v0)
    hist = torch.zeros(L.sgr_test_sort_hist_words(n), dtype=torch.int32, device="cuda")
    tmp = torch.zeros(L.sgr_test_scan_tmp_words(hist.numel()), dtype=torch.int32, device="cuda")
    if n > 1:  # a first sort of other data leaves its look-back table behind: the second must not read it as its own
        kk = torch.from_numpy(np.roll(keys, 1).view(np.int32)).cuda()
        check(L.sgr_test_sort32(_vp(kk), _vp(torch.zeros_like(kk)), _vp(v0.clone()), _vp(torch.zeros_like(v0)), n, end_bit,
                                max_bits, _vp(hist), _vp(tmp), None))
    cur = check(L.sgr_test_sort32(_vp(k0), _vp(k1), _vp(v0), _vp(v1), n, end_bit, max_bits, _vp(hist), _vp(tmp), None))
    torch.cuda.synchronize()
    ks = (k1 if cur else k0).cpu().numpy().view(np.uint32)
    vs = (v1 if cur else v0).cpu().numpy().view(np.uint32)
    order = np.argsort(keys, kind="stable")
    assert (ks == keys[order]).all()
    assert (vs == vals[order]).all()


@pytest.fixture
def sort_mode():
    """Selects the radix sort's form (three launches per pass / the one-sweep A/B form) for one test."""
    from street_gaussians_amd import _C
    prev = _C.test_switches()

    def set_mode(one_sweep):
        if one_sweep and not _C.has_variants():
            pytest.skip("the one-sweep sort is an A/B design outside the shipped library (tools/build_variant.py -DSGR_WITH_VARIANTS=1)")
        _C.test_switches((prev & ~_C.USE_ONESWEEP) | (_C.USE_ONESWEEP if one_sweep else 0))
    yield set_mode
    _C.test_switches(prev)


def test_wave_sum_dpp_equals_shuffle():
    L, check = _lib()
    nw = 257
    g = torch.Generator().manual_seed(0)
    # integers: every summation order is exact, so DPP, shuffle and numpy must agree bit for bit
    x = torch.randint(-1000, 1000, (nw * 64,), generator=g).float()
    xd = x.cuda()
    a = torch.zeros(nw, device="cuda")
    b = torch.zeros(nw, device="cuda")
    check(L.sgr_test_wave_sum(_vp(xd), _vp(a), _vp(b), nw, None))
    torch.cuda.synchronize()
    ref = x.reshape(nw, 64).sum(1).numpy()
    assert (a.cpu().numpy() == ref).all()
    assert (b.cpu().numpy() == ref).all()


def test_lds_returning_atomics_serve_equal_addresses_in_lane_order():
    """What the per-tile LDS sort's ranking relies on (csrc/sgr_tile_sort.hip): within ONE wave64 ds_add_rtn_u32 the lanes
    that hit the same counter get 0, 1, 2, ... back in ASCENDING LANE ORDER.  Patterns: all lanes on one counter, two / four /
    sixteen counters in every interleaving, random counters."""
    L, check = _lib()
    g = torch.Generator().manual_seed(3)
    pats = [torch.zeros(64, dtype=torch.int64), torch.arange(64) % 2, torch.arange(64) // 32, torch.arange(64) % 4,
            torch.arange(64) // 16, torch.arange(64) % 16, torch.arange(64), 63 - torch.arange(64), (63 - torch.arange(64)) // 8]
    pats += [torch.randint(0, k, (64,), generator=g) for k in (2, 3, 5, 8, 16, 32, 64) for _ in range(200)]
    pat = torch.stack(pats).to(torch.int32)
    trials = pat.shape[0]
    out = torch.zeros(trials, 64, dtype=torch.int32, device="cuda")
    pd = pat.cuda().contiguous()
    check(L.sgr_test_lds_atomic_order(_vp(pd), _vp(out), trials, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    p = pat.numpy()
    # expected: rank of the lane among the lanes with the same counter, in lane order
    want = np.zeros_like(p)
    for t in range(trials):
        seen = {}
        for l in range(64):
            want[t, l] = seen.get(p[t, l], 0)
            seen[p[t, l]] = want[t, l] + 1
    assert (got == want).all(), int((got != want).sum())


def test_parity_mode_elementary_functions_have_the_bits_of_expf_and_division():
    """SGR_EXACT mode runs expf and T / (1 - alpha) written out without their range handling (sgr_math.h: sgr_expf_ref,
    sgr_div_by).  They must have the bits of the device library's expf and of hipcc's IEEE `/`: dense sweeps of the operand
    ranges the blend kernels produce, plus the special arguments the shortened expf still has to get right."""
    L, check = _lib()
    n = 1 << 24
    g = torch.Generator().manual_seed(5)
    # expf: every power the kernels can form -- (-inf, 0] densely around the range where alpha >= 1/255 is possible
    # (x >= -5.6), the whole underflow region, huge negative arguments, positive ones up to the overflow select, NaN
    x = torch.cat([-8.0 * torch.rand(n // 2, generator=g), -120.0 * torch.rand(n // 4, generator=g),
                   -torch.exp(88.0 * torch.rand(n // 8, generator=g)), 88.7 * torch.rand(n // 8 - 16, generator=g),
                   torch.tensor([0.0, -0.0, -float("inf"), float("nan"), -1e38, -3.4e38, -87.3, -88.0, -103.2, -103.3,
                                 -104.0, -126.0, -149.0, -150.0, -1e-45, -1e-38])]).float()
    assert x.numel() == n
    # quotients: numerator T or T_final in [0, 1], denominator 1 - alpha in [0.01, 1 - 1/255] (alpha = min(0.99, .) >= 1/255)
    # (numerators are 0 or >= 2^-24: T_final = 1 - alpha_out with alpha_out <= 1; a subnormal one would need v_div_scale)
    a = torch.cat([torch.rand(n // 2, generator=g), 1e-4 * torch.rand(n // 4, generator=g),
                   torch.clamp(torch.rand(n // 4, generator=g) ** 8, min=2.0 ** -24)]).float()
    a[:4] = torch.tensor([0.0, 1.0, 1e-4, 2.0 ** -24])
    b = (0.01 + (1.0 - 1.0 / 255.0 - 0.01) * torch.rand(n, generator=g)).float()
    b[:n // 16] = 1.0 - torch.clamp(torch.rand(n // 16, generator=g) * 0.99, min=1.0 / 255.0)  # as the kernel forms it
    xd, ad, bd = x.cuda(), a.cuda(), b.cuda()
    outs = [torch.empty(n, device="cuda") for _ in range(4)]
    check(L.sgr_test_exact_math(n, _vp(xd), _vp(outs[0]), _vp(outs[1]), _vp(ad), _vp(bd), _vp(outs[2]), _vp(outs[3]), None))
    torch.cuda.synchronize()
    e_lib, e_ref, d_lib, d_ref = (o.cpu().view(torch.int32) for o in outs)
    nan = torch.isnan(x)
    # The written-out form (round 6: argument clamped at -86, integer exponent add, no other range handling) has the bits of
    # expf wherever the blend kernels use G -- power <= 0 with alpha = opacity * G >= 1/255, i.e. x >= -5.6 -- and in fact for
    # every x in [-86, 87].  Below -86 (and for NaN) it returns expf(-86) = 4.5e-38: "nothing", like expf's subnormals and
    # zeros there, and never a NaN (a NaN alpha would pass the kernels' !(alpha < thr) test).
    tiny = x < -86.0
    low = outs[1].cpu()[tiny | nan]
    assert (low == low[0]).all() and 4e-38 < float(low[0]) < 5e-38, low[:8].tolist()
    assert (outs[0].cpu()[tiny] <= 5e-38).all()
    bad = (e_lib != e_ref) & ~nan & ~tiny & (x <= 87.0)
    assert not bad.any(), f"expf: {int(bad.sum())} of {n} differ, first at x = {x[bad][:5].tolist()}"
    bad = d_lib != d_ref
    assert not bad.any(), f"division: {int(bad.sum())} of {n} differ, first at {a[bad][:5].tolist()} / {b[bad][:5].tolist()}"
    # and the quotient really is the correctly rounded one
    q64 = (a.double() / b.double()).float()
    assert (outs[3].cpu() == q64).all()


def test_fast_exp_stays_well_inside_the_guard_band_of_the_parity_backward():
    """The parity mode's backward takes G = exp(power) from exp2(power * log2 e) and falls back to the accurate expf for a
    visit with a pixel whose alpha is within 4e-6 (relative) of the 1/255 threshold (csrc/sgr_blend_bwd.hip,
    SGR_EXACT_BWD_FAST).  That keeps every blend / skip decision the forward's as long as the fast G is much closer than
    4e-6 to the true one WHERE ALPHA CAN SIT AT THE THRESHOLD: alpha = o * G = 1/255 with o <= 1 means power >= -ln 255 =
    -5.54.  Swept here with the GPU's own exp2 (f32) against float64: measured 4.3e-7 at worst (the rounding of the product
    below 8, the f32 constant, 0.08e-6 from the exp2 instruction itself), held to 6e-7 -- a seventh of the band."""
    g = torch.Generator().manual_seed(11)
    x = -(torch.rand(1 << 24, generator=g) * 5.6).cuda()  # power in (-5.6, 0]
    log2e = torch.tensor(1.4426950408889634, dtype=torch.float32, device="cuda")
    fast = torch.exp2(x * log2e)  # f32 product rounded once, then the exp2 instruction: the kernel's sequence
    ref = torch.exp(x.double())
    rel = ((fast.double() - ref).abs() / ref).max().item()
    assert rel < 6.0e-7, rel
    # and the accurate path the guard falls back to is expf itself (bit-checked in the test above)
