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


@pytest.mark.parametrize("n,end_bit,dup", [(1, 14, False), (4097, 14, True), (2_500_000, 14, True), (1_000_000, 32, False),
                                           (20_000_003, 15, False)])
@pytest.mark.parametrize("one_sweep", [False, True])
def test_sort_pairs32_stable(n, end_bit, dup, one_sweep, sort_mode):
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
                                _vp(hist), _vp(tmp), None))
    cur = check(L.sgr_test_sort32(_vp(k0), _vp(k1), _vp(v0), _vp(v1), n, end_bit, _vp(hist), _vp(tmp), None))
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
