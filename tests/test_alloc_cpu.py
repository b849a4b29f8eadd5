"""Host logic of the allocation size ladder and the densify pool (no GPU): street_gaussians_amd/_alloc.py, densify.Pool."""
import torch

from street_gaussians_amd import _alloc, densify


def test_ladder_repeats_sizes_with_bounded_slack():
    prev = 0
    seen = set()
    for n in list(range(1, 4096, 97)) + [int(1.05 ** k * (1 << 20)) for k in range(0, 200)]:
        r = _alloc.ladder(n)
        assert r >= n
        if n < (1 << 20):
            assert r == n  # small requests are left to the allocator's own pools
        else:
            assert r - n <= n // 8 + 1  # at most 1/8 of slack
            assert r % (1 << (n.bit_length() - 4)) == 0  # a multiple of 1/8 of the power of two below the request
        assert r >= prev or n < (1 << 20)  # monotone
        prev = r if n >= (1 << 20) else prev
        seen.add(r)
    # a geometric walk of +5 % per step visits far fewer block sizes than steps: consecutive sizes repeat
    walk = [_alloc.ladder(int(1.05 ** k * (200 << 20))) for k in range(60)]
    assert len(set(walk)) <= 40
    assert _alloc.ladder(_alloc.ladder(123456789)) == _alloc.ladder(123456789)  # idempotent


def test_ladder_backed_tensor_keeps_its_shape_and_dtype():
    t = _alloc.empty((300_000, 3), torch.float32, "cpu")  # 3.6 MB: above the 1 MiB floor
    assert t.shape == (300_000, 3) and t.dtype == torch.float32 and t.is_contiguous()
    assert t.untyped_storage().nbytes() == _alloc.ladder(300_000 * 3 * 4)
    s = _alloc.empty((10, 3), torch.float32, "cpu")
    assert s.untyped_storage().nbytes() == 120
    t.zero_()
    assert float(t.sum()) == 0.0


def test_pool_reserves_once_and_grows_in_steps():
    pool = densify.Pool("cpu", factor=1.25)
    est = densify.live_bytes_estimate(10_000)
    assert est > 10_000 * 236  # at least the parameters' own bytes at SH3
    a = pool.reserve(10_000)
    assert a >= int(1.25 * est)
    assert pool.reserve(9_000) == a  # smaller request: nothing happens
    b = pool.reserve(10_400)         # slightly larger: grows by at least 1/8, not by the difference
    assert b >= a + a // 8
    assert pool.reserve(10_400) == b
