"""Size ladder for the library's device allocations.

A training loop with adaptive density control changes the number of Gaussians every few hundred iterations; every tensor
whose size follows P (gradients, the rasterizer's scratch, the densify step's outputs) then asks torch's caching allocator
for a block size it has not seen, the cached blocks of the old size stay behind unused, and the request ends in a device
allocation -- tens of milliseconds for a multi-GB block on some hosts (measured: BENCH_r04 configs[4], 7-40 device
allocations inside a 30-iteration region).  Rounding the BACKING size of those tensors up to a ladder of eight steps per
octave makes consecutive sizes repeat: P + 5 % asks for the block P used (or the next step, once per ~2 densify steps),
at a cost of at most 12.5 % of slack.  The tensor itself keeps its exact shape (a view of the rounded storage) -- which
also means ``untyped_storage().nbytes()`` exceeds ``numel() * 4`` and ``torch.save`` would write the slack: use
``checkpoint.save`` / ``checkpoint.exact_storage`` (it clones such tensors into exact storages) for anything that goes to
disk, and do not ``resize_`` these tensors."""
from __future__ import annotations

import math

import torch

_MIN = 1 << 20  # below 1 MiB the allocator's own small-block pools do the job


def ladder(nbytes: int) -> int:
    """nbytes rounded up to a multiple of 1/8 of the power of two below it (exact below 1 MiB)."""
    nbytes = int(nbytes)
    if nbytes < _MIN:
        return nbytes
    step = 1 << (nbytes.bit_length() - 4)
    return (nbytes + step - 1) // step * step


def empty(shape, dtype, device) -> torch.Tensor:
    """torch.empty(shape) backed by a ladder-sized storage."""
    shape = tuple(int(s) for s in shape)
    n = math.prod(shape) if shape else 1
    item = torch.empty((), dtype=dtype).element_size()
    nb = n * item
    if nb < _MIN:
        return torch.empty(shape, dtype=dtype, device=device)
    buf = torch.empty(ladder(nb), dtype=torch.uint8, device=device)
    return buf[:nb].view(dtype).view(shape)
