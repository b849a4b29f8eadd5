"""CPU tests for the colour-loss row: the torch restatement is pinned against the reference's own
l1_loss / ssim (extracted from its source, which cannot be imported whole) and the C ABI must be exported."""
import ctypes as C
import os
import re

import pytest
import torch

import torch_ref_loss as ref

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LU = "/root/reference/lib/utils/loss_utils.py"


def _reference_namespace():
    src = open(LU).read()
    ns = {"torch": torch, "F": torch.nn.functional, "exp": __import__("math").exp, "Variable": lambda t: t}
    for name in ("l1_loss", "gaussian", "create_window", "ssim", "_ssim"):  # the reference's own source, executed in place
        m = re.search(rf"^def {name}\(.*?(?=^def |\Z)", src, re.S | re.M)
        exec(m.group(0), ns)
    return ns


@pytest.mark.skipif(not os.path.exists(LU), reason="reference checkout not present on this machine")
@pytest.mark.parametrize("masked", [False, True])
def test_restated_losses_match_the_reference(masked):
    ns = _reference_namespace()
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(3, 37, 53, generator=g), torch.rand(3, 37, 53, generator=g)
    mask = (torch.rand(1, 37, 53, generator=g) < 0.7) if masked else None
    assert torch.equal(ref.l1_loss(a, b, mask), ns["l1_loss"](a, b, mask))
    assert torch.equal(ref.ssim(a, b, mask=mask), ns["ssim"](a, b, mask=mask))


def test_loss_abi_symbols_are_exported():
    from street_gaussians_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip("libsgr_hip.so not built")
    L = C.CDLL(_native.LIB_PATH)
    decl = open(os.path.join(ROOT, "include", "sgr_loss.h")).read()
    names = re.findall(r"^(?:int|size_t) (sgr_\w+)\(", decl, re.M)
    assert len(names) == 12
    for n in names:
        assert hasattr(L, n), n
