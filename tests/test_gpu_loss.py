"""`-m gpu` parity tests of the fused colour losses (street_gaussians_amd/losses.py + csrc/sgr_loss.hip) against the
torch restatement of the reference (tests/torch_ref_loss.py) evaluated in float64 on the CPU."""
import pytest
import torch

import torch_ref_loss as ref
from street_gaussians_amd import losses

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,H,W,masked", [(3, 64, 96, False), (3, 131, 77, True), (1, 16, 16, False), (3, 5, 7, True),
                                          (3, 320, 480, True)])
def test_l1_and_ssim_match_reference(C, H, W, masked):
    g = torch.Generator().manual_seed(H * 1000 + W)
    a = torch.rand(C, H, W, generator=g)
    b = (a + 0.2 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
    mask = (torch.rand(1, H, W, generator=g) < 0.8) if masked else None
    a64 = a.double().requires_grad_(True)
    l1_ref = ref.l1_loss(a64, b.double(), mask)
    ss_ref = ref.ssim(a64, b.double(), mask=mask)
    (0.8 * l1_ref + 0.2 * (1.0 - ss_ref)).backward()
    ag = a.cuda().requires_grad_(True)
    mg = None if mask is None else mask.cuda()
    l1 = losses.l1_loss(ag, b.cuda(), mg)
    ss = losses.ssim(ag, b.cuda(), mask=mg)
    (0.8 * l1 + 0.2 * (1.0 - ss)).backward()
    assert abs(l1.item() - l1_ref.item()) <= 2e-6 * abs(l1_ref.item())
    assert abs(ss.item() - ss_ref.item()) <= 5e-6
    gr, gg = a64.grad, ag.grad.cpu().double()
    scale = float(gr.abs().max())
    assert float((gg - gr).abs().max()) <= 2e-5 * scale
    if masked:
        assert float(gg[:, ~mask[0]].abs().max()) == 0.0
    # bit-reproducible
    ag2 = a.cuda().requires_grad_(True)
    (0.8 * losses.l1_loss(ag2, b.cuda(), mg) + 0.2 * (1.0 - losses.ssim(ag2, b.cuda(), mask=mg))).backward()
    assert torch.equal(ag2.grad, ag.grad)


def test_loss_argument_checks():
    from street_gaussians_amd._native import SgrError
    x = torch.rand(3, 8, 8, device="cuda")
    with pytest.raises(NotImplementedError):
        losses.ssim(x, x, window_size=7)
    with pytest.raises(SgrError, match="no CPU path"):
        losses.l1_loss(x.cpu(), x.cpu())
    assert float(losses.ssim(x, x)) == pytest.approx(1.0, abs=1e-6)
    assert float(losses.l1_loss(x, x)) == 0.0
