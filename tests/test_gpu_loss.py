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


@pytest.mark.parametrize("H,W,seed", [(64, 96, 0), (131, 77, 1), (320, 480, 2)])
def test_accumulation_and_lidar_terms_match_reference(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    acc = torch.rand(1, H, W, generator=g)
    acc[0, :2] = 0.0          # clamp active (lower) -> zero gradient
    acc[0, 2:4] = 1.0         # clamp active (upper)
    sky = torch.rand(1, H, W, generator=g) < 0.3
    depth = torch.rand(1, H, W, generator=g) * 40
    lidar = torch.where(torch.rand(1, H, W, generator=g) < 0.4, torch.rand(1, H, W, generator=g) * 60, torch.zeros(1, H, W))
    mask = torch.rand(1, H, W, generator=g) < 0.9
    # float32 like the reference runs it: the clamp bound 1 - 1e-6 is not representable, and -log(1 - acc) at the
    # bound (13.80 in float32, 13.82 in float64) is part of what the reference computes
    a64, d64 = acc.clone().requires_grad_(True), depth.clone().requires_grad_(True)
    rs, ro = ref.sky_loss(a64, sky), ref.obj_acc_loss(a64, sky)
    rl = ref.lidar_depth_loss(d64, a64.clamp(min=0.05), lidar, mask)
    (rs + 0.5 * ro + 0.1 * rl).backward()
    ag, dg = acc.cuda().requires_grad_(True), depth.cuda().requires_grad_(True)
    s, o = losses.sky_loss(ag, sky.cuda()), losses.obj_acc_loss(ag, sky.cuda())
    l = losses.lidar_depth_loss(dg, ag.clamp(min=0.05), lidar.cuda(), mask.cuda())
    (s + 0.5 * o + 0.1 * l).backward()
    for got, want in ((s, rs), (o, ro), (l, rl)):
        assert abs(got.item() - want.item()) <= 5e-6 * abs(want.item()), (got.item(), want.item())
    for got, want in ((ag.grad, a64.grad), (dg.grad, d64.grad)):
        scale = float(want.abs().max())
        assert float((got.cpu() - want).abs().max()) <= 3e-5 * scale
    assert float(ag.grad[0, :2].abs().max()) == 0.0  # acc = 0: both clamps are active


def test_lidar_selection_edge_cases():
    # no valid pixel -> mean of an empty tensor is NaN (torch); all errors equal -> ties share the slots
    z = torch.zeros(1, 8, 8, device="cuda")
    assert torch.isnan(losses.lidar_depth_loss(z + 1, z + 1, z, None))
    d = (z + 3.0).requires_grad_(True)
    l = losses.lidar_depth_loss(d, z + 1, z + 1, None)      # every error is exactly 2
    assert l.item() == pytest.approx(2.0)
    l.backward()
    assert float(d.grad.sum()) == pytest.approx(1.0, rel=1e-5)  # d(mean)/d(errors) sums to one whichever ties are kept


def test_lidar_topk_count_is_python_int_of_double_product():
    """train.py:128 takes int(0.95 * n) in double: for n = 100 that is 95 (0.95f * 100 would truncate to 94).  With the
    errors 1..100 the mean of the 95 smallest is 48, of the 94 smallest 47.5."""
    for n, k in ((100, 95), (20, 19), (40, 38), (2000, 1900)):
        err = torch.arange(1, n + 1, dtype=torch.float32, device="cuda")[torch.randperm(n, device="cuda")]
        depth = (err + 10.0).reshape(1, 1, n)
        lidar = torch.full((1, 1, n), 10.0, device="cuda")
        l = losses.lidar_depth_loss(depth, torch.ones_like(depth), lidar, None)
        assert int(0.95 * n) == k
        assert l.item() == pytest.approx((k + 1) / 2.0, rel=1e-6), (n, l.item())


def test_loss_argument_checks():
    from street_gaussians_amd._native import SgrError
    x = torch.rand(3, 8, 8, device="cuda")
    with pytest.raises(NotImplementedError):
        losses.ssim(x, x, window_size=7)
    with pytest.raises(SgrError, match="no CPU path"):
        losses.l1_loss(x.cpu(), x.cpu())
    assert float(losses.ssim(x, x)) == pytest.approx(1.0, abs=1e-6)
    assert float(losses.l1_loss(x, x)) == 0.0


@pytest.mark.parametrize("H,W,masked,lam", [(64, 96, False, 0.2), (131, 77, True, 0.2), (320, 480, True, 0.5)])
def test_fused_color_loss_matches_the_two_term_reference(H, W, masked, lam):
    """losses.color_loss = train.py:100-104 in one op: same value and gradient as the two reference terms."""
    g = torch.Generator().manual_seed(5)
    img, gt = torch.rand(3, H, W, generator=g), torch.rand(3, H, W, generator=g)
    mask = (torch.rand(1, H, W, generator=g) < 0.7) if masked else None
    x64 = img.double().requires_grad_(True)
    want = (1.0 - lam) * 1.0 * ref.l1_loss(x64, gt.double(), mask) + lam * (1.0 - ref.ssim(x64, gt.double(), mask=mask))
    want.backward()
    x = img.cuda().requires_grad_(True)
    got, l1 = losses.color_loss(x, gt.cuda(), None if mask is None else mask.cuda(), lambda_dssim=lam, return_l1=True)
    got.backward()
    assert abs(got.item() - want.item()) <= 5e-6 * abs(want.item())
    assert abs(l1.item() - ref.l1_loss(img.double(), gt.double(), mask).item()) <= 5e-6
    scale = float(x64.grad.abs().max())
    assert float((x.grad.cpu() - x64.grad).abs().max()) <= 3e-5 * scale
    # and equals the sum of the two separate ops' gradients
    y = img.cuda().requires_grad_(True)
    two = (1.0 - lam) * losses.l1_loss(y, gt.cuda(), None if mask is None else mask.cuda()) + \
        lam * (1.0 - losses.ssim(y, gt.cuda(), mask=None if mask is None else mask.cuda()))
    two.backward()
    assert torch.allclose(x.grad, y.grad, rtol=1e-5, atol=1e-9)
