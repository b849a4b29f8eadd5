"""Float reference of the colour losses for street_gaussians_amd/losses.py, written with plain torch ops (any device /
dtype) -- test infrastructure only.  It computes what /root/reference/lib/utils/loss_utils.py computes (l1_loss :21-37,
gaussian / create_window :70-78, ssim / _ssim :80-125); tests/test_loss_cpu.py checks that bit for bit against the
reference's own functions when /root/reference is present."""
import math

import torch
import torch.nn.functional as F

WINDOW, SIGMA = 11, 1.5
K1, K2 = 0.01, 0.03


def l1_loss(network_output, gt, mask=None):
    """Mean absolute difference over the C values of the selected pixels; images (C, H, W), mask (1, H, W) bool."""
    diff = (network_output - gt).abs().permute(1, 2, 0)  # (H, W, C): masked selection keeps whole pixels
    if mask is not None:
        diff = diff[mask[0]]
    return diff.mean()


def window_1d(size=WINDOW, sigma=SIGMA):
    """Normalised Gaussian taps, built as the reference builds them (Python doubles -> float32 tensor -> / sum)."""
    half = size // 2
    taps = torch.Tensor([math.exp(-(i - half) ** 2 / float(2 * sigma ** 2)) for i in range(size)])
    return taps / taps.sum()


def window_2d(channels, size=WINDOW):
    col = window_1d(size).unsqueeze(1)
    return col.mm(col.t()).float()[None, None].expand(channels, 1, size, size).contiguous()


def ssim(img1, img2, window_size=WINDOW, size_average=True, mask=None):
    """Mean SSIM with a depthwise window_size x window_size Gaussian window and zero padding; both images are zeroed
    outside the mask first."""
    channels = img1.size(-3)
    if mask is not None:
        zero = torch.zeros_like(img1)
        img1, img2 = torch.where(mask, img1, zero), torch.where(mask, img2, torch.zeros_like(img2))
    kernel = window_2d(channels, window_size).to(img1.device).type_as(img1)

    def blur(t):
        return F.conv2d(t, kernel, padding=window_size // 2, groups=channels)

    m1, m2 = blur(img1), blur(img2)
    m1m1, m2m2, m1m2 = m1.pow(2), m2.pow(2), m1 * m2
    v1 = blur(img1 * img1) - m1m1
    v2 = blur(img2 * img2) - m2m2
    cov = blur(img1 * img2) - m1m2
    c1, c2 = K1 ** 2, K2 ** 2
    smap = ((2 * m1m2 + c1) * (2 * cov + c2)) / ((m1m1 + m2m2 + c1) * (v1 + v2 + c2))
    return smap.mean() if size_average else smap.mean(1).mean(1).mean(1)


def sky_loss(acc, sky_mask):
    """train.py:107-109 (before the optional per-camera scale)."""
    a = torch.clamp(acc, min=1e-6, max=1. - 1e-6)
    return torch.where(sky_mask, -torch.log(1 - a), -torch.log(a)).mean()


def obj_acc_loss(acc_obj, obj_bound):
    """train.py:116-121."""
    a = torch.clamp(acc_obj, min=1e-6, max=1. - 1e-6)
    inside = -(a * torch.log(a) + (1. - a) * torch.log(1. - a))
    return torch.where(obj_bound, inside, -torch.log(1. - a)).mean()


def lidar_depth_loss(depth, acc, lidar_depth, mask, keep=0.95):
    """train.py:124-131."""
    valid = torch.logical_and(lidar_depth > 0., mask)
    expected = depth / (acc + 1e-10)
    err = torch.abs(expected[valid] - lidar_depth[valid])
    err, _ = torch.topk(err, int(keep * err.size(0)), largest=False)
    return err.mean()
