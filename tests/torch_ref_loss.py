"""Restatement of the reference's colour losses in plain torch ops (any device / dtype) -- test infrastructure for
street_gaussians_amd/losses.py.  Follows /root/reference/lib/utils/loss_utils.py line by line; tests/test_loss_cpu.py
pins it against the reference's own functions when /root/reference is present."""
from math import exp

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt, mask=None):  # loss_utils.py:21-37
    network_output = network_output.permute(1, 2, 0)
    gt = gt.permute(1, 2, 0)
    if mask is not None:
        mask = mask.squeeze(0)
        network_output = network_output[mask]
        gt = gt[mask]
    return torch.abs(network_output - gt).mean()


def gaussian(window_size, sigma):  # :70-72
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def create_window(window_size, channel):  # :74-78
    _1D_window = gaussian(window_size, 1.5).unsqueeze(1)
    _2D_window = _1D_window.mm(_1D_window.t()).float().unsqueeze(0).unsqueeze(0)
    return _2D_window.expand(channel, 1, window_size, window_size).contiguous()


def ssim(img1, img2, window_size=11, size_average=True, mask=None):  # :80-96
    channel = img1.size(-3)
    window = create_window(window_size, channel)
    if mask is not None:
        img1 = torch.where(mask, img1, torch.zeros_like(img1))
        img2 = torch.where(mask, img2, torch.zeros_like(img2))
    window = window.to(img1.device).type_as(img1)
    return _ssim(img1, img2, window, window_size, channel, size_average)


def _ssim(img1, img2, window, window_size, channel, size_average=True):  # :98-125
    mu1 = F.conv2d(img1, window, padding=window_size // 2, groups=channel)
    mu2 = F.conv2d(img2, window, padding=window_size // 2, groups=channel)
    mu1_sq = mu1.pow(2)
    mu2_sq = mu2.pow(2)
    mu1_mu2 = mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=window_size // 2, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=window_size // 2, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=window_size // 2, groups=channel) - mu1_mu2
    C1 = 0.01 ** 2
    C2 = 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)
