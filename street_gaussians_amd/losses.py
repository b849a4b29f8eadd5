"""Drop-in `l1_loss` and `ssim` for /root/reference/lib/utils/loss_utils.py:21-37 and :80-125:
same names, arguments and results, each backed by one forward and one backward HIP kernel (csrc/sgr_loss.hip) instead
of ~30 MIOpen / elementwise launches.  `train.py:100-104` keeps its two lines; only the import changes:

    from street_gaussians_amd.losses import l1_loss, ssim

Tensors must live on the GPU; there is no CPU implementation in the product."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _native
from ._native import SgrError, check


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _prep(img, name):
    if not img.is_cuda:
        raise SgrError(f"{name} must be a HIP (cuda) tensor: there is no CPU path")
    if img.dim() == 4:
        img = img.reshape(-1, img.shape[-2], img.shape[-1])
    if img.dim() != 3:
        raise RuntimeError(f"{name} must have dimensions (C, H, W)")
    return img.to(torch.float32).contiguous()


def _prep_mask(mask, H, W):
    if mask is None:
        return None
    m = mask.reshape(-1, H, W)
    if m.shape[0] != 1:
        raise RuntimeError("mask must have dimensions (1, H, W)")
    return m[0].to(torch.uint8).contiguous()


class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, mask):
        L = _native.lib()
        Cc, H, W = img1.shape
        dev = img1.device
        need = img1.requires_grad
        out = torch.empty(1, dtype=torch.float32, device=dev)
        ws = torch.empty(L.sgr_ssim_workspace_floats(Cc, H, W), dtype=torch.float32, device=dev)
        partials = torch.empty(3 * Cc * H * W, dtype=torch.float32, device=dev) if need else None
        with torch.cuda.device(dev):
            check(L.sgr_ssim_forward(Cc, H, W, _p(img1), _p(img2), _p(mask), _p(out), _p(partials), _p(ws), _stream(dev)))
        ctx.save_for_backward(img1, img2, mask if mask is not None else torch.empty(0, device=dev), partials
                              if partials is not None else torch.empty(0, device=dev))
        ctx.has_mask = mask is not None
        return out[0]

    @staticmethod
    def backward(ctx, upstream):
        img1, img2, mask, partials = ctx.saved_tensors
        Cc, H, W = img1.shape
        dev = img1.device
        if partials.numel() == 0:
            raise RuntimeError("ssim forward ran without requires_grad")
        up = upstream.reshape(1).to(torch.float32).contiguous()
        grad = torch.empty_like(img1)
        with torch.cuda.device(dev):
            check(_native.lib().sgr_ssim_backward(Cc, H, W, _p(img1), _p(img2), _p(mask) if ctx.has_mask else None,
                                                  _p(partials), _p(up), _p(grad), _stream(dev)))
        return grad, None, None


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mask):
        L = _native.lib()
        Cc, H, W = a.shape
        dev = a.device
        out = torch.empty(2, dtype=torch.float32, device=dev)
        ws = torch.empty(L.sgr_l1_workspace_floats(Cc, H, W), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.sgr_l1_forward(Cc, H, W, _p(a), _p(b), _p(mask), _p(out), _p(ws), _stream(dev)))
        ctx.save_for_backward(a, b, mask if mask is not None else torch.empty(0, device=dev), out)
        ctx.has_mask = mask is not None
        return out[0]

    @staticmethod
    def backward(ctx, upstream):
        a, b, mask, out = ctx.saved_tensors
        Cc, H, W = a.shape
        dev = a.device
        up = upstream.reshape(1).to(torch.float32).contiguous()
        grad = torch.empty_like(a)
        with torch.cuda.device(dev):
            check(_native.lib().sgr_l1_backward(Cc, H, W, _p(a), _p(b), _p(mask) if ctx.has_mask else None, _p(out),
                                                _p(up), _p(grad), _stream(dev)))
        return grad, None, None


class _BCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acc, mask, mode):
        L = _native.lib()
        dev, n = acc.device, acc.numel()
        out = torch.empty(1, dtype=torch.float32, device=dev)
        ws = torch.empty(L.sgr_l1_workspace_floats(1, 1, 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.sgr_bce_forward(n, mode, _p(acc), _p(mask), _p(out), _p(ws), _stream(dev)))
        ctx.save_for_backward(acc, mask)
        ctx.mode = mode
        return out[0]

    @staticmethod
    def backward(ctx, upstream):
        acc, mask = ctx.saved_tensors
        dev = acc.device
        up = upstream.reshape(1).to(torch.float32).contiguous()
        grad = torch.empty_like(acc)
        with torch.cuda.device(dev):
            check(_native.lib().sgr_bce_backward(acc.numel(), ctx.mode, _p(acc), _p(mask), _p(up), _p(grad), _stream(dev)))
        return grad, None, None


class _Lidar(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, acc, lidar, mask, keep):
        L = _native.lib()
        dev, n = depth.device, depth.numel()
        out = torch.empty(4, dtype=torch.float32, device=dev)
        work = torch.empty(L.sgr_lidar_work_bytes(n), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(L.sgr_lidar_depth_forward(n, _p(depth), _p(acc), _p(lidar), _p(mask), float(keep), _p(out), _p(work),
                                            _stream(dev)))
        ctx.save_for_backward(depth, acc, lidar, out, work)
        return out[0]

    @staticmethod
    def backward(ctx, upstream):
        depth, acc, lidar, out, work = ctx.saved_tensors
        dev = depth.device
        up = upstream.reshape(1).to(torch.float32).contiguous()
        gd, ga = torch.empty_like(depth), torch.empty_like(acc)
        with torch.cuda.device(dev):
            check(_native.lib().sgr_lidar_depth_backward(depth.numel(), _p(depth), _p(acc), _p(lidar), _p(out), _p(work),
                                                         _p(up), _p(gd), _p(ga), _stream(dev)))
        return gd, ga, None, None, None


def _flat(t, name, like=None):
    if not t.is_cuda:
        raise SgrError(f"{name} must be a HIP (cuda) tensor: there is no CPU path")
    if like is not None and t.numel() != like.numel():
        raise RuntimeError(f"{name} must have as many elements as the rendered map")
    return t.to(torch.float32).contiguous()


def _flat_mask(m, like):
    if m is None:
        return None
    if m.numel() != like.numel():
        raise RuntimeError("mask must have as many elements as the rendered map")
    return m.to(torch.uint8).contiguous()


def sky_loss(acc: torch.Tensor, sky_mask: torch.Tensor) -> torch.Tensor:
    """train.py:107-109: clamp(acc, 1e-6, 1-1e-6), then mean of -log(1-acc) on sky pixels and -log(acc) elsewhere."""
    a = _flat(acc, "acc")
    return _BCE.apply(a, _flat_mask(sky_mask, a), 0)


def obj_acc_loss(acc_obj: torch.Tensor, obj_bound: torch.Tensor) -> torch.Tensor:
    """train.py:116-121: entropy of the clamped object accumulation inside the boxes, -log(1-acc) outside, mean."""
    a = _flat(acc_obj, "acc_obj")
    return _BCE.apply(a, _flat_mask(obj_bound, a), 1)


def lidar_depth_loss(depth: torch.Tensor, acc: torch.Tensor, lidar_depth: torch.Tensor, mask: Optional[torch.Tensor] = None,
                     keep: float = 0.95) -> torch.Tensor:
    """train.py:124-131: mean of the int(keep * n) smallest |depth / (acc + 1e-10) - lidar_depth| over the n pixels
    with lidar_depth > 0 and mask.  Gradients flow to depth and acc."""
    d = _flat(depth, "depth")
    return _Lidar.apply(d, _flat(acc, "acc", d), _flat(lidar_depth, "lidar_depth", d).detach(), _flat_mask(mask, d), keep)


class _ColorLoss(torch.autograd.Function):
    """(1 - lambda_dssim) * lambda_l1 * l1_loss + lambda_dssim * (1 - ssim)   (train.py:100-104) as one op: the two
    forward kernels, then ONE backward kernel that writes dL/dimage -- the upstream gradient the rasterizer's backward
    reads -- without materialising the two per-term gradient images and their sum."""

    @staticmethod
    def forward(ctx, img, gt, mask, lambda_dssim, lambda_l1):
        L = _native.lib()
        Cc, H, W = img.shape
        dev = img.device
        out_s = torch.empty(1, dtype=torch.float32, device=dev)
        out_l = torch.empty(2, dtype=torch.float32, device=dev)
        ws = torch.empty(max(L.sgr_ssim_workspace_floats(Cc, H, W), L.sgr_l1_workspace_floats(Cc, H, W)),
                         dtype=torch.float32, device=dev)
        partials = torch.empty(3 * Cc * H * W, dtype=torch.float32, device=dev) if img.requires_grad else None
        with torch.cuda.device(dev):
            check(L.sgr_l1_forward(Cc, H, W, _p(img), _p(gt), _p(mask), _p(out_l), _p(ws), _stream(dev)))
            check(L.sgr_ssim_forward(Cc, H, W, _p(img), _p(gt), _p(mask), _p(out_s), _p(partials), _p(ws), _stream(dev)))
        ctx.save_for_backward(img, gt, mask if mask is not None else torch.empty(0, device=dev),
                              partials if partials is not None else torch.empty(0, device=dev), out_l)
        ctx.has_mask = mask is not None
        ctx.w_l1, ctx.w_ssim = (1.0 - float(lambda_dssim)) * float(lambda_l1), -float(lambda_dssim)
        loss = ctx.w_l1 * out_l[0] + float(lambda_dssim) * (1.0 - out_s[0])
        ctx.mark_non_differentiable(out_l)
        return loss, out_l

    @staticmethod
    def backward(ctx, upstream, _unused):
        img, gt, mask, partials, out_l = ctx.saved_tensors
        Cc, H, W = img.shape
        dev = img.device
        if partials.numel() == 0:
            raise RuntimeError("color_loss forward ran without requires_grad")
        up = upstream.reshape(1).to(torch.float32).contiguous()
        grad = torch.empty_like(img)
        with torch.cuda.device(dev):
            check(_native.lib().sgr_color_loss_backward(Cc, H, W, _p(img), _p(gt), _p(mask) if ctx.has_mask else None,
                                                        _p(partials), _p(out_l), ctx.w_l1, ctx.w_ssim, _p(up), _p(grad),
                                                        _stream(dev)))
        return grad, None, None, None, None


def color_loss(image: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None, lambda_dssim: float = 0.2,
               lambda_l1: float = 1.0, return_l1: bool = False):
    """train.py:100-104 in one op: ``(1 - lambda_dssim) * lambda_l1 * l1_loss(image, gt, mask) + lambda_dssim *
    (1 - ssim(image, gt, mask=mask))``; its backward is a single kernel producing dL/dimage for the rasterizer.
    With ``return_l1`` also returns the (detached) L1 term train.py logs (``scalar_dict['l1_loss']``)."""
    a, b = _prep(image, "image"), _prep(gt, "gt")
    if a.shape != b.shape:
        raise RuntimeError("image and gt must have the same shape")
    loss, out_l = _ColorLoss.apply(a, b.detach(), _prep_mask(mask, a.shape[1], a.shape[2]), lambda_dssim, lambda_l1)
    return (loss, out_l[0]) if return_l1 else loss


def l1_loss(network_output: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """loss_utils.py:21-37: mean |network_output - gt| over the C values of the pixels selected by mask (1, H, W)."""
    a, b = _prep(network_output, "network_output"), _prep(gt, "gt")
    if a.shape != b.shape:
        raise RuntimeError("network_output and gt must have the same shape")
    return _L1.apply(a, b.detach(), _prep_mask(mask, a.shape[1], a.shape[2]))


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True,
         mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """loss_utils.py:80-125: mean SSIM (11x11 Gaussian window, sigma 1.5, zero padding), both images zeroed where the
    mask is False.  Gradient flows to img1 only (the ground truth is data)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("only window_size=11, size_average=True (the values train.py uses) are implemented")
    a, b = _prep(img1, "img1"), _prep(img2, "img2")
    if a.shape != b.shape:
        raise RuntimeError("img1 and img2 must have the same shape")
    return _SSIM.apply(a, b.detach(), _prep_mask(mask, a.shape[1], a.shape[2]))
