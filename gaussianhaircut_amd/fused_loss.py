"""Fused photometric loss of the stage-1 step (csrc/ghr_loss.h): masked L1 + (1 - SSIM) + mask L1 in two HIP kernels
instead of 10 MIOpen depthwise convolutions + ~40 elementwise kernels (src/train_gaussians.py:126-140,
src/utils/loss_utils.py:19-26,91-121)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .diff_gaussian_rasterization import _ptr, _stream


class _PhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, mask, gt_image, gt_mask, w_l1, w_ssim, w_mask):
        assert image.is_cuda, "fused loss has no CPU path"
        _, H, W = image.shape
        image_c, mask_c = image.detach().float().contiguous(), mask.detach().float().contiguous()
        gt_image_c, gt_mask_c = gt_image.detach().float().contiguous(), gt_mask.detach().float().contiguous()
        dev = image.device
        with torch.cuda.device(dev):
            maps = torch.empty((9, H, W), dtype=torch.float32, device=dev)
            sums = torch.empty(768, dtype=torch.float32, device=dev)  # GHR_LOSS_SUMS
            loss = torch.empty((), dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().ghr_loss_forward(_stream(), W, H, _ptr(image_c), _ptr(mask_c), _ptr(gt_image_c),
                                                   _ptr(gt_mask_c), w_l1, w_ssim, w_mask, _ptr(maps), _ptr(sums),
                                                   ctypes.c_void_p(loss.data_ptr())))
        ctx.save_for_backward(image_c, mask_c, gt_image_c, gt_mask_c, maps)
        ctx.w = (w_l1, w_ssim, w_mask)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        image, mask, gt_image, gt_mask, maps = ctx.saved_tensors
        _, H, W = image.shape
        dev = image.device
        with torch.cuda.device(dev):
            d_image = torch.empty_like(image)
            d_mask = torch.empty_like(mask)
            gl = grad_loss.detach().float().contiguous()
            _lib.check(_lib.lib().ghr_loss_backward(_stream(), W, H, _ptr(image), _ptr(mask), _ptr(gt_image),
                                                    _ptr(gt_mask), _ptr(maps), ctypes.c_void_p(gl.data_ptr()),
                                                    ctx.w[0], ctx.w[1], ctx.w[2], _ptr(d_image), _ptr(d_mask)))
        return d_image, d_mask, None, None, None, None, None


def photometric_loss(image, mask, gt_image, gt_mask, w_l1, w_ssim, w_mask):
    """w_l1 * l1_loss(image, gt, mask=gt_mask[1:]) + w_ssim * (1 - ssim(image*m, gt*m)) + w_mask * l1_loss(mask, gt_mask)."""
    return _PhotometricLoss.apply(image, mask, gt_image, gt_mask, float(w_l1), float(w_ssim), float(w_mask))
