"""Fused loss of the stage-1 step (csrc/ghr_loss.h): masked L1 + (1 - SSIM) + mask L1 + orientation loss in two HIP
kernels instead of 10 MIOpen depthwise convolutions + ~60 elementwise kernels (src/train_gaussians.py:126-140,
src/utils/loss_utils.py:19-47,91-121, and the orientation-angle post-processing of
src/gaussian_renderer/__init__.py:100-105)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .diff_gaussian_rasterization import _on_device, _ptr, _stream


def _off(t: torch.Tensor, plane: int, n: int):
    return ctypes.c_void_p(t.data_ptr() + 4 * plane * n)


def _f32c(t):
    return t.detach().float().contiguous()


def _args(W, H, image, mask, dir2d, oconf, gt_image, gt_mask, gt_angle, gt_oconf, w, unmasked=False, gt_stats=None):
    a = _lib.LossArgs()
    a.W, a.H = int(W), int(H)
    a.image, a.mask, a.dir2d, a.orient_conf = image, mask, dir2d, oconf
    a.gt_image, a.gt_mask = _ptr(gt_image), _ptr(gt_mask)
    a.gt_orient_angle = _ptr(gt_angle) if gt_angle is not None else None
    a.gt_orient_conf = _ptr(gt_oconf) if gt_oconf is not None else None
    a.w_l1, a.w_ssim, a.w_mask, a.w_orient = [float(x) for x in w]
    a.unmasked_colours = int(bool(unmasked))
    a.gt_stats = _ptr(gt_stats) if gt_stats is not None else None
    return a


def gt_ssim_stats(gt_image, gt_mask, mask_colours=True):
    """[2,3,H,W] SSIM window moments of the (masked) ground truth: constants of a training view.  Pass them to
    ``stage1_loss(..., gt_stats=)`` and its forward convolves three window moments instead of five; the values are
    bit-identical to what it would compute itself."""
    assert gt_image.is_cuda, "fused loss has no CPU path"
    _, H, W = gt_image.shape
    gi, gm = _f32c(gt_image), _f32c(gt_mask)
    with _on_device(gi.device):
        out = torch.empty((2, 3, H, W), dtype=torch.float32, device=gi.device)
        a = _args(W, H, None, None, None, None, gi, gm, None, None, (0.0, 0.0, 0.0, 0.0), not mask_colours)
        _lib.check(_lib.lib().ghr_loss_gt_stats(_stream(), ctypes.byref(a), _ptr(out)))
    return out


class _Stage1LossPacked(torch.autograd.Function):
    """Loss on the rasterizer's packed [10,H,W] output; backward returns the packed [10,H,W] gradient in one pass
    (no split / cat / zero-fill kernels between the loss and the rasterizer backward)."""

    @staticmethod
    def forward(ctx, renders, gt_image, gt_mask, gt_angle, gt_oconf, w_l1, w_ssim, w_mask, w_orient, unmasked=False,
                gt_stats=None):
        assert renders.is_cuda, "fused loss has no CPU path"
        C, H, W = renders.shape
        assert C == _lib.NUM_CHANNELS
        r = _f32c(renders)
        n = H * W
        gt_image_c, gt_mask_c = _f32c(gt_image), _f32c(gt_mask)
        orient = float(w_orient) != 0.0
        gt_angle_c = _f32c(gt_angle) if orient else None
        gt_oconf_c = _f32c(gt_oconf) if orient else None
        dev = renders.device
        with _on_device(dev):
            maps = torch.empty((9, H, W), dtype=torch.float32, device=dev)
            sums = torch.empty(_lib.loss_sums_floats(W, H), dtype=torch.float32, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            if gt_stats is not None:
                assert gt_stats.shape == (2, 3, H, W) and gt_stats.is_contiguous() and gt_stats.dtype == torch.float32
            a = _args(W, H, _off(r, 0, n), _off(r, 3, n), _off(r, 5, n), _off(r, 8, n), gt_image_c, gt_mask_c,
                      gt_angle_c, gt_oconf_c, (w_l1, w_ssim, w_mask, w_orient if orient else 0.0), unmasked, gt_stats)
            _lib.check(_lib.lib().ghr_loss_forward(_stream(), ctypes.byref(a), _ptr(maps), _ptr(sums),
                                                   ctypes.c_void_p(loss.data_ptr())))
        ctx.save_for_backward(r, gt_image_c, gt_mask_c, maps, sums, *([gt_angle_c, gt_oconf_c] if orient else []))
        ctx.w = (w_l1, w_ssim, w_mask, w_orient if orient else 0.0)
        ctx.unmasked = bool(unmasked)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        r, gt_image, gt_mask, maps, sums, *o = ctx.saved_tensors
        gt_angle, gt_oconf = (o[0], o[1]) if o else (None, None)
        _, H, W = r.shape
        n = H * W
        dev = r.device
        with _on_device(dev):
            d = torch.empty_like(r)
            gl = _f32c(grad_loss)
            a = _args(W, H, _off(r, 0, n), _off(r, 3, n), _off(r, 5, n), _off(r, 8, n), gt_image, gt_mask, gt_angle,
                      gt_oconf, ctx.w, ctx.unmasked)
            _lib.check(_lib.lib().ghr_loss_backward(_stream(), ctypes.byref(a), _ptr(maps), _ptr(sums),
                                                    ctypes.c_void_p(gl.data_ptr()), _off(d, 0, n), _off(d, 3, n),
                                                    _off(d, 5, n), _off(d, 8, n), _off(d, 7, n), _off(d, 9, n)))
        return d, None, None, None, None, None, None, None, None, None, None


class _PhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, mask, gt_image, gt_mask, w_l1, w_ssim, w_mask):
        assert image.is_cuda, "fused loss has no CPU path"
        _, H, W = image.shape
        image_c, mask_c = _f32c(image), _f32c(mask)
        gt_image_c, gt_mask_c = _f32c(gt_image), _f32c(gt_mask)
        dev = image.device
        with _on_device(dev):
            maps = torch.empty((9, H, W), dtype=torch.float32, device=dev)
            sums = torch.empty(_lib.loss_sums_floats(W, H), dtype=torch.float32, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            a = _args(W, H, _ptr(image_c), _ptr(mask_c), None, None, gt_image_c, gt_mask_c, None, None,
                      (w_l1, w_ssim, w_mask, 0.0))
            _lib.check(_lib.lib().ghr_loss_forward(_stream(), ctypes.byref(a), _ptr(maps), _ptr(sums),
                                                   ctypes.c_void_p(loss.data_ptr())))
        ctx.save_for_backward(image_c, mask_c, gt_image_c, gt_mask_c, maps, sums)
        ctx.w = (w_l1, w_ssim, w_mask, 0.0)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        image, mask, gt_image, gt_mask, maps, sums = ctx.saved_tensors
        _, H, W = image.shape
        dev = image.device
        with _on_device(dev):
            d_image = torch.empty_like(image)
            d_mask = torch.empty_like(mask)
            gl = _f32c(grad_loss)
            a = _args(W, H, _ptr(image), _ptr(mask), None, None, gt_image, gt_mask, None, None, ctx.w)
            _lib.check(_lib.lib().ghr_loss_backward(_stream(), ctypes.byref(a), _ptr(maps), _ptr(sums),
                                                    ctypes.c_void_p(gl.data_ptr()), _ptr(d_image), _ptr(d_mask), None,
                                                    None, None, None))
        return d_image, d_mask, None, None, None, None, None


def photometric_loss(image, mask, gt_image, gt_mask, w_l1, w_ssim, w_mask):
    """w_l1 * l1_loss(image, gt, mask=gt_mask[1:]) + w_ssim * (1 - ssim(image*m, gt*m)) + w_mask * l1_loss(mask, gt_mask)."""
    return _PhotometricLoss.apply(image, mask, gt_image, gt_mask, float(w_l1), float(w_ssim), float(w_mask))


def stage1_loss(renders, gt_image, gt_mask, gt_orient_angle, gt_orient_conf, w_l1, w_ssim, w_mask, w_orient,
                mask_colours=True, gt_stats=None):
    """The whole loss of src/train_gaussians.py:126-140 on the packed [10,H,W] rasterizer output ``renders``
    (channels: rgb 0-2, mask 3-4, 2D direction 5-6, orientation confidence 8):
    photometric terms + ``w_orient * or_loss(orient_angle, gt_orient_angle, orient_conf, weight=gt_orient_conf,
    mask=gt_mask[:1])`` with a NaN orientation term dropped.  ``mask_colours=False``: L1 / SSIM on the whole image, the
    strand-stage form (src/train_strands.py:128-129).  ``gt_stats``: optional ``gt_ssim_stats(gt_image, gt_mask,
    mask_colours)`` of this view (same result, 40 % less window arithmetic in the forward)."""
    return _Stage1LossPacked.apply(renders, gt_image, gt_mask, gt_orient_angle, gt_orient_conf, float(w_l1),
                                   float(w_ssim), float(w_mask), float(w_orient), not mask_colours, gt_stats)
