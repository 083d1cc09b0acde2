"""Losses of the stage-1 step (reference: ``src/utils/loss_utils.py:19-47,91-121``; used train_gaussians.py:126-140)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt, weight=None, mask=None):
    err = (network_output - gt).abs()
    if mask is not None:
        err = err * mask
    if weight is not None:
        return (err * weight).sum() / weight.sum()
    return err.mean()


def or_loss(network_output, gt, confs=None, weight=None, mask=None):
    """Orientation loss on angles in [0,1) with wrap-around (loss_utils.py:31-47)."""
    weight = torch.ones_like(gt[:1]) if weight is None else weight
    d = network_output - gt
    err = torch.minimum(d.abs(), torch.minimum((d - 1).abs(), (d + 1).abs())) * math.pi
    if confs is not None:
        err = err * confs - (confs + 1e-7).log()
    if mask is not None:
        err = err * mask
    return (err * weight).sum() / weight.sum()


_window_cache = {}


def _window(size: int, channel: int, like: torch.Tensor) -> torch.Tensor:
    key = (size, channel, like.device, like.dtype)
    w = _window_cache.get(key)
    if w is None:
        g = torch.tensor([math.exp(-(i - size // 2) ** 2 / (2 * 1.5 ** 2)) for i in range(size)], dtype=torch.float32)
        g = (g / g.sum()).unsqueeze(1)
        w = (g @ g.t()).expand(channel, 1, size, size).contiguous().to(device=like.device, dtype=like.dtype)
        _window_cache[key] = w
    return w


def ssim(img1, img2, window_size=11, size_average=True):
    """Gaussian-window SSIM, sigma 1.5 (loss_utils.py:91-121)."""
    ch = img1.size(-3)
    w = _window(window_size, ch, img1)
    pad = window_size // 2

    def blur(t):
        return F.conv2d(t, w, padding=pad, groups=ch)

    mu1, mu2 = blur(img1), blur(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = blur(img1 * img1) - mu1_sq
    s2 = blur(img2 * img2) - mu2_sq
    s12 = blur(img1 * img2) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return m.mean() if size_average else m
