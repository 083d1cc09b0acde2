"""`-m "not gpu"` layer for the fused loss and fused Adam: their per-pixel / per-element arithmetic is `__host__
__device__` (csrc/ghr_loss.h: orient_pixel, ssim_point; csrc/ghr_adam.h: adam_update) and runs here on the CPU through
the host-sim scaffold, against PyTorch autograd of the reference's formulas (src/utils/loss_utils.py:31-47,91-121;
src/gaussian_renderer/__init__.py:100-105) and torch.optim.Adam."""
import ctypes
import math

import numpy as np
import pytest
import torch

from tests import helpers as hp


@pytest.fixture(scope="module")
def sim():
    return hp.HostSim().L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_orientation_pixel_value_and_gradients_match_autograd(sim):
    g = torch.Generator().manual_seed(0)
    n = 4000
    d = torch.randn(n, 2, generator=g, dtype=torch.float64) * 0.4
    conf = torch.rand(n, generator=g, dtype=torch.float64) * 2 + 0.05
    gt = torch.rand(n, generator=g, dtype=torch.float64)
    m = (torch.rand(n, generator=g) > 0.2).double()
    d.requires_grad_(True)
    conf.requires_grad_(True)
    # the reference chain: normalize -> mirror -> clamp -> acos / pi (gaussian_renderer/__init__.py:100-105), then
    # or_loss per pixel (loss_utils.py:31-47) without the final weighted mean
    u = d / d.norm(dim=1, keepdim=True).clamp_min(1e-12)
    mirror = torch.where(u[:, 0] < 0, -torch.ones_like(u[:, 0]), torch.ones_like(u[:, 0]))
    c = u[:, 1].clamp(-1 + 1e-3, 1 - 1e-3) * mirror
    angle = torch.acos(c) / math.pi
    diff = angle - gt
    lmin = torch.minimum(diff.abs(), torch.minimum((diff - 1).abs(), (diff + 1).abs()))
    loss = (lmin * math.pi * conf - torch.log(conf + 1e-7)) * m
    loss.sum().backward()
    out = np.zeros((n, 4), np.float32)
    f = lambda t: np.ascontiguousarray(t.detach().numpy().astype(np.float32))
    a = [f(d[:, 0]), f(d[:, 1]), f(conf), f(gt), f(m)]
    sim.ghrsim_orient_pixel(n, *[_p(x) for x in a], _p(out))
    # away from the kinks (clamp edges, wrapped-difference switches, mirror flip) value and derivatives are smooth
    dd = diff.detach().numpy()
    uu = u.detach().numpy()
    ok = (np.abs(np.abs(dd) - 0.5) > 1e-3) & (np.abs(dd) > 1e-3) & (np.abs(uu[:, 0]) > 1e-3) & (np.abs(uu[:, 1]) < 0.998)
    assert ok.mean() > 0.9
    ref = np.stack([loss.detach().numpy(), d.grad[:, 0].numpy(), d.grad[:, 1].numpy(), conf.grad.numpy()], 1)
    err = np.abs(out[ok] - ref[ok])
    assert (err <= 2e-4 * (1.0 + np.abs(ref[ok]))).all(), err.max()


def test_ssim_point_value_and_partials_match_autograd(sim):
    g = torch.Generator().manual_seed(1)
    n = 5000
    mu1 = torch.rand(n, generator=g, dtype=torch.float64).requires_grad_(True)
    mu2 = torch.rand(n, generator=g, dtype=torch.float64)
    e11 = (mu1.detach() ** 2 + torch.rand(n, generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    e22 = mu2 ** 2 + torch.rand(n, generator=g, dtype=torch.float64) * 0.05
    e12 = (mu1.detach() * mu2 + (torch.rand(n, generator=g, dtype=torch.float64) - 0.5) * 0.04).requires_grad_(True)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    s1, s2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    val = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))  # loss_utils.py:117
    val.sum().backward()
    out = np.zeros((n, 4), np.float32)
    f = lambda t: np.ascontiguousarray(t.detach().numpy().astype(np.float32))
    sim.ghrsim_ssim_point(n, _p(f(mu1)), _p(f(mu2)), _p(f(e11)), _p(f(e22)), _p(f(e12)), _p(out))
    ref = np.stack([val.detach().numpy(), mu1.grad.numpy(), e11.grad.numpy(), e12.grad.numpy()], 1)
    scale = np.abs(ref).max(0)
    assert (np.abs(out - ref) <= 3e-5 * scale[None, :] + 1e-4 * np.abs(ref)).all()


def test_adam_update_matches_torch_adam(sim):
    g = torch.Generator().manual_seed(2)
    n = 3000
    p0 = torch.randn(n, generator=g)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=1.6e-4, eps=1e-15)
    p = p0.numpy().copy()
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        pt.grad = grad.clone()
        opt.step()
        gn = np.ascontiguousarray(grad.numpy())
        sim.ghrsim_adam(n, _p(p), _p(gn), _p(m), _p(v), ctypes.c_float(1.6e-4), ctypes.c_double(0.9),
                        ctypes.c_double(0.999), ctypes.c_float(1e-15), step)
        np.testing.assert_allclose(p, pt.detach().numpy(), rtol=2e-6, atol=1e-7)
