"""Strand polylines -> segment Gaussians (csrc/ghr_strands.h, ABI 18): the product's per-strand / per-row functions against the
PyTorch form of ``initialize_gaussians_hair`` (src/scene/gaussian_model_strands.py:435-452), which
tests/test_reference_golden.py pins to the reference's own module.  CPU: the functions through the host simulator; GPU: the
kernels through the C ABI and the model's autograd function."""
import ctypes

import numpy as np
import pytest
import torch

from gaussianhaircut_amd.scene import gaussian_model_strands as gms
from gaussianhaircut_amd.scene.gaussian_model_strands import GaussianModelStrands


def _strands(S, n_seg, seed, dev="cpu", degenerate=True):
    g = torch.Generator().manual_seed(seed)
    unit = torch.nn.functional.normalize
    origins = unit(torch.randn(S, 1, 3, generator=g), dim=-1)
    dirs = torch.randn(S, n_seg, 3, generator=g) * 0.003 + unit(torch.randn(S, 1, 3, generator=g), dim=-1) * 0.01
    if degenerate and S > 2 and n_seg > 2:
        dirs[0, 1] = 0.0                                  # a zero-length segment: F.normalize's eps, norm's backward at 0
        dirs[1, 0] = torch.tensor([-0.01, 0.0, 0.0])      # anti-parallel to the x axis: quaternion (0, 0, 0, 0)
        dirs[2, 2] = torch.tensor([1e-13, 0.0, 0.0])      # below the eps
    feats = torch.randn(S * n_seg, 16, 3, generator=g) * 0.1
    return origins.to(dev), dirs.to(dev), feats.to(dev)


def _torch_form(origins, dirs, scale, cots=None, double=False):
    """The PyTorch form (and its autograd gradients for the cotangents ``cots``)."""
    m = GaussianModelStrands(3, scale=scale)
    dt = torch.float64 if double else torch.float32
    m.pts_origins = origins.to(dt)
    m._dirs = torch.nn.Parameter(dirs.to(dt).clone())
    m._initialize_gaussians_hair_torch()
    out = (m._xyz, m._rotation, m._scaling)
    if cots is None:
        return [o.detach() for o in out], None
    (sum((o * c.to(dt)).sum() for o, c in zip(out, cots))).backward()
    return [o.detach() for o in out], m._dirs.grad


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


@pytest.mark.parametrize("S,n_seg", [(7, 99), (3, 1), (40, 5), (2, 300)])
def test_hostsim_strand_functions_match_the_torch_form(hostsim, S, n_seg):
    origins, dirs, _ = _strands(S, n_seg, seed=S + n_seg)
    scale = 1e-3
    P = S * n_seg
    xyz, rot, sc = np.zeros((P, 3), np.float32), np.zeros((P, 4), np.float32), np.zeros((P, 3), np.float32)
    o_np, d_np = _np(origins.reshape(S, 3)), _np(dirs)
    hostsim.L.ghrsim_strand_build(S, n_seg, _p(o_np), _p(d_np), ctypes.c_float(scale), _p(xyz), _p(rot), _p(sc))
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g), torch.randn(P, 3, generator=g)]
    (xyz_t, rot_t, sc_t), gd = _torch_form(origins, dirs, scale, cots)
    # (ATen's CPU cumsum accumulates a float row in DOUBLE; its device scan, like the kernel, in float: bit-equal on the GPU only)
    seq = np.cumsum(d_np.astype(np.float32), axis=1, dtype=np.float32)
    pts = o_np[:, None, :] + np.concatenate([np.zeros((S, 1, 3), np.float32), seq], axis=1)
    assert np.array_equal(xyz, ((pts[:, 1:] + pts[:, :-1]) * np.float32(0.5)).reshape(-1, 3))
    assert np.allclose(xyz, xyz_t.numpy(), rtol=0, atol=2e-6)
    assert np.allclose(rot, rot_t.numpy(), rtol=0, atol=3e-7) and np.array_equal(rot[:, 1], np.zeros(P, np.float32))
    assert np.allclose(sc, sc_t.numpy(), rtol=3e-7, atol=0)
    c_np = [_np(c) for c in cots]
    for mask in ((1, 1, 1), (1, 0, 0), (0, 1, 1), (0, 0, 0)):
        used = [c if k else None for c, k in zip(cots, mask)]
        out = np.full((S, n_seg, 3), np.nan, np.float32)
        hostsim.L.ghrsim_strand_build_backward(S, n_seg, _p(d_np), *[_p(c) if k else None for c, k in zip(c_np, mask)], _p(out))
        _, gd64 = _torch_form(origins, dirs, scale, [c if k else torch.zeros_like(c) for c, k in zip(cots, mask)], double=True)
        ref = gd64.numpy()
        ok = np.ones((S, n_seg), bool)
        if S > 2 and n_seg > 2:
            ok[2, 2] = False  # |dir| < eps: the double chain's clamp and the fp32 kernels see different sides of 1e-12
        tol = 2e-5 * np.abs(ref[ok]).max() if np.abs(ref[ok]).max() > 0 else 0.0
        assert np.isfinite(out).all()
        assert np.abs(out[ok] - ref[ok]).max() <= tol, (mask, np.abs(out[ok] - ref[ok]).max(), tol)
        del used
    # fp32 autograd of the PyTorch form is no closer to the double chain than the kernels' arithmetic
    assert np.isfinite(gd.numpy()[ok]).all()


def test_strands_per_block_fits_the_lds_for_every_length():
    # the launch's LDS request: 2 x spb x n_seg x 12 B <= 48 KB up to GHR_STRAND_MAX_SEG (mirrors csrc/ghr_strands.h)
    from gaussianhaircut_amd import _lib
    for n_seg in (1, 2, 13, 99, 100, 455, 456, 1365, 1366, _lib.STRAND_MAX_SEG):
        spb = max(1, min(256 // 3, 32768 // (24 * n_seg)))
        assert 2 * spb * n_seg * 12 <= 48 * 1024 and 3 * spb <= 256


def test_model_uses_the_torch_form_on_the_cpu_and_keeps_pts():
    origins, dirs, feats = _strands(5, 9, seed=1)
    m = GaussianModelStrands(3).create_from_strands(origins, dirs, feats)
    assert m._pts.shape == (5, 10, 3) and torch.equal(m._pts[:, 0], origins[:, 0])
    assert not gms._strand_build_applies(m.pts_origins, m._dirs)


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("S,n_seg", [(30000, 99), (1000, 1), (513, 300), (3, 2048), (100, 7)])
def test_gpu_strand_build_matches_the_torch_form(S, n_seg):
    dev = torch.device("cuda:0")
    origins, dirs, feats = _strands(S, n_seg, seed=3, dev=dev)
    P = S * n_seg
    g = torch.Generator().manual_seed(6)
    cots = [torch.randn(P, k, generator=g).to(dev) for k in (3, 4, 3)]
    (xyz_t, rot_t, sc_t), gd_t = _torch_form(origins, dirs, 1e-3, cots)
    _, gd64 = _torch_form(origins, dirs, 1e-3, cots, double=True)
    d = torch.nn.Parameter(dirs.clone())
    assert gms._strand_build_applies(origins, d)
    xyz, rot, sc, rows = gms._StrandBuild.apply(origins, d, 1e-3)
    assert torch.equal(xyz, xyz_t)
    assert rows.shape == (P, 3) and rows.data_ptr() == d.data_ptr() and type(rows.grad_fn).__name__.startswith("_StrandBuild")
    assert torch.allclose(rot, rot_t, rtol=0, atol=3e-7) and float(rot.detach()[:, 1].abs().max()) == 0.0
    assert torch.allclose(sc, sc_t, rtol=3e-7, atol=0)
    (sum((o * c).sum() for o, c in zip((xyz, rot, sc), cots))).backward()
    ok = torch.ones(S, n_seg, dtype=torch.bool, device=dev)
    if S > 2 and n_seg > 2:
        ok[2, 2] = False
    ref = gd64[ok]
    scale = float(ref.abs().max())
    err_k, err_t = float((d.grad[ok].double() - ref).abs().max()), float((gd_t[ok].double() - ref).abs().max())
    assert torch.isfinite(d.grad).all()
    assert err_k <= max(2e-5 * scale, 2.0 * err_t), (err_k, err_t, scale)
    # partial cotangents (an output nobody differentiated)
    d2 = torch.nn.Parameter(dirs.clone())
    xyz2, _, _, _ = gms._StrandBuild.apply(origins, d2, 1e-3)
    (xyz2 * cots[0]).sum().backward()
    _, gx64 = _torch_form(origins, dirs, 1e-3, [cots[0], torch.zeros_like(cots[1]), torch.zeros_like(cots[2])], double=True)
    assert float((d2.grad.double() - gx64).abs().max()) <= 2e-5 * float(gx64.abs().max())
    # the direction rows as a fourth output (round 6): their cotangent is added by the kernel, last -- the bits of autograd
    # summing the node's gradient and the rows' own (what the graph did while `_dir` was a plain view of `_dirs`)
    c_rows = torch.randn(P, 3, generator=g).to(dev)
    d3, d4 = torch.nn.Parameter(dirs.clone()), torch.nn.Parameter(dirs.clone())
    o3 = gms._StrandBuild.apply(origins, d3, 1e-3)
    (sum((o * c).sum() for o, c in zip(o3[:3], cots)) + (o3[3] * c_rows).sum()).backward()
    o4 = gms._StrandBuild.apply(origins, d4, 1e-3)
    (sum((o * c).sum() for o, c in zip(o4[:3], cots)) + (d4.reshape(-1, 3) * c_rows).sum()).backward()
    assert torch.equal(d3.grad, d4.grad)
    d5 = torch.nn.Parameter(dirs.clone())
    (gms._StrandBuild.apply(origins, d5, 1e-3)[3] * c_rows).sum().backward()   # the rows alone
    assert torch.equal(d5.grad.reshape(-1, 3)[ok.reshape(-1)], c_rows[ok.reshape(-1)])


@pytest.mark.gpu
def test_gpu_strand_model_builds_through_the_kernel_and_refuses_bad_shapes():
    from gaussianhaircut_amd import _lib
    if not gms.FUSED_STRAND_BUILD:
        pytest.skip("GHR_FUSED_STRAND_BUILD=0: the model takes the PyTorch form")
    dev = torch.device("cuda:0")
    origins, dirs, feats = _strands(50, 20, seed=4, dev=dev)
    m = GaussianModelStrands(3).create_from_strands(origins, dirs, feats)
    ref = GaussianModelStrands(3)
    ref.pts_origins, ref._dirs = m.pts_origins, m._dirs
    ref._initialize_gaussians_hair_torch()
    assert m._xyz.grad_fn is not None and type(m._xyz.grad_fn).__name__.startswith("_StrandBuild")
    assert torch.equal(m._xyz, ref._xyz) and torch.equal(m._dir, ref._dir)
    assert torch.equal(m._pts, ref._pts)
    assert torch.allclose(m.get_scaling, ref.get_scaling, rtol=3e-7, atol=0)
    L = _lib.lib()
    z = torch.zeros(16, device=dev)
    p = ctypes.c_void_p(z.data_ptr())
    assert L.ghr_strand_build(None, 1, _lib.STRAND_MAX_SEG + 1, p, p, 1e-3, p, p, p) == _lib.GHR_E_INVALID
    assert L.ghr_strand_build(None, 1, 1, None, p, 1e-3, p, p, p) == _lib.GHR_E_INVALID
    assert L.ghr_strand_build(None, 0, 5, None, None, 1e-3, None, None, None) == 0
    assert L.ghr_strand_build_backward(None, 1, 1, p, None, None, None, None) == _lib.GHR_E_INVALID
