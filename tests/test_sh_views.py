"""SH-coefficient gradients from per-view factors (ABI 19: ``ghr_model_args.d_rgb`` + ``ghr_sh_grad_from_views``): what the
data-parallel step sends instead of the 48 SH gradient floats per Gaussian.  CPU: the product's per-Gaussian functions through
the host simulator -- the rebuilt gradients are the BITS the projection backward stores for one view, and the bits a single
process accumulating several views ends with; the SH-free buffers may be NULL.  GPU tests: tests/test_gpu_dist_shared.py
(two ranks) and tests/test_gpu_fused.py (kernel against the stored gradients)."""
import ctypes

import numpy as np
import pytest
import torch

from gaussianhaircut_amd.scene.cameras import ring_cameras
from gaussianhaircut_amd.utils import synthetic as syn
from tests import helpers as hp


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _backward(hostsim, model, cam, gacc, radii, P, K, with_sh=True):
    keep = []
    a = hp.model_args_from(model, cam, keep)
    outs = dict(d_means2D=np.zeros((P, 3), np.float32), d_xyz=np.zeros((P, 3), np.float32), d_ls=np.zeros((P, 3), np.float32),
                d_rot=np.zeros((P, 4), np.float32), d_op=np.zeros(P, np.float32), d_label=np.zeros(P, np.float32),
                d_conf=np.zeros(P, np.float32), d_fdc=np.zeros((P, 1, 3), np.float32),
                d_frest=np.zeros((P, K - 1, 3), np.float32))
    d_rgb = np.full((P, 3), np.nan, np.float32)
    hostsim.L.ghrsim_set_d_rgb(_p(d_rgb))
    try:
        names = ("d_means2D", "d_xyz", "d_ls", "d_rot", "d_op", "d_label", "d_conf", "d_fdc", "d_frest")
        hostsim.L.ghrsim_project_backward3(ctypes.byref(a), _p(radii), _p(gacc),
                                           *[_p(outs[k]) if (with_sh or k != "d_fdc") else None for k in names], None, None, 0)
    finally:
        hostsim.L.ghrsim_set_d_rgb(None)
    return outs, d_rgb


@pytest.mark.parametrize("cfg,deg", [("tiny", 3), ("tiny_strands", 2), ("tiny", 0)])
def test_rebuilt_sh_gradients_have_the_bits_of_the_stored_and_of_the_accumulated_ones(hostsim, cfg, deg):
    spec = syn.CONFIGS[cfg]
    model = syn.make_model(spec)
    model.active_sh_degree = deg
    P, K = spec.P, 16
    cams = ring_cameras(3, spec.W, spec.H)
    g = torch.Generator().manual_seed(3)
    xyz = np.ascontiguousarray(model.get_xyz.detach().numpy().astype(np.float32))
    per_view, tables, campos = [], [], []
    for v, cam in enumerate(cams):
        with torch.no_grad():
            model.get_conic(cam)
            model.get_mean_2d(cam)
        keepv = model.filter_points(cam).numpy()
        gacc = (torch.randn(P, 16, generator=g).numpy() * keepv[:, None]).astype(np.float32)
        radii = (keepv * 1).astype(np.int32)
        if v == 1:
            radii[: P // 3] = 0  # culled in this view: no gradient, whatever the lines hold
        outs, d_rgb = _backward(hostsim, model, cam, gacc, radii, P, K)
        assert np.isfinite(d_rgb).all() and np.abs(d_rgb[radii == 0]).max(initial=0.0) == 0.0
        per_view.append(outs)
        tables.append(d_rgb)
        campos.append(cam.camera_center.numpy().astype(np.float32))
        # the other gradients do not depend on whether the SH gradients are stored
        if v == 0:
            outs_b, d_rgb_b = _backward(hostsim, model, cam, gacc, radii, P, K, with_sh=False)
            assert np.array_equal(d_rgb, d_rgb_b)
            for k in ("d_means2D", "d_xyz", "d_ls", "d_rot", "d_op", "d_label", "d_conf"):
                assert np.array_equal(outs[k], outs_b[k]), k
    g_views = np.ascontiguousarray(np.stack(tables))             # [V, P, 3]
    cpos = np.ascontiguousarray(np.stack(campos))
    for n_views in (1, 2, 3):
        d_dc, d_rest = np.full((P, 1, 3), np.nan, np.float32), np.full((P, K - 1, 3), np.nan, np.float32)
        hostsim.L.ghrsim_sh_grad_from_views(P, deg, K, _p(xyz), n_views, _p(cpos), _p(g_views), ctypes.c_longlong(3 * P),
                                            _p(d_dc), _p(d_rest))
        # one process accumulating the same views: view 0 assigned, the others added (fp32, in order)
        acc_dc, acc_rest = per_view[0]["d_fdc"].copy(), per_view[0]["d_frest"].copy()
        for v in range(1, n_views):
            acc_dc = (per_view[v]["d_fdc"] + acc_dc).astype(np.float32)
            acc_rest = (per_view[v]["d_frest"] + acc_rest).astype(np.float32)
        assert np.array_equal(d_dc, acc_dc) and np.array_equal(d_rest, acc_rest), n_views   # (+0 == -0)
        assert np.abs(acc_dc).max() > 0
        if deg < 3:
            assert np.abs(d_rest[:, (deg + 1) ** 2 - 1:]).max() == 0.0
        if deg > 0:
            assert np.abs(d_rest[:, : (deg + 1) ** 2 - 1]).max() > 0


def test_rebuild_skips_views_without_a_gradient_and_carries_non_finite_ones(hostsim):
    P, K = 4, 16
    xyz = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    cpos = np.array([[0, 0, 0], [0, 0, -4]], np.float32)  # view 0 sits ON Gaussian 0: its direction is undefined there
    g = np.zeros((2, P, 3), np.float32)
    g[1] = 1.0
    g[0, 1] = 2.0
    g[0, 3, 1] = np.nan
    d_dc, d_rest = np.zeros((P, 1, 3), np.float32), np.zeros((P, K - 1, 3), np.float32)
    hostsim.L.ghrsim_sh_grad_from_views(P, 3, K, _p(xyz), 2, _p(cpos), _p(g), ctypes.c_longlong(3 * P), _p(d_dc), _p(d_rest))
    assert np.isfinite(d_dc[:3]).all() and np.isfinite(d_rest[:3]).all()      # Gaussian 0: only view 1 contributes
    assert np.allclose(d_dc[0, 0], 0.28209479177387814) and np.allclose(d_dc[1, 0], 3 * 0.28209479177387814)
    assert np.isnan(d_dc[3, 0, 1]) and np.isfinite(d_dc[3, 0, 0])
