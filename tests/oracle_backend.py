"""TEST-ONLY rasterizer backend: an autograd.Function with the product op's signature that runs the CPU oracle.
Patched over ``gaussianhaircut_amd.diff_gaussian_rasterization.rasterize_gaussians`` by CPU tests so the host-side
plumbing (render(), training_step, gradient bucket, gloo data parallel) can be exercised without a GPU.  The
product never imports this."""
from __future__ import annotations

import contextlib

import numpy as np
import torch

import oracle


def _np(t):
    if t is None or t.numel() == 0:
        return None
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)


LAST = {}


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, conic, rs):
        kw = dict(scales=_np(scales), rotations=_np(rotations), scale_modifier=rs.scale_modifier,
                  cov3D_precomp=_np(cov3D), conic_precomp=_np(conic))
        out, radii, st = oracle.rasterize_forward(_np(rs.bg), _np(means3D), _np(colors), _np(opacities),
                                                  _np(rs.viewmatrix), _np(rs.projmatrix), rs.tanfovx, rs.tanfovy,
                                                  rs.image_height, rs.image_width, **kw)
        ctx.st, ctx.rs, ctx.kw = st, rs, kw
        LAST["state"] = st  # tests that compare stage by stage read the oracle's K1 / binning / K7 state here
        LAST["colors"] = _np(colors)
        ctx.save_for_backward(means3D, colors, scales, rotations, cov3D, conic)
        r = torch.from_numpy(radii)
        ctx.mark_non_differentiable(r)
        return torch.from_numpy(out), r

    @staticmethod
    def backward(ctx, g_out, _):
        means3D, colors, scales, rotations, cov3D, conic = ctx.saved_tensors
        rs = ctx.rs
        g = oracle.rasterize_backward(ctx.st, _np(rs.bg), _np(means3D), _np(colors), _np(rs.viewmatrix),
                                      _np(rs.projmatrix), rs.tanfovx, rs.tanfovy, _np(g_out), **ctx.kw)
        t = {k: torch.from_numpy(v) for k, v in g.items()}
        gc = t["dL_dconic"]
        g_conic = torch.stack([gc[:, 0, 0], 2 * gc[:, 0, 1], gc[:, 1, 1]], dim=-1)

        def opt(x, ref):
            return x if ref.numel() else None

        return (t["dL_dmeans3D"], t["dL_dmeans2D"], None, t["dL_dcolors"], t["dL_dopacity"],
                opt(t["dL_dscales"], scales), opt(t["dL_drotations"], rotations), opt(t["dL_dcov3D"], cov3D),
                opt(g_conic, conic), None)


@contextlib.contextmanager
def oracle_rasterizer():
    import gaussianhaircut_amd.diff_gaussian_rasterization as dgr
    orig = dgr.rasterize_gaussians
    dgr.rasterize_gaussians = lambda *a: _OracleRasterize.apply(*a)
    try:
        yield
    finally:
        dgr.rasterize_gaussians = orig


# ---- the same op in IEEE double over the fp32 oracle's lists: the arbiter of the full-size loss leg -----------------
class _OracleRasterize64(torch.autograd.Function):
    """Pipeline mode (A) only: conic, colours, opacity supplied; the pixel mean is the projection of means3D
    (forward.cu:203-212), its gradient goes to the `means2D` sink (NDC, factors 0.5 W / 0.5 H: backward.cu:464-465)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, conic, rs, st32):
        assert conic is not None and conic.numel() and means3D.dtype == torch.float64
        H, W = int(rs.image_height), int(rs.image_width)
        pm = rs.projmatrix.detach().double()
        m = means3D.detach()
        hom = m @ pm[:3, :] + pm[3:4, :]
        ndc = hom[:, :2] / (hom[:, 3:4] + 0.0000001)
        xy = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=-1).numpy()
        co = torch.cat([conic.detach(), opacities.detach().reshape(-1, 1)], dim=1).numpy()
        col = colors.detach().numpy()
        bg = rs.bg.detach().double().numpy()
        out, final_T, n_contrib = oracle.render_forward64(st32.ranges, st32.point_list, xy, col, co, bg, H, W)
        ctx.pack = (st32, xy, co, col, bg, final_T, n_contrib, H, W)
        LAST["n_contrib64"] = n_contrib
        radii = torch.from_numpy(st32.radii.copy())
        ctx.mark_non_differentiable(radii)
        return torch.from_numpy(out), radii

    @staticmethod
    def backward(ctx, g_out, _):
        st32, xy, co, col, bg, final_T, n_contrib, H, W = ctx.pack
        g = oracle.render_backward64(st32.ranges, st32.point_list, bg, xy, co, col, final_T, n_contrib,
                                     g_out.detach().numpy(), H, W)
        t = {k: torch.from_numpy(v) for k, v in g.items()}
        gc = t["dL_dconic"]
        g_conic = torch.stack([gc[:, 0, 0], 2 * gc[:, 0, 1], gc[:, 1, 1]], dim=-1)
        return (None, t["dL_dmeans2D"], None, t["dL_dcolors"], t["dL_dopacity"], None, None, None, g_conic, None, None)


@contextlib.contextmanager
def oracle_rasterizer64(st32):
    """render() with double tensors composited in double over the lists of the fp32 oracle state `st32`."""
    import gaussianhaircut_amd.diff_gaussian_rasterization as dgr
    orig = dgr.rasterize_gaussians
    dgr.rasterize_gaussians = lambda *a: _OracleRasterize64.apply(*a, st32)
    try:
        yield
    finally:
        dgr.rasterize_gaussians = orig


def double_chain(model, cam, keep):
    """(model, camera) in IEEE double with the same raw parameters; the cull decision is the fp32 chain's (`keep`), so
    that both chains hand the rasterizer the same Gaussians in the same order."""
    import copy
    from gaussianhaircut_amd.scene.gaussian_model import GaussianModel
    m = GaussianModel(model.max_sh_degree)
    m.active_sh_degree = model.active_sh_degree
    for n in ("_xyz", "_scaling", "_rotation", "_opacity", "_label", "_orient_conf", "_features_dc", "_features_rest"):
        setattr(m, n, torch.nn.Parameter(getattr(model, n).detach().double().clone()))
    m.filter_points = lambda _cam: keep
    c = copy.copy(cam)
    for k, v in list(c.__dict__.items()):
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            setattr(c, k, v.detach().double())
    return m, c
