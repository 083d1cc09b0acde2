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
