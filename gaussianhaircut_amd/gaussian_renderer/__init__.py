"""``render()`` / ``render_hair()`` -- same signatures and return dict as the reference's
``src/gaussian_renderer/__init__.py:23-113,116-214``; the rasterizer behind them is the gfx950 HIP library.

Channel layout of the 10-feature splat (``:64-74``): ``[rgb(3) | hair label(1) | foreground(1) | dir2D(3) |
orientation confidence(1) | view depth(1)]``; outputs are split ``[3, 2, 3, 1, 1]`` (``:100``) and the 2D strand
direction is turned into an orientation angle in [0,1) (``:102-105``).

``pc`` / ``pc_hair`` only need the reference's model interface (get_conic, get_mean_2d, get_direction_2d,
get_depths, filter_points, get_xyz, get_opacity, get_features, get_label, get_orient_conf, cov, ...), so the
reference's own model classes work here too -- and they, too, take the fused HIP path when they keep the reference's
parametrisation (``is_free_gaussian_model``), with constant or trainable cameras.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from ..diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from ..utils.sh_utils import eval_sh


def _tan_half(fov) -> float:
    """tan(FoV/2) as a host float (the reference's ``torch.tan(FoV * 0.5).item()``).  A device-resident FoV tensor costs
    a blocking D2H read that drains the launch queue, so the value is remembered on the tensor object and re-read only
    when the tensor was written since (its autograd version counter moves with every in-place update)."""
    if not isinstance(fov, torch.Tensor):
        return math.tan(float(fov) * 0.5)
    cached = getattr(fov, "_ghr_tan_half", None)
    if cached is None or cached[0] != fov._version:
        cached = (fov._version, math.tan(float(fov.detach()) * 0.5))
        try:
            fov._ghr_tan_half = cached
        except AttributeError:
            pass
    return cached[1]


def _raster_settings(cam, bg_color, scaling_modifier, sh_degree, debug):
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=_tan_half(cam.FoVx), tanfovy=_tan_half(cam.FoVy), bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=sh_degree,
        campos=cam.camera_center, prefiltered=True, debug=debug)


def _sh_to_rgb(deg, shs_view, xyz, campos):
    d = xyz - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(deg, shs_view, d) + 0.5, 0.0)


def orient_angle_from(cov2d):
    """Reference :102-105: rendered 2D strand direction -> orientation angle / pi in [0,1)."""
    dir2d = F.normalize(cov2d[:2], dim=0)
    mirror = torch.where(dir2d[[0]] < 0, -torch.ones_like(dir2d[[0]]), torch.ones_like(dir2d[[0]]))
    return torch.acos(dir2d[[1]].clamp(-1 + 1e-3, 1 - 1e-3) * mirror) / math.pi


class RenderPackage(dict):
    """The reference's return dict.  ``orient_angle`` (10 small PyTorch kernels at full resolution) and
    ``visibility_filter`` (``radii > 0``) are computed on first access: the fused stage-1 loss derives the angle inside
    its own kernel from ``renders_packed`` (the packed [10,H,W] rasterizer output, kept as an attribute) and the
    measured step reads neither."""

    _LAZY = ("orient_angle", "visibility_filter")

    def __init__(self, renders, cov2d, **kw):
        super().__init__(**kw)
        self._cov2d = cov2d
        self.renders_packed = renders
        self.count = None  # fused path: num_rendered (int) or a PendingCount
        self.densify_stats_done = False  # fused path with pipe.densify_stats: the backward pass keeps the statistics

    def _materialise(self, k=None):
        if k in (None, "orient_angle") and not dict.__contains__(self, "orient_angle"):
            dict.__setitem__(self, "orient_angle", orient_angle_from(self._cov2d))
        if k in (None, "visibility_filter") and not dict.__contains__(self, "visibility_filter"):
            dict.__setitem__(self, "visibility_filter", dict.__getitem__(self, "radii") > 0)

    def __getitem__(self, k):
        if k in self._LAZY:
            self._materialise(k)
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        if k in self._LAZY:
            self._materialise(k)
        return dict.get(self, k, default)

    def __contains__(self, k):
        return k in self._LAZY or dict.__contains__(self, k)

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)


def _package(renders, screenspace_points, radii):
    image, mask, cov2d, orient_conf, _ = renders.split([3, 2, 3, 1, 1], dim=0)
    return RenderPackage(renders, cov2d, render=image, mask=mask, orient_conf=orient_conf,
                         viewspace_points=screenspace_points, radii=radii)


_CAMERA_TENSORS = ("world_view_transform", "full_proj_transform", "camera_center", "projection_matrix", "FoVx", "FoVy")


def camera_requires_grad(cam) -> bool:
    """True when autograd is recording and any tensor of the camera the projection reads is being trained.  The
    reference optimises camera pose and FoV by default (``src/arguments/__init__.py:61-62``, ``src/scene/cameras.py:
    83-151``, stepped at ``src/train_gaussians.py:183-196``); their gradients flow through get_conic / get_mean_2d /
    get_direction_2d / get_depths and the SH view direction.  The fused path returns them too (ABI 17: the projection
    backward reduces the camera cotangents, ``fused._camera_grads``), so such a camera no longer forces the generic path."""
    if not torch.is_grad_enabled():
        return False
    for name in _CAMERA_TENSORS:
        t = getattr(cam, name, None)
        if isinstance(t, torch.Tensor) and t.requires_grad:
            return True
    return False


_RAW_FIELDS = ("_xyz", "_scaling", "_rotation", "_opacity", "_label", "_orient_conf", "_features_dc", "_features_rest")
_FREE_ACTIVATIONS = (("scaling_activation", torch.exp), ("opacity_activation", torch.sigmoid),
                     ("label_activation", torch.sigmoid), ("orient_conf_activation", torch.exp),
                     ("rotation_activation", F.normalize))


def is_free_gaussian_model(pc) -> bool:
    """Whether ``pc`` is parametrised like the reference's free-Gaussian ``scene.gaussian_model.GaussianModel``
    (``src/scene/gaussian_model.py:30-43,107-141``): the eight raw tensors, exp / sigmoid / normalize activations (checked by
    identity: the reference binds ``torch.exp`` etc. in ``setup_functions``), SH degree counters -- this package's own class,
    the reference's, or any class that keeps that interface.  Strand models (their per-Gaussian quantities are DERIVED from
    strand parameters: ``_dirs``, ``initialize_gaussians_hair``) are not."""
    from ..scene.gaussian_model import GaussianModel
    if type(pc) is GaussianModel:
        return True
    # (not `_dir`: the reference's free-Gaussian get_direction_2d caches an attribute of that name, gaussian_model.py:389)
    if hasattr(pc, "initialize_gaussians_hair") or hasattr(pc, "_dirs"):
        return False
    if not all(isinstance(getattr(pc, f, None), torch.Tensor) for f in _RAW_FIELDS):
        return False
    if not all(getattr(pc, n, None) is f for n, f in _FREE_ACTIVATIONS):
        return False
    try:
        K = (int(pc.max_sh_degree) + 1) ** 2
        P = pc._xyz.shape[0]
        return (0 <= int(pc.active_sh_degree) <= int(pc.max_sh_degree) <= 3 and
                tuple(pc._features_dc.shape) == (P, 1, 3) and tuple(pc._features_rest.shape) == (P, K - 1, 3) and
                tuple(pc._xyz.shape) == (P, 3) and tuple(pc._scaling.shape) == (P, 3) and
                tuple(pc._rotation.shape) == (P, 4) and pc._opacity.numel() == P and pc._label.numel() == P and
                pc._orient_conf.numel() == P)
    except Exception:
        return False


def _use_fused(pc, pipe, cam=None) -> bool:
    """The fused HIP path covers the free-Gaussian parametrisation (exp / sigmoid / normalize activations, longest-axis
    direction; ``is_free_gaussian_model``) on a ROCm device, through constant AND trainable cameras; anything else (strand
    models here, exotic classes) takes the generic path below, which only relies on the model's public interface and on
    PyTorch autograd."""
    return (getattr(pipe, "fused_projection", True) and is_free_gaussian_model(pc) and pc._xyz.is_cuda)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0):
    """Render the scene (reference :23-113).  ``bg_color`` (10 floats) must be on the GPU."""
    if _use_fused(pc, pipe, viewpoint_camera):
        from .fused import render_model_fused
        # pipe.defer_count (set by trainer.training_step, which owns the recovery): queue the view without ever
        # reading num_rendered back; the package then carries a PendingCount in `.count`
        # pipe.densify_stats: the view's backward pass also keeps the model's densification statistics (the package says so:
        # trainer.densification_step then skips its own PyTorch form)
        dens = bool(getattr(pipe, "densify_stats", False)) and torch.is_grad_enabled()
        renders, radii, screenspace_points, count = render_model_fused(
            viewpoint_camera, pc, bg_color, scaling_modifier, getattr(pipe, "debug", False),
            defer_count=getattr(pipe, "defer_count", False) and not getattr(pipe, "debug", False), densify_stats=dens,
            fuse_adam=bool(getattr(pipe, "fuse_adam", False)))
        pkg = _package(renders, screenspace_points, radii)
        pkg.count = count
        pkg.densify_stats_done = dens
        return pkg
    conic = pc.get_conic(viewpoint_camera, scaling_modifier)  # must precede direction / filter (cached state)
    screenspace_points = pc.get_mean_2d(viewpoint_camera)
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rasterizer = GaussianRasterizer(_raster_settings(viewpoint_camera, bg_color, scaling_modifier,
                                                     pc.active_sh_degree, getattr(pipe, "debug", False)))
    xyz = pc.get_xyz
    n_coef = (pc.max_sh_degree + 1) ** 2
    shs_view = pc.get_features.transpose(1, 2).reshape(-1, 3, n_coef)
    rgb = _sh_to_rgb(pc.active_sh_degree, shs_view, xyz, viewpoint_camera.camera_center)
    cov3D = pc.cov
    dir2d = pc.get_direction_2d(viewpoint_camera)
    label = pc.get_label
    colors = torch.cat([rgb, label, torch.ones_like(label), dir2d, pc.get_orient_conf,
                        pc.get_depths(viewpoint_camera)], dim=-1)

    keep = pc.filter_points(viewpoint_camera)
    radii = torch.zeros_like(xyz[:, 0]).int()
    renders, _radii = rasterizer(means3D=xyz[keep], means2D=screenspace_points[keep], shs=None,
                                 colors_precomp=colors[keep], opacities=pc.get_opacity[keep], scales=None,
                                 rotations=None, cov3D_precomp=cov3D[keep], conic_precomp=conic[keep])
    radii[keep] = _radii
    return _package(renders, screenspace_points, radii)


def _has(obj, name: str) -> bool:
    """``hasattr`` that does not EVALUATE a property: ``hasattr(pc_hair, "get_orient_conf")`` runs the property -- an exp kernel
    over every strand Gaussian plus its autograd node, twice per strand-stage iteration (round 6: tools/strand_torch_profile.py)."""
    if name in getattr(obj, "__dict__", ()):
        return True
    cls = type(obj)
    if hasattr(cls, name):
        return True
    return hasattr(cls, "__getattr__") and hasattr(obj, name)  # (classes that answer attributes dynamically, e.g. nn.Module)


def _use_fused_hair(pc, pc_hair, pipe, cam=None) -> bool:
    """Fused strand-stage path: a free-Gaussian head with its ``*_precomp`` attributes (``src/train_strands.py:65-73``) and a
    strand model exposing the explicit per-Gaussian quantities of ``src/scene/gaussian_model_strands.py:230-452`` (``_dir``,
    ``get_scaling``, ``_rotation``, ``get_orient_conf``, SH features) on a ROCm device; constant or trainable camera."""
    hair_ok = all(_has(pc_hair, n) for n in ("_dir", "_rotation", "_features_dc", "_features_rest", "get_scaling",
                                               "get_orient_conf", "get_xyz", "active_sh_degree"))
    head_ok = is_free_gaussian_model(pc) and all(_has(pc, n) for n in ("xyz_precomp", "opacity_precomp", "scaling_precomp",
                                                                        "rotation_precomp", "mask_precomp", "shs_view"))
    return bool(getattr(pipe, "fused_projection", True) and hair_ok and head_ok and pc_hair.get_xyz.is_cuda)


def render_hair(viewpoint_camera, pc, pc_hair, pipe, bg_color: torch.Tensor, scaling_modifier=1.0):
    """Frozen head Gaussians (``*_precomp`` attributes of ``pc``) + trainable hair strands (reference :116-214)."""
    if _use_fused_hair(pc, pc_hair, pipe, viewpoint_camera):
        from .fused import render_hair_fused
        renders, radii, screenspace_points = render_hair_fused(viewpoint_camera, pc, pc_hair, bg_color,
                                                               scaling_modifier, getattr(pipe, "debug", False),
                                                               fuse_adam=bool(getattr(pipe, "fuse_adam", False)))
        return _package(renders, screenspace_points, radii)
    head = pc.mask_precomp
    conic = torch.cat([pc.get_conic(viewpoint_camera, scaling_modifier)[head],
                       pc_hair.get_conic(viewpoint_camera, scaling_modifier)])
    screenspace_points = torch.cat([pc.get_mean_2d(viewpoint_camera)[head].detach(),
                                    pc_hair.get_mean_2d(viewpoint_camera)], dim=0)
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rasterizer = GaussianRasterizer(_raster_settings(viewpoint_camera, bg_color, scaling_modifier,
                                                     pc_hair.active_sh_degree, getattr(pipe, "debug", False)))
    xyz = torch.cat([pc.xyz_precomp, pc_hair.get_xyz])
    opacity = torch.cat([pc.opacity_precomp, pc_hair.get_opacity])
    keep = torch.cat([pc.filter_points(viewpoint_camera)[head], pc_hair.filter_points(viewpoint_camera)])
    scales = torch.cat([pc.scaling_precomp, pc_hair.get_scaling])
    rotations = torch.cat([pc.rotation_precomp, pc_hair.get_rotation])

    n_coef = (pc.max_sh_degree + 1) ** 2
    shs_view = torch.cat([pc.shs_view, pc_hair.get_features.transpose(1, 2).reshape(-1, 3, n_coef)])
    rgb = _sh_to_rgb(pc_hair.active_sh_degree, shs_view, xyz, viewpoint_camera.camera_center)
    zeros1 = torch.zeros_like(pc.xyz_precomp[:, :1])
    label = torch.cat([zeros1, pc_hair.get_label])
    dir2d = torch.cat([torch.zeros_like(pc.xyz_precomp), pc_hair.get_direction_2d(viewpoint_camera)])
    orient_conf = torch.cat([zeros1, pc_hair.get_orient_conf])
    depth = torch.cat([pc.get_depths(viewpoint_camera)[head], pc_hair.get_depths(viewpoint_camera)])
    colors = torch.cat([rgb, label, torch.ones_like(label), dir2d, orient_conf, depth], dim=-1)

    radii = torch.zeros_like(xyz[:, 0]).int()
    renders, _radii = rasterizer(means3D=xyz[keep], means2D=screenspace_points[keep], shs=None,
                                 colors_precomp=colors[keep], opacities=opacity[keep], scales=scales[keep],
                                 rotations=rotations[keep], cov3D_precomp=None, conic_precomp=conic[keep])
    radii[keep] = _radii
    return _package(renders, screenspace_points, radii)
