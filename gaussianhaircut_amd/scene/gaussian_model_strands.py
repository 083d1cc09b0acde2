"""Strand-parametrised hair Gaussians: host-side mirror of the projection helpers of the reference's
``src/scene/gaussian_model_strands.py`` (:230-452).  Each strand is a polyline of ``n_seg`` direction vectors; every
segment becomes one Gaussian (mid-point, longest axis = half the segment length along the segment, parallel-transport
quaternion; ``initialize_gaussians_hair`` :435-452).  The learned strand *generators* of the reference depend on the
un-vendored NeuralHaircut checkpoints and are out of scope; strands here are explicit tensors.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib
from ..utils.general_utils import build_scaling_rotation, parallel_transport
from .gaussian_model import GaussianModel


FUSED_STRAND_BUILD = os.environ.get("GHR_FUSED_STRAND_BUILD", "1") != "0"


def _strand_build_applies(origins, dirs) -> bool:
    return (dirs.is_cuda and dirs.dtype == torch.float32 and dirs.dim() == 3 and dirs.shape[-1] == 3 and
            0 < dirs.shape[1] <= _lib.STRAND_MAX_SEG and dirs.is_contiguous() and origins.dtype == torch.float32 and
            origins.device == dirs.device and tuple(origins.shape) == (dirs.shape[0], 1, 3) and not origins.requires_grad)


class _StrandBuild(torch.autograd.Function):
    """(origins [S,1,3], dirs [S,n_seg,3]) -> xyz [P,3], rotation [P,4], scaling [P,3] of the P = S n_seg segment Gaussians, and
    the direction rows [P,3] (a view of dirs): as an output of THIS node their cotangent -- the rasterizer's d_dir3d -- arrives
    in its backward and is added by the kernel, instead of autograd summing two paths into ``_dirs`` with a pass of its own."""

    @staticmethod
    def forward(ctx, origins, dirs, scale):
        from ..diff_gaussian_rasterization import _on_device, _ptr, _stream
        S, n_seg = int(dirs.shape[0]), int(dirs.shape[1])
        P = S * n_seg
        f32 = dict(dtype=torch.float32, device=dirs.device)
        origins = origins.contiguous()
        xyz, rot, scaling = torch.empty((P, 3), **f32), torch.empty((P, 4), **f32), torch.empty((P, 3), **f32)
        with _on_device(dirs.device):
            _lib.check(_lib.lib().ghr_strand_build(_stream(), S, n_seg, _ptr(origins), _ptr(dirs), float(scale), _ptr(xyz),
                                                   _ptr(rot), _ptr(scaling)))
        ctx.save_for_backward(dirs)
        ctx.set_materialize_grads(False)  # an output nobody differentiated arrives as None (the kernel takes NULL), not as zeros
        return xyz, rot, scaling, dirs.view(-1, 3)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_xyz, d_rot, d_scaling, d_dir_rows):
        from ..diff_gaussian_rasterization import _on_device, _ptr, _stream
        (dirs,) = ctx.saved_tensors
        if not ctx.needs_input_grad[1] or (d_xyz is None and d_rot is None and d_scaling is None and d_dir_rows is None):
            return None, None, None
        S, n_seg = int(dirs.shape[0]), int(dirs.shape[1])
        cots = [None if g is None else g.contiguous().float() for g in (d_xyz, d_rot, d_scaling, d_dir_rows)]
        d_dirs = torch.empty_like(dirs)
        with _on_device(dirs.device):
            _lib.check(_lib.lib().ghr_strand_build_backward_ex(_stream(), S, n_seg, _ptr(dirs),
                                                               *[None if g is None else _ptr(g) for g in cots], _ptr(d_dirs)))
        return None, d_dirs, None


class GaussianModelStrands(GaussianModel):
    conic_eps = 1e-7  # gaussian_model_strands.py:355

    def __init__(self, sh_degree: int, scale: float = 1e-3):
        super().__init__(sh_degree)
        self.scale = scale

    def create_from_strands(self, origins, dirs, features, orient_conf_log=None, spatial_lr_scale: float = 1.0):
        """origins: (S,1,3) roots, dirs: (S,n_seg,3) segment vectors, features: (S*n_seg, K, 3)."""
        self.spatial_lr_scale = spatial_lr_scale
        self.pts_origins = origins.detach().clone().float()
        self._dirs = nn.Parameter(dirs.detach().clone().float().requires_grad_(True))
        P = dirs.shape[0] * dirs.shape[1]
        dev = dirs.device
        self._features_dc = nn.Parameter(features[:, :1, :].detach().clone().float().contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(features[:, 1:, :].detach().clone().float().contiguous().requires_grad_(True))
        self._orient_conf = nn.Parameter((orient_conf_log if orient_conf_log is not None
                                          else torch.zeros(P, 1, device=dev)).float().requires_grad_(True))
        self.initialize_gaussians_hair()
        return self

    def initialize_gaussians_hair(self):
        """gaussian_model_strands.py:435-452.  On a ROCm device one HIP kernel each way (``ghr_strand_build``, csrc/ghr_strands.h)
        instead of ~55 PyTorch kernels per iteration; the polyline points ``_pts`` are then made on first use."""
        self._dir = self._dirs.reshape(-1, 3)
        if FUSED_STRAND_BUILD and _strand_build_applies(self.pts_origins, self._dirs):
            self.__dict__.pop("_pts_value", None)
            self._xyz, self._rotation, self._scaling, self._dir = _StrandBuild.apply(self.pts_origins, self._dirs,
                                                                                      float(self.scale))
            return
        self._initialize_gaussians_hair_torch()

    @property
    def _pts(self):
        """[S, n_seg + 1, 3] polyline points (gaussian_model_strands.py:436)."""
        if "_pts_value" not in self.__dict__:
            self._pts_value = self.pts_origins + torch.cat([torch.zeros_like(self.pts_origins),
                                                            torch.cumsum(self._dirs, dim=1)], dim=1)
        return self._pts_value

    @_pts.setter
    def _pts(self, value):
        self._pts_value = value

    def _initialize_gaussians_hair_torch(self):
        pts = self.pts_origins + torch.cat([torch.zeros_like(self.pts_origins), torch.cumsum(self._dirs, dim=1)], dim=1)
        self._pts = pts
        self._dir = self._dirs.reshape(-1, 3)
        self._xyz = ((pts[:, 1:] + pts[:, :-1]) * 0.5).reshape(-1, 3)
        x_axis = torch.zeros_like(self._xyz)
        x_axis[:, 0] = 1.0
        self._rotation = parallel_transport(x_axis, self._dir).reshape(-1, 4)
        scaling = torch.ones_like(self._xyz) * self.scale
        self._scaling = torch.cat([self._dir.norm(dim=-1, keepdim=True) * 0.5, scaling[:, 1:]], dim=-1)

    # activations differ from the free-Gaussian model: scaling / rotation are already in linear space
    @property
    def get_scaling(self):
        return self._scaling

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_opacity(self):  # gaussian_model_strands.py:131-136: strands are opaque hair
        return torch.ones_like(self.get_xyz[:, :1])

    @property
    def get_label(self):
        return torch.ones_like(self.get_xyz[:, :1])

    def get_covariance(self, scaling_modifier=1, return_full_covariance=False):
        from ..utils.general_utils import strip_symmetric
        self.scaling = self.get_scaling
        M = build_scaling_rotation(self.scaling * scaling_modifier, self._rotation)
        self.cov_full = M.transpose(1, 2) @ M
        self.cov = strip_symmetric(self.cov_full)
        return self.cov_full if return_full_covariance else self.cov

    def get_direction_2d(self, viewpoint_camera):
        """normalize(dir) @ T (gaussian_model_strands.py:396-433)."""
        T = self._projection_jacobian(viewpoint_camera)
        return (F.normalize(self._dir, dim=-1)[:, None, :] @ T)[:, 0]

    def param_groups(self, training_args):
        """gaussian_model_strands.py:578-589 (directions, SH, orientation confidence)."""
        return [
            # the reference names the strand-direction group "xyz" so that update_learning_rate() schedules it (:582,:596-602)
            {'params': [self._dirs], 'lr': training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {'params': [self._features_dc], 'lr': training_args.feature_lr, "name": "f_dc"},
            {'params': [self._features_rest], 'lr': training_args.feature_lr / 20.0, "name": "f_rest"},
            {'params': [self._orient_conf], 'lr': training_args.orient_conf_lr, "name": "orient_conf"},
        ]


# The reference's latent stage (src/scene/gaussian_model_latent_strands.py) has the same projection helpers line for line;
# what differs is where the strand polylines come from (a latent texture decoded by an un-vendored strand prior: out of
# scope).  For the hot path the two classes are the same object.
GaussianModelLatentStrands = GaussianModelStrands
GaussianModelCurves = GaussianModelStrands  # the reference's class name in this module (src/scene/gaussian_model_strands.py:31)
