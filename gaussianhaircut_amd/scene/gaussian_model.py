"""``GaussianModel``: parameters, activations, the PyTorch-side projection maths that feeds the rasterizer, and the
Adam parameter groups of the measured training step.

Host-side mirror of the reference's ``src/scene/gaussian_model.py`` (same attribute / method names and call-order
contract so ``render()`` and a reference-shaped training loop work unchanged):

* activations / getters                      gaussian_model.py:30-141
* ``filter_points``                          :143-228   (python restatement of K1's cull + radius + rect)
* ``get_covariance`` / ``_2d`` / ``get_conic`` :230-315
* ``get_mean_2d`` / ``get_depths`` / ``get_direction_2d``  :317-393
* ``training_setup`` / ``update_learning_rate``           :426-456

* densify / clone / split / prune / reset_opacity           :560-741   (``scene/densification.py``)
* ``save_ply`` / ``load_ply``                                :458-579   (``scene/ply_io.py``)

Device-agnostic (the reference hard-codes ``device="cuda"``, :235,:384).  Out of scope: ``create_from_pcd`` (needs
simple_knn).  ``capture``/``restore`` are symmetric (the reference's restore() unpacks 14 of capture()'s 15 fields,
:65-100).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from ..utils.general_utils import build_rotation, get_expon_lr_func, inverse_sigmoid, strip_symmetric
from .densification import DensificationMixin
from .ply_io import PlyMixin

BLOCK_X = BLOCK_Y = 16


class GaussianModel(DensificationMixin, PlyMixin):
    conic_eps = 1e-12  # gaussian_model.py:312 (strand models use 1e-7, gaussian_model_strands.py:355)

    def setup_functions(self):
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.label_activation = torch.sigmoid
        self.inverse_label_activation = inverse_sigmoid
        self.rotation_activation = F.normalize
        self.orient_conf_activation = torch.exp
        self.orient_conf_inverse_activation = torch.log

    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        empty = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = empty
        self._opacity = self._orient_conf = self._label = empty
        self.max_radii2D = self.xyz_gradient_accum = self.denom = empty
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.setup_functions()

    # ------------------------------------------------------------------ construction
    def create_from_tensors(self, xyz, features, log_scales, rotations, opacity_logit, label_logit=None,
                            orient_conf_log=None, spatial_lr_scale: float = 1.0):
        """Initialise from raw tensors (synthetic stand-in for create_from_pcd, gaussian_model.py:399-424).
        ``features``: (P, (deg+1)^2, 3) SH coefficients, DC first."""
        P = xyz.shape[0]
        dev = xyz.device
        self.spatial_lr_scale = spatial_lr_scale

        def par(t):
            return nn.Parameter(t.detach().clone().float().contiguous().requires_grad_(True))

        self._xyz = par(xyz)
        self._features_dc = par(features[:, :1, :])
        self._features_rest = par(features[:, 1:, :])
        self._scaling = par(log_scales)
        self._rotation = par(rotations)
        self._opacity = par(opacity_logit.reshape(P, 1))
        self._label = par(label_logit.reshape(P, 1) if label_logit is not None else torch.zeros(P, 1, device=dev))
        self._orient_conf = par(orient_conf_log.reshape(P, 1) if orient_conf_log is not None
                                else torch.zeros(P, 1, device=dev))
        self.max_radii2D = torch.zeros(P, device=dev)
        return self

    def capture(self, collective: bool = False):
        """The reference's checkpoint tuple (src/scene/gaussian_model.py:84-99).  Under data parallelism with the sharded
        FusedAdam update the optimizer state is only complete after a collective: ``collective=True`` (every rank makes
        the call) or ``optimizer.sync_moments()`` on every rank first -- a lone ``capture()`` raises StaleMomentsError
        instead of hanging (optim.FusedAdam.state_dict)."""
        opt = None
        if self.optimizer:
            from ..optim import FusedAdam
            opt = self.optimizer.state_dict(collective=collective) if isinstance(self.optimizer, FusedAdam) \
                else self.optimizer.state_dict()
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling,
                self._rotation, self._opacity, self._orient_conf, self._label, self.max_radii2D,
                self.xyz_gradient_accum, self.denom, opt, self.spatial_lr_scale)

    def restore(self, model_args, training_args=None):
        (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
         self._opacity, self._orient_conf, self._label, self.max_radii2D, xyz_gradient_accum, denom, opt_dict,
         self.spatial_lr_scale) = model_args
        if training_args is not None:
            self.training_setup(training_args)
            self.xyz_gradient_accum, self.denom = xyz_gradient_accum, denom
            if opt_dict is not None:
                self.optimizer.load_state_dict(opt_dict)

    # ------------------------------------------------------------------ getters (gaussian_model.py:107-141)
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    @property
    def get_label(self):
        return self.label_activation(self._label)

    @property
    def get_orient_conf(self):
        return self.orient_conf_activation(self._orient_conf)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------------ projection maths
    @staticmethod
    def _view_space(xyz, viewmatrix):
        return xyz @ viewmatrix[:3, :3] + viewmatrix[3:4, :3]

    @staticmethod
    def _tan_half_fov(cam):
        return torch.tan(torch.as_tensor(cam.FoVx) * 0.5), torch.tan(torch.as_tensor(cam.FoVy) * 0.5)

    def _projection_jacobian(self, cam):
        """T = W @ J of gaussian_model.py:264-290 (== the kernel's T, forward.cu:82-99)."""
        h, w = int(cam.image_height), int(cam.image_width)
        tan_fovx, tan_fovy = self._tan_half_fov(cam)
        focal_y, focal_x = h / (2.0 * tan_fovy), w / (2.0 * tan_fovx)
        view = cam.world_view_transform
        t = self._view_space(self.get_xyz, view)
        tz = t[:, 2]
        limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
        tx = torch.clamp(t[:, 0] / tz, min=-limx, max=limx) * tz
        ty = torch.clamp(t[:, 1] / tz, min=-limy, max=limy) * tz
        z = torch.zeros_like(tz)
        # rows of J^T: [fx/tz, 0, 0], [0, fy/tz, 0], [-fx tx/tz^2, -fy ty/tz^2, 0]
        J = torch.stack([torch.stack([focal_x / tz, z, z], dim=-1),
                         torch.stack([z, focal_y / tz, z], dim=-1),
                         torch.stack([-(focal_x * tx) / (tz * tz), -(focal_y * ty) / (tz * tz), z], dim=-1)], dim=1)
        return view[None, :3, :3] @ J

    def get_covariance(self, scaling_modifier=1, return_full_covariance=False):
        """Sigma = (S R)^T (S R); caches ``scaling``, ``R``, ``cov_full``, ``cov`` (gaussian_model.py:230-250)."""
        self.scaling = self.get_scaling
        s = self.scaling * scaling_modifier
        self.R = build_rotation(self._rotation)
        M = s[:, :, None] * self.R
        self.cov_full = M.transpose(1, 2) @ M
        self.cov = strip_symmetric(self.cov_full)
        return self.cov_full if return_full_covariance else self.cov

    def get_covariance_2d(self, viewpoint_camera, scaling_modifier=1):
        """cov2D = T^t Sigma^t T + 0.3 I (gaussian_model.py:252-301)."""
        self.proj_transform_cov = self._projection_jacobian(viewpoint_camera)
        cov_full = self.get_covariance(scaling_modifier, return_full_covariance=True)
        T = self.proj_transform_cov
        full = T.transpose(1, 2) @ cov_full.transpose(1, 2) @ T
        a = full[:, 0, 0] + 0.3
        b = full[:, 0, 1]
        c = full[:, 1, 1] + 0.3
        self.cov2d_full = full
        self.cov2d = torch.stack([a, b, c], dim=-1)
        return self.cov2d

    def get_conic(self, viewpoint_camera, scaling_modifier=1):
        """Inverse 2D covariance (a, b, c) with the reference's epsilon (gaussian_model.py:303-315)."""
        self.cov2d = self.get_covariance_2d(viewpoint_camera, scaling_modifier)
        det = self.cov2d[:, [0]] * self.cov2d[:, [2]] - self.cov2d[:, [1]] ** 2
        det_inv = 1.0 / (det + self.conic_eps)
        self.conic = torch.stack([self.cov2d[:, 2], -self.cov2d[:, 1], self.cov2d[:, 0]], dim=-1) * det_inv
        return self.conic

    def get_mean_2d(self, viewpoint_camera):
        """NDC mean p/(w + 1e-7) (gaussian_model.py:317-337; forward.cu:203-205)."""
        proj = viewpoint_camera.full_proj_transform
        p_hom = self.get_xyz @ proj[:3, :] + proj[3:4, :]
        p_w = 1.0 / (p_hom[:, [3]] + 0.0000001)
        self.xyz_proj = p_hom[:, :3] * p_w
        return self.xyz_proj

    def get_depths(self, viewpoint_camera):
        return self._view_space(self.get_xyz, viewpoint_camera.world_view_transform)[:, -1:]

    def _direction_3d(self):
        """Longest principal axis scaled by its length (gaussian_model.py:384-388).  Needs get_covariance() first."""
        j = self.scaling.argsort(dim=-1, descending=True)[:, 0]
        idx = torch.arange(self.scaling.shape[0], device=self.scaling.device)
        return self.R[idx, j] * self.scaling[idx, j][:, None]

    def get_direction_2d(self, viewpoint_camera):
        """dir3D @ T (gaussian_model.py:344-393).  Call-order: get_conic() first (cached scaling / R)."""
        T = self._projection_jacobian(viewpoint_camera)
        self._dir = self._direction_3d()
        return (self._dir[:, None, :] @ T)[:, 0]

    @torch.no_grad()
    def filter_points(self, viewpoint_camera):
        """Python restatement of K1's cull: z > 0.2, det != 0, non-empty tile rect (gaussian_model.py:143-228).
        Call-order: get_conic() and get_mean_2d() first (uses cached cov2d / xyz_proj)."""
        z = self._view_space(self.get_xyz, viewpoint_camera.world_view_transform)[:, [2]]
        a, b, c = self.cov2d[:, [0]], self.cov2d[:, [1]], self.cov2d[:, [2]]
        det = a * c - b ** 2
        keep = torch.logical_and(z > 0.2, det != 0)
        mid = 0.5 * (a + c)
        root = torch.clamp(mid ** 2 - det, min=0.1) ** 0.5
        radius = torch.ceil(3 * torch.maximum(mid + root, mid - root) ** 0.5)
        W, H = viewpoint_camera.image_width, viewpoint_camera.image_height
        px = ((self.xyz_proj[:, [0]] + 1) * W - 1.0) * 0.5
        py = ((self.xyz_proj[:, [1]] + 1) * H - 1.0) * 0.5
        gx, gy = (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y
        x0 = torch.clamp(((px - radius) / BLOCK_X).int(), min=0, max=gx)
        y0 = torch.clamp(((py - radius) / BLOCK_Y).int(), min=0, max=gy)
        x1 = torch.clamp(((px + radius + BLOCK_X - 1) / BLOCK_X).int(), min=0, max=gx)
        y1 = torch.clamp(((py + radius + BLOCK_Y - 1) / BLOCK_Y).int(), min=0, max=gy)
        self.points_mask = torch.logical_and(keep, (x1 - x0) * (y1 - y0) != 0).squeeze(-1)
        return self.points_mask

    # ------------------------------------------------------------------ optimisation (gaussian_model.py:426-456)
    def param_groups(self, training_args):
        groups = [
            {'params': [self._xyz], 'lr': training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {'params': [self._features_dc], 'lr': training_args.feature_lr, "name": "f_dc"},
            {'params': [self._features_rest], 'lr': training_args.feature_lr / 20.0, "name": "f_rest"},
            {'params': [self._opacity], 'lr': training_args.opacity_lr, "name": "opacity"},
            {'params': [self._label], 'lr': training_args.label_lr, "name": "label"},
            {'params': [self._scaling], 'lr': training_args.scaling_lr, "name": "scaling"},
            {'params': [self._rotation], 'lr': training_args.rotation_lr, "name": "rotation"},
        ]
        if training_args.train_orient_conf:
            groups.append({'params': [self._orient_conf], 'lr': training_args.orient_conf_lr, "name": "orient_conf"})
        return groups

    def training_setup(self, training_args, fused=None):
        """``fused=None`` picks the single-kernel HIP Adam (optim.FusedAdam) when the parameters live on a ROCm
        device, ``torch.optim.Adam`` otherwise; both use the reference's groups, lrs and eps = 1e-15."""
        self.percent_dense = training_args.percent_dense
        P, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        if fused is None:
            fused = self.get_xyz.is_cuda
        if fused:
            from ..optim import FusedAdam
            self.optimizer = FusedAdam(self.param_groups(training_args), eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(self.param_groups(training_args), lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=training_args.position_lr_init * self.spatial_lr_scale,
                                                    lr_final=training_args.position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=training_args.position_lr_delay_mult,
                                                    max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group['lr'] = self.xyz_scheduler_args(iteration)
                return group['lr']

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """gaussian_model.py:739-741."""
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1,
                                                             keepdim=True)
        self.denom[update_filter] += 1

    @torch.no_grad()
    def precompute_head(self):
        """Freeze the non-hair ("head", label < 0.5) Gaussians for the strand stage: the ``*_precomp`` attributes
        ``render_hair()`` reads (reference: src/train_strands.py:65-73)."""
        self.mask_precomp = self.get_label[..., 0] < 0.5
        m = self.mask_precomp
        self.xyz_precomp = self.get_xyz[m].detach()
        self.opacity_precomp = self.get_opacity[m].detach()
        self.scaling_precomp = self.get_scaling[m].detach()
        self.rotation_precomp = self.get_rotation[m].detach()
        self.cov3D_precomp = self.get_covariance(1.0)[m].detach()
        n_coef = (self.max_sh_degree + 1) ** 2
        self.shs_view = self.get_features[m].detach().transpose(1, 2).reshape(-1, 3, n_coef)
        return self

    def leaf_parameters(self):
        ps = [self._xyz, self._features_dc, self._features_rest, self._opacity, self._label, self._scaling,
              self._rotation, self._orient_conf]
        return [p for p in ps if isinstance(p, nn.Parameter)]


class OptimizationParams:
    """Defaults of the reference's ``OptimizationParams`` (``src/arguments/__init__.py:85-122``) that the step uses."""
    iterations = 30_000
    position_lr_init = 0.00016
    position_lr_final = 0.0000016
    position_lr_delay_mult = 0.01
    position_lr_max_steps = 30_000
    feature_lr = 0.0025
    opacity_lr = 0.05
    label_lr = 0.05
    orient_conf_lr = 0.05
    scaling_lr = 0.005
    rotation_lr = 0.001
    percent_dense = 0.01
    lambda_dl1 = 0.8
    lambda_dssim = 0.2
    lambda_dmask = 0.2
    lambda_dorient = 0.0
    lambda_dsds = 0.0
    densification_interval = 100
    opacity_reset_interval = 3000
    densify_from_iter = 500
    densify_until_iter = 15_000
    densify_grad_threshold = 0.0002
    opacity_reg_from_iter = 30_000
    gaussian_pruning_threshold = 0.5
    train_orient_conf = True
    use_gt_orient_conf = True
