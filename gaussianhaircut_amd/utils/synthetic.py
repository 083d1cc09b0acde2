"""Deterministic synthetic workloads of SURVEY.md 8(d) / BASELINE.md 3 (there is no network for real captures).

All randomness comes from a CPU ``torch.Generator`` so a given (config, seed) is bit-identical on every machine;
tensors are moved to the target device afterwards.

cfg 1: 10k random Gaussians, 256x256          cfg 2: 100k random, 1920x1080
cfg 3: 500k strand-aligned (5051 strands x 99 segments, truncated to 500 000), 1080p
cfg 4: cfg-3 model, 32 ring cameras            cfg 5: 2M strand-aligned (20 203 x 99)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from ..scene.cameras import Camera, make_camera, parity_camera
from ..scene.gaussian_model import GaussianModel
from .general_utils import inverse_sigmoid, parallel_transport

BG10 = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 100.0]  # train_gaussians.py:68 (black background, far depth)


def background(device="cpu") -> torch.Tensor:
    return torch.tensor(BG10, dtype=torch.float32, device=device)


@dataclass
class WorkloadSpec:
    name: str
    P: int
    W: int
    H: int
    seed: int
    kind: str  # "random" | "strands"
    log_scale_mean: float = math.log(0.02)
    n_strands: int = 0


CONFIGS: Dict[str, WorkloadSpec] = {
    "cfg1": WorkloadSpec("cfg1_10k_256", 10_000, 256, 256, 0, "random", math.log(0.02)),
    "cfg2": WorkloadSpec("cfg2_100k_1080p", 100_000, 1920, 1080, 1, "random", math.log(0.01)),
    "cfg3": WorkloadSpec("cfg3_500k_strands_1080p", 500_000, 1920, 1080, 2, "strands", n_strands=5051),
    "cfg5": WorkloadSpec("cfg5_2M_strands_1080p", 2_000_000, 1920, 1080, 4, "strands", n_strands=20203),
    # small parity-test shapes
    "tiny": WorkloadSpec("tiny_2k_128x96", 2_000, 128, 96, 7, "random", math.log(0.04)),
    "ragged": WorkloadSpec("ragged_3k_200x136", 3_000, 200, 136, 8, "random", math.log(0.05)),
    "tiny_strands": WorkloadSpec("tiny_strands_64x99_320x240", 64 * 99, 320, 240, 9, "strands", n_strands=64),
}


def _randn(gen, *shape):
    return torch.randn(*shape, generator=gen, dtype=torch.float32)


def _rand(gen, *shape):
    return torch.rand(*shape, generator=gen, dtype=torch.float32)


def random_gaussian_params(P: int, seed: int, log_scale_mean: float, sh_degree: int = 3):
    """xyz ~ U[-1.3,1.3]^3; log-scale ~ N(mean, .5^2) shared + N(0,.3^2) per axis; random unit quaternion;
    opacity = sigmoid(N(0,1.5^2)); SH ~ N(0,.1^2) (+DC offset); label/orient_conf logits ~ N(0,1)/N(0,.3)."""
    g = torch.Generator().manual_seed(seed)
    xyz = (_rand(g, P, 3) * 2 - 1) * 1.3
    log_scales = log_scale_mean + 0.5 * _randn(g, P, 1) + 0.3 * _randn(g, P, 3)
    rot = F.normalize(_randn(g, P, 4), dim=-1)
    opacity_logit = 1.5 * _randn(g, P, 1)
    K = (sh_degree + 1) ** 2
    feats = 0.1 * _randn(g, P, K, 3)
    feats[:, 0, :] += 0.5 * _randn(g, P, 3) + 0.3
    label_logit = _randn(g, P, 1)
    orient_conf_log = 0.3 * _randn(g, P, 1)
    return dict(xyz=xyz, log_scales=log_scales, rotations=rot, opacity_logit=opacity_logit, features=feats,
                label_logit=label_logit, orient_conf_log=orient_conf_log)


def strand_polylines(n_strands: int, n_seg: int, seed: int, step: float = 0.01, rho: float = 0.95):
    """Roots uniform on the unit sphere; each strand a random walk of ``n_seg`` steps of length ``step`` whose
    direction follows an AR(1) process with coefficient ``rho`` (SURVEY.md 8(d) cfg 3)."""
    g = torch.Generator().manual_seed(seed)
    roots = F.normalize(_randn(g, n_strands, 3), dim=-1)
    d = F.normalize(roots + 0.3 * _randn(g, n_strands, 3), dim=-1)
    noise = _randn(g, n_strands, n_seg, 3)
    dirs = torch.empty(n_strands, n_seg, 3)
    for k in range(n_seg):
        d = F.normalize(rho * d + math.sqrt(1 - rho * rho) * 0.35 * noise[:, k], dim=-1)
        dirs[:, k] = d * step
    return roots[:, None, :], dirs


def strand_gaussian_params(n_strands: int, P: int, seed: int, n_seg: int = 99, width: float = 1e-3,
                           sh_degree: int = 3):
    """Free Gaussians initialised exactly like the reference's strand segments (gaussian_model_strands.py:435-452):
    xyz = segment mid-point, scale = (|dir|/2, w, w), quaternion = parallel_transport(x, dir); opacity ~ 1, label ~ 1."""
    origins, dirs = strand_polylines(n_strands, n_seg, seed)
    pts = origins + torch.cat([torch.zeros_like(origins), torch.cumsum(dirs, dim=1)], dim=1)
    xyz = ((pts[:, 1:] + pts[:, :-1]) * 0.5).reshape(-1, 3)[:P]
    d = dirs.reshape(-1, 3)[:P]
    x_axis = torch.zeros_like(d)
    x_axis[:, 0] = 1
    rot = parallel_transport(x_axis, d)
    scales = torch.cat([d.norm(dim=-1, keepdim=True) * 0.5, torch.full((P, 2), width)], dim=-1)
    g = torch.Generator().manual_seed(seed + 1000)
    K = (sh_degree + 1) ** 2
    feats = 0.1 * _randn(g, P, K, 3)
    feats[:, 0, :] += 0.4
    big = torch.full((P, 1), 0.999)
    return dict(xyz=xyz, log_scales=torch.log(scales), rotations=rot, opacity_logit=inverse_sigmoid(big),
                features=feats, label_logit=inverse_sigmoid(big), orient_conf_log=torch.zeros(P, 1))


def make_model(spec: WorkloadSpec, device="cpu", sh_degree: int = 3) -> GaussianModel:
    if spec.kind == "random":
        p = random_gaussian_params(spec.P, spec.seed, spec.log_scale_mean, sh_degree)
    else:
        p = strand_gaussian_params(spec.n_strands, spec.P, spec.seed, sh_degree=sh_degree)
    p = {k: v.to(device) for k, v in p.items()}
    m = GaussianModel(sh_degree)
    m.create_from_tensors(p["xyz"], p["features"], p["log_scales"], p["rotations"], p["opacity_logit"],
                          p["label_logit"], p["orient_conf_log"], spatial_lr_scale=1.0)
    m.active_sh_degree = sh_degree
    return m


def make_view(spec: WorkloadSpec, device="cpu", cam: str = "front") -> Camera:
    """``cam``: one of scene.cameras.PARITY_CAMERAS ("front" = the SURVEY 8(d) camera, identity rotation)."""
    if cam == "front":
        return make_camera(spec.W, spec.H, fovy_deg=40.0, distance=4.0, device=device)
    return parity_camera(cam, spec.W, spec.H, device=device)


@torch.no_grad()
def raster_inputs(spec: WorkloadSpec, device="cpu", model: Optional[GaussianModel] = None,
                  cam=None) -> Dict[str, object]:
    """Everything the rasterizer op consumes in pipeline mode (A) for one view, built with the host-side projection
    (same tensors ``render()`` would pass, gaussian_renderer/__init__.py:58-96), without autograd."""
    model = model or make_model(spec, device)
    if cam is None or isinstance(cam, str):
        cam = make_view(spec, device, cam or "front")
    conic = model.get_conic(cam)
    means2D = model.get_mean_2d(cam)
    from .sh_utils import eval_sh
    xyz = model.get_xyz
    K = (model.max_sh_degree + 1) ** 2
    shs_view = model.get_features.transpose(1, 2).reshape(-1, 3, K)
    d = F.normalize(xyz - cam.camera_center[None], dim=-1)
    rgb = torch.clamp_min(eval_sh(model.active_sh_degree, shs_view, d) + 0.5, 0.0)
    label = model.get_label
    colors = torch.cat([rgb, label, torch.ones_like(label), model.get_direction_2d(cam), model.get_orient_conf,
                        model.get_depths(cam)], dim=-1)
    keep = model.filter_points(cam)
    return dict(
        P=int(keep.sum().item()), W=spec.W, H=spec.H,
        means3D=xyz[keep].contiguous(), means2D=means2D[keep].contiguous(), colors=colors[keep].contiguous(),
        opacities=model.get_opacity[keep].contiguous(), cov3D=model.cov[keep].contiguous(),
        conic=conic[keep].contiguous(), scales=model.get_scaling[keep].contiguous(),
        rotations=model.get_rotation[keep].contiguous(), bg=background(device),
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        tanfovx=math.tan(float(cam.FoVx) * 0.5), tanfovy=math.tan(float(cam.FoVy) * 0.5),
        campos=cam.camera_center, keep=keep)


def grad_image(spec: WorkloadSpec, seed: int, device="cpu", C: int = 10) -> torch.Tensor:
    """dL/dout ~ N(0,1)/N (SURVEY.md 8(d) cfg 2)."""
    g = torch.Generator().manual_seed(seed)
    return (_randn(g, C, spec.H, spec.W) / (spec.H * spec.W)).to(device)
