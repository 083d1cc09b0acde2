"""Minimal camera carrying exactly the fields ``render()`` and the model's projection helpers consume
(reference: ``src/scene/cameras.py:72-80`` / ``MiniCam``): image size, FoV, ``world_view_transform`` (= W2C^T),
``full_proj_transform`` (= view @ proj, row-vector convention) and ``camera_center``; plus optional ground truth."""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from ..utils.graphics_utils import getProjectionMatrix, getWorld2View2


class Camera:
    def __init__(self, R, T, FoVx, FoVy, width, height, znear=0.01, zfar=100.0, device="cpu", image_name="synthetic"):
        self.image_width, self.image_height = int(width), int(height)
        # tensors, like the reference (trainable FoV there): render() calls torch.tan(FoV * 0.5).item()
        self.FoVx = torch.tensor(float(FoVx), dtype=torch.float32, device=device)
        self.FoVy = torch.tensor(float(FoVy), dtype=torch.float32, device=device)
        self.znear, self.zfar = znear, zfar
        self.image_name = image_name
        self.R, self.T = np.asarray(R, dtype=np.float64), np.asarray(T, dtype=np.float64)   # as src/scene/cameras.py:37-38
        w2c = torch.tensor(getWorld2View2(np.asarray(R), np.asarray(T)), dtype=torch.float32)
        self.world_view_transform = w2c.transpose(0, 1).contiguous().to(device)
        self.projection_matrix = getProjectionMatrix(znear, zfar, float(FoVx), float(FoVy)).transpose(0, 1).to(device)
        self.full_proj_transform = (self.world_view_transform @ self.projection_matrix).contiguous()
        self.camera_center = torch.inverse(self.world_view_transform.cpu())[3, :3].to(device)
        self.original_image: Optional[torch.Tensor] = None
        self.original_mask: Optional[torch.Tensor] = None
        self.original_orient_angle: Optional[torch.Tensor] = None
        self.original_orient_conf: Optional[torch.Tensor] = None

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self


def make_camera(width, height, fovy_deg=40.0, distance=4.0, device="cpu") -> Camera:
    """camera at (0,0,-distance) looking down +z; FoVx from the aspect ratio."""
    fovy = math.radians(fovy_deg)
    fovx = 2 * math.atan(math.tan(fovy / 2) * width / height)
    return Camera(np.eye(3), np.array([0.0, 0.0, distance]), fovx, fovy, width, height, device=device)


def ring_cameras(n, width, height, radius=4.0, fovy_deg=40.0, device="cpu", roll_deg=0.0):
    """SURVEY.md 8(d) cfg 4: azimuth 360*k/n, elevation 10*sin(2*pi*k/n) degrees, looking at the origin.
    ``roll_deg`` turns every camera about its own viewing axis (COLMAP poses are never upright: the parity tests use it
    so that all nine entries of the view rotation are non-trivial)."""
    cams = []
    fovy = math.radians(fovy_deg)
    fovx = 2 * math.atan(math.tan(fovy / 2) * width / height)
    for k in range(n):
        az, el = 2 * math.pi * k / n, math.radians(10.0) * math.sin(2 * math.pi * k / n)
        c = radius * np.array([math.cos(el) * math.sin(az), math.sin(el), -math.cos(el) * math.cos(az)])
        fwd = -c / np.linalg.norm(c)
        right = np.cross(np.array([0.0, 1.0, 0.0]), fwd)
        right /= np.linalg.norm(right)
        up = np.cross(fwd, right)
        if roll_deg:
            cr, sr = math.cos(math.radians(roll_deg)), math.sin(math.radians(roll_deg))
            right, up = cr * right + sr * up, -sr * right + cr * up
        R_c2w = np.stack([right, up, fwd], axis=1)  # columns = camera axes in world
        T = -R_c2w.T @ c
        cams.append(Camera(R_c2w, T, fovx, fovy, width, height, device=device, image_name="ring%03d" % k))
    return cams


PARITY_CAMERAS = ("front", "ring5", "ring13roll")


def parity_camera(name, width, height, device="cpu") -> Camera:
    """The cameras every oracle- / reference-pinned parity test runs: the SURVEY front camera (identity rotation), view 5
    of BASELINE configs[3]'s 32-camera ring (azimuth 56 deg: a rotation about y plus a small pitch), and view 13 of that
    ring rolled by 20 deg about its viewing axis (a full 3x3 rotation, as world_view_transform built from COLMAP poses is
    in the reference: src/scene/cameras.py:72-80).  With R = I a transposed W in computeCov2D (forward.cu:74-113,
    backward.cu:144-274) or in the model's T = W J would be invisible."""
    if name == "front":
        return make_camera(width, height, device=device)
    if name == "ring5":
        return ring_cameras(32, width, height, device=device)[5]
    if name == "ring13roll":
        return ring_cameras(32, width, height, device=device, roll_deg=20.0)[13]
    raise KeyError(name)
