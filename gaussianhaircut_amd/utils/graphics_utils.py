"""Camera matrices (reference: ``src/utils/graphics_utils.py``), device-agnostic."""
from __future__ import annotations

import math

import numpy as np
import torch


def getWorld2View2(R, t, translate=np.array([.0, .0, .0]), scale=1.0):
    """4x4 world->camera from a camera-to-world rotation R and translation t (graphics_utils.py:38-49)."""
    Rt = np.eye(4)
    Rt[:3, :3] = np.asarray(R).T
    Rt[:3, 3] = t
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))


def getProjectionMatrix(znear, zfar, fovX, fovY, cx=0, cy=0, device=None):
    """OpenGL-style perspective matrix with z in [0,1] and w = z_view (graphics_utils.py:51-72)."""
    fovX, fovY = torch.as_tensor(fovX, dtype=torch.float32), torch.as_tensor(fovY, dtype=torch.float32)
    tx, ty = torch.tan(fovX / 2), torch.tan(fovY / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4, dtype=torch.float32)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[0, 2] = -cx
    P[1, 2] = -cy
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P.to(device) if device is not None else P


def fov2focal(fov, pixels):
    return pixels / (2 * (torch.tan(fov / 2) if isinstance(fov, torch.Tensor) else math.tan(fov / 2)))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))
