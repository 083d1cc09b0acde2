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
            if isinstance(v, torch.nn.Parameter):
                self.__dict__[k] = torch.nn.Parameter(v.detach().to(device), requires_grad=v.requires_grad)
            elif isinstance(v, torch.Tensor):
                self.__dict__[k] = v.to(device)
        return self


def ortho2rotation(poses: torch.Tensor) -> torch.Tensor:
    """6D rotation parametrisation -> rotation matrix with columns (x, y, z): Gram-Schmidt of the two 3-vectors, z = x cross y
    (semantics of the reference's ``ortho2rotation``, src/scene/cameras.py:170-197, incl. its clamp(|x|^2, 1e-8) + 1e-10)."""
    x_raw, y_raw = poses[..., 0:3], poses[..., 3:6]
    x = torch.nn.functional.normalize(x_raw, dim=-1)
    factor = (x * y_raw).sum(-1, keepdim=True) / (torch.clamp((x ** 2).sum(-1, keepdim=True), min=1e-8) + 1e-10)
    y = torch.nn.functional.normalize(y_raw - factor * x, dim=-1)
    z = torch.cross(x, y, dim=-1)
    return torch.stack([x, y, z], -1)


class TrainableCamera(Camera):
    """A camera whose pose and field of view are functions of trainable residuals, as the reference trains them by default
    (src/arguments/__init__.py:61-62; parametrisation of src/scene/cameras.py:85-151, the ``use_barf = False`` branch):
    ``world_view_transform = (W2C @ [[R(rotation_res), translation_res], [0, 1]])^T``, ``FoV = FoV0 + fov_res``.  The reference
    rebuilds every matrix on every property access (and inverts a 4x4 for the camera centre); here ``tensors()`` builds all
    five once per call under autograd -- ``render()`` asks for them once per view (``fused.camera_inputs``) -- and the
    properties are thin views of the same computation for code that reads them one by one.  BARF's se(3) parametrisation
    (utils/camera_opt_utils.py) is out of scope: any camera class whose five tensors carry a graph works the same way."""

    def __init__(self, R, T, FoVx, FoVy, width, height, znear=0.01, zfar=100.0, device="cpu", image_name="synthetic",
                 trainable_cameras=True, trainable_intrinsics=True):
        base = Camera(R, T, FoVx, FoVy, width, height, znear, zfar, device, image_name)
        dev = torch.device(device)
        fixed = dict(base.__dict__)
        self._colmap_transform = fixed.pop("world_view_transform").transpose(0, 1).contiguous()   # W2C
        self._FoVx, self._FoVy = fixed.pop("FoVx"), fixed.pop("FoVy")
        for k in ("projection_matrix", "full_proj_transform", "camera_center"):
            fixed.pop(k)
        self.__dict__.update(fixed)  # image size, znear / zfar, R / T, name, ground-truth slots
        self.trainable_cameras, self.trainable_intrinsics = bool(trainable_cameras), bool(trainable_intrinsics)
        self._rotation_res = torch.nn.Parameter(torch.eye(3, 3)[:2].reshape(-1).clone().to(dev), requires_grad=self.trainable_cameras)
        self._translation_res = torch.nn.Parameter(torch.zeros(3, device=dev), requires_grad=self.trainable_cameras)
        self._fov_res = torch.nn.Parameter(torch.zeros(2, device=dev), requires_grad=self.trainable_intrinsics)
        P0 = torch.zeros(4, 4)
        P0[2, 3], P0[2, 2], P0[3, 2] = 1.0, zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)  # (already transposed)
        self._proj_const = P0.to(dev)
        e = torch.zeros(2, 4, 4)
        e[0, 0, 0] = e[1, 1, 1] = 1.0
        self._proj_slots = e.to(dev)
        self._bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)

    def parameters(self):
        return [self._rotation_res, self._translation_res, self._fov_res]

    def tensors(self):
        """(world_view_transform, full_proj_transform, camera_center, FoVx, FoVy, projection_matrix), one graph."""
        R_a = ortho2rotation(self._rotation_res)
        residual = torch.cat([torch.cat([R_a, self._translation_res[:, None]], dim=1), self._bottom], dim=0)
        view = (self._colmap_transform @ residual).transpose(0, 1)
        fov = torch.stack([self._FoVx, self._FoVy]) + self._fov_res
        inv_tan = 1.0 / torch.tan(fov * 0.5)   # P[0,0] = 2 n / (2 n tan(FoVx / 2)), graphics_utils.py:64-65
        proj = self._proj_const + (self._proj_slots * inv_tan[:, None, None]).sum(0)
        full = view @ proj
        # camera centre: the reference inverts the 4x4 (cameras.py:150); for a rigid transform inverse(view)[3, :3] = -t R^T
        center = -(view[3, :3] @ view[:3, :3].transpose(0, 1))
        return view, full, center, fov[0], fov[1], proj

    world_view_transform = property(lambda self: self.tensors()[0])
    full_proj_transform = property(lambda self: self.tensors()[1])
    camera_center = property(lambda self: self.tensors()[2])
    FoVx = property(lambda self: self.tensors()[3])
    FoVy = property(lambda self: self.tensors()[4])
    projection_matrix = property(lambda self: self.tensors()[5])


def make_camera(width, height, fovy_deg=40.0, distance=4.0, device="cpu") -> Camera:
    """camera at (0,0,-distance) looking down +z; FoVx from the aspect ratio."""
    fovy = math.radians(fovy_deg)
    fovx = 2 * math.atan(math.tan(fovy / 2) * width / height)
    return Camera(np.eye(3), np.array([0.0, 0.0, distance]), fovx, fovy, width, height, device=device)


def ring_cameras(n, width, height, radius=4.0, fovy_deg=40.0, device="cpu", roll_deg=0.0, cls=None):
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
        cams.append((cls or Camera)(R_c2w, T, fovx, fovy, width, height, device=device, image_name="ring%03d" % k))
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
