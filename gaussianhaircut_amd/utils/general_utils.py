"""Small tensor helpers used on the measured step (reference: ``src/utils/general_utils.py``).

Device-agnostic re-implementations (the reference hard-codes ``device="cuda"``, general_utils.py:66,84,112).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:
    """logit; general_utils.py:19-20."""
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate schedule with optional warm-up (general_utils.py:30-63)."""
    log_a, log_b = (math.log(lr_init), math.log(lr_final)) if lr_init > 0 and lr_final > 0 else (0.0, 0.0)

    def schedule(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        warm = 1.0
        if lr_delay_steps > 0:
            warm = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return warm * np.exp(log_a * (1 - t) + log_b * t)

    return schedule


_UPPER = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))


def strip_symmetric(sym: torch.Tensor) -> torch.Tensor:
    """(P,3,3) symmetric -> (P,6) [xx, xy, xz, yy, yz, zz] (general_utils.py:65-78)."""
    return torch.stack([sym[:, r, c] for r, c in _UPPER], dim=-1)


strip_lowerdiag = strip_symmetric


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """Quaternion (w,x,y,z), normalised here, to the matrix the reference builds at general_utils.py:79-112.

    Note the layout: element [i, j] is the standard rotation matrix's [j, i] (the reference fills the transpose so
    that ``S @ R`` matches the kernel's column-major glm product, forward.cu:134-140).
    """
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)], dim=-1),
        torch.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], dim=-1),
        torch.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], dim=-1),
    ]
    return torch.stack(rows, dim=1)


def build_scaling_rotation(s: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """diag(s) @ R (general_utils.py:114-123)."""
    return s[:, :, None] * build_rotation(r)


def dot(a, b, dim=-1, keepdim=True):
    return (a * b).sum(dim=dim, keepdim=keepdim)


def parallel_transport(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Un-normalised quaternion rotating direction a onto b (general_utils.py:150-160)."""
    a = F.normalize(a, dim=-1)
    b = F.normalize(b, dim=-1)
    return torch.cat([1 + dot(a, b), torch.cross(a, b, dim=-1)], dim=-1)
