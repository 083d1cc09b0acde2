"""Pins the oracle's hand-derived backward (restated from backward.cu) with an INDEPENDENT derivation: a small fp64
PyTorch renderer differentiated by autograd.  Cases avoid threshold crossings (no alpha clamp at 0.99, no T < 1e-4
stop, the alpha >= 1/255 mask is treated as a constant), where the CUDA semantics are piecewise smooth."""
import math

import numpy as np
import torch

from gaussianhaircut_amd.scene.cameras import make_camera
from gaussianhaircut_amd.utils.general_utils import build_rotation


def torch_render(means2D_pix, conic, opacity, colors, order_per_tile, bg, W, H, active):
    """fp64 compositing with the reference's formulas; `active[e, pix]` = pair passed both skips in the oracle."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    out = torch.zeros(colors.shape[1], H, W, dtype=torch.float64)
    T = torch.ones(H, W, dtype=torch.float64)
    for k, g in enumerate(order_per_tile):
        dx = means2D_pix[g, 0] - xs
        dy = means2D_pix[g, 1] - ys
        power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
        alpha = opacity[g] * torch.exp(power) * active[k]
        out = out + colors[g][:, None, None] * (alpha * T)[None]
        T = T * (1 - alpha)
    return out + T[None] * bg[:, None, None]


def test_render_backward_matches_autograd(oracle_mod):
    torch.manual_seed(0)
    W = H = 16  # single tile
    P = 5
    cam = make_camera(W, H)
    tx, ty = math.tan(float(cam.FoVx) / 2), math.tan(float(cam.FoVy) / 2)
    xyz = torch.tensor([[0.05, 0.02, 0.0], [-0.1, 0.05, 0.3], [0.1, -0.1, 0.6], [0.0, 0.1, 0.9], [-0.05, -0.05, 1.2]])
    scales = torch.rand(P, 3) * 0.15 + 0.15
    rot = torch.nn.functional.normalize(torch.randn(P, 4), dim=-1)
    opac = torch.rand(P) * 0.5 + 0.2
    colors = torch.rand(P, 10)
    bg = torch.rand(10)
    view, proj = cam.world_view_transform, cam.full_proj_transform
    out, radii, st = oracle_mod.rasterize_forward(bg.numpy(), xyz.numpy(), colors.numpy(), opac.numpy(), view.numpy(),
                                                  proj.numpy(), tx, ty, H, W, scales=scales.numpy(),
                                                  rotations=rot.numpy())
    assert (radii > 0).all() and st.fragile.sum() == 0
    assert st.final_T.min() > 0.01, "case must not saturate (a stop needs T < 0.01)"
    dL = torch.randn(10, H, W, dtype=torch.float64)
    g = oracle_mod.rasterize_backward(st, bg.numpy(), xyz.numpy(), colors.numpy(), view.numpy(), proj.numpy(), tx, ty,
                                      dL.numpy().astype(np.float32), scales=scales.numpy(), rotations=rot.numpy())

    # ---- independent fp64 graph: same projection maths as the kernel (mode B), written with torch ops -------------
    x64 = xyz.double().requires_grad_(True)
    s64 = scales.double().requires_grad_(True)
    q64 = rot.double().requires_grad_(True)
    o64 = opac.double().requires_grad_(True)
    c64 = colors.double().requires_grad_(True)
    V, Pm = view.double(), proj.double()
    t = x64 @ V[:3, :3] + V[3:4, :3]
    hom = x64 @ Pm[:3, :] + Pm[3:4, :]
    ndc = hom[:, :2] / (hom[:, 3:4] + 1e-7)
    pix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], dim=-1)
    fx, fy = W / (2 * tx), H / (2 * ty)
    # unnormalised-quaternion rotation as in forward.cu:121-138 (here q is unit, so build_rotation agrees)
    # NOT build_rotation(): that normalises q, and d/dq through the normalisation differs from the kernel's
    # gradient (dnormvdv is commented out at backward.cu:340).  Same matrix layout (S @ R), raw q.
    qr, qx, qy, qz = q64[:, 0], q64[:, 1], q64[:, 2], q64[:, 3]
    R = torch.stack([
        torch.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy + qr * qz), 2 * (qx * qz - qr * qy)], -1),
        torch.stack([2 * (qx * qy - qr * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz + qr * qx)], -1),
        torch.stack([2 * (qx * qz + qr * qy), 2 * (qy * qz - qr * qx), 1 - 2 * (qx * qx + qy * qy)], -1)], dim=1)
    M = s64[:, :, None] * R
    Sigma = M.transpose(1, 2) @ M
    tz = t[:, 2]
    z = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, z, z], -1), torch.stack([z, fy / tz, z], -1),
                     torch.stack([-(fx * t[:, 0]) / tz ** 2, -(fy * t[:, 1]) / tz ** 2, z], -1)], dim=1)
    Tm = V[None, :3, :3] @ J
    cov = Tm.transpose(1, 2) @ Sigma.transpose(1, 2) @ Tm
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], dim=-1)
    order = st.point_list[st.ranges[0, 0]:st.ranges[0, 1]].astype(np.int64).tolist()
    # active mask taken from the oracle's own decisions (alpha >= 1/255), constant w.r.t. the inputs
    ys, xs = np.mgrid[0:H, 0:W]
    act = []
    for gidx in order:
        dx, dy = st.xy[gidx, 0] - xs, st.xy[gidx, 1] - ys
        co = st.conic_opacity[gidx]
        power = -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
        act.append(torch.from_numpy(((co[3] * np.exp(power) >= 1 / 255) & (power <= 0)).astype(np.float64)))
    img = torch_render(pix, conic, o64, c64, order, bg.double(), W, H, act)
    assert np.abs(img.detach().numpy() - out).max() < 2e-5
    (img * dL).sum().backward()

    def close(a, b, tol=2e-4):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return np.abs(a - b).max() <= tol * (np.abs(b).max() + 1e-12)

    assert close(g["dL_dcolors"], c64.grad.numpy())
    assert close(g["dL_dopacity"][:, 0], o64.grad.numpy())
    assert close(g["dL_dscales"], s64.grad.numpy())
    assert close(g["dL_drotations"], q64.grad.numpy(), tol=5e-4)
    assert close(g["dL_dmeans3D"], x64.grad.numpy(), tol=5e-4)


def test_mode_a_gradients_match_autograd(oracle_mod):
    """Pipeline mode: d/d(mean2D NDC), d/d(conic) (with the wrapper's 2x on the off-diagonal), d/d(opacity), d/d(colors)."""
    torch.manual_seed(1)
    W = H = 16
    P = 4
    cam = make_camera(W, H)
    tx, ty = math.tan(float(cam.FoVx) / 2), math.tan(float(cam.FoVy) / 2)
    xyz = torch.tensor([[0.02, 0.0, 0.0], [-0.08, 0.06, 0.4], [0.07, -0.05, 0.8], [0.0, 0.0, 1.3]])
    A = torch.randn(P, 2, 2) * 0.1 + torch.eye(2) * 0.35
    S = A @ A.transpose(1, 2) + 0.02 * torch.eye(2)  # SPD conics with pixel-scale extent
    conic = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 1, 1]], -1).contiguous()
    opac = torch.rand(P) * 0.4 + 0.3
    colors = torch.rand(P, 10)
    bg = torch.rand(10)
    view, proj = cam.world_view_transform, cam.full_proj_transform
    cov3D = torch.zeros(P, 6)
    out, radii, st = oracle_mod.rasterize_forward(bg.numpy(), xyz.numpy(), colors.numpy(), opac.numpy(), view.numpy(),
                                                  proj.numpy(), tx, ty, H, W, cov3D_precomp=cov3D.numpy(),
                                                  conic_precomp=conic.numpy())
    assert (radii > 0).all() and st.fragile.sum() == 0 and st.final_T.min() > 0.01
    dL = torch.randn(10, H, W, dtype=torch.float64)
    g = oracle_mod.rasterize_backward(st, bg.numpy(), xyz.numpy(), colors.numpy(), view.numpy(), proj.numpy(), tx, ty,
                                      dL.numpy().astype(np.float32), cov3D_precomp=cov3D.numpy(),
                                      conic_precomp=conic.numpy())
    assert np.abs(g["dL_dmeans3D"]).max() == 0 and np.abs(g["dL_dcov3D"]).max() == 0  # dormant in mode A

    ndc = torch.from_numpy(np.stack([(2 * st.xy[:, 0] + 1) / W - 1, (2 * st.xy[:, 1] + 1) / H - 1], -1)).double()
    ndc.requires_grad_(True)
    k64 = conic.double().requires_grad_(True)
    o64 = opac.double().requires_grad_(True)
    c64 = colors.double().requires_grad_(True)
    pix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], -1)
    order = st.point_list[st.ranges[0, 0]:st.ranges[0, 1]].astype(np.int64).tolist()
    ys, xs = np.mgrid[0:H, 0:W]
    act = []
    for gidx in order:
        dx, dy = st.xy[gidx, 0] - xs, st.xy[gidx, 1] - ys
        co = st.conic_opacity[gidx]
        power = -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
        act.append(torch.from_numpy(((co[3] * np.exp(power) >= 1 / 255) & (power <= 0)).astype(np.float64)))
    img = torch_render(pix, k64, o64, c64, order, bg.double(), W, H, act)
    assert np.abs(img.detach().numpy() - out).max() < 2e-5
    (img * dL).sum().backward()

    def close(a, b, tol=2e-4):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return np.abs(a - b).max() <= tol * (np.abs(b).max() + 1e-12)

    assert close(g["dL_dmeans2D"][:, :2], ndc.grad.numpy())
    gc = g["dL_dconic"]
    assert close(np.stack([gc[:, 0, 0], 2 * gc[:, 0, 1], gc[:, 1, 1]], -1), k64.grad.numpy())
    assert close(g["dL_dopacity"][:, 0], o64.grad.numpy())
    assert close(g["dL_dcolors"], c64.grad.numpy())
